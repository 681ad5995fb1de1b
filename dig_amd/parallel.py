"""Data parallelism for the MI355X engine: one process per GPU, torch.distributed 'nccl' backend (= RCCL over xGMI).

Replaces the reference's SyncBatchNorm.convert + DistributedDataParallel pair (run_mae_pretraining_moco.py:389-392):
  * gradients live in ONE flat fp32 arena; as soon as the hand-written backward finishes a stage (decoder, heads,
    encoder block i, ...) its contiguous arena range is all-reduced asynchronously on RCCL's stream while earlier
    layers are still computing -- 17 large collectives per step instead of bucket bookkeeping over 183 tensors;
    xGMI is point-to-point (7 links/GPU), so fewer, larger messages are the right shape;
  * BatchNorm statistics ([2,C] sum / sum-of-squares vectors) are all-reduced between the `stats` and `apply`
    kernels (SyncBN semantics: statistics over the global batch);
  * MoCo keys are all-gathered once per step as a single [2, 4B, dim] message (the reference gathers k1 and k2
    separately, modeling_pretrain_moco_mim_ori.py:551-552,580-591);
  * the mean over ranks is taken by seeding the backward with loss / world (utils.NativeScalerWithGradNormCount): no scaling pass;
  * parameters / buffers are broadcast from rank 0 at construction (what DDP's constructor does).
"""
import torch
import torch.distributed as dist


class DistComm:
    def __init__(self, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._pending = []
        self._ready = set()                  # buckets of a group (grad_ready) that are final and wait for their neighbours
        self.log = None                      # a list: every collective issued is recorded as (op, numel) -- the order on the communicator
        #                                       must be the same on every rank or the first multi-GPU step dead-locks (tests/golden/collective_order_tiny.json)

    def _note(self, op, t):
        if self.log is not None:
            self.log.append((op, int(t.numel())))

    def all_reduce_(self, t):
        self._note("all_reduce", t)
        dist.all_reduce(t, group=self.group)
        return t

    def all_gather_cat(self, t):
        out = torch.empty((self.world,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        self._note("all_gather", t)
        if dist.get_backend(self.group) == "gloo":           # (test backend: no all_gather_into_tensor for device tensors)
            dist.all_gather(list(out.unsqueeze(1).unbind(0)), t.contiguous(), group=self.group)
        else:
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def grad_ready(self, model, key):
        """Called by the backward as soon as every gradient in bucket `key` is final.  Buckets of one group (MoCo_ViT.bucket_groups: small
        neighbours in the arena) go out together, when the last of them is final: 14 gradient messages per step instead of 17."""
        group = model.bucket_groups().get(key) if hasattr(model, "bucket_groups") else None
        if group is not None:
            self._ready.add(key)
            if not all(k in self._ready for k in group):
                return
            self._ready.difference_update(group)
            rng = [model.bucket_range(k) for k in group]
            lo, hi = min(r[0] for r in rng), max(r[1] for r in rng)
            key = "+".join(group)
        else:
            lo, hi = model.bucket_range(key)
        g = model.flat_grads[lo:hi]
        self._note("all_reduce_async:" + key, g)
        self._pending.append((dist.all_reduce(g, group=self.group, async_op=True), g))

    def finish_grad_sync(self, model):
        """Wait for the outstanding bucket all-reduces.  The 1/world averaging costs nothing: NativeScalerWithGradNormCount seeds the
        backward with loss / world, so every rank's gradients arrive pre-divided and the SUM all-reduce is the mean (a power-of-two
        scale commutes with every rounding on the way: bit-identical to scaling the 174 MB arena afterwards, minus that pass)."""
        if self._ready:
            raise RuntimeError(f"gradient buckets {sorted(self._ready)} were final but their group never completed: a backward stage did not report")
        for work, _ in self._pending:
            work.wait()
        self._pending = []


class DistributedDataParallel(torch.nn.Module):
    """Wrapper exposing `.module` like torch's DDP; the collectives are issued by the model's own backward."""

    def __init__(self, module, process_group=None, broadcast=True):
        super().__init__()
        self.module = module
        module.comm = DistComm(process_group)
        if getattr(module, "drop_path_rate", 0.0):
            # stochastic-depth mask keys: every rank draws its own (the reference seeds torch with seed + rank, run_mae_pretraining_moco.py:313)
            module.drop_seed = (int(module.drop_seed) + 0x9E3779B97F4A7C15 * module.comm.rank) & ((1 << 64) - 1)
        if broadcast:
            for k in ("online", "momentum", "bn_stats", "bn_count"):
                dist.broadcast(module._flat[k], src=0, group=process_group)
            if hasattr(module, "mark_weights_changed"):
                module.mark_weights_changed()            # (the arena was just rewritten behind the optimizer's back: bf16 shadow and transposed copies are stale)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
