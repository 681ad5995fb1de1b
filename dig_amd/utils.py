"""Host-side training utilities with the reference's names and semantics (reference: utils/utils.py).

cosine_scheduler (:522-538), adjust_moco_momentum (:540-543), NativeScalerWithGradNormCount (:477-504, bf16 needs no
loss scaling so `scale` is the constant 1.0 and the work it triggers -- backward, global grad norm, optional clip,
optimizer step -- runs as flat-arena kernels), get_grad_norm_ (:507-519), SmoothedValue / MetricLogger (:30-282),
init_distributed_mode (:375-407), save_model / auto_load_model (:546-651)."""
import datetime
import glob
import json
import math
import os
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist

from . import ops


# ------------------------------------------------------------------------------------------------ schedules
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Per-iteration cosine schedule; like the reference, linear warm-up is only emitted when warmup_epochs > 0
    (warmup_steps then overrides its length)."""
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    print("Set warmup steps = %d" % warmup_iters)
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


def adjust_moco_momentum(epoch, args):
    return 1. - 0.5 * (1. + math.cos(math.pi * epoch / args.epochs)) * (1. - args.moco_m)


# ------------------------------------------------------------------------------------------------ distributed
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    """One process per GPU; RANK / WORLD_SIZE / LOCAL_RANK from the launcher env; backend 'nccl' (= RCCL on ROCm)."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ.get('LOCAL_RANK', 0))
        if os.environ.get("DIG_SHARE_GPU") == "1":           # harness test only: every rank on device 0 (with DIG_DIST_BACKEND=gloo)
            args.gpu = 0
    else:
        print('Not using distributed mode')
        args.distributed = False
        args.rank, args.world_size, args.gpu = 0, 1, 0
        return
    args.distributed = True
    torch.cuda.set_device(args.gpu)
    args.dist_backend = os.environ.get("DIG_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    extra = {"device_id": torch.device("cuda", args.gpu)} if args.dist_backend == "nccl" else {}
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank, **extra)
    dist.barrier()


# ------------------------------------------------------------------------------------------------ grad norm / scaler
def get_grad_norm_(parameters=None, norm_type: float = 2.0, model=None):
    """Global L2 norm of all gradients = sqrt(sum over the flat gradient arena of g^2) (pads are zero)."""
    if norm_type != 2.0:
        raise NotImplementedError("only the L2 norm is used by the pre-training recipe")
    g = model.flat_grads
    ws = getattr(model, "_norm_ws", None)
    if ws is None or ws.device != g.device:
        ws = model._norm_ws = torch.empty(1024 + 1, device=g.device, dtype=torch.float32)
    ops.sumsq(g, ws[:1024], ws[1024:])
    return ws[1024].sqrt()


class NativeScalerWithGradNormCount:
    state_dict_key = "amp_scaler"

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True, adamw_dev_scalars=None):
        model = optimizer.model
        comm = getattr(model, "comm", None)
        world = getattr(comm, "world", 1) if comm is not None else 1
        # data parallel: the backward is seeded with loss / world, so the SUM all-reduce of the gradient arena is DDP's mean
        (loss * (1.0 / world) if world > 1 else loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        if comm is not None:
            comm.finish_grad_sync(model)
        norm = get_grad_norm_(model=model)
        scale = 1.0
        if clip_grad is not None and clip_grad > 0:
            # clip_grad_norm_ semantics: scale by min(1, max_norm / (norm + 1e-6)); one host read, as in torch
            scale = min(1.0, float(clip_grad) / (float(norm) + 1e-6))
        # the squared norm doubles as the update's gate: after a non-finite loss the optimizer launch changes nothing (GradScaler's
        # inf-skip, utils/utils.py:498-504), so the weights a caller sees after the late `sys.exit(1)` of the engine are the last good ones
        if adamw_dev_scalars is not None:           # captured step (dig_amd/step_graph.py): schedule values come from device memory
            optimizer.step(grad_scale=scale, finite_gate=model._norm_ws[1024:], dev_scalars=adamw_dev_scalars)
        else:
            optimizer.step(grad_scale=scale, finite_gate=model._norm_ws[1024:])
        return norm

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass


# ------------------------------------------------------------------------------------------------ logging
class SmoothedValue:
    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return float(np.median(np.array(self.deque))) if self.deque else 0.0

    @property
    def avg(self):
        return float(np.mean(np.array(self.deque))) if self.deque else 0.0

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(float(v))

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join("{}: {}".format(n, str(m)) for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=None):
        header = header or ''
        start = time.time()
        end = time.time()
        iter_time, data_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        n = len(iterable)
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                print(self.delimiter.join([header, f"[{i}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]))
            end = time.time()
        total = time.time() - start
        print('{} Total time: {} ({:.4f} s / it)'.format(header, str(datetime.timedelta(seconds=int(total))), total / max(n, 1)))


# ------------------------------------------------------------------------------------------------ checkpoints
class _JsonlScalarWriter:
    """SummaryWriter's add_scalar / flush / close on a JSON-lines file (`scalars.jsonl` in the log directory: one {"tag", "value", "step",
    "wall_time"} object per line) -- what TensorboardLogger writes through when neither tensorboardX nor torch.utils.tensorboard can be imported."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a", encoding="utf-8")

    def add_scalar(self, tag, value, step=None):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": None if step is None else int(step), "wall_time": time.time()}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


class TensorboardLogger(object):
    """utils/utils.py:285-306 of the reference: the scalar writer `run_mae_pretraining_moco.py --log_dir` builds on rank 0 and hands to
    train_one_epoch (`update(head=..., **scalars)` per step, `set_step()`, `flush()` per epoch).  The reference writes through tensorboardX;
    here: tensorboardX if it is installed, else torch.utils.tensorboard, else a JSON-lines file with the same tags and steps."""

    def __init__(self, log_dir):
        writer = None
        try:
            from tensorboardX import SummaryWriter
            writer = SummaryWriter(logdir=log_dir)
        except ImportError:
            try:
                from torch.utils.tensorboard import SummaryWriter
                writer = SummaryWriter(log_dir=log_dir)
            except ImportError:
                writer = _JsonlScalarWriter(log_dir)
        self.writer = writer
        self.step = 0

    def set_step(self, step=None):
        if step is not None:
            self.step = step
        else:
            self.step += 1

    def update(self, head='scalar', step=None, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            if not isinstance(v, (float, int)):
                raise TypeError(f"TensorboardLogger.update: {k} is {type(v).__name__}, not a scalar")
            self.writer.add_scalar(head + "/" + k, v, self.step if step is None else step)

    def flush(self):
        self.writer.flush()


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def load_state_dict(model, state_dict, prefix='', ignore_missing="relative_position_index"):
    """utils/utils.py:428-480: non-strict load used for `--finetune` checkpoints.  Keys `prefix + name` that the model owns (same
    shape) are loaded; returns (missing_keys, unexpected_keys) and prints them as the reference does.  A shape mismatch raises,
    as `_load_from_state_dict` collects it into error_msgs."""
    own = model.param_shapes()
    picked, unexpected, errors = {}, [], []
    for k, v in state_dict.items():
        if not k.startswith(prefix):
            unexpected.append(k)
            continue
        name = k[len(prefix):]
        if name in own:
            if tuple(v.shape) != tuple(own[name]):
                errors.append(f"size mismatch for {name}: checkpoint {tuple(v.shape)} vs model {tuple(own[name])}")
            else:
                picked[name] = v
        elif not (name.endswith("position_table") or name.startswith("patch_embed.") or name in ("pos_embed", "encoder.pos_embed")):
            unexpected.append(k)                   # (buffers / RecModel's aliases of encoder.patch_embed, model_builder.py:92-93)
    missing = [n for n in own if n not in picked and not any(ig in n for ig in ignore_missing.split('|'))]
    if errors:
        raise RuntimeError("\n".join(errors))
    model.load_state_dict(picked, strict=False)
    if missing:
        print("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if unexpected:
        print("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, unexpected))
    return missing, unexpected


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """checkpoint-{epoch}.pth with the reference's top-level keys {model, optimizer, epoch, scaler, args}."""
    if not is_main_process():
        return
    os.makedirs(args.output_dir, exist_ok=True)
    path = os.path.join(args.output_dir, 'checkpoint-%s.pth' % str(epoch))
    ck = {'model': {k: v.detach().cpu() for k, v in model_without_ddp.state_dict().items()},
          'optimizer': _to_cpu(optimizer.state_dict()),
          'epoch': epoch, 'scaler': loss_scaler.state_dict(), 'args': vars(args) if hasattr(args, "__dict__") else args}
    # dropout / stochastic-depth mask keys (an extra top-level key: the reference ignores it).  The pre-training model carries them only under
    # --drop_path > 0: with the README recipe its checkpoint has exactly the reference's five keys
    if hasattr(model_without_ddp, "drop_step") and not (hasattr(model_without_ddp, "dpr") and not getattr(model_without_ddp, "drop_path_rate", 0.0)
                                                       and hasattr(model_without_ddp, "use_moco_target")):
        ck['dig_amd'] = {'drop_seed': int(model_without_ddp.drop_seed), 'drop_step': int(model_without_ddp.drop_step)}   # reference ignores it)
    torch.save(ck, path)


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """Resume from args.resume, or (auto_resume) from the highest-numbered checkpoint-N.pth in output_dir."""
    if getattr(args, "auto_resume", False) and not getattr(args, "resume", ""):
        latest = -1
        for ck in glob.glob(os.path.join(args.output_dir, 'checkpoint-*.pth')):
            t = ck.split('-')[-1].split('.')[0]
            if t.isdigit():
                latest = max(int(t), latest)
        if latest >= 0:
            args.resume = os.path.join(args.output_dir, 'checkpoint-%d.pth' % latest)
        print("Auto resume checkpoint: %s" % getattr(args, "resume", ""))
    if getattr(args, "resume", ""):
        ck = torch.load(args.resume, map_location='cpu', weights_only=False)
        model_without_ddp.load_state_dict(ck['model'])
        if 'dig_amd' in ck and hasattr(model_without_ddp, "drop_step"):
            model_without_ddp.drop_seed, model_without_ddp.drop_step = int(ck['dig_amd']['drop_seed']), int(ck['dig_amd']['drop_step'])
        print("Resume checkpoint %s" % args.resume)
        if 'optimizer' in ck and 'epoch' in ck:
            optimizer.load_state_dict(ck['optimizer'])
            args.start_epoch = ck['epoch'] + 1
            if 'scaler' in ck:
                loss_scaler.load_state_dict(ck['scaler'])
            print("With optim & sched!")
