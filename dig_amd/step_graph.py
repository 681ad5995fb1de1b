"""The pre-training step as a HIP graph.

One iteration of train_one_epoch launches ~665 kernels on two HIP streams; the host needs ~11.6 ms per 25 ms step to queue them
(ctypes call + tensor bookkeeping per launch) and keeps ahead of the GPU only because nothing in the step reads back.  Captured
once and replayed, the same launches cost the host one hipGraphLaunch.  What makes the step capturable:

* every per-step scalar is read from device memory at run time: AdamW's (lr, weight decay) pairs and bias corrections
  (dig_adamw_step_dev), the EMA momentum (dig_ema_update_dev), the contrastive loss weight (a device scalar in the loss sum);
* inputs live in static buffers (the loader's batch is copied in before each replay), outputs are one static vector of the ten
  values the engine logs;
* nothing inside the step synchronises with the host (the lagged read-back of round 1), allocations come from the graph's private
  pool, the second stream forks from and joins the capturing stream through events (torch's wait_stream), which HIP records as graph
  dependencies.

Measured on one MI355X (tools/gpu_step_graph_probe.py, ViT-S, B = 128, A/B in one process): eager 25.0 ms per step with 10.3 ms of
host work to queue it; replayed 26.3 ms with 1.9 ms of host work.  The replay is SLOWER on the GPU: ROCm's graph executor spreads
the nodes of a forked graph over DEBUG_HIP_FORCE_GRAPH_QUEUES internal queues (4 by default: 35.1 ms; 8: 42.0 ms; 3: 36.9 ms; 2:
26.3 ms; 1: 28.6 ms) without regard to which capture stream a node came from, so the long data-gradient chain hops between queues
at every fork and pays a cross-queue barrier each time; stream priorities are lost as well.  A graph captured from ONE stream
replays exactly as fast as the eager launches (27.77 vs 27.80 ms, 0.34 ms of host work), but one stream costs 2.8 ms per step
against two (`model.overlap_streams = False` selects that form).  dig_amd/__init__.py therefore sets the queue count to 2, and the captured step is OPT-IN (`DIG_STEP_GRAPH=1` or
`model.step_graph = True`): it is the mode for a host that cannot keep 10 ms per step free (a busy data-loading process, a slow
core), not a throughput gain.  bench.py reports both modes (`step_graph` in the JSON line).  Grouping the weight-gradient launches
into one or two forks per encoder block did not help either mode (eager 25.0 -> 25.5 ms, replay unchanged).

Not captured: data-parallel runs (RCCL calls stay eager), gradient clipping (torch's clip_grad_norm_ semantics need one host read),
and anything that changes the launch sequence -- each distinct sequence (contrastive branch on / off, batch size, masked tokens per
sample, normalised targets) gets its own graph, after `WARMUP` eager steps of that sequence (module loading, hipFuncSetAttribute,
workspace growth all happen there).  Replay is bit-identical to the eager step (tests/test_gpu_step.py::test_graphed_step_equals_eager).
"""
import os

import torch

from . import ops

ENABLED = os.environ.get("DIG_STEP_GRAPH", "0") == "1"
WARMUP = 3


class StepGraph:
    """Static buffers + one captured graph per launch sequence, kept on the model (`core._step_graph`)."""
    N_SCALARS = 16           # [0:6] AdamW (lr0, wd0, lr1, wd1, 1/bc1, 1/sqrt(bc2)); [6:8] EMA (m, 1-m); [8] contrastive loss weight

    def __init__(self, device):
        self.device = device
        self.scalars = torch.zeros(self.N_SCALARS, device=device, dtype=torch.float32)
        # the host runs at most two steps ahead of the GPU (the engine resolves step n-1's read-back before it queues step n+1), so a
        # ring of four pinned staging vectors is never overwritten before its copy has run
        self._pinned = [torch.zeros(self.N_SCALARS, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._n = 0
        self.inputs = {}                 # input signature -> (images, aug, mask) static buffers
        self.graphs = {}                 # key -> (CUDAGraph, outputs)
        self.eager_runs = {}             # key -> eager executions so far
        self.replays = 0

    def set_scalars(self, values):
        pin = self._pinned[self._n % len(self._pinned)]
        self._n += 1
        pin[:len(values)] = torch.tensor(values, dtype=torch.float32)
        self.scalars.copy_(pin, non_blocking=True)

    def set_inputs(self, images, aug, mask):
        sig = tuple((tuple(t.shape), t.dtype) for t in (images, aug, mask))
        bufs = self.inputs.get(sig)
        if bufs is None:
            bufs = self.inputs[sig] = tuple(torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in (images, aug, mask))
        for b, t in zip(bufs, (images, aug, mask)):
            b.copy_(t, non_blocking=True)
        return sig, bufs

    def run(self, key, body):
        """body() performs one step from the static buffers and returns its output tensors; the first WARMUP calls per key run it
        eagerly, the next one captures it, every later one replays."""
        hit = self.graphs.get(key)
        if hit is None:
            n = self.eager_runs.get(key, 0)
            if n < WARMUP:
                self.eager_runs[key] = n + 1
                return body()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = body()
            hit = self.graphs[key] = (graph, out)          # capturing executed nothing: the replay below is this call's step
        hit[0].replay()
        self.replays += 1
        return hit[1]


def usable(core, model, optimizer, loss_scaler, max_norm):
    """The captured form covers the single-process recipe; everything else takes the eager path."""
    from .optim_factory import FusedAdamW
    from .utils import NativeScalerWithGradNormCount
    if not getattr(core, "step_graph", ENABLED) or model is not core:
        return False
    comm = getattr(core, "comm", None)
    if comm is not None and (comm.world > 1 or getattr(comm, "world_override", False)):
        return False
    if max_norm is not None and max_norm > 0:
        return False
    if not (getattr(core, "use_moco_target", True) and getattr(core, "use_pixel_target", True)):
        return False                                  # (the single-objective models run eagerly)
    return type(optimizer) is FusedAdamW and type(loss_scaler) is NativeScalerWithGradNormCount and optimizer.model is core


def get(core, device):
    sg = getattr(core, "_step_graph", None)
    if sg is None or sg.device != device:
        sg = core._step_graph = StepGraph(device)
    return sg


def adamw_scalars(optimizer):
    """(lr0, wd0, lr1, wd1, 1/bc1, 1/sqrt(bc2)) of the step the optimizer is about to take."""
    g0, g1 = optimizer.param_groups
    b1, b2 = g0["betas"]
    c1, c2 = ops.adamw_bias_corrections(b1, b2, optimizer._step + 1)
    return [g0["lr"], g0["weight_decay"], g1["lr"], g1["weight_decay"], c1, c2]
