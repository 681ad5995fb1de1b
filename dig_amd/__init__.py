"""dig_amd: MI355X-native (gfx950) engine for the DiG SimMIM+MoCo-v3 pre-training step."""
__version__ = "0.1.0"

import os as _os

# The engine runs the momentum branch and the weight-gradient GEMMs on a second HIP stream.  ROCclr multiplexes all streams
# of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); RCCL's own streams use those up, after which the side
# stream shares a queue with the main stream and every launch serialises (measured: 28.2 -> 30.9 ms per step under
# torch.distributed).  Must be set before the HIP runtime initialises, i.e. before the first device call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# The captured training step (dig_amd/step_graph.py, opt-in) is a two-branch graph; ROCm's graph executor deals a forked graph's nodes over
# this many internal queues, and with more than two the data-gradient chain hops queues at every fork (ViT-S step replay: 26.3 ms
# with 2 queues, 36.9 with 3, 35.1 with the default 4, 42.0 with 8).  The runtime reads the variable when it initialises, so it is set here --
# but only for a process that asked for the captured step (DIG_STEP_GRAPH=1): every other process keeps the runtime's default.  A caller that
# turns the captured step on programmatically (`model.step_graph = True`: bench.py's extra measurement) sets the variable itself before the
# first device call.
if _os.environ.get("DIG_STEP_GRAPH", "0") == "1":
    _os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
