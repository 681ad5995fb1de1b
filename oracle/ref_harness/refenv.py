"""Import environment for the *reference* DiG code (test infrastructure, build container only).

/root/reference is imported unmodified; what it cannot import in this image (timm, turtle/tkinter,
torch._six, tensorboardX) is provided by the stand-ins in ./shim, and its CUDA-only calls are patched
to CPU no-ops, exactly as SURVEY.md Appendix B describes.  Nothing here is shipped or imported by the
product package, and nothing here travels to the GPU box as a dependency of tests/bench.
"""
import math
import os
import sys
import types

REF = os.environ.get("DIG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def setup(rank=0, world_size=1, port=29541):
    sys.dont_write_bytecode = True
    shim = os.path.join(HERE, "shim")
    for p in (REF, shim):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, shim)
    import torch
    import torch.distributed as dist
    six = types.ModuleType("torch._six")
    six.inf = math.inf
    sys.modules["torch._six"] = six
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    from utils import utils as ref_utils
    ref_utils.SmoothedValue.synchronize_between_processes = lambda self: None
    return ref_utils


def create_ref_model(name="pretrain_simmim_moco_ori_vit_small_patch4_32x128", **over):
    import modeling_pretrain_moco_mim_ori  # noqa: F401  (registers factories)
    from timm.models import create_model
    kw = dict(pretrained=False, drop_path_rate=0.0, drop_block_rate=None, mlp_dim=4096, dim=256, T=0.2,
              num_windows=4, encoder_type='vit', queue_size=65536, patchnet_name='no_patchtrans')
    kw.update(over)
    return create_model(name, **kw)
