"""Records the order of collectives one pre-training step of the tiny model issues on its communicator (one-rank RCCL group, collective code
paths forced) and writes tests/golden/collective_order_tiny.json -- the list tests/test_gpu_step.py::test_rccl_path_world1_matches_local_path
asserts.  Run on the GPU box: python tools/gpu_collective_order.py gpurun_out/collective_order_tiny.json [tiny | vit_small]   (then copy it to tests/golden/)"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import dig_oracle as O  # noqa: E402
from gpu_util import build_model, engine_args  # noqa: E402
from dig_amd.parallel import DistributedDataParallel  # noqa: E402
from dig_amd.optim_factory import create_optimizer  # noqa: E402
from dig_amd.engine_for_pretraining_moco import train_one_epoch  # noqa: E402
from dig_amd.utils import NativeScalerWithGradNormCount  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
which = sys.argv[2] if len(sys.argv) > 2 else "tiny"
if which == "tiny":
    cfg, Bn = O.DiGConfig(**O.TINY), 4
    P, S = O.det_state(cfg, 21)
else:                                       # ViT-S: the grouped weight-gradient launches shift every block's bucket by one launch
    cfg, Bn = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128"), 8
    P, S = O.init_state(cfg, 21)
hp = O.StepHyper(lr=1e-3)
im, au, mk = O.synthetic_batch(Bn, cfg, 900)
m = build_model(cfg, P, S)
ddp = DistributedDataParallel(m)
m.comm.world_override = True
args = engine_args(hp)
opt = create_optimizer(args, ddp)
logs = []
for s in range(3):
    m.comm.log = []
    train_one_epoch(ddp, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device("cuda:0"), s,
                    NativeScalerWithGradNormCount(), None, patch_size=4, normlize_target=False, start_steps=s,
                    lr_schedule_values=np.full(4, hp.lr), wd_schedule_values=np.full(4, hp.weight_decay), args=args)
    logs.append([[op, n] for op, n in m.comm.log])
assert logs[1] == logs[2], "the order of collectives must not depend on the step"
with open(sys.argv[1], "w") as f:
    json.dump({"config": f"{which}, B = {Bn}, one rank, collective paths forced", "step": logs[1], "first_step": logs[0]}, f, indent=0)
dist.destroy_process_group()
