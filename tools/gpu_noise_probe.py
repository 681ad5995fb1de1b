import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch, dig_oracle as O
from gpu_util import build_model, run_engine_steps
cfg = O.DiGConfig(**O.TINY); hp = O.StepHyper(lr=1e-3)
im, au, mk = O.synthetic_batch(4, cfg, 900)
gs = []
for _ in range(3):
    m = build_model(cfg, *O.det_state(cfg, 21))
    run_engine_steps(m, [(im, au, mk)], hp)
    gs.append(m.flat_grads.clone())
print("run-to-run rel diff:", ((gs[0]-gs[1]).norm()/gs[0].norm()).item(), ((gs[0]-gs[2]).norm()/gs[0].norm()).item())
m = build_model(cfg, *O.det_state(cfg, 21)); m.overlap_streams = False
run_engine_steps(m, [(im, au, mk)], hp)
print("overlap off vs on:", ((gs[0]-m.flat_grads).norm()/gs[0].norm()).item())
