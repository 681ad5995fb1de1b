"""The pre-training step eager vs replayed from the captured HIP graph (dig_amd/step_graph.py), full size (ViT-S, B = 128):
wall ms per step and host ms per step spent queueing it, A/B inside one process (alternating blocks)."""
import sys, types, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
import dig_amd.utils as U
from dig_amd.registry import create_model
from dig_amd.optim_factory import create_optimizer
from dig_amd.engine_for_pretraining_moco import train_one_epoch
from dig_amd.datasets import RandomMaskingGenerator
dev = torch.device("cuda:0")
model = create_model("pretrain_simmim_moco_ori_vit_small_patch4_32x128", pretrained=False, drop_path_rate=0.0, drop_block_rate=None,
                     mlp_dim=4096, dim=256, T=0.2, num_windows=4, encoder_type='vit', queue_size=65536, patchnet_name='no_patchtrans')
model.to(dev)
B = 128
args = types.SimpleNamespace(num_view=2, moco_m=0.99, use_moco_m_cos=1, epochs=10, contrast_start_epoch=0, contrast_warmup_steps=0,
                             loss_weight_contrast=0.1, loss_weight_pixel=1.0, only_mim_on_ori_img=True, eval_freq=500, opt='adamw',
                             lr=1.5e-4 * B / 256, weight_decay=0.1, opt_eps=1e-8, opt_betas=[0.9, 0.999])
opt = create_optimizer(args, model)
scaler = U.NativeScalerWithGradNormCount()
lr_s, wd_s = np.full(4000, args.lr), np.full(4000, 0.1)
batches = bench.synth_batches(4, B, dev, 1234)
gen = RandomMaskingGenerator((8, 32), 0.7, num_view=2, seed=1234, device=dev)
pos = [0]
def run(n):
    loader = bench.FreshMaskLoader(batches, n, gen)
    model._host_launch = (0.0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_one_epoch(model, None, None, loader, None, opt, dev, 0, scaler, None, patch_size=4, normlize_target=False,
                    start_steps=pos[0], lr_schedule_values=lr_s, wd_schedule_values=wd_s, args=args)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pos[0] += n
    return dt / n * 1e3, model._host_launch[0] / n * 1e3
out = sys.stdout
sys.stdout = sys.stderr
import os
from dig_amd import step_graph
if os.environ.get("PROBE_SERIAL") == "1":            # one stream: is the replay of a linear graph as fast as the eager launches?
    model.overlap_streams = False
model.step_graph = True
run(8)                                   # eager first step, warm-up, capture
for rep in range(int(os.environ.get('PROBE_REPS', '3'))):
    for mode in (False, True):
        model.step_graph = mode
        run(3)
        w, h = run(30)
        print(f"{'graph' if mode else 'eager'}: {w:7.3f} ms/step wall, host {h:6.3f} ms/step to queue it", file=out, flush=True)
print("graphs captured:", len(model._step_graph.graphs), "replays:", model._step_graph.replays, file=out)
