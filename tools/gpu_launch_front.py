"""Where is the host launch front relative to the GPU? host time vs GPU event time at entry of each encoder pass."""
import sys, types, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
import dig_amd.utils as U
from dig_amd.registry import create_model
from dig_amd.optim_factory import create_optimizer
from dig_amd.engine_for_pretraining_moco import train_one_epoch
from dig_amd import engine_core
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
lr_s, wd_s = np.full(1000, args.lr), np.full(1000, 0.1)
batches = bench.synth_batches(4, B, dev, 1234)
def run(n, start):
    loader = [batches[i % 4] for i in range(n)]
    return train_one_epoch(model, None, None, loader, None, opt, dev, 0, scaler, None, patch_size=4, normlize_target=False,
                           start_steps=start, lr_schedule_values=lr_s, wd_schedule_values=wd_s, args=args)
sys.stdout = sys.stderr
run(5, 0)
torch.cuda.synchronize()
log = []
base_ev = torch.cuda.Event(enable_timing=True)
def mark(tag):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    log.append((tag, time.perf_counter(), ev))
def wrap(name):
    orig = getattr(engine_core._Step, name)
    def f(self, *a, **k):
        mark(name + ":in"); r = orig(self, *a, **k); mark(name + ":out"); return r
    setattr(engine_core._Step, name, f)
for n in ("encoder_forward", "encoder_backward", "forward", "backward"):
    wrap(n)
base_ev.record(); t0 = time.perf_counter()
run(6, 5)
torch.cuda.synchronize()
for tag, th, ev in log:
    print(f"{tag:24s} host {1e3*(th-t0):8.2f} ms   gpu {base_ev.elapsed_time(ev):8.2f} ms")
