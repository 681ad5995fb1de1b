"""Soak: N steps of the full-size step on a fixed set of synthetic batches; loss trend, memory growth, step-time drift."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dig_amd.utils as U
from dig_amd.registry import create_model
from dig_amd.optim_factory import create_optimizer
from dig_amd.engine_for_pretraining_moco import train_one_epoch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = create_model("pretrain_simmim_moco_ori_vit_small_patch4_32x128", pretrained=False, drop_path_rate=0.0, drop_block_rate=None,
                     mlp_dim=4096, dim=256, T=0.2, num_windows=4, encoder_type='vit', queue_size=65536, patchnet_name='no_patchtrans').to(dev)
B = 128
args = types.SimpleNamespace(num_view=2, moco_m=0.99, use_moco_m_cos=1, epochs=10, contrast_start_epoch=0, contrast_warmup_steps=0,
                             loss_weight_contrast=0.1, loss_weight_pixel=1.0, only_mim_on_ori_img=True, eval_freq=500, opt='adamw',
                             lr=1.5e-4 * B / 256, weight_decay=0.1, opt_eps=1e-8, opt_betas=[0.9, 0.999])
opt = create_optimizer(args, model)
scaler = U.NativeScalerWithGradNormCount()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lr_s = np.concatenate([np.linspace(0, args.lr, 20), np.full(N, args.lr)])   # (the reference cosine_scheduler asserts with warmup_epochs=0)
wd_s = np.full(N + 8, 0.1)
batches = bench.synth_batches(8, B, dev, 99)
sys.stdout = open(os.devnull, "w")
out = []
for chunk in range(N // 50):
    torch.cuda.synchronize(); t = time.perf_counter()
    st = train_one_epoch(model, None, None, [batches[i % 8] for i in range(50)], None, opt, dev, 0, scaler, None, patch_size=4,
                         normlize_target=False, start_steps=chunk * 50, lr_schedule_values=lr_s, wd_schedule_values=wd_s, args=args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    out.append((chunk, st["loss"], st["loss_pixel"], st["loss_contrast"], st["grad_norm"], dt * 1e3, torch.cuda.memory_allocated() / 2**20,
                torch.cuda.max_memory_allocated() / 2**20))
sys.stdout = sys.__stdout__
for o in out:
    print("steps %4d-%4d  loss %.4f  pixel %.4f  contrast %.4f  gnorm %.3f  %.2f ms/step  alloc %.0f MiB  peak %.0f MiB" % (o[0] * 50, o[0] * 50 + 49, *o[1:]))
assert all(np.isfinite(o[1]) for o in out) and out[-1][2] < out[0][2] and abs(out[-1][6] - out[1][6]) < 64
print("SOAK_OK")
