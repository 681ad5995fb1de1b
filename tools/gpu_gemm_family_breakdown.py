"""Per-shape breakdown of the dig_gemm_bf16 launches of one pre-training step (overlap off, HIP events per launch): which shapes hold the
time of each family and how far each is from the two roofs."""
import os, sys, types, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dig_amd import ops
import dig_amd.utils as U
from dig_amd.registry import create_model
from dig_amd.optim_factory import create_optimizer
from dig_amd.engine_for_pretraining_moco import train_one_epoch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = create_model("pretrain_simmim_moco_ori_vit_small_patch4_32x128", pretrained=False, drop_path_rate=0.0, drop_block_rate=None, mlp_dim=4096,
                     dim=256, T=0.2, num_windows=4, encoder_type='vit', queue_size=65536, patchnet_name='no_patchtrans')
model.to(dev)
B = 128
args = types.SimpleNamespace(num_view=2, moco_m=0.99, use_moco_m_cos=1, epochs=10, contrast_start_epoch=0, contrast_warmup_steps=0, loss_weight_contrast=0.1,
                             loss_weight_pixel=1.0, only_mim_on_ori_img=True, eval_freq=500, opt='adamw', lr=1.5e-4 * B / 256, weight_decay=0.1,
                             opt_eps=1e-8, opt_betas=[0.9, 0.999])
opt = create_optimizer(args, model); scaler = U.NativeScalerWithGradNormCount()
batches = bench.synth_batches(2, B, dev, 1234)
lr_s, wd_s = np.full(64, args.lr), np.full(64, 0.1)
def run(n, start):
    return train_one_epoch(model, None, None, [batches[i % 2] for i in range(n)], None, opt, dev, 0, scaler, None, patch_size=4, normlize_target=False,
                           start_steps=start, lr_schedule_values=lr_s, wd_schedule_values=wd_s, args=args)
sys.stdout = open(os.devnull, "w")
run(3, 0)
model.overlap_streams = False
rec = []
orig = ops.gemm
def timed(A, Bm, I, J, R, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(A, Bm, I, J, R, **kw); e1.record()
    fam = "wgrad" if kw.get("ta") else ("dgrad" if kw.get("tb") else "fwd")
    byt = 2.0 * I * R + 2.0 * J * R + (8.0 * I * J if kw.get("ta") else I * J * (4.0 if kw.get("out_kind") == ops.OUT_F32 else 2.0)
                                        + 2.0 * I * J * ((kw.get("pre") is not None) + (kw.get("resid") is not None)))
    tag = f"{fam} I={I} J={J} R={R} act={kw.get('act', 0)} resid={int(kw.get('resid') is not None)} pre={int(kw.get('pre') is not None)} bk={kw.get('bk', 0)}"
    rec.append((tag, 2.0 * I * J * R, byt, e0, e1)); return out
ops.gemm = timed
run(2, 3)
ops.gemm = orig
torch.cuda.synchronize()
sys.stdout = sys.__stdout__
agg = collections.OrderedDict()
for tag, fl, byt, e0, e1 in rec:
    d = agg.setdefault(tag, [0, 0.0, 0.0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += fl; d[3] += byt
tot = sum(v[1] for v in agg.values())
for tag, (n, t, fl, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{t / 2 * 1e3:6.3f} ms/step {n // 2:3d} x {t / n * 1e6:7.1f} us  {fl / t / 1e12:6.0f} TF/s ({fl / t / 2.5e15 * 100:4.1f} % mfma)  {byt / t / 1e9:6.0f} GB/s ({byt / t / 8e12 * 100:4.1f} % hbm)  {tag}")
print(f"all GEMM launches: {tot / 2 * 1e3:.2f} ms per step")
