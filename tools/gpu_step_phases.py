"""GPU time per phase of the UNPROFILED step (events on the caller's stream at the phase boundaries of dig_amd/engine_core.py): rocprofv3
serialises the two HIP streams of the forward and slows the host, so its timeline overstates the step by ~3 ms; this is the same
breakdown without a profiler attached."""
import sys, types
import numpy as np, torch
sys.path.insert(0, ".")
import bench
import dig_amd.utils as U
from dig_amd import engine_core
from dig_amd.registry import create_model
from dig_amd.optim_factory import create_optimizer
from dig_amd.engine_for_pretraining_moco import train_one_epoch
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
out = sys.stdout
sys.stdout = sys.stderr
run(30, 0)
torch.cuda.synchronize()
engine_core.PHASE_MARKS = []
n = 40
run(n, 30)
torch.cuda.synchronize()
marks = engine_core.PHASE_MARKS
per = len(marks) // n
names = [m[0] for m in marks[:per]]
acc = np.zeros(per)
for s in range(1, n):                                   # phase k = from mark k-1 to mark k; phase 0 = from the previous step's last mark
    for k in range(per):
        prev = marks[s * per + k - 1][1]
        acc[k] += prev.elapsed_time(marks[s * per + k][1])
sys.stdout = out
print(f"step {acc.sum() / (n - 1):.2f} ms (GPU, between marks; {n - 1} steps)")
# marks inside the encoder blocks ("blk: ...") are summed over the blocks; an event record costs the stream ~3 us (tools/experiments/
# boundary_lab.hip), so the per-launch figures carry that much of the tool itself
blk = {}
for k in range(per):
    if names[k].startswith("blk: "):
        blk[names[k]] = blk.get(names[k], 0.0) + acc[k] / (n - 1)
shown = set()
for k in range(per):
    if names[k].startswith("blk: "):
        if names[k] not in shown:
            shown.add(names[k])
            print(f"  {blk[names[k]]:6.2f} ms  12 x '{names[k]}' = {blk[names[k]] / 12 * 1e3:6.1f} us per block (stream time up to the end of that launch)")
        continue
    print(f"  {acc[k] / (n - 1):6.2f} ms  up to '{names[k]}'" + ("   (= optimizer of the previous step, loader, input glue)" if k == 0 else ""))
