"""Where does one stream of the step wait for the other?  Every Stream.wait_stream(other) of a few unprofiled steps is bracketed with
timing events on both streams: stall = when `other` reached the point - when the waiting stream arrived there (if positive)."""
import sys, types, time, traceback, collections
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
    train_one_epoch(model, None, None, loader, None, opt, dev, 0, scaler, None, patch_size=4, normlize_target=False,
                    start_steps=pos[0], lr_schedule_values=lr_s, wd_schedule_values=wd_s, args=args)
    pos[0] += n
out = sys.stdout
sys.stdout = sys.stderr
run(8)
torch.cuda.synchronize()
log = []
orig = torch.cuda.Stream.wait_stream
main_id = torch.cuda.current_stream(dev).cuda_stream
def wait_stream(self, other):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(self); b.record(other)
    fr = traceback.extract_stack(limit=3)[0]
    log.append((f"{fr.name}:{fr.lineno}", self.cuda_stream == main_id, a, b))
    return orig(self, other)
torch.cuda.Stream.wait_stream = wait_stream
base = torch.cuda.Event(enable_timing=True); base.record()
N = 6
t0 = time.perf_counter(); run(N); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N * 1e3
torch.cuda.Stream.wait_stream = orig
agg = collections.OrderedDict()
for where, is_main, a, b in log:
    stall = max(0.0, a.elapsed_time(b))
    k = (where, "main waits for side" if is_main else "side waits for main")
    c = agg.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += stall
print(f"step {dt:.2f} ms (with the bracketing events); stalls per step by call site:", file=out, flush=True)
for (where, who), (c, t) in agg.items():
    if t / N > 0.02:
        print(f"  {where:32s} {who:20s} {c / N:6.1f} waits  {t / N:7.3f} ms", file=out, flush=True)
out.flush()
del log, agg, a, b
torch.cuda.synchronize()
