"""Host-side cost of queueing one step (cProfile over train_one_epoch; the GPU runs behind the launch front)."""
import cProfile, pstats, sys, types, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
import dig_amd.utils as U
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
sys.stdout = sys.stderr
run(5, 0)
torch.cuda.synchronize()
# pure host queueing cost: neuter the readback waits by making the GPU irrelevant -> measure with the profiler
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable(); run(10, 5); pr.disable()
print("10 steps wall", time.perf_counter() - t)
st = pstats.Stats(pr, stream=sys.stderr); st.sort_stats("tottime").print_stats(28)
