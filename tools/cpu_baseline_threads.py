"""How many threads give the CPU oracle its best rate on this host?  (bench.py / bench_finetune.py report the CPU baseline.)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dig_oracle as O
cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tr = O.OracleTrainer(cfg, seed=0)
im, au, mk = O.synthetic_batch(Bc, cfg, 1234)
hp = O.StepHyper(lr=1.5e-4 * Bc / 256)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
for nt in (8, 16, 32, 64, 128):
    if nt > (os.cpu_count() or 8): break
    torch.set_num_threads(nt)
    tr.step(im, au, mk, hp)
    t0 = time.perf_counter(); n = 0
    while n < 1 or (time.perf_counter() - t0 < 12 and n < 6):
        tr.step(im, au, mk, hp); n += 1
    dt = time.perf_counter() - t0
    print(f"threads {nt:4d}: {n * Bc / dt:7.2f} samples/s ({dt / n:.2f} s per step of {Bc})", flush=True)
