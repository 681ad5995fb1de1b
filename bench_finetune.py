"""Benchmark of the fine-tune training step (BASELINE config 5) in the format of bench.py.

    python bench_finetune.py --gpus 1 --steps K --warmup W [--decoder tf_decoder|attention] [--no-drop]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_finetune.py --gpus N ...

One "step" = forward (teacher forcing) + SeqCrossEntropyLoss + backward + gradient all-reduce + AdamW with layer-wise lr decay 0.75 of
`simmim_vit_small_patch4_32x128` + the recognition decoder on a synthetic batch of 256 crops per GPU (97 classes, 25 positions), with
the README regularisers (`--drop 0.1 --attn_drop_rate 0.1 --drop_path 0.1`, decoder dropout 0.1) unless --no-drop; inputs resident in
HBM.  Prints ONE JSON line (rank 0) with the same `roofline` / `cpu_baseline` objects as bench.py (the headline benchmark of the
repository stays bench.py: the pre-training step)."""
import argparse
import contextlib
import json
import os
import sys
import time
import types

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench import GemmProbe, PEAK_BF16, PEAK_HBM                                   # noqa: E402


def synth(B, T, device, seed):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(device)
    lens = torch.from_numpy(rng.randint(3, T + 1, size=B))
    tg = torch.from_numpy(rng.randint(0, 94, size=(B, T)))
    for b in range(B):
        tg[b, int(lens[b]) - 1] = 94
        tg[b, int(lens[b]):] = 95
    return images, tg, lens


def cpu_baseline(decoder, budget_s):
    """The fp32 CPU oracle of this step (oracle/finetune_oracle.py / attn_decoder_oracle.py, pinned to the reference) on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dig_oracle as O
    import decode_oracle as D
    ecfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    Bc = 8
    images, tg, lens = synth(Bc, 25, "cpu", 77)
    if decoder == "tf_decoder":
        import finetune_oracle as F
        c = D.DecoderConfig()
        P = {**D.det_encoder_state(ecfg, 1), **D.det_decoder_state(c, 2)}
        groups = F.param_groups(P, ecfg.depth, 0.75, 0.05)
        state = {}

        def step(i):
            _, grads, _ = F.loss_and_grads(P, ecfg, c, images, tg, lens)
            F.adamw_step(P, grads, state, i + 1, 1e-4, groups)
    else:
        import attn_decoder_oracle as A
        import finetune_oracle as F
        c = A.AttnDecConfig()
        P = {**D.det_encoder_state(ecfg, 1), **A.det_state(c, 2)}
        groups = F.param_groups(P, ecfg.depth, 0.75, 0.05)
        state = {}

        def step(i):
            _, grads, _ = A.loss_and_grads(P, ecfg, c, images, tg, lens)
            F.adamw_step(P, grads, state, i + 1, 1e-4, groups)
    # 16 threads: the best count for the torch CPU oracle on the MI355X host (tools/cpu_baseline_threads.py); with the default of 128
    # the many small decoder ops spend their time in fork/join (0.02 images/s instead of ~8)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))
    step(0)
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < 16):
        step(n + 1)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n * Bc / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} oracle steps (fp32 torch CPU restatement of the reference fine-tune step, rates 0) at batch {Bc}, same model"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="crops per GPU")
    ap.add_argument("--decoder", default="tf_decoder", choices=["tf_decoder", "attention"])
    ap.add_argument("--no-drop", action="store_true", help="every drop rate 0 (the deterministic step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()

    import dig_amd.utils as U
    from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer, FlatGradComm
    from dig_amd.attn_recognizer import AttnRecModelTrain

    sys.stdout.flush()
    saved_fd1 = os.dup(1)
    os.dup2(2, 1)
    stdout = sys.stdout
    sys.stdout = sys.stderr
    dargs = types.SimpleNamespace()
    U.init_distributed_mode(dargs)
    world, rank = U.get_world_size(), U.get_rank()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", dargs.gpu)
    torch.cuda.set_device(dev)
    torch.manual_seed(0 + rank)
    p = 0.0 if a.no_drop else 0.1
    B = a.batch
    margs = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25, drop=p,
                                  attn_drop_rate=p, drop_path=p, opt="adamw", lr=1e-4 * B * world / 256, weight_decay=0.05, opt_eps=1e-8,
                                  opt_betas=[0.9, 0.999])
    model = RecModelTrain(margs, decoder_dropout=p) if a.decoder == "tf_decoder" else AttnRecModelTrain(margs)
    model.to(dev)
    model.train()
    if world > 1:
        model.comm = FlatGradComm(model)
    nl = model.get_num_layers()
    asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
    opt = create_optimizer(margs, model, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    for grp in opt.param_groups:
        grp["lr"] = margs.lr * grp["lr_scale"]
    crit, scaler = SeqCrossEntropyLoss(), U.NativeScalerWithGradNormCount()
    batches = [synth(B, 25, dev, 1234 + rank + 100 * i) for i in range(4)]

    def run(n, start):
        loss = None
        for i in range(n):
            images, tg, lens = batches[(start + i) % len(batches)]
            opt.zero_grad()
            loss = crit(model((images, tg, lens))[0], tg, lens)
            scaler(loss, opt, clip_grad=None, parameters=None)
        return loss

    if a.warmup > 0:
        run(a.warmup, 0)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = run(a.steps, a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # roofline of the dominant kernel family: two extra instrumented steps on EVERY rank (they hold the step's collective)
    model.overlap_streams = False
    probe = GemmProbe() if rank == 0 else contextlib.nullcontext()
    with probe:
        run(2, a.warmup + a.steps)
    model.overlap_streams = True
    roof = None
    if rank == 0:
        summ = probe.summary()
        dom = max(summ, key=lambda k: summ[k]["seconds"])
        d = summ[dom]
        tf, gbs = d["flops"] / d["seconds"] / 1e12, d["bytes"] / d["seconds"] / 1e9
        hbm_bound = d["bytes"] / PEAK_HBM > d["flops"] / PEAK_BF16
        mfma_view = {"achieved": tf, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": tf / (PEAK_BF16 / 1e12)}
        hbm_view = {"achieved": gbs, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": gbs / (PEAK_HBM / 1e9)}
        roof = {"bound": "hbm" if hbm_bound else "mfma", **(hbm_view if hbm_bound else mfma_view), "traffic": None,
                "kernel": f"dig_gemm_bf16[{dom}] (gemm_kernel / gemm_wide_kernel, v_mfma_f32_32x32x16_bf16)", "mfma": mfma_view, "hbm": hbm_view,
                "flops_per_launch": d["flops"] / d["launches"], "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                "avg_launch_us": d["seconds"] / d["launches"] * 1e6, "launches_per_step": d["launches"] // 2,
                "by_variant": {k: {"TFLOP/s": v["flops"] / v["seconds"] / 1e12, "GB/s": v["bytes"] / v["seconds"] / 1e9,
                                   "ms_per_step": v["seconds"] / 2 * 1e3, "launches_per_step": v["launches"] // 2} for k, v in summ.items()}}
    loss_value = float(loss.item())
    sys.stdout.flush()
    sys.stderr.flush()
    os.dup2(saved_fd1, 1)
    sys.stdout = stdout
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    head = "tf_decoder (6 layers, d 512, 8 heads)" if a.decoder == "tf_decoder" else "GRU attention head (sDim = attDim = 512)"
    regs = "all drop rates 0" if a.no_drop else "--drop 0.1 --attn_drop_rate 0.1 --drop_path 0.1" + (", decoder dropout 0.1" if a.decoder == "tf_decoder" else "")
    line = {"metric": "fine-tune images/sec (32x128 crops, 97 classes, 25 positions) ViT-S/4 + recognition decoder", "value": a.steps * B * world / dt,
            "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"simmim_vit_small_patch4_32x128 + {head}: train_class_batch + SeqCrossEntropyLoss + backward + AdamW (layer decay "
                                   f"0.75), {regs}, {B} crops/GPU, random-init weights", "global_batch": B * world, "parallelism": f"dp{world}",
                       "loss": loss_value},
            "roofline": roof}
    if not a.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(a.decoder, a.cpu_budget)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
