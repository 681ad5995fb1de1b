"""Benchmark of the DiG pre-training step on MI355X (the driver's contract).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full `train_one_epoch` iteration of `pretrain_simmim_moco_ori_vit_small_patch4_32x128` on a synthetic
batch of 128 samples per GPU (each sample = original 32x128 crop + augmented view, 179/256 patches masked), README
loss weights (MIM 1.0 + MoCo 0.1), forward + backward + gradient all-reduce + fused AdamW, inputs resident in HBM.
Prints ONE JSON line (rank 0).  Extra objects: `roofline` for the dominant kernel (algorithmic FLOPs / measured
launch time from HIP events on the launch stream) and `cpu_baseline` (the fp32 CPU oracle timed on this host).
"""
import argparse
import json
import os
import sys
import time
import types

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # before HIP initialises: see dig_amd/__init__.py
# this process also measures the captured (HIP-graph) form of the step beside the eager one (`step_graph` in the line): the graph executor's
# queue count for it (dig_amd/__init__.py sets it only under DIG_STEP_GRAPH=1); eager launches -- the headline -- do not read it
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = {"small": 99.6e9, "base": 171e9, "tiny": None}      # SURVEY.md §8(d), MIM+MoCo
ENC_FWD_FLOP = {"small": 12.089e9, "base": 20.951e9, "tiny": None}     # encoder forward per 32x128 image
# the single-objective factories (modeling_pretrain_moco_mim_ori.py:627-681, 709-763, 818-871): credited work per sample by the same rules --
# Gen-only: encoder forward + backward of the masked view + the decoder on its 179 masked rows = 3 F + 0.129 G (BASELINE configs[1]'s credit,
# here without the momentum networks and heads the SimMIM+MoCo model still runs for its meters); Dis-only: online forward + backward and
# momentum forward of two views + projector / predictor / InfoNCE = 8 F + 1.39 G
KIND_FACTORY = {"simmim_moco": "pretrain_simmim_moco_ori", "simmim": "pretrain_simmim_ori", "moco": "pretrain_moco_ori"}
KIND_TEXT = {"simmim": "Gen-only model (use_moco_target=False): encoder + final LayerNorm + SimMIM decoder, MIM loss on the original view",
             "moco": "Dis-only model (use_pixel_target=False): MoCo-v3 on two unmasked views, no pix_projector, no decoder"}


def flop_per_sample(model, kind):
    if kind == "simmim_moco":
        return FLOP_PER_SAMPLE[model]
    f = ENC_FWD_FLOP[model]
    if f is None:
        return None
    return 3 * f + 0.129e9 if kind == "simmim" else 8 * f + 1.39e9
WORKLOAD_TEXT = {
    "mim_moco": "BASELINE configs[2]'s step on this many GPUs: SimMIM (w=1.0) + MoCo-v3 (w=0.1)",
    "mim_only": "BASELINE configs[1]: loss_weight_contrast=0 (the contrastive forward still runs for the loss_contrast / accuracy meters; "
                "its backward and the augmented view's encoder backward are exact zeros in the reference and are not launched)",
}
PEAK_BF16 = 2.5e15
PEAK_HBM = 8.0e12                                                      # /opt/skills/guides/MI355X_MICROARCH.md
PMC_FILE = "r06_pmc_traffic.json"                                      # rocprofv3 --pmc passes of this command (tools/collect_profiles.sh), stamped
#                                                                        with the hash of the kernel SOURCES they were measured on


class FreshMaskLoader:
    """`n` steps over a small set of resident image batches; every step gets a FRESH mask from the device
    generator of dig_amd.datasets (Philox4x32-10, exactly 179 of 256 per view -- the engine zeroes view 1's mask as the reference does)."""

    def __init__(self, batches, n, gen):
        self.batches, self.n, self.gen = batches, n, gen

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            (im, au, _), a, b = self.batches[i % len(self.batches)]
            yield ([im, au, self.gen(im.shape[0]).to(torch.float64)], a, b)


def synth_batches(n, B, device, seed):
    """U(-1,1) crops from Generator(seed), RandomMaskingGenerator((8,32), 0.7, num_view=2) masks."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        im = (torch.rand((B, 3, 32, 128), generator=g) * 2 - 1).to(device)
        au = (torch.rand((B, 3, 32, 128), generator=g) * 2 - 1).to(device)
        mk = np.zeros((B, 2, 256), dtype=np.float64)
        for b in range(B):
            for v in range(2):
                row = np.hstack([np.zeros(256 - 179), np.ones(179)])
                rng.shuffle(row)
                mk[b, v] = row
        out.append(([im, au, torch.from_numpy(mk).to(device)], torch.ones(1), torch.ones(1)))
    return out


class GemmProbe:
    """Times every matrix-core launch of a step -- dig_gemm_bf16, the fused attention sub-block (dig_attn_block_fwd), the attention kernels
    (dig_attn_fwd / _bwd), the fused MLP launches (dig_mlp_chain_fwd / _fwd_ln / _bwd) and the grouped weight-gradient launch
    (dig_wgrad_group) -- with the library's launch probe (dig_probe_start / dig_probe_stop, csrc/probe.hip): each
    launch carries the start / stop events of hipExtLaunchKernel ON ITS LAUNCH STREAM, i.e. the kernel's own begin and end on the device,
    which is what rocprofv3's kernel trace reports.  With the two-stream overlap left ON these are the durations IN THE STEP; with
    `model.overlap_streams = False` the durations of the kernels one at a time.  The wrappers below only book what each launch computes
    (family, algorithmic FLOP and bytes), in call order -- the order in which the probe returns the durations."""

    def __init__(self):
        from dig_amd import ops
        self.ops = ops
        self.rec = []
        self._saved = {}

    def _bracket(self, variant, flops, byt, fn):
        self.rec.append((variant, flops, byt))
        return fn()

    def __enter__(self):
        ops = self.ops
        sv = self._saved
        ops.L.call("dig_probe_start")
        sv["gemm"], sv["chain"], sv["chain_ln"], sv["chain_bwd"], sv["wg_launch"] = (ops.gemm, ops.mlp_chain_fwd, ops.mlp_chain_fwd_ln,
                                                                                      ops.mlp_chain_bwd, ops.WgradGroup.launch)

        def timed(A, B, I, J, R, **kw):
            # family = operand form + tile code, i.e. one kernel template as rocprofv3 lists it (fwd:544 = gemm_pwide_kernel<4, ...>,
            # dgrad:264 = gemm_wide_kernel<false, true, 0, 4, 3, ...>, ...; tools/pmc_traffic.py maps kernel names to the same keys)
            variant = ("wgrad" if kw.get("ta") else ("dgrad" if kw.get("tb") else "fwd")) + ":" + str(kw.get("bk") or (64 if not (kw.get("ta") or kw.get("tb")) else 32))
            # algorithmic HBM bytes of the launch: both operands once, the output and the residual once (weight gradients: fp32
            # read-modify-write of dW).  NOT counted: the saved pre-activation of fc1 (an implementation choice of the backward),
            # the split-R slabs
            byt = 2.0 * I * R + 2.0 * J * R
            if kw.get("ta"):
                byt += 8.0 * I * J
            else:
                byt += I * J * (4.0 if kw.get("out_kind") == ops.OUT_F32 else 2.0)
                byt += 2.0 * I * J * (kw.get("resid") is not None)
            return self._bracket(variant, 2.0 * I * J * R, byt, lambda: sv["gemm"](A, B, I, J, R, **kw))
        ops.gemm = timed

        def timed_chain(x, w1, b1, w2, b2, resid, save=False):
            R, D = x.shape
            Fh = w1.shape[0]
            # algorithmic HBM bytes of the fused fc1 -> GELU -> fc2 launch: x, residual and output rows once, both weight matrices once;
            # the online form also writes what the reference's autograd keeps for the backward (pre-activation and GELU output)
            byt = 2.0 * R * D * 3 + 2.0 * 2 * D * Fh + (2.0 * 2 * R * Fh if save else 0.0)
            return self._bracket("mlp_chain_online" if save else "mlp_chain_momentum", 4.0 * R * D * Fh, byt, lambda: sv["chain"](x, w1, b1, w2, b2, resid, save=save))
        ops.mlp_chain_fwd = timed_chain

        def timed_chain_ln(x, ln_g, ln_b, eps, w1, b1, w2, b2, nln_g=None, nln_b=None, save=False, resid=None, drop=None):
            R, D = x.shape
            Fh = w1.shape[0]
            # the fused launch with its LayerNorms: raw rows in, output rows and the next block's normalised rows out, both weight matrices
            # once; the online form also writes what the reference's autograd keeps (norm2's output, the pre-activation, the GELU output)
            byt = 2.0 * R * D * (2 + (nln_g is not None)) + 2.0 * 2 * D * Fh + ((2.0 * R * D + 2.0 * 2 * R * Fh) if save else 0.0)
            return self._bracket("mlp_chain_online" if save else "mlp_chain_momentum", 4.0 * R * D * Fh, byt,
                                 lambda: sv["chain_ln"](x, ln_g, ln_b, eps, w1, b1, w2, b2, nln_g, nln_b, save=save, resid=resid, drop=drop))
        ops.mlp_chain_fwd_ln = timed_chain_ln

        def timed_chain_bwd(dy, w2t, pre, w1t, colsum=True, out=None):
            R, D = dy.shape
            Fh = w2t.shape[0]
            # fused data gradient of the MLP: dy and dx rows, the saved pre-activation in, d(pre-activation) out (the fc1 weight gradient
            # reads it), both weight matrices once
            byt = 2.0 * R * D * 2 + 2.0 * R * Fh * 2 + 2.0 * 2 * D * Fh
            return self._bracket("mlp_chain_bwd", 4.0 * R * D * Fh, byt, lambda: sv["chain_bwd"](dy, w2t, pre, w1t, colsum=colsum, out=out))
        ops.mlp_chain_bwd = timed_chain_bwd
        sv["chain_bwd_ln"] = ops.mlp_chain_bwd_ln

        def timed_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, ln_g, ln_mean, ln_rstd, colsum=True, out=None, projt=None, dctx=None):
            R, D = dy.shape
            Fh = w2t.shape[0]
            pj = projt is not None
            byt = 2.0 * R * D * (3 + pj) + 2.0 * R * Fh * 2 + 2.0 * 2 * D * Fh + 2.0 * D * D * pj      # + the rows norm2 normalised (, dctx, Wproj)
            return self._bracket("mlp_chain_bwd", 4.0 * R * D * Fh + 2.0 * R * D * D * pj, byt,
                                 lambda: sv["chain_bwd_ln"](dy, w2t, pre, w1t, x_mid, ln_g, ln_mean, ln_rstd, colsum=colsum, out=out, projt=projt,
                                                            dctx=dctx))
        ops.mlp_chain_bwd_ln = timed_chain_bwd_ln

        def attn_block_rec(R, D, save, n_img, heads):
            # the fused attention sub-block: qkv Linear + scores + context + proj Linear; algorithmic bytes = ln1, x and x_mid rows, both weight
            # matrices once; the online form also writes what the backward reads (qkv, ctx, lse)
            fl = 2.0 * R * D * 3 * D + 4.0 * R * 256 * D + 2.0 * R * D * D
            byt = 2.0 * R * D * 3 + 2.0 * 4 * D * D + ((2.0 * R * 3 * D + 2.0 * R * D + 4.0 * n_img * heads * 256) if save else 0.0)
            return ("attn_block_online" if save else "attn_block_momentum", fl, byt)
        sv["attn_block"] = ops.attn_block_fwd

        def attn_rec(bwd, R, D):
            # attention proper, per (image, head): scores + context (forward: 2 matrix products of 2 x 256 x 256 x 64), backward: scores again,
            # dP, dV, dK, dQ (5); bytes: qkv (+ ctx, dctx, lse) in, ctx / dqkv out
            return ("attn_bwd" if bwd else "attn_fwd", (10.0 if bwd else 4.0) * R * 256 * D, 2.0 * R * D * (9 if bwd else 4))
        sv["attn_fwd"], sv["attn_bwd"] = ops.attn_fwd, ops.attn_bwd
        sv["attn_bwd_proj"] = ops.attn_bwd_proj

        def attn_proj_rec(R, D):
            # the attention backward with the projection's data gradient in the launch: + 2 R D D FLOP; dy in instead of d(ctx) in, + the weight
            f, b = attn_rec(True, R, D)[1:]
            return ("attn_bwd_proj", f + 2.0 * R * D * D, b + 2.0 * D * D)

        def timed_attn_bwd_proj(qkv, ctx, dy, projt, lse, n_img, heads, D, scale, bias_sums=False):
            self.rec.append(attn_proj_rec(qkv.shape[0], D))
            return sv["attn_bwd_proj"](qkv, ctx, dy, projt, lse, n_img, heads, D, scale, bias_sums=bias_sums)
        ops.attn_bwd_proj = timed_attn_bwd_proj

        def timed_attn_fwd(qkv, n_img, heads, D, drop=None, q_rows=256):
            self.rec.append(attn_rec(False, qkv.shape[0], D))
            return sv["attn_fwd"](qkv, n_img, heads, D, drop=drop, q_rows=q_rows)

        def timed_attn_bwd(qkv, ctx, dctx, lse, n_img, heads, D, scale, bias_sums=False, drop=None, q_rows=256):
            self.rec.append(attn_rec(True, qkv.shape[0], D))
            return sv["attn_bwd"](qkv, ctx, dctx, lse, n_img, heads, D, scale, bias_sums=bias_sums, drop=drop, q_rows=q_rows)
        ops.attn_fwd, ops.attn_bwd = timed_attn_fwd, timed_attn_bwd

        def timed_attn_block(ln1, x, qkv_w, qkv_b, proj_w, proj_b, n_img, heads, D, scale, save=False):
            self.rec.append(attn_block_rec(ln1.shape[0], D, save, n_img, heads))
            return sv["attn_block"](ln1, x, qkv_w, qkv_b, proj_w, proj_b, n_img, heads, D, scale, save=save)
        ops.attn_block_fwd = timed_attn_block

        probe = self

        def timed_wg_launch(grp):
            if not grp.cur:
                if grp.pending is None:
                    return sv["wg_launch"](grp)
                return sv["wg_launch"](grp)                   # the fold-only launch that ends a backward: booked by the call hook below
            R = grp.rows
            fl = sum(2.0 * R * dw.shape[0] * dw.shape[1] for _, _, dw, _ in grp.cur)
            # grouped weight gradients: both operands of every problem once, fp32 read-modify-write of each dW (the slabs are not algorithmic)
            byt = sum(2.0 * R * (dw.shape[0] + dw.shape[1]) + 8.0 * dw.shape[0] * dw.shape[1] for _, _, dw, _ in grp.cur)
            probe._in_wg = True                               # (the launch goes through ops.L.call: the call hook below must not book it again)
            try:
                return probe._bracket("wgrad_group", fl, byt, lambda: sv["wg_launch"](grp))
            finally:
                probe._in_wg = False
        ops.WgradGroup.launch = timed_wg_launch

        # the one-call-per-encoder-block path (dig_encoder_block_fwd / _bwd): the same launches, issued inside the library -- booked here in
        # the order the call issues them (csrc/encoder_block.inc), with the same family / FLOP / byte rules as the wrappers above
        sv["call"] = ops.L.call

        def gemm_rec(form, tile, I, J, R, resid=False):
            byt = 2.0 * I * R + 2.0 * J * R + 2.0 * I * J + 2.0 * I * J * resid
            return (form + ":" + str(tile), 2.0 * I * J * R, byt)

        def call(name, *args):
            if name == "dig_encoder_block_fwd":
                b = args[0]._obj
                R, D, Fh = b.rows, b.D, b.F
                if b.fuse_attn and ops.attn_block_supported(b.heads, D):
                    self.rec.append(attn_block_rec(R, D, bool(b.save), b.n_img, b.heads))
                else:
                    self.rec.append(gemm_rec("fwd", b.tile_qkv, R, 3 * D, D))
                    self.rec.append(attn_rec(False, R, D))
                    self.rec.append(gemm_rec("fwd", b.tile_proj, R, D, D, resid=True))
                byt = 2.0 * R * D * (2 + bool(b.next_n1_g)) + 2.0 * 2 * D * Fh + ((2.0 * R * D + 2.0 * 2 * R * Fh) if b.save else 0.0)
                self.rec.append(("mlp_chain_online" if b.save else "mlp_chain_momentum", 4.0 * R * D * Fh, byt))
            elif name == "dig_encoder_block_bwd":
                b = args[0]._obj
                R, D, Fh = b.rows, b.D, b.F
                # (fuse_ln2: norm2's backward rides in the launch -- one more [R, D] operand, x_mid; dy and the output are the chain's own)
                proj_in = bool(b.fuse_ln2 and b.projt)                        # ... and the projection's data gradient: its FLOP, dctx out, the weight
                self.rec.append(("mlp_chain_bwd", 4.0 * R * D * Fh + 2.0 * R * D * D * proj_in,
                                 2.0 * R * D * ((3 if b.fuse_ln2 else 2) + proj_in) + 2.0 * R * Fh * 2 + 2.0 * 2 * D * Fh + 2.0 * D * D * proj_in))
                proj_attn = bool(not proj_in and b.attn_proj and b.proj_wt and ops.attn_bwd_proj_supported(D))   # (csrc/encoder_block.inc)
                if not proj_in and not proj_attn:
                    self.rec.append(gemm_rec("dgrad", b.tile_dgrad, R, D, D))
                self.rec.append(attn_proj_rec(R, D) if proj_attn else attn_rec(True, R, D))
                shapes = ((D, Fh), (Fh, D), (D, D), (3 * D, D))
                if not b.wg_defer:                                            # (deferred plan: the grouped launch is issued by the caller, booked below)
                    self.rec.append(("wgrad_group", sum(2.0 * R * o * i for o, i in shapes), sum(2.0 * R * (o + i) + 8.0 * o * i for o, i in shapes)))
                self.rec.append(gemm_rec("dgrad", b.tile_dgrad, R, D, 3 * D))
            elif name == "dig_wgrad_group" and args[1] == 0:
                self.rec.append(("wgrad_fold", 0.0, 0.0))                     # the fold-only launch that ends a backward
            elif name == "dig_wgrad_group" and not getattr(self, "_in_wg", False):   # a grouped launch issued directly (the deferred plan): its problem table
                pr = (ops._WgProb * int(args[1])).from_address(int(args[0]))
                R = int(args[4])
                self.rec.append(("wgrad_group", sum(2.0 * R * q.I * q.J for q in pr), sum(2.0 * R * (q.I + q.J) + 8.0 * q.I * q.J for q in pr)))
            return sv["call"](name, *args)
        ops.L.call = call
        return self

    def __exit__(self, *a):
        import ctypes
        ops, sv = self.ops, self._saved
        ops.gemm, ops.mlp_chain_fwd, ops.mlp_chain_fwd_ln, ops.mlp_chain_bwd = sv["gemm"], sv["chain"], sv["chain_ln"], sv["chain_bwd"]
        ops.mlp_chain_bwd_ln = sv["chain_bwd_ln"]
        ops.WgradGroup.launch = sv["wg_launch"]
        ops.attn_block_fwd = sv["attn_block"]
        ops.attn_fwd, ops.attn_bwd = sv["attn_fwd"], sv["attn_bwd"]
        ops.attn_bwd_proj = sv["attn_bwd_proj"]
        ops.L.call = sv["call"]
        torch.cuda.synchronize()
        cap = len(self.rec) + 64
        buf = (ctypes.c_float * cap)()
        lib = ops.L.lib()
        lib.dig_probe_stop.restype = ctypes.c_int
        n = lib.dig_probe_stop(buf, cap)
        # the fold-only launch of the grouped weight gradients (no problems) is a probed launch the wrappers do not book: it is the one
        # dig_wgrad_group call per backward whose Python-side record is missing -- booked here as ("wgrad_fold", 0 FLOP)
        if n != len(self.rec):
            raise RuntimeError(f"launch probe: {n} probed launches, {len(self.rec)} booked")
        self.us = [float(buf[i]) for i in range(n)]

    def summary(self):
        agg = {}
        for (variant, fl, byt), us in zip(self.rec, self.us):
            d = agg.setdefault(variant, [0.0, 0.0, 0, 0.0])
            d[0] += fl
            d[1] += us * 1e-6
            d[2] += 1
            d[3] += byt
        return {k: {"flops": v[0], "seconds": v[1], "launches": v[2], "bytes": v[3]} for k, v in agg.items()}


KERNEL_TEXT = {
    "mlp_chain_online": "dig_mlp_chain_fwd_ln, online form (mlp_chain_kernel<1, true>: norm2 -> fc1 -> GELU -> fc2 + residual -> next norm1 in one launch, writes what the backward reads)",
    "mlp_chain_momentum": "dig_mlp_chain_fwd_ln, momentum form (mlp_chain_kernel<0, true>: norm2 -> fc1 -> GELU -> fc2 + residual -> next norm1 in one launch, no side outputs)",
    "mlp_chain_bwd": "dig_mlp_chain_bwd_ln (mlp_chain_kernel<2>: the MLP's two data gradients x GELU' and norm2's backward in one launch)",
    "attn_bwd": "dig_attn_bwd (attn_bwd_kernel: dq, dk, dv of the softmax attention given d(ctx), one workgroup per (image, head), + the q / v bias sums)",
    "attn_bwd_proj": "dig_attn_bwd_proj (attn_bwd_kernel<.., PROJ>: the same launch with the projection's data gradient in front -- d(ctx) = dx_mid Wproj computed per "
                     "workgroup, never written)",
    "attn_fwd": "dig_attn_fwd (attn_fwd_kernel: softmax(q k^T) v, one workgroup per (image, head))",
    "attn_block_online": "dig_attn_block_fwd, online form (attn_block_kernel<true>: qkv Linear -> softmax(q k^T) v -> proj Linear + residual in one launch, one "
                         "workgroup per image; writes qkv, ctx, lse for the backward)",
    "attn_block_momentum": "dig_attn_block_fwd, momentum form (attn_block_kernel<false>: the same launch with nothing kept)",
    "wgrad_group": "dig_wgrad_group (wgrad_wide_kernel / wgrad_group_kernel: the four weight gradients of a block in one launch, 4 x 3 MFMA blocks per wave, "
                   "LDS-DMA ring, slabs folded by the next launch)",
}


def cpu_baseline(model_name, budget_s=20.0):
    """The fp32 CPU oracle (oracle/dig_oracle.py, pinned to the reference by tests/golden) on this host's cores, at the two batch sizes
    SURVEY.md 8(d) names: B = 4 (BASELINE configs[0], the reference's own CPU-runnable case) and B = 128 (the per-GPU batch of the
    metric: one untimed + three timed steps of ~30 s each -- the default run stays within a few minutes)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dig_oracle as O
    cfg = O.make_config(model_name)
    # thread count: measured in this run.  One step at batch 16 per candidate (after an untimed one), the best rate wins -- on the MI355X host
    # (2 x EPYC 9575F, 256 hardware threads) that has been 16 threads (tools/cpu_baseline_threads.py: 6.5 / 8.5 / 7.4 / 3.2 / 1.6 samples/s
    # at 8 / 16 / 32 / 64 / 128): torch's default of 128 threads spends its time in fork/join.  ~10 s of the budget.
    tr = O.OracleTrainer(cfg, seed=0)
    ncpu = os.cpu_count() or 8
    sweep = {}
    if budget_s >= 10:
        im_s, au_s, mk_s = O.synthetic_batch(16, cfg, 4321)
        hp_s = O.StepHyper(lr=1.5e-4 * 16 / 256)
        for nt in (8, 16, 32, 64):
            if nt > ncpu:
                break
            torch.set_num_threads(nt)
            tr.step(im_s, au_s, mk_s, hp_s)
            t0 = time.perf_counter()
            tr.step(im_s, au_s, mk_s, hp_s)
            sweep[nt] = 16 / (time.perf_counter() - t0)
    best_nt = max(sweep, key=sweep.get) if sweep else max(1, min(16, ncpu))
    torch.set_num_threads(best_nt)

    def timed(Bc, max_steps, budget):
        im, au, mk = O.synthetic_batch(Bc, cfg, 1234)
        hp = O.StepHyper(lr=1.5e-4 * Bc / 256)
        n, t0 = 0, time.perf_counter()
        while True:
            tr.step(im, au, mk, hp)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget or n >= max_steps:
                return n * Bc / dt, n, dt
    timed(4, 1, 0.0)                                                     # warm-up (thread pool, allocator)
    v4, n4, t4 = timed(4, 6, budget_s * 0.25)
    if budget_s >= 10:
        timed(128, 1, 0.0)                                               # one untimed step at the measured batch (first-touch allocation of the
        v128, n128, t128 = timed(128, 3, 1e9)                            #  32x larger tensors, thread-pool growth), then three timed ones
    else:
        v128, n128, t128 = None, 0, 0.0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": v128 if v128 is not None else v4, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            # the UNMODIFIED reference (engine_for_pretraining_moco.train_one_epoch + MoCo_ViT, fp32, gloo world 1,
            # the same synthetic generator) timed where it can be imported -- the build container, not this host; reported beside the port
            "reference_build_container": {"value": 2.24, "unit": "images/sec", "cores": 8, "batch": 128, "steps": 2, "seconds_per_step": 57.2,
                                          "batch_4": {"value": 2.0, "seconds_per_step": 2.02, "steps": 3},
                                          "host_cpu": "8 vCPU Intel Xeon (family 6 model 207, 2.1 GHz, AVX-512), torch 2.10.0 CPU, 8 threads",
                                          "source": "BASELINE.md section 2 (survey-time run of the reference through oracle/ref_harness)"},
            "host_cpu": cpu_model, "host_logical_cpus": os.cpu_count(),
            "batch_128": {"value": v128, "steps": n128, "seconds": t128}, "batch_4": {"value": v4, "steps": n4, "seconds": t4},
            "thread_sweep_batch16": {str(k): round(v, 2) for k, v in sweep.items()},
            "threads_note": f"{best_nt} torch threads: the best of the in-run sweep (`thread_sweep_batch16`, samples/s of one warm step at batch 16 per "
                            "candidate); all 256 logical CPUs are slower, torch's fork/join dominates (tools/cpu_baseline_threads.py)",
            "sample": f"fp32 torch CPU restatement of the reference step (oracle/dig_oracle.py), same model/recipe: {n128} warm timed steps at batch 128 "
                      f"(`value`; one untimed step at that batch first), {n4} steps at batch 4 (BASELINE configs[0])"}


def _self_launch(n):
    """Re-executes this command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on 127.0.0.1 with a free port.  With
    DIG_SHARE_GPU=1 (test harness: every rank on device 0) the backend defaults to gloo -- RCCL refuses two ranks on one device."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("DIG_SHARE_GPU") == "1":
        env.setdefault("DIG_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    if r.returncode:
        raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # >= 100 timed steps after >= 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="samples per GPU")
    ap.add_argument("--model", default="small", choices=["tiny", "small", "base"])
    ap.add_argument("--model-kind", default="simmim_moco", choices=["simmim_moco", "simmim", "moco"],
                    help="simmim_moco = the headline model (pretrain_simmim_moco_ori_*); simmim / moco = the reference's single-objective "
                         "factories (pretrain_simmim_ori_* Gen-only, pretrain_moco_ori_* Dis-only): their own JSON line, not the headline")
    ap.add_argument("--num-windows", type=int, default=4, help="README: 4; the reference's argparse default is 5 (uneven pooling windows)")
    ap.add_argument("--patchnet-name", default="no_patchtrans", choices=["no_patchtrans", "regular", "conv"],
                    help="README: no_patchtrans; the reference's argparse default is regular (PatchNet with its 2-block patch transformer)")
    ap.add_argument("--drop-path", type=float, default=0.0, help="stochastic depth rate (--drop_path of the reference driver; README: 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mim-only", action="store_true", help="skip the extra measurement of the other workload")
    ap.add_argument("--workload", default="mim_moco", choices=["mim_moco", "mim_only"],
                    help="headline workload: mim_moco = BASELINE configs[2]'s step (SimMIM + MoCo-v3, the default at every N so that the "
                         "N=1/2/4/8 series is one workload); mim_only = configs[1] (loss_weight_contrast=0).  The other one is reported "
                         "in the same JSON line under `mim_only` / `mim_moco`.")
    ap.add_argument("--no-step-graph", action="store_true", help="skip the extra measurement of the HIP-graph replay of the step")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    a = ap.parse_args()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL -- the reference's own launch line is
        # `torch.distributed.launch --nproc_per_node=8`, README.md:53), rank 0's single JSON line comes back on our stdout
        return _self_launch(a.gpus)

    import dig_amd.utils as U
    from dig_amd.registry import create_model
    from dig_amd.optim_factory import create_optimizer
    from dig_amd.engine_for_pretraining_moco import train_one_epoch
    from dig_amd.parallel import DistributedDataParallel

    # logs (ours, and banners native libraries such as RCCL write to fd 1) -> stderr; stdout carries the single JSON line
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
    model_name = f"{KIND_FACTORY[a.model_kind]}_vit_{a.model}_patch4_32x128"
    if a.model_kind != "simmim_moco":
        a.no_mim_only = True                          # (one objective: there is no "other workload" of the same model)
    model = create_model(model_name, pretrained=False, drop_path_rate=a.drop_path, drop_block_rate=None, mlp_dim=4096, dim=256, T=0.2,
                         num_windows=a.num_windows, encoder_type='vit', queue_size=65536, patchnet_name=a.patchnet_name)
    readme = a.num_windows == 4 and a.patchnet_name == "no_patchtrans" and a.drop_path == 0.0
    if not readme:
        a.no_cpu_baseline = True                      # (the CPU port is timed on the README recipe)
    model.to(dev)
    force_dist = world > 1 or (os.environ.get("DIG_FORCE_DIST") == "1" and torch.distributed.is_initialized())
    run_model = DistributedDataParallel(model) if force_dist else model
    B = a.batch
    W_MAIN, W_OTHER = (0.1, 0.0) if a.workload == "mim_moco" else (0.0, 0.1)
    args = types.SimpleNamespace(num_view=2, moco_m=0.99, use_moco_m_cos=1, epochs=10, contrast_start_epoch=0, contrast_warmup_steps=0,
                                 loss_weight_contrast=W_MAIN, loss_weight_pixel=1.0, only_mim_on_ori_img=True, eval_freq=500, opt='adamw',
                                 lr=1.5e-4 * B * world / 256, weight_decay=0.1, opt_eps=1e-8, opt_betas=[0.9, 0.999])
    opt = create_optimizer(args, model)
    scaler = U.NativeScalerWithGradNormCount()
    total = a.warmup + 3 * a.steps + 24
    lr_s, wd_s = np.full(total + 8, args.lr), np.full(total + 8, 0.1)
    batches = synth_batches(4, B, dev, 1234 + rank)

    from dig_amd.datasets import RandomMaskingGenerator
    mask_gen = RandomMaskingGenerator((8, 32), 0.7, num_view=2, seed=1234 + rank, device=dev)

    def run(n, start):
        loader = FreshMaskLoader(batches, n, mask_gen)
        return train_one_epoch(run_model, None, None, loader, None, opt, dev, 0, scaler, None, patch_size=4, normlize_target=False,
                               start_steps=start, lr_schedule_values=lr_s, wd_schedule_values=wd_s, args=args)

    if rank != 0:
        sys.stdout = open(os.devnull, "w")
    if a.warmup > 0:
        run(a.warmup, 0)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    model._host_launch = (0.0, 0)
    t0 = time.perf_counter()
    stats = run(a.steps, a.warmup)
    torch.cuda.synchronize()
    host_ms = model._host_launch[0] / max(model._host_launch[1], 1) * 1e3      # host time spent queueing a step (outside the GPU's critical path)
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # ---- the other BASELINE workload on the same model state: reported beside the headline
    mim_only = None
    if not a.no_mim_only:
        args.loss_weight_contrast = W_OTHER
        run(2, a.warmup + a.steps)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(a.steps, a.warmup + a.steps + 2)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt1], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt1 = float(t.item())
        args.loss_weight_contrast = W_MAIN
        mim_only = {"workload": WORKLOAD_TEXT["mim_only" if a.workload == "mim_moco" else "mim_moco"],
                    "value": a.steps * B * world / dt1, "unit": "images/sec", "ms_per_step": dt1 / a.steps * 1e3, "steps": a.steps}
    # ---- the same workload replayed from the captured HIP graph (dig_amd/step_graph.py; single process only): reported beside the headline
    graphed = None
    if world == 1 and not force_dist and not a.no_step_graph:
        pos = a.warmup + a.steps + (0 if a.no_mim_only else a.steps + 2)
        model.step_graph = True
        run(6, pos)                                       # eager warm-up of the captured body + capture
        torch.cuda.synchronize()
        model._host_launch = (0.0, 0)
        t2 = time.perf_counter()
        run(a.steps, pos + 6)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        graphed = {"ms_per_step": dt2 / a.steps * 1e3, "value": a.steps * B / dt2, "unit": "images/sec",
                   "host_ms_per_step": model._host_launch[0] / max(model._host_launch[1], 1) * 1e3,
                   "replays": model._step_graph.replays, "note": "opt-in (DIG_STEP_GRAPH=1): same launches, bit-identical results, one "
                   "hipGraphLaunch per step; ROCm's graph executor runs the two-branch graph slower than the two eager streams"}
        model.step_graph = False
    # ---- one-rank RCCL path (DIG_FORCE_DIST=1 under torch.distributed.run with one rank): the collectives of a step, counted on an extra step
    dist_info = None
    if force_dist and rank == 0 and getattr(model, "comm", None) is not None and hasattr(model.comm, "log"):
        model.comm.log = []
        run(1, a.warmup + a.steps)
        torch.cuda.synchronize()
        lg, model.comm.log = model.comm.log, None
        kinds = {}
        for op, n in lg:
            k = op.split(":")[0]
            kinds.setdefault(k, [0, 0])
            kinds[k][0] += 1
            kinds[k][1] += n
        dist_info = {"world": world, "collectives_per_step": len(lg),
                     "by_kind": {k: {"count": v[0], "elements": v[1]} for k, v in kinds.items()},
                     "note": "every collective DistComm issues in one step (all-reduce of the BatchNorm statistics, fused key all-gather, 17 gradient buckets)"}
    elif force_dist:
        run(1, a.warmup + a.steps)                    # (every rank runs the same extra step: it contains collectives)
    # ---- roofline of the dominant kernel family: instrumented extra steps (outside the timed region)
    # (EVERY rank runs these steps -- they contain the step's collectives; only rank 0 instruments its launches)
    # `roofline.frac` = the family's algorithmic FLOP / its launch durations IN THE STEP (both streams running, as in the timed region and
    # under rocprofv3); the same launches one at a time (stream overlap off) are reported beside it as `roofline.alone`.
    roof = None
    import contextlib
    model.step_graph = False                          # (launched eagerly: a replayed graph has no per-launch brackets)
    pos = a.warmup + a.steps
    run(1, pos)
    probe_in = GemmProbe() if rank == 0 else contextlib.nullcontext()
    with probe_in:
        run(2, pos + 1)
    model.overlap_streams = False                     # kernels one at a time
    run(1, pos + 3)                                   # one plain step in this mode first: the caller's stream pool gets the blocks the
    probe = GemmProbe() if rank == 0 else contextlib.nullcontext()   # high-priority stream's pool held, so no hipMalloc sits inside a bracket
    with probe:
        run(2, pos + 4)
    model.overlap_streams = True
    if rank == 0:
        summ_in, summ = probe_in.summary(), probe.summary()
        # The dominant family = the kernel template instantiation (as rocprofv3 lists them) with the most kernel time per step, measured one
        # kernel at a time -- the serialisation rocprofv3's kernel trace applies, so this is the top row of its summary -- with no tie-break on
        # the work a family carries: the kernel that OWNS the step, whatever its efficiency.  `frac` is that family's FLOP over its IN-STEP
        # durations (both streams running, as in the timed region).  Beside it: `best` = the most efficient family among those with >= 5 % of
        # the matrix-core kernel time, `weighted` = all matrix-core families together (sum of FLOP / sum of in-step durations).
        cand = [k for k in summ if summ[k]["flops"] > 0]
        dom = max(cand, key=lambda k: summ[k]["seconds"])
        d, da = summ_in.get(dom, summ[dom]), summ[dom]
        tf = d["flops"] / d["seconds"] / 1e12
        gbs = d["bytes"] / d["seconds"] / 1e9
        # HBM bytes per launch: rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE) over this same command,
        # collected with tools/collect_profiles.sh and committed with the hash of the kernel sources they were measured on; reported only
        # while that hash still matches the sources this run was built from
        traffic, traffic_src, step_bytes = None, None, None
        try:
            from dig_amd import build as dig_build
            src_hash = dig_build.source_hash()
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
                tj = json.load(f)
            if tj.get("src_sha256_16") == src_hash:
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
                if tj.get("steps_in_run"):                      # all kernel families of the step: measured HBM bytes per step
                    step_bytes = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in tj["kernels"].values()) / tj["steps_in_run"]
                traffic_src = f"profiles/{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, kernel sources {src_hash})"
            else:
                traffic_src = f"profiles/{PMC_FILE} was collected on kernel sources {tj.get('src_sha256_16')}, this tree has {src_hash}: not reported"
        except Exception as e:  # noqa: BLE001
            traffic_src = f"profiles/{PMC_FILE}: {type(e).__name__}"
        # the bounding roofline of this path is bf16 MFMA; the HBM view of the same launches is kept beside it
        mfma_view = {"achieved": tf, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": tf / (PEAK_BF16 / 1e12)}
        hbm_view = {"achieved": gbs, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": gbs / (PEAK_HBM / 1e9)}
        tfa = da["flops"] / da["seconds"] / 1e12
        kname = KERNEL_TEXT.get(dom, f"dig_gemm_bf16[{dom}] (operand form : tile code of include/dig_hip.h; gemm_kernel / gemm_wide_kernel / gemm_pwide_kernel, "
                                     "v_mfma_f32_32x32x16_bf16, fused bias/GELU/residual epilogues)")

        def by_variant(sm):
            return {k: {"TFLOP/s": v["flops"] / max(v["seconds"], 1e-12) / 1e12, "GB/s": v["bytes"] / max(v["seconds"], 1e-12) / 1e9,
                        "ms_per_step": v["seconds"] / 2 * 1e3, "launches_per_step": v["launches"] // 2, "avg_launch_us": v["seconds"] / v["launches"] * 1e6,
                        "flops_per_launch": v["flops"] / v["launches"]}
                    for k, v in sm.items()}
        tot_s = sum(summ_in[k]["seconds"] for k in summ_in if summ_in[k]["flops"] > 0)
        tot_f = sum(summ_in[k]["flops"] for k in summ_in)
        big = [k for k in summ_in if summ_in[k]["flops"] > 0 and summ_in[k]["seconds"] >= 0.05 * tot_s]
        best = max(big, key=lambda k: summ_in[k]["flops"] / summ_in[k]["seconds"])
        best_tf = summ_in[best]["flops"] / summ_in[best]["seconds"] / 1e12
        tot_sa = sum(summ[k]["seconds"] for k in summ if summ[k]["flops"] > 0)
        roof = {"bound": "mfma", **mfma_view, "traffic": traffic,
                "best": {"family": best, "achieved": best_tf, "unit": "TFLOP/s", "frac": best_tf / (PEAK_BF16 / 1e12),
                         "note": "the most efficient family with >= 5 % of the matrix-core kernel time (in the step)"},
                "weighted": {"achieved": tot_f / tot_s / 1e12, "unit": "TFLOP/s", "frac": tot_f / tot_s / PEAK_BF16,
                             "alone_frac": tot_f / tot_sa / PEAK_BF16, "ms_per_step": tot_s / 2 * 1e3,
                             "note": "all matrix-core launches of the step: sum of FLOP / sum of in-step durations (alone_frac: one kernel at a time)"},
                "kernel": kname, "family": dom, "measured": "start / stop events of hipExtLaunchKernel on every launch of the family, on its launch stream (dig_probe_start / _stop), in two extra "
                            "steps with both streams running: the kernels' device-side durations in the step, as rocprofv3's kernel trace reports them",
                "traffic_source": traffic_src,
                "hbm": hbm_view, "step_hbm_bytes": step_bytes,
                "flops_per_launch": d["flops"] / d["launches"], "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                "avg_launch_us": d["seconds"] / d["launches"] * 1e6, "launches_per_step": d["launches"] // 2,
                "alone": {"achieved": tfa, "unit": "TFLOP/s", "frac": tfa / (PEAK_BF16 / 1e12), "avg_launch_us": da["seconds"] / da["launches"] * 1e6,
                          "note": "the same launches with the stream overlap off (one kernel at a time)"},
                "by_variant": by_variant(summ_in), "by_variant_alone": by_variant(summ)}
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    try:                                   # RCCL prints its banner through C stdio: its buffer must drain while fd 1 still points at stderr,
        import ctypes                      # or the banner lands on the real stdout behind the JSON line at exit
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(saved_fd1, 1)
    sys.stdout = stdout
    if rank != 0:
        return
    value = a.steps * B * world / dt
    fl = flop_per_sample(a.model, a.model_kind)
    if a.model_kind != "simmim_moco":
        WORKLOAD_TEXT[a.workload] = KIND_TEXT[a.model_kind]
    line = {"metric": "pretrain images/sec (32x128, 2-view, mask 0.7) ViT-S/4", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{model_name}: full train_one_epoch step, {WORKLOAD_TEXT[a.workload]}; dim 256, mlp 4096, m 0.99 cos, "
                                   f"T 0.2, {a.num_windows} windows ({a.patchnet_name}), drop_path {a.drop_path}, mask 0.7, 2 views, AdamW wd 0.1, {B} samples/GPU, "
                                   "random-init weights",
                       "global_batch": B * world, "parallelism": f"dp{world}", "loss": float(stats.get("loss", float('nan')))},
            "step_mfma_frac": (value / world * fl / PEAK_BF16) if fl else None, "flop_per_sample": fl,
            # the whole step against the HBM roof: PMC-measured bytes of every kernel family per step (profiles/r05_pmc_traffic.json,
            # same kernel sources) / step time / 8 TB/s -- the roof that actually prices this model width
            "step_hbm_frac": (roof["step_hbm_bytes"] / (dt / a.steps) / PEAK_HBM) if roof and roof.get("step_hbm_bytes") and B == 128 and a.model == "small" and a.workload == "mim_moco" else None,
            "host_ms_per_step": host_ms, "step_graph": graphed, "dist": dist_info,
            "roofline": roof, ("mim_only" if a.workload == "mim_moco" else "mim_moco"): mim_only}
    if not a.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(model_name, a.cpu_budget)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
