"""Forward and hand-written backward of MoCo_ViT as a sequence of HIP kernel launches.

The math follows the reference step by step (citations: modeling_pretrain_moco_mim_ori.py:488-577 MoCo_ViT.forward,
modeling_pretrain_vit.py:89-106 encoder, modeling_finetune.py:87-158 attention/block/MLP, :444-461 InfoNCE); what is
new is the execution model: no autograd graph over ops, one custom autograd node for the whole model whose
backward is the explicit reverse sequence below, activations in bf16, gradients accumulated straight into the flat
fp32 gradient arena, per-stage callbacks so the data-parallel wrapper can start the RCCL all-reduce of a finished
parameter range while earlier layers are still in backward.
"""
import ctypes
import itertools
import os
import weakref
from typing import Optional, Tuple

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
FWD_MODE = os.environ.get("DIG_FWD_MODE", "flip")                   # two-stream plan of the forward (single process): "flip" (default) or "side"
CHAIN_BWD_EVERY = max(1, int(os.environ.get("DIG_CHAIN_BWD_EVERY", "1")))   # with DIG_MLP_CHAIN_MASK bit 2: the fused MLP backward in every k-th block only
CHAIN_BWD_PHASE = int(os.environ.get("DIG_CHAIN_BWD_PHASE", "0"))
BATCH_REDUCE = os.environ.get("DIG_BATCH_REDUCE", "0") == "1"       # an encoder block's eleven reduction launches as two: fewer launches,
#                                                                     0.1-0.15 ms SLOWER per step -> opt-in
FUSED_QV_BIAS_SUMS = os.environ.get("DIG_FUSED_QV_BIAS", "1") != "0"
# weight gradients of an encoder block on the grouped kernel (csrc/wgrad.hip): "block" = one launch per block (default: the block's
# accumulators are dumped once), "pair" = MLP pair | attention pair, "off" = the tiled split-R launches + their slab sums
WGRAD_GROUPING = os.environ.get("DIG_WGRAD_GROUPING", "block")
# The grouped weight-gradient launch owns the chip (one 8-wave workgroup per CU with every register and 140 KiB of LDS), and so do the fused
# MLP backward and the 256-row data-gradient tiles: beside it on a second stream the chain's next kernel only waits for CUs (its in-step
# duration grows by the launch's length, the step does not get shorter).  "1": the launch sits IN the data-gradient chain, on the caller's
# stream, between attention backward and the qkv data gradient -- back-to-back kernel boundaries instead of cross-stream events; the small
# reductions (bias / LayerNorm-parameter column sums) stay on the second stream.
WGRAD_INLINE = os.environ.get("DIG_WGRAD_INLINE", "1") == "1"
# Block-call path: all twelve grouped weight-gradient launches behind the LAST data gradient instead of inside each block's chain.  "auto"
# (default): deferred in a single process (nothing waits for a block's gradients before the optimizer: 19.96 -> 19.83 ms per step, A/B on one
# box), inline under a process group (a block's bucket must be final as early as possible: its all-reduce overlaps the rest of the backward).
# "1" / "0" force either plan.  Deferring keeps every block's gradient temporaries alive until the end of the backward (~0.45 GB per ViT-S block).
# "auto" also checks the device's free memory against what the deferred plan keeps alive (depth x the block's bf16 temporaries + the saved
# activations no longer released block by block: ~5.4 GB more at the end of the backward for ViT-S at B = 128) and stays inline when that
# would not leave a quarter of the free memory.
WGRAD_DEFER = os.environ.get("DIG_WGRAD_DEFER", "auto").strip().lower()
if WGRAD_DEFER not in ("auto", "0", "1"):
    raise ValueError(f"DIG_WGRAD_DEFER={WGRAD_DEFER!r}: one of auto, 0, 1")
# Block-call path, single process: the blocks' parameter-gradient reductions (five small launches per block on the second stream: 61 per step, each
# taking CU slots from the chip-owning kernel of the data-gradient chain that runs beside it) held back and folded in ONE
# dig_colsum_partials_multi launch behind the last data gradient (108 segments; the same sums).  "0": per block, on the second stream.
RED_DEFER = os.environ.get("DIG_RED_DEFER", "1") != "0"
BWD_SINGLE_STREAM = os.environ.get("DIG_BWD_SINGLE", "0") == "1"    # lab switch: the whole backward on the caller's stream (sum of solo kernel times)


# Lab: GPU time stamps at the phase boundaries of a step without a profiler (tools/gpu_step_phases.py sets PHASE_MARKS = []): events on the
# caller's stream, resolved by the tool after a synchronise.
PHASE_MARKS = None


def _mark(name, dev):
    if PHASE_MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        PHASE_MARKS.append((name, ev))


class LocalComm:
    world, rank = 1, 0

    def all_reduce_(self, t):
        return t

    def all_gather_cat(self, t):
        return t

    def grad_ready(self, model, key):
        pass


LOCAL = LocalComm()


class _EncWeights:
    """Per-encoder (online / momentum) accessors resolved once per arena binding."""

    def __init__(self, model, prefix, arena):
        self.blocks = []
        w16, f32 = model._w(arena), model._f32
        g32 = model._g32 if arena == "online" else {}
        for i in range(model.depth):
            b = f"{prefix}blocks.{i}."
            d = {k: f32[b + k] for k in ("norm1.weight", "norm1.bias", "attn.proj.bias", "norm2.weight", "norm2.bias",
                                         "mlp.fc1.bias", "mlp.fc2.bias")}
            d.update({k: w16[b + k] for k in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")})
            d["qkv_bias"] = model._qkv_bias[b + "attn."]
            if g32:
                d["g"] = {k: g32[b + k] for k in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.proj.weight",
                                                   "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight",
                                                   "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")}
                d["g"]["qkv_bias"] = model._qkv_bias_grad[b + "attn."]
            self.blocks.append(d)
        self.pe_w = f32[prefix + "patch_embed.proj.weight"].view(model.D, 48)
        self.pe_b = f32[prefix + "patch_embed.proj.bias"]
        self.mask_token = f32[prefix + "mask_token"].view(model.D)
        if g32:
            self.g_pe_w = g32[prefix + "patch_embed.proj.weight"].view(model.D, 48)
            self.g_pe_b = g32[prefix + "patch_embed.proj.bias"]
            self.g_mask_token = g32[prefix + "mask_token"].view(model.D)


def _weights(model):
    cache = getattr(model, "_wcache", None)
    mom = getattr(model, "use_moco_target", True)
    if cache is None or cache[0] != model._views_version or model._shadow.get("online") is None or (mom and model._shadow.get("momentum") is None):
        model.shadow("online")
        if mom:
            model.shadow("momentum")
        model._w16 = None
        cache = (model._views_version, _EncWeights(model, "encoder.", "online"),
                 _EncWeights(model, "momentum_encoder.", "momentum") if mom else None)
        model._wcache = cache
    return cache[1], cache[2]


class _BlockSaved:
    """What dig_encoder_block_fwd(save = 1) left for the backward of one block: two buffers (bf16 tensors, fp32 statistics) and the block's
    input rows x / ln1 / mean / rstd, which live in the PREVIOUS block's buffers (or, for block 0, in tensors of their own).  The block-call
    backward reads addresses (`ptr`); the per-entry-point backward asks for tensors()."""
    __slots__ = ("b16", "b32", "off", "rows", "D", "F", "n_img", "heads", "inp", "inp_ptr")

    def __init__(self, b16, b32, off, rows, D, F, n_img, heads, inp, inp_ptr):
        self.b16, self.b32, self.off, self.rows, self.D, self.F, self.n_img, self.heads = b16, b32, off, rows, D, F, n_img, heads
        self.inp = inp                   # keeps x, ln1, mu1, rs1 alive: 4 tensors (block 0) or the previous block's (b16, b32)
        self.inp_ptr = inp_ptr           # their addresses: (x, ln1, mu1, rs1)

    def ptr(self, name):
        return (self.b32 if name in ("lse", "mu2", "rs2", "nmu", "nrs") else self.b16).data_ptr() + self.off[name]

    def _v16(self, buf, base, byte_off, cols):
        a = (byte_off - base) // 2
        return buf[a:a + self.rows * cols].view(self.rows, cols)

    def view(self, name):
        R, D, F = self.rows, self.D, self.F
        if name == "lse":
            a = self.off["lse"] // 4
            return self.b32[a:a + self.n_img * self.heads * 256].view(self.n_img * self.heads, 256)
        if name in ("mu2", "rs2", "nmu", "nrs"):
            a = self.off[name] // 4
            return self.b32[a:a + R]
        cols = {"qkv": 3 * D, "pre": F, "act": F}.get(name, D)
        a = self.off[name] // 2
        return self.b16[a:a + R * cols].view(R, cols)

    def tensors(self):
        """(x, ln1, mu1, rs1, qkv, ctx, lse, x_mid, ln2, mu2, rs2, pre, act) as tensors (views)."""
        if len(self.inp) == 4:
            x, ln1, mu1, rs1 = self.inp
        else:
            p = self.inp[2]
            x, ln1, mu1, rs1 = p.view("out"), p.view("nln"), p.view("nmu"), p.view("nrs")
        return (x, ln1, mu1, rs1) + tuple(self.view(k) for k in ("qkv", "ctx", "lse", "x_mid", "ln2", "mu2", "rs2", "pre", "act"))


def _saved_tensors(s):
    return s.tensors() if isinstance(s, _BlockSaved) else s


class _Step:
    def __init__(self, model):
        self._keep = []                             # tensors the side stream still reads (see _on_side)
        self._keep_marks = []                       # (event on the side stream, length of _keep it covers): see _release_kept
        self.m = model
        self.comm = model.comm or LOCAL

    # ------------------------------------------------------------------ encoder
    def _path_specs(self, plan, site_off):
        """[(attention-branch DropSpec, MLP-branch DropSpec) or None per block] for stochastic depth under `plan` (dropout.DropPlan)."""
        from . import dropout as DR
        M = self.m
        out = []
        for i, p in enumerate(M.dpr):
            if not p:
                out.append(None)
                continue
            out.append((plan.spec(0, 0.0, path_site=DR.enc_site(i + site_off, 2), path_p=p, rows_per_sample=M.N),
                        plan.spec(0, 0.0, path_site=DR.enc_site(i + site_off, 4), path_p=p, rows_per_sample=M.N)))
        return out

    def encoder_forward(self, ew, images, aug, mask_u8, save, views=2, path=None):
        """views = 1: the encoder over `images` only (Gen-only models with only_mim_on_ori_img: the samples of a ViT batch are independent, and
        nothing downstream reads the augmented view's rows).  path: per-block stochastic-depth specs (_path_specs) or None."""
        M = self.m
        B, D, H, N = images.shape[0], M.D, M.H, M.N
        R = views * B * N
        x = torch.empty((R, D), device=images.device, dtype=BF16)
        for half, im in enumerate((images, aug)[:views]):
            ops.L.call("dig_patch_embed_fwd", ops.L.ptr(im), ops.L.ptr(ew.pe_w), ops.L.ptr(ew.pe_b),
                       ops.L.ptr(mask_u8[half * B:(half + 1) * B]), ops.L.ptr(ew.mask_token), ops.L.ptr(M._pos),
                       ops.L.ptr(x[half * B * N:(half + 1) * B * N]), B, M.gh, M.gw, D, ops.L.stream())
        saved = []
        scale = (D // H) ** -0.5
        chain = ops.mlp_chain_supported(D, M.F, R) and bool(ops.MLP_CHAIN_MASK & (2 if save else 1))
        chain_ln = chain and ops.MLP_CHAIN_LN and M.F <= 2048
        nxt = None                                                      # (ln1, mean, rstd) of this block, made by the previous block's launch
        if chain_ln and ops.BLOCK_CALLS and D == H * 64 and path is None:
            return self._encoder_forward_calls(ew, x, views * B, save, single_view=views == 1)
        for i, blk in enumerate(ew.blocks):
            ln1, mu1, rs1 = nxt if nxt is not None else ops.layernorm_fwd(x, blk["norm1.weight"], blk["norm1.bias"], M.ln_eps)
            nxt = None
            ds = path[i] if path is not None else None                 # x + drop_path(branch): per-sample keep / scale in the producing epilogue
            fused_attn = ops.attn_block_supported(H, D, B if views == 1 else None) and ds is None
            if fused_attn:
                # qkv Linear -> attention -> proj Linear + residual in one launch (csrc/attn_block.hip); qkv / lse exist only where kept
                x_mid, ctx, qkv, lse = ops.attn_block_fwd(ln1, x, blk["attn.qkv.weight"], blk["qkv_bias"], blk["attn.proj.weight"],
                                                          blk["attn.proj.bias"], views * B, H, D, scale, save=save)
            else:
                qkv = ops.linear_fwd(ln1, blk["attn.qkv.weight"], bias=blk["qkv_bias"], alpha=scale, alpha_cols=D)
                ctx, lse = ops.attn_fwd(qkv, views * B, H, D)
            if chain_ln:
                # norm2 -> fc1 -> GELU -> fc2 (+ residual) -> the NEXT block's norm1 in one launch: between two blocks the residual stream
                # is written once and no LayerNorm launch remains (the first block's norm1 is the only stand-alone one); norm2 is taken on
                # the way into the MLP launch
                nb = ew.blocks[i + 1] if i + 1 < len(ew.blocks) else None
                if not fused_attn:
                    x_mid = ops.linear_fwd(ctx, blk["attn.proj.weight"], bias=blk["attn.proj.bias"], resid=x, drop=ds[0] if ds else None)
                r = ops.mlp_chain_fwd_ln(x_mid, blk["norm2.weight"], blk["norm2.bias"], M.ln_eps, blk["mlp.fc1.weight"], blk["mlp.fc1.bias"],
                                         blk["mlp.fc2.weight"], blk["mlp.fc2.bias"], nb["norm1.weight"] if nb else None,
                                         nb["norm1.bias"] if nb else None, save=save, drop=ds[1] if ds else None)
                ln2, mu2, rs2 = r["ln"], r["ln_mean"], r["ln_rstd"]
                if save:
                    saved.append((x, ln1, mu1, rs1, qkv, ctx, lse, x_mid, ln2, mu2, rs2, r["pre"], r["act"]))
                if nb is not None:
                    nxt = (r["nln"], r["nln_mean"], r["nln_rstd"])
                x = r["out"]
                continue
            if not fused_attn:
                x_mid = ops.linear_fwd(ctx, blk["attn.proj.weight"], bias=blk["attn.proj.bias"], resid=x, drop=ds[0] if ds else None)
            ln2, mu2, rs2 = ops.layernorm_fwd(x_mid, blk["norm2.weight"], blk["norm2.bias"], M.ln_eps)
            if chain and ds is None:
                # fc1 -> GELU -> fc2 (+ residual) in one launch: the [R, F] hidden tensor is never a GEMM operand in HBM; the online
                # branch still writes the pre-activation and the GELU output (the backward's inputs), the momentum branch nothing
                if save:
                    x_out, pre, act = ops.mlp_chain_fwd(ln2, blk["mlp.fc1.weight"], blk["mlp.fc1.bias"], blk["mlp.fc2.weight"],
                                                        blk["mlp.fc2.bias"], x_mid, save=True)
                else:
                    pre = act = None
                    x_out = ops.mlp_chain_fwd(ln2, blk["mlp.fc1.weight"], blk["mlp.fc1.bias"], blk["mlp.fc2.weight"], blk["mlp.fc2.bias"], x_mid)
            else:
                pre = torch.empty((R, M.F), device=x.device, dtype=BF16) if save else None
                act = ops.linear_fwd(ln2, blk["mlp.fc1.weight"], bias=blk["mlp.fc1.bias"], act=1, pre=pre)
                x_out = ops.linear_fwd(act, blk["mlp.fc2.weight"], bias=blk["mlp.fc2.bias"], resid=x_mid, drop=ds[1] if ds else None)
            if save:
                saved.append((x, ln1, mu1, rs1, qkv, ctx, lse, x_mid, ln2, mu2, rs2, pre, act))
            x = x_out
        return x, saved

    def _encoder_forward_calls(self, ew, x, n_img, save, single_view=False):
        """The block loop of encoder_forward with ONE FFI crossing per block (dig_encoder_block_fwd: qkv GEMM -> attention -> proj GEMM +
        residual -> norm2 + MLP + residual + the next block's norm1) and two allocations per block (a bf16 buffer, an fp32 one) instead of
        four crossings and thirteen allocations.  Same kernels, same arguments, same order: bit-identical to the loop above."""
        M = self.m
        D, Fh, H = M.D, M.F, M.H
        R, dev = x.shape[0], x.device
        off, n16, n32 = ops.block_fwd_layout(R, D, Fh, n_img, H, save)
        blk0 = ew.blocks[0]
        ln1, mu1, rs1 = ops.layernorm_fwd(x, blk0["norm1.weight"], blk0["norm1.bias"], M.ln_eps)
        inp, inp_ptr = (x, ln1, mu1, rs1), (x.data_ptr(), ln1.data_ptr(), mu1.data_ptr(), rs1.data_ptr())
        stream = ops.L.stream()
        saved = []
        nb_blocks = len(ew.blocks)
        key = ("fwd_call", bool(save), R, n_img)
        fuse_attn = int(ops.attn_block_supported(H, D, n_img if single_view else None))
        prev = None
        for i, blk in enumerate(ew.blocks):
            st = blk.get(key)
            if st is None:
                nb = ew.blocks[i + 1] if i + 1 < nb_blocks else None
                st = blk[key] = ops.BlockFwd(
                    n_img=n_img, heads=H, D=D, F=Fh, rows=R, save=int(bool(save)),
                    tile_qkv=ops.fwd_tile_code(R, 3 * D, D) or ops.GEMM_BK_FWD, tile_proj=ops.fwd_tile_code(R, D, D, has_resid=True) or ops.GEMM_BK_FWD,
                    fuse_attn=fuse_attn, eps=M.ln_eps, scale=(D // H) ** -0.5,
                    qkv_w=blk["attn.qkv.weight"].data_ptr(), qkv_b=blk["qkv_bias"].data_ptr(), proj_w=blk["attn.proj.weight"].data_ptr(),
                    proj_b=blk["attn.proj.bias"].data_ptr(), n2_g=blk["norm2.weight"].data_ptr(), n2_b=blk["norm2.bias"].data_ptr(),
                    fc1_w=blk["mlp.fc1.weight"].data_ptr(), fc1_b=blk["mlp.fc1.bias"].data_ptr(), fc2_w=blk["mlp.fc2.weight"].data_ptr(),
                    fc2_b=blk["mlp.fc2.bias"].data_ptr(), next_n1_g=nb["norm1.weight"].data_ptr() if nb else None,
                    next_n1_b=nb["norm1.bias"].data_ptr() if nb else None)
            b16 = torch.empty(n16 // 2, device=dev, dtype=BF16)
            b32 = torch.empty(n32 // 4, device=dev, dtype=F32)
            p16, p32 = b16.data_ptr(), b32.data_ptr()
            st.fuse_attn = fuse_attn                                  # (a switch, like fuse_ln2 / wg_defer of the backward: read on every call)
            st.x, st.ln1 = inp_ptr[0], inp_ptr[1]
            st.qkv, st.ctx, st.x_mid, st.out, st.nln = p16 + off["qkv"], p16 + off["ctx"], p16 + off["x_mid"], p16 + off["out"], p16 + off["nln"]
            st.lse = p32 + off["lse"]
            if save:
                st.ln2, st.pre, st.act = p16 + off["ln2"], p16 + off["pre"], p16 + off["act"]
                st.mu2, st.rs2, st.nmu, st.nrs = p32 + off["mu2"], p32 + off["rs2"], p32 + off["nmu"], p32 + off["nrs"]
            ops.L.call("dig_encoder_block_fwd", ctypes.byref(st), stream)
            cur = _BlockSaved(b16, b32, off, R, D, Fh, n_img, H, inp, inp_ptr)
            if save:
                saved.append(cur)
                inp, inp_ptr = (b16, b32, cur), (p16 + off["out"], p16 + off["nln"], p32 + off["nmu"], p32 + off["nrs"])
            else:
                inp, inp_ptr = (b16, b32), (p16 + off["out"], p16 + off["nln"], 0, 0)     # (no chain of blocks: the previous buffers go back to the pool)
            prev = cur
        return prev.view("out"), saved

    def _encoder_backward_calls(self, ew, saved, dx, wT, plan, n_img, R):
        """The block loop of encoder_backward with ONE FFI crossing per block (dig_encoder_block_bwd: the data-gradient chain with the grouped
        weight gradients in it on this stream, the five parameter-gradient reductions on the side stream behind one event) and two
        allocations per block.  Same kernels, same arguments, same order as the loop in encoder_backward.  Returns the gradient w.r.t. the
        patch embedding's output rows."""
        M = self.m
        D, Fh, H = M.D, M.F, M.H
        dev = dx.device
        main, side = self._streams(dev)
        off, n16, n32 = ops.block_bwd_layout(R, D, Fh, n_img)
        grp = plan["group"]
        stream, side_h = ops.L.stream(), ctypes.c_void_p(side.cuda_stream)
        probs = ((ops._WgProb * 4)(), (ops._WgProb * 4)())
        wmap_ptr = plan["wmap"].data_ptr()
        tile = ops.dgrad_tile_code(R, D) or ops.GEMM_BK_BWD
        tile_direct = ops.dgrad_direct_tile_code(R, D)
        key = ("bwd_call", R, n_img)
        dy_ptr, dy_owner = dx.data_ptr(), dx
        prev_block, n_launch, slabs_prev = None, 0, None
        deferred = []                                                    # deferred plan: (problem table, temporaries kept alive) per block
        defer = WGRAD_DEFER == "1" or (WGRAD_DEFER == "auto" and self.comm is LOCAL and self._defer_fits(dev, M.depth * n16))
        red_defer = bool(RED_DEFER and defer and self.comm is LOCAL and ops.MLP_CHAIN_LNB and M.depth * 9 <= ops.COLSUM_MAX_SEGS)
        red_segs, red_keep = [], []                                      # (partials address, destination, stride, partial rows, columns) of the held-back reductions
        lib = ops.L.lib()
        n_b, n_l2, n_l1 = lib.dig_mlp_chain_colsum_rows(R), lib.dig_mlp_chain_ln_parts(R), lib.dig_layernorm_bwd_parts(R)
        for i in reversed(range(M.depth)):
            blk, g, sv = ew.blocks[i], ew.blocks[i]["g"], saved[i]
            saved[i] = None
            st = blk.get(key)
            if st is None:
                gb = g["qkv_bias"]
                for k in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
                    assert g[k].is_contiguous()
                st = blk[key] = ops.BlockBwd(
                    n_img=n_img, heads=H, D=D, F=Fh, rows=R, tile_dgrad=tile, scale=(D // H) ** -0.5,
                    qkv_w=blk["attn.qkv.weight"].data_ptr(), proj_w=blk["attn.proj.weight"].data_ptr(),
                    n1_g=blk["norm1.weight"].data_ptr(), n1_b=blk["norm1.bias"].data_ptr(), n2_g=blk["norm2.weight"].data_ptr(),
                    n2_b=blk["norm2.bias"].data_ptr(),
                    g_n1_g=g["norm1.weight"].data_ptr(), g_n1_b=g["norm1.bias"].data_ptr(), g_qkv_w=g["attn.qkv.weight"].data_ptr(),
                    g_q_b=gb.data_ptr(), g_v_b=gb[2 * D:].data_ptr(), g_proj_w=g["attn.proj.weight"].data_ptr(),
                    g_proj_b=g["attn.proj.bias"].data_ptr(), g_n2_g=g["norm2.weight"].data_ptr(), g_n2_b=g["norm2.bias"].data_ptr(),
                    g_fc1_w=g["mlp.fc1.weight"].data_ptr(), g_fc1_b=g["mlp.fc1.bias"].data_ptr(), g_fc2_w=g["mlp.fc2.weight"].data_ptr(),
                    g_fc2_b=g["mlp.fc2.bias"].data_ptr(),
                    wg_fn=plan["fn"], wg_wa=plan["wa"], wg_splits=plan["splits"], wg_n_wg=plan["n_wg"], wg_fold_splits=plan["splits"],
                    wg_trans=(ctypes.c_int * 4)(*plan["trans"]))
            w2t, w1t, projt, qkvt = wT[i]
            st.w2t, st.w1t = w2t.data_ptr(), w1t.data_ptr()
            st.projt = projt.data_ptr() if (ops.MLP_CHAIN_LNB and ops.MLP_CHAIN_PROJ) else None
            st.tile_direct = tile_direct                              # (a switch: read on every call)
            st.attn_proj = int(ops.ATTN_BWD_PROJ and ops.attn_bwd_proj_supported(D) and not ops.attn_bwd_mode())   # (a switch, read on every call)
            st.proj_wt = projt.data_ptr() if (tile_direct or st.attn_proj) else None
            st.qkv_wt = qkvt.data_ptr() if tile_direct else None
            st.x, st.ln1, st.mu1, st.rs1 = sv.inp_ptr
            for k in ("qkv", "ctx", "lse", "x_mid", "ln2", "mu2", "rs2", "pre", "act"):
                setattr(st, k, sv.ptr(k))
            t16 = torch.empty(n16 // 2, device=dev, dtype=BF16)
            t32 = torch.empty(n32 // 4, device=dev, dtype=F32)
            p16, p32 = t16.data_ptr(), t32.data_ptr()
            st.dy = dy_ptr
            st.dln2, st.dpre, st.dctx, st.dqkv = p16 + off["dln2"], p16 + off["dpre"], p16 + off["dctx"], p16 + off["dqkv"]
            st.bparts, st.ws1, st.ws2, st.qs, st.vs = p32 + off["bparts"], p32 + off["ws1"], p32 + off["ws2"], p32 + off["qs"], p32 + off["vs"]
            slabs = grp._slabs(plan["slab_bytes"])
            grp.set ^= 1
            st.wg_map, st.wg_slabs = wmap_ptr, slabs.data_ptr()
            st.wg_defer = int(defer)
            st.fuse_ln2 = int(ops.MLP_CHAIN_LNB)
            st.defer_red = int(red_defer)
            if red_defer:
                # what dig_encoder_block_bwd would have launched on the second stream (csrc/encoder_block.inc), as segments of one launch
                gb = g["qkv_bias"]
                red_segs.append((st.bparts, g["mlp.fc1.bias"], Fh, n_b, Fh))
                for k, dst in enumerate((g["norm2.weight"], g["norm2.bias"], g["mlp.fc2.bias"])):
                    red_segs.append((st.ws2 + 4 * k * D, dst, 3 * D, n_l2, D))
                red_segs.append((st.qs, gb[:D], D, n_img, D))
                red_segs.append((st.vs, gb[2 * D:], D, n_img, D))
                for k, dst in enumerate((g["norm1.weight"], g["norm1.bias"], g["attn.proj.bias"])):
                    red_segs.append((st.ws1 + 4 * k * D, dst, 3 * D, n_l1, D))
                red_keep.append(t32)
            if defer:
                own = (ops._WgProb * 4)()
                deferred.append((own, t16, sv, i))
                st.wg_probs = ctypes.addressof(own)
                st.wg_fold_n = 0
                st.wg_fold_probs = st.wg_fold_slabs = None
                st.side = side_h
                ops.L.call("dig_encoder_block_bwd", ctypes.byref(st), stream)
                if side is not main and not red_defer:
                    self._keep.append(t32)
                dy_ptr, dy_owner = p16 + off["dctx"], t16
                self._mark_kept(dev)
                self._release_kept(dev)
                continue
            st.wg_probs = ctypes.addressof(probs[n_launch & 1])
            st.wg_fold_n = 4 if n_launch else 0
            st.wg_fold_probs = ctypes.addressof(probs[(n_launch & 1) ^ 1]) if n_launch else None
            st.wg_fold_slabs = slabs_prev.data_ptr() if n_launch else None
            st.side = side_h
            ops.L.call("dig_encoder_block_bwd", ctypes.byref(st), stream)
            n_launch += 1
            slabs_prev = slabs
            if side is not main:
                self._keep.append(t32)                                   # the side stream's reductions read it
            dy_ptr, dy_owner = p16 + off["dctx"], t16                    # the next block's incoming gradient lives in this block's buffer
            del sv
            self._mark_kept(dev)
            self._release_kept(dev)
            # (block i's slabs are folded by the NEXT launch, so the bucket that is final here is block i + 1's)
            if prev_block is not None:
                self._grad_ready(dev, f"encoder.blocks.{prev_block}")
            prev_block = i
        if red_segs:
            # every block's bias / LayerNorm partial rows in one launch, on this stream, in front of the weight gradients (nothing runs beside it)
            segs = (ops._ColsumSeg * len(red_segs))(*[ops._ColsumSeg(int(p_), d_.data_ptr(), st_, n_, c_) for p_, d_, st_, n_, c_ in red_segs])
            ops.L.call("dig_colsum_partials_multi", segs, len(red_segs), stream)
            del red_keep
        if defer:
            # the twelve launches now, in block order (each folds its predecessor's slabs), every bucket behind its fold
            prev_probs = None
            for own, _t16, _sv, i in deferred:
                slabs = grp._slabs(plan["slab_bytes"])
                grp.set ^= 1
                ops.L.call("dig_wgrad_group", ctypes.addressof(own), 4, ctypes.addressof(prev_probs) if prev_probs is not None else None,
                           4 if prev_probs is not None else 0, int(R), plan["splits"], wmap_ptr, plan["n_wg"], ops.L.ptr(slabs),
                           ops.L.ptr(slabs_prev) if slabs_prev is not None else None, plan["splits"], plan["fn"], plan["wa"], stream)
                if prev_block is not None:
                    self._grad_ready(dev, f"encoder.blocks.{prev_block}")
                prev_probs, slabs_prev, prev_block = own, slabs, i
            probs = (prev_probs, prev_probs)
            n_launch = 1
        # fold of the last launch's slabs (a fold-only launch), then the last bucket
        ops.L.call("dig_wgrad_group", None, 0, ctypes.addressof(probs[(n_launch & 1) ^ 1]), 4, int(R), 1, None, ops.WGRAD_GROUP_SLOTS, None,
                   ops.L.ptr(slabs_prev), plan["splits"], plan["fn"], plan["wa"], stream)
        if prev_block is not None:
            self._grad_ready(dev, f"encoder.blocks.{prev_block}")
        a = off["dctx"] // 2
        return dy_owner[a:a + R * D].view(R, D)

    def _defer_fits(self, dev, extra_bytes):
        """The deferred weight-gradient plan keeps `extra_bytes` of gradient temporaries (and every block's saved activations) alive until
        the end of the backward: taken only while that leaves three quarters of what the device (and torch's pool) has free.  The answer
        is cached per size: one hipMemGetInfo per new shape, not per step."""
        cache = self.m.__dict__.setdefault("_defer_fits_cache", {})
        key = (dev.index, extra_bytes)
        if key not in cache:
            free, _ = torch.cuda.mem_get_info(dev)
            pooled = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            cache[key] = extra_bytes <= (free + pooled) // 4
        return cache[key]

    def mlp_weight_transposes(self, ew, fresh=False):
        """K-contiguous copies of the MLP weights for the fused backward (W2^T [F, D], W1^T [D, F]; 1.2 MB each) and of the projection weight.
        Normally the previous step's optimizer launch has written them (dig_adamw_step_tr, MoCo_ViT.transposed_weight_table): `fresh` ->
        no launch at all.  Otherwise (first step, after load_state_dict / .to(), under graph capture, DIG_ADAMW_FOLD=0) three launches
        rebuild them from this step's bf16 weight shadow, into the same persistent buffers where the model has them."""
        tr = self.m.transposed_weight_table() if hasattr(self.m, "transposed_weight_table") else None
        self.head_wT = {}
        if tr is not None and fresh:
            self.head_wT = tr[6]
            return tr[4]
        outs = tr[4] if tr is not None else None
        w2t = ops.transpose_bf16_multi([b["mlp.fc2.weight"] for b in ew.blocks], [o[0] for o in outs] if outs else None)   # one launch per weight shape
        w1t = ops.transpose_bf16_multi([b["mlp.fc1.weight"] for b in ew.blocks], [o[1] for o in outs] if outs else None)
        projt = ops.transpose_bf16_multi([b["attn.proj.weight"] for b in ew.blocks], [o[2] for o in outs] if outs else None)
        qkvt = ops.transpose_bf16_multi([b["attn.qkv.weight"] for b in ew.blocks], [o[3] for o in outs] if outs else None)   # (direct-form qkv data gradient)
        if tr is not None:                                              # (the heads' copies: one small launch each, only behind a foreign write)
            w16 = self.m._w("online")
            for name, dst in tr[6].items():
                ops.transpose_bf16(w16[name], dst)
            self.head_wT = tr[6]
        return list(zip(w2t, w1t, projt, qkvt))

    # ------------------------------------------------------------------ two-stream helpers (backward)
    def _streams(self, dev):
        M = self.m
        main = torch.cuda.current_stream(dev)
        side = M._bwd_side_stream(dev) if (getattr(M, "overlap_streams", True) and not BWD_SINGLE_STREAM) else main
        return main, side

    def _on_side(self, dev, fn, *tensors):
        """Run fn (weight-gradient GEMMs, column sums: consumers of tensors the main chain has just produced) on the side
        stream, after everything the main stream has queued so far."""
        main, side = self._streams(dev)
        if side is main:
            fn()
            return
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()
        # the side stream reads tensors of the caller's stream's pool: instead of record_stream (an event on the side stream per block when
        # it is released: ~120 per backward, 0.5+ ms of queue time) they are kept alive until backward() has joined the two streams
        self._keep.extend(tensors)

    def _mark_kept(self, dev):
        """Everything _on_side() has kept so far is dead once the side stream passes this point."""
        main, side = self._streams(dev)
        if side is main or not self._keep:
            return
        ev = torch.cuda.Event()
        ev.record(side)
        self._keep_marks.append((ev, len(self._keep)))

    def _release_kept(self, dev, lag=2):
        """Drop the kept tensors of marks at least `lag` marks old: the caller's stream waits for that mark (the side stream is never two
        encoder blocks behind, so the wait is already satisfied when it is reached) and the blocks go back to its pool -- without this
        every block's gradient temporaries stayed resident until the end of backward (~0.7 GB per ViT-S block at B = 256)."""
        main, side = self._streams(dev)
        while len(self._keep_marks) > lag:
            ev, n = self._keep_marks.pop(0)
            main.wait_event(ev)
            del self._keep[:n]
            self._keep_marks = [(e, m - n) for e, m in self._keep_marks]

    def _grad_ready(self, dev, key):
        """Bucket `key` is final once both streams pass this point.  With a process group the all-reduce is issued from the
        side stream after it has waited for the main chain, so the main chain never stalls on a collective."""
        main, side = self._streams(dev)
        if side is not main and (self.comm.world > 1 or getattr(self.comm, "world_override", False)):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.comm.grad_ready(self.m, key)
        else:
            self.comm.grad_ready(self.m, key)

    def encoder_backward(self, ew, saved, dx, images, aug, mask_u8, views=2):
        """dx: bf16 [R, D] gradient w.r.t. the encoder output (consumed).
        The data-gradient chain (dgrad GEMMs, attention backward, LayerNorm backward) runs on the caller's stream; the
        weight-gradient GEMMs and bias column sums only consume (dy, saved activation) pairs, so they are issued on a
        second HIP stream and overlap the chain (both are latency-bound kernels that leave CU resources idle)."""
        M = self.m
        B, D, H, N = images.shape[0], M.D, M.H, M.N
        scale = (D // H) ** -0.5
        dev = dx.device
        main, side = self._streams(dev)

        def on_side(fn, *tensors):
            self._on_side(dev, fn, *tensors)

        chain_any = ops.mlp_chain_supported(D, M.F, views * B * N) and bool(ops.MLP_CHAIN_MASK & 4)
        chain = chain_any
        wT = getattr(self, "wT", None)
        if chain and wT is None:
            wT = self.mlp_weight_transposes(ew)
        elif wT is not None:
            for pair_ in wT:                                            # made on the side stream in forward(), read here on the main one
                for t in pair_:
                    t.record_stream(main)
        # The four weight gradients of a block as ONE grouped launch on the side stream (csrc/wgrad.hip), issued as soon as the block's last
        # operand (dqkv) exists; its slabs are folded into the gradient arena by the next block's launch (or the flush after block 0), so
        # block i's bucket is final -- and its all-reduce is issued -- one launch later.  Shapes the grouped kernel does not take (tiny test
        # models) and the batched-reduction mode keep the per-layer launches.
        Rg = (B * N) if views == 1 else (2 * B * N)
        grouped = (ops.WGRAD_GROUP and WGRAD_GROUPING != "off" and
                   all(ops.wgrad_group_route(o, i_, Rg) is not None for o, i_ in ((M.F, D), (D, M.F), (3 * D, D), (D, D))))
        path = getattr(self, "path_on", None)
        if (grouped and chain_any and ops.BLOCK_CALLS and WGRAD_INLINE and WGRAD_GROUPING == "block" and CHAIN_BWD_EVERY == 1 and FUSED_QV_BIAS_SUMS
                and path is None and not BATCH_REDUCE and PHASE_MARKS is None and all(isinstance(s_, _BlockSaved) for s_ in saved) and dx.is_contiguous()):
            plan = ops.wgrad_block_plan(dev, Rg, D, M.F)
            if plan is not None:
                dx = self._encoder_backward_calls(ew, saved, dx, wT, plan, B if views == 1 else 2 * B, Rg)
                grouped = None                                           # the blocks are done: only the patch embedding is left
        grp = ops.WgradGroup(dev) if grouped else None
        prev_block = None
        # single process, per-entry-point path (the 512-wide model, single-view encoders): the blocks' bias / LayerNorm column-sum launches held
        # back as in _encoder_backward_calls -- collected over ALL blocks and folded in one dig_colsum_partials_multi launch behind the loop
        vred = ops.GradReduceBatch() if (RED_DEFER and grp and self.comm is LOCAL and path is None and not BATCH_REDUCE and PHASE_MARKS is None
                                         and grouped is not None) else None

        def launch_group(*tensors):
            if WGRAD_INLINE:
                grp.launch()                                             # in the data-gradient chain itself (see WGRAD_INLINE)
            else:
                on_side(grp.launch, *tensors)

        for i in reversed(range(M.depth if grouped is not None else 0)):
            blk, g = ew.blocks[i], ew.blocks[i]["g"]
            x, ln1, mu1, rs1, qkv, ctx, lse, x_mid, ln2, mu2, rs2, pre, act = _saved_tensors(saved[i])
            saved[i] = None
            chain = chain_any and (i % CHAIN_BWD_EVERY == CHAIN_BWD_PHASE % CHAIN_BWD_EVERY)
            if views == 1:            # only view 0 carries a gradient (zero contrastive weight): rows [0, B*N) of everything
                Rh = B * N
                x, ln1, mu1, rs1, qkv, ctx, x_mid, ln2, mu2, rs2, pre, act = (t[:Rh] for t in (x, ln1, mu1, rs1, qkv, ctx, x_mid, ln2, mu2, rs2, pre, act))
                lse = lse[:B * H]
            # x_out = x_mid + fc2(gelu(fc1(ln2)))
            # The reductions this block leaves behind (four split-R slab sums, five bias / LayerNorm-parameter column sums) are
            # collected in `red` and issued as two launches after the block's last weight-gradient GEMM (ops.GradReduceBatch).
            red = ops.GradReduceBatch() if BATCH_REDUCE else None
            wg = red.wgrad if red else ops.linear_wgrad
            csum = red.colsum_partials if red else ops.colsum_partials
            held = []                                                        # operands of the grouped launch (kept alive until the streams join)
            # With the grouped launch, everything this block hands to the second stream -- the bias / LayerNorm-parameter column sums: five
            # small launches that only consume what the chain has produced -- goes over in ONE hand-over at the end of the block (one
            # wait_stream + one stream switch on the host instead of five; the kernels are off the critical path either way)
            late, late_t = [], []

            def side_later(fn, *tensors):
                if grp:
                    late.append(fn)
                    late_t.extend(tensors)
                else:
                    on_side(fn, *tensors)
            if grp:
                def wg(dy_, x_, dw_):
                    if not grp.add(dy_, x_, dw_):                    # (never inside an assert: python -O would drop the weight gradient)
                        raise RuntimeError("grouped weight gradient: a problem of this block does not fit the group's plan")
                    held.extend((dy_, x_))
            # stochastic depth: a dropped branch back-propagates the residual gradient under the same per-sample mask (dz); its bias gradient is
            # the column sum of the MASKED gradient, so the LayerNorm kernel's fused residual column sum is switched off for it
            ds = path[i] if path is not None else None
            dz = ops.dropout_apply(dx, ds[1]) if ds else dx
            if ds:
                on_side(lambda dz=dz: ops.colsum(dz, g["mlp.fc2.bias"]), dz)
            if grp:
                wg(dz, act, g["mlp.fc2.weight"])
            else:
                on_side(lambda dz=dz: wg(dz, act, g["mlp.fc2.weight"]), dz, act)
            if chain:
                # data gradient through fc2, GELU' and fc1 in one launch (d(pre-activation) leaves it as a side output for the fc1
                # weight gradient, with its column sums = the fc1 bias gradient)
                w2t, w1t, projt, qkvt = wT[i]
                dctx = None
                if ds:
                    dln2, dact, bparts = ops.mlp_chain_bwd(dz, w2t, pre, w1t)
                    lnp = None
                elif ops.MLP_CHAIN_LNB and red is None and ops.MLP_CHAIN_PROJ:
                    dx_mid, dact, bparts, lnp, dctx = ops.mlp_chain_bwd_ln(dx, w2t, pre, w1t, x_mid, blk["norm2.weight"], mu2, rs2, projt=projt)
                    dln2 = dx_mid
                elif ops.MLP_CHAIN_LNB and red is None:
                    # ... with norm2's backward in the same launch: dx_mid = dx + LN2'(d ln2) leaves it, the three parameter-gradient
                    # sums of norm2 / fc2's bias as partial rows
                    dx_mid, dact, bparts, lnp = ops.mlp_chain_bwd_ln(dx, w2t, pre, w1t, x_mid, blk["norm2.weight"], mu2, rs2)
                    dln2 = dx_mid
                else:
                    dln2, dact, bparts = ops.mlp_chain_bwd(dx, w2t, pre, w1t)
                    lnp = None
                _mark("blk: fused MLP backward", dev)
            else:
                dact, bparts = ops.linear_dgrad(dz, blk["mlp.fc2.weight"], gelu_pre=pre, colsum=True)   # d(pre-activation): GELU' and
                dln2 = None                                                                              # the fc1 bias sums fused
            if grp:
                wg(dact, ln2, g["mlp.fc1.weight"])
                if vred:
                    vred.colsum_partials(bparts, g["mlp.fc1.bias"])
                else:
                    side_later(lambda: csum(bparts, g["mlp.fc1.bias"]), bparts)
                if WGRAD_GROUPING == "pair":
                    launch_group(*held)
            else:
                on_side(lambda: (wg(dact, ln2, g["mlp.fc1.weight"]), csum(bparts, g["mlp.fc1.bias"])), dact, ln2, bparts)   # (0.3 ms/step vs a 201 MB pass)
            if dln2 is None:
                dln2 = ops.dgrad_direct(dact, wT[i][1]) if wT is not None else None        # (direct form on fc1.weight^T where it pays)
                if dln2 is None:
                    dln2 = ops.linear_dgrad(dact, blk["mlp.fc1.weight"])
            if chain and lnp is not None and vred:
                vred.layernorm_finalize_parts(lnp, g["norm2.weight"], g["norm2.bias"], g["mlp.fc2.bias"])
            elif chain and lnp is not None:
                side_later(lambda lnp=lnp, g=g: ops.layernorm_finalize_parts(lnp, g["norm2.weight"], g["norm2.bias"], g["mlp.fc2.bias"]), lnp)
            else:
                dx_mid, fin2, ws2 = ops.layernorm_bwd(dln2, x_mid, blk["norm2.weight"], blk["norm2.bias"], mu2, rs2, dx, g["norm2.weight"],
                                                      g["norm2.bias"], out=dln2, dres_colsum=None if ds else g["mlp.fc2.bias"], defer=True)
                if red or vred:                                               # norm2 grads + colsum(dx) = fc2 bias grad: off the chain
                    (red or vred).layernorm_finalize(ws2, x_mid.shape[0], D, g["norm2.weight"], g["norm2.bias"], None if ds else g["mlp.fc2.bias"])
                else:
                    side_later(fin2, ws2)
            # x_mid = x + proj(attn(ln1))
            dzp = ops.dropout_apply(dx_mid, ds[0]) if ds else dx_mid
            if ds:
                on_side(lambda dzp=dzp: ops.colsum(dzp, g["attn.proj.bias"]), dzp)
            if grp:
                wg(dzp, ctx, g["attn.proj.weight"])
            else:
                on_side(lambda dzp=dzp: wg(dzp, ctx, g["attn.proj.weight"]), dzp, ctx)
            _mark("blk: LayerNorm backward (norm2)", dev)
            # (the projection's data gradient inside the attention backward launch where that form exists: no GEMM, no d(ctx) rows)
            proj_attn = ((not chain or dctx is None) and ops.ATTN_BWD_PROJ and FUSED_QV_BIAS_SUMS and wT is not None and not ds
                         and ops.attn_bwd_proj_supported(D) and not ops.attn_bwd_mode())
            if (not chain or dctx is None) and not proj_attn:
                # (direct form on proj.weight^T where it pays: both operands K-contiguous, bit-identical to the transpose-read form)
                dctx = ops.dgrad_direct(dzp, wT[i][2]) if wT is not None else None
                if dctx is None:
                    dctx = ops.linear_dgrad(dzp, blk["attn.proj.weight"])
            _mark("blk: proj data gradient", dev)
            gb = g["qkv_bias"]
            if FUSED_QV_BIAS_SUMS:
                # q_bias / v_bias gradients: per-image column sums of dQ (already carrying the q scale) and dV leave the attention
                # kernel as [2B, D] fp32 partials (DPP row reductions of the accumulators, no extra pass over the 150 MB dqkv);
                # K has no bias
                if proj_attn:
                    dqkv, qs, vs = ops.attn_bwd_proj(qkv, ctx, dzp, wT[i][2], lse, views * B, H, D, scale, bias_sums=True)
                    dctx = None                                             # (no d(ctx) rows: the qkv data gradient below gets a buffer of its own)
                else:
                    dqkv, qs, vs = ops.attn_bwd(qkv, ctx, dctx, lse, views * B, H, D, scale, bias_sums=True)
                _mark("blk: attention backward", dev)
                if grp:
                    wg(dqkv, ln1, g["attn.qkv.weight"])
                    launch_group(*held)
                    _mark("blk: grouped weight gradients", dev)
                    if vred:
                        vred.colsum_partials(qs, gb[:D]); vred.colsum_partials(vs, gb[2 * D:])
                    else:
                        side_later(lambda: (csum(qs, gb[:D]), csum(vs, gb[2 * D:])), qs, vs)
                else:
                    on_side(lambda: (wg(dqkv, ln1, g["attn.qkv.weight"]),
                                     csum(qs, gb[:D]), csum(vs, gb[2 * D:])), dqkv, ln1, qs, vs)
            else:
                dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, views * B, H, D, scale)
                if grp:
                    wg(dqkv, ln1, g["attn.qkv.weight"])
                    launch_group(*held)
                    side_later(lambda: (ops.colsum(dqkv, gb[:D], cols=D), ops.colsum(dqkv[:, 2 * D:], gb[2 * D:], cols=D)), dqkv)
                else:
                    on_side(lambda: (wg(dqkv, ln1, g["attn.qkv.weight"]),
                                     ops.colsum(dqkv, gb[:D], cols=D), ops.colsum(dqkv[:, 2 * D:], gb[2 * D:], cols=D)), dqkv, ln1)
            dln1 = ops.dgrad_direct(dqkv, wT[i][3], out=dctx) if wT is not None else None
            if dln1 is None:
                dln1 = ops.linear_dgrad(dqkv, blk["attn.qkv.weight"], out=dctx)
            _mark("blk: qkv data gradient", dev)
            dx, fin1, ws1 = ops.layernorm_bwd(dln1, x, blk["norm1.weight"], blk["norm1.bias"], mu1, rs1, dx_mid, g["norm1.weight"],
                                              g["norm1.bias"], out=dln1, dres_colsum=None if ds else g["attn.proj.bias"], defer=True)
            _mark("blk: LayerNorm backward (norm1)", dev)
            if vred:
                vred.layernorm_finalize(ws1, x.shape[0], D, g["norm1.weight"], g["norm1.bias"], g["attn.proj.bias"])
            elif red:                                                         # norm1 grads + colsum(dx_mid) = proj bias grad
                red.layernorm_finalize(ws1, x.shape[0], D, g["norm1.weight"], g["norm1.bias"], None if ds else g["attn.proj.bias"])
                side_later(red.flush, *red.tensors())
            else:
                side_later(fin1, ws1)
            if late:
                fns = list(late)
                on_side(lambda: [f() for f in fns], *late_t)
            del dact, pre, act, dln2, dqkv, dctx, held, late, late_t, dz, dzp
            self._mark_kept(dev)
            self._release_kept(dev)
            # this block's gradients are final once BOTH streams pass this point: the bucket's all-reduce is issued from the
            # side stream after it has waited for the main chain, so the main chain itself never stalls on the collective.
            # (Grouped weight gradients: block i's slabs are folded by the NEXT launch, so the bucket that is final here is block i + 1's.)
            if grp:
                if prev_block is not None:
                    self._grad_ready(dev, f"encoder.blocks.{prev_block}")
                prev_block = i
            else:
                self._grad_ready(dev, f"encoder.blocks.{i}")
        if vred:
            vred.flush()                                                     # (vectors only: one launch per 112 segments, on this stream)
        if grp:
            if WGRAD_INLINE:
                grp.flush()
            else:
                on_side(grp.flush)
            if prev_block is not None:
                self._grad_ready(dev, f"encoder.blocks.{prev_block}")
        for half, im in enumerate((images, aug)[:views]):
            ops.patch_embed_bwd_mfma(dx[half * B * N:(half + 1) * B * N], im, mask_u8[half * B:(half + 1) * B], ew.g_pe_w, ew.g_pe_b,
                                     ew.g_mask_token, D, M.gh, M.gw)
        main.wait_stream(side)
        self.comm.grad_ready(M, "encoder.embed")

    # ------------------------------------------------------------------ BN-MLP heads
    def mlp_forward(self, x, pre, arena, save):
        """_build_mlp stack (modeling_pretrain_moco_mim_ori.py:463-482): Linear(no bias) -> BN(train, cross-rank stats)
        -> ReLU ... ; the last BN has no affine parameters."""
        M = self.m
        dims = M.mlps[pre]
        w16, f32 = M._w(arena), M._f32
        n_local = x.shape[0]
        n_total = float(n_local * self.comm.world)
        saved = []
        for l, (d1, d2) in enumerate(dims):
            last = l == len(dims) - 1
            h = ops.linear_fwd(x, w16[f"{pre}.{3 * l}.weight"])
            gamma = None if last else f32[f"{pre}.{3 * l + 1}.weight"]
            beta = None if last else f32[f"{pre}.{3 * l + 1}.bias"]
            rm, rv, i_bn = M._bn_views[f"{pre}.{3 * l + 1}"]
            if self.comm is LOCAL and ops.bn_fused_supported(n_local, d2):
                # a single rank: nothing sits between statistics and apply -- a few-row layer (the heads on 8 B pooled rows) is one launch
                y, mean, rstd = ops.bn_fwd_fused(h, M.bn_eps, gamma, beta, relu=not last, running=(rm, rv, M.bn_momentum))
            else:
                sums = torch.empty((2, d2), device=x.device, dtype=F32)
                ops.bn_stats(h, sums)
                self.comm.all_reduce_(sums)
                y, mean, rstd = ops.bn_fwd_apply(h, sums, n_total, M.bn_eps, gamma, beta, relu=not last, running=(rm, rv, M.bn_momentum))
            self._bn_touched.append(i_bn)
            if save:
                saved.append((x, h, mean, rstd))
            x = y
        return x, saved

    def mlp_forward_pair(self, xa, pre_a, arena_a, save_a, xb, pre_b, arena_b, save_b):
        """Two BN-MLP stacks of the same layer widths (the online head and its momentum twin) run layer by layer in lock step, so
        that their cross-rank BatchNorm statistics travel in ONE all-reduce per layer ([2 stacks, 2, C] instead of two [2, C]
        messages): 14 -> 8 latency-bound collectives per forward, all issued from one stream in program order.  Same arithmetic
        per stack as mlp_forward (the all-reduce is elementwise), used when there is a process group."""
        M = self.m
        dims = M.mlps[pre_a]
        assert dims == M.mlps[pre_b]
        f32 = M._f32
        outs = []
        xs = [xa, xb]
        saved = [[], []]
        for l, (d1, d2) in enumerate(dims):
            last = l == len(dims) - 1
            hs = [ops.linear_fwd(xs[k], M._w(ar)[f"{pre}.{3 * l}.weight"]) for k, (pre, ar) in enumerate(((pre_a, arena_a), (pre_b, arena_b)))]
            sums = torch.empty((2, 2, d2), device=xa.device, dtype=F32)
            for k in range(2):
                ops.bn_stats(hs[k], sums[k])
            self.comm.all_reduce_(sums)
            for k, (pre, save) in enumerate(((pre_a, save_a), (pre_b, save_b))):
                n_total = float(xs[k].shape[0] * self.comm.world)
                gamma = None if last else f32[f"{pre}.{3 * l + 1}.weight"]
                beta = None if last else f32[f"{pre}.{3 * l + 1}.bias"]
                rm, rv, i_bn = M._bn_views[f"{pre}.{3 * l + 1}"]
                y, mean, rstd = ops.bn_fwd_apply(hs[k], sums[k], n_total, M.bn_eps, gamma, beta, relu=not last, running=(rm, rv, M.bn_momentum))
                self._bn_touched.append(i_bn)
                if save:
                    saved[k].append((xs[k], hs[k], mean, rstd))
                xs[k] = y
        return (xs[0], saved[0]), (xs[1], saved[1])

    def mlp_backward(self, dy, pre, saved, need_dx=True, dx_out=None):
        M = self.m
        dims = M.mlps[pre]
        w16, f32, g32 = M._w("online"), M._f32, M._g32
        n_total = float(saved[0][0].shape[0] * self.comm.world)
        for l in reversed(range(len(dims))):
            d1, d2 = dims[l]
            last = l == len(dims) - 1
            x, h, mean, rstd = saved[l]
            gamma = None if last else f32[f"{pre}.{3 * l + 1}.weight"]
            beta = None if last else f32[f"{pre}.{3 * l + 1}.bias"]
            if self.comm is LOCAL and ops.bn_fused_supported(h.shape[0], d2):
                dh = ops.bn_bwd_fused(dy, h, mean, rstd, gamma, beta, not last, None if last else g32[f"{pre}.{3 * l + 1}.bias"],
                                      None if last else g32[f"{pre}.{3 * l + 1}.weight"])
            else:
                sums = torch.empty((2, d2), device=dy.device, dtype=F32)
                # (the LOCAL sums are the affine gradients: accumulated by the statistics launch itself)
                ops.bn_bwd_stats(dy, h, mean, rstd, gamma, beta, not last, sums, None if last else g32[f"{pre}.{3 * l + 1}.bias"],
                                 None if last else g32[f"{pre}.{3 * l + 1}.weight"])
                self.comm.all_reduce_(sums)
                dh = ops.bn_bwd_apply(dy, h, mean, rstd, gamma, beta, not last, sums, n_total)
            # (every head weight is used once per forward: right after zero_grad() its gradient can be written instead of added)
            asg = getattr(self, "_assign", False)
            self._on_side(dy.device, lambda: ops.linear_wgrad(dh, x, g32[f"{pre}.{3 * l}.weight"], assign=asg), dh, x)
            if l > 0 or need_dx:
                wt = getattr(self, "head_wT", {}).get(f"{pre}.{3 * l}.weight") if ops.HEAD_DGRAD_DIRECT else None
                if wt is not None:
                    # direct form on the K-contiguous copy the optimizer launch left (the forward's tile plan for this shape)
                    dy = ops.linear_fwd(dh, wt, out=dx_out if l == 0 else None)
                else:
                    dy = ops.linear_dgrad(dh, w16[f"{pre}.{3 * l}.weight"], out=dx_out if l == 0 else None)
        return dy

    # ------------------------------------------------------------------ full forward
    def forward(self, images, aug, mask_b2n, m, mim_views=1, training=False):
        """training: a backward will follow (set by the autograd node: inside its forward() autograd's grad mode is off, so
        torch.is_grad_enabled() says nothing).
        mim_views: 1 = only_mim_on_ori_img (the README recipe: the decoder runs on view 0's masked rows), 2 = both views' masked rows
        (modeling_pretrain_moco_mim_ori.py:572-577)."""
        M = self.m
        dev = images.device
        B, D, N, nw = images.shape[0], M.D, M.N, M.num_windows
        comm = self.comm
        self.B = B
        self._bn_touched = []
        images = images.contiguous().float()
        aug = aug.contiguous().float()
        if mask_b2n.dtype == torch.uint8 and mask_b2n.dim() == 2 and tuple(mask_b2n.shape) == (2 * B, N):
            mask_u8 = mask_b2n if mask_b2n.is_contiguous() else mask_b2n.contiguous()      # prepared by the engine: view-major rows already
        else:
            mask_u8 = mask_b2n.permute(1, 0, 2).reshape(2 * B, N).to(torch.uint8).contiguous()    # rows 0..B-1 = view 0 (:497)
        if not M.use_pixel_target:
            # Dis-only: `if not self.use_pixel_target: vis_mask_pos = None` (modeling_pretrain_moco_mim_ori.py:493-494) -- no token is replaced
            zm = getattr(M, "_zero_mask", None)
            if zm is None or zm.shape != (2 * B, N) or zm.device != dev:
                zm = M._zero_mask = torch.zeros((2 * B, N), device=dev, dtype=torch.uint8)
            mask_u8 = zm
        self.images, self.aug, self.mask_u8 = images, aug, mask_u8
        ew_on, ew_mo = _weights(M)
        pp = M.has_pix_projector
        _mark("forward: start", dev)
        # the bf16 operand shadow of the online arena (and the transposed weight copies below): left by the previous step's optimizer launch
        # unless something else has written the parameters since (MoCo_ViT.weights_fresh)
        fresh = M.weights_fresh() and not torch.cuda.is_current_stream_capturing()
        if not fresh:
            ops.cast_f32_to_bf16(M._flat["online"], M.shadow("online"))
        # stochastic depth (--drop_path): this step's keys; the online encoder's specs are kept for the backward
        self.path_on = self.path_mo = None
        if M.drop_path_rate > 0 and M.training:
            from . import dropout as DR
            plan = DR.DropPlan(M.drop_seed, M.drop_step)
            M.drop_step += 1
            self.path_on, self.path_mo = self._path_specs(plan, 0), self._path_specs(plan, 64)
        if not M.use_moco_target:
            return self._forward_gen_only(ew_on, images, aug, mask_u8, mim_views, training, fresh)
        # ---- momentum branch (no grad) on a second HIP stream: it depends only on the pre-step online weights (fp32
        # arena, read-only here) and the inputs, so it overlaps the online forward.  EMA with the current online weights
        # comes first (:526).
        main = torch.cuda.current_stream(dev)
        side = M._fwd_stream(dev) if (getattr(M, "overlap_streams", True) and FWD_MODE != "serial") else main
        dist_mode = comm.world > 1 or getattr(comm, "world_override", False)
        mode = FWD_MODE if (not dist_mode and side is not main) else "side"

        regular = M.patchnet == 'regular'

        def extract(masked, enc_rows, pre, arena, save):
            """patch_extractor (PatchNet.forward, :189-205): the pooled windows of [masked view | augmented view]; with --patchnet_name regular
            they then attend over all tokens of their image (dig_amd/patchnet.py)."""
            if M.patchnet == 'conv':
                # ConvPatchNet (:207-260): no window pooling in front -- the token maps of both views through the convolution stack, one row per image
                from . import convpatchnet
                feat = torch.cat([masked, enc_rows[B * N:]]) if masked.data_ptr() != enc_rows.data_ptr() else enc_rows
                return convpatchnet.forward(self, feat, pre, arena, 2 * B, save)
            pooled = torch.empty((2 * B * nw, D), device=dev, dtype=BF16)
            ops.window_pool_fwd(masked, pooled[:B * nw], B, M.gh, M.gw, nw, D)
            ops.window_pool_fwd(enc_rows[B * N:], pooled[B * nw:], B, M.gh, M.gw, nw, D)
            if not regular:
                return pooled, None
            from . import patchnet
            # (the image tokens of both views as one [2 B N, D] matrix: without a pix_projector that is the encoder output itself)
            feat = torch.cat([masked, enc_rows[B * N:]]) if masked.data_ptr() != enc_rows.data_ptr() else enc_rows
            return patchnet.forward(self, feat, pooled, pre, arena, 2 * B, save)

        def online_heads(enc):
            # (`if hasattr(self, 'pix_projector')`, :500-510: the Dis-only models pool the encoder's own rows of both views)
            masked2, self.saved_pix = self.mlp_forward(enc[:B * N], "pix_projector", "online", True) if pp else (enc[:B * N], None)
            pooled, self.saved_pnet = extract(masked2, enc, "patch_extractor", "online", True)
            q, self.saved_proj = self.mlp_forward(pooled, "encoder_projection_layer", "online", True)
            q, self.saved_pred = self.mlp_forward(q, "predictor", "online", True)
            return q

        def momentum_heads(enc_m):
            masked_m = self.mlp_forward(enc_m[:B * N], "pix_projector_m", "momentum", False)[0] if pp else enc_m[:B * N]
            pooled_m, _ = extract(masked_m, enc_m, "momentum_patch_extractor", "momentum", False)
            k, _ = self.mlp_forward(pooled_m, "momentum_projection_layer", "momentum", False)
            return k

        def momentum_branch(heads=True):
            ops.ema_update(M._flat["momentum"], M._flat["online"], M.shadow("momentum"), M.n_ema, m)
            self.wT = None
            if training and ((ops.mlp_chain_supported(D, M.F, 2 * B * N) and (ops.MLP_CHAIN_MASK & 4)) or
                             (ops.DGRAD_DIRECT and 2 * B * N >= 8192)):
                # K-contiguous copies of the online weights for the backward (the fused MLP backward's operands; the direct-form data
                # gradients): left by the optimizer launch, or rebuilt here in front of the momentum encoder from this step's shadow
                self.wT = self.mlp_weight_transposes(ew_on, fresh)
            enc_m, _ = self.encoder_forward(ew_mo, images, aug, mask_u8, False, path=self.path_mo)
            return enc_m, (momentum_heads(enc_m) if heads else None)

        def decoder():
            # SimMIM decoder on the masked tokens (:560-570; the reference decodes all rows then selects): view 0 only, or both views
            return self._decoder_forward(mask_u8, images, mim_views), None

        vis_out = None
        if mode == "flip":
            # The big forward kernels own the whole chip one at a time (persistent GEMM tiles and the fused MLP chain: one workgroup per CU
            # with all of its LDS / registers), so the two branches serialise kernel by kernel whatever streams they are on, and the question
            # is only WHO goes first: the online branch on the HIGH-priority stream, the momentum branch on the caller's (23.76 -> 23.58 ms).
            # Workgroup dispatch is strictly by priority: the momentum branch's first kernel starts when the online branch is ~0.5 ms into its
            # ~60 head kernels (rocprofv3 timeline).  Measured alternatives, same box: both encoders on one stream with the online heads and
            # the SimMIM decoder on the other at higher (24.32 ms), equal (24.33) or lower (24.21) priority against 23.74 -- a small kernel
            # that holds a few CUs when a persistent one-workgroup-per-CU kernel starts delays that whole kernel by its own length, 72 times.
            hi_st = M._side_stream(dev)
            hi_st.wait_stream(main)
            with torch.cuda.stream(hi_st):
                enc, self.saved_enc = self.encoder_forward(ew_on, images, aug, mask_u8, True, path=self.path_on)
                self.enc = enc
                qs = online_heads(enc)
            # (Measured and not kept, round 4: the SimMIM decoder right behind the online heads instead of after the join: 21.06 vs 21.07 ms.)
            # (Measured and not kept, round 4: the online heads + SimMIM decoder held back until both encoders are done, so that they run
            #  beside the momentum heads instead of between the encoders: 21.84 vs 21.77 ms, two A/B pairs.)
            enc_m, ks = momentum_branch()
            del enc_m
            main.wait_stream(hi_st)
        else:
            # momentum branch (no grad) on a second HIP stream: it depends only on the pre-step online weights (fp32 arena, read-only
            # here) and the inputs.  EMA with the current online weights comes first (:526).
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc_m, ks = momentum_branch(heads=not dist_mode)
            enc, self.saved_enc = self.encoder_forward(ew_on, images, aug, mask_u8, True, path=self.path_on)
            self.enc = enc
            if not dist_mode:
                qs = online_heads(enc)
                main.wait_stream(side)
                if side is not main:
                    ks.record_stream(main)
                del enc_m
            else:
                # with a process group the heads of both branches run together on this stream, after both encoders: one BatchNorm-
                # statistics all-reduce per layer PAIR, issued in program order (no collective of the momentum branch in front of the
                # online branch's first one on RCCL's in-order stream)
                main.wait_stream(side)
                enc_m.record_stream(main)
                if pp:
                    (masked2, self.saved_pix), (masked_m, _) = self.mlp_forward_pair(enc[:B * N], "pix_projector", "online", True,
                                                                                      enc_m[:B * N], "pix_projector_m", "momentum", False)
                else:
                    masked2, self.saved_pix, masked_m = enc[:B * N], None, enc_m[:B * N]
                pooled, self.saved_pnet = extract(masked2, enc, "patch_extractor", "online", True)
                pooled_m, _ = extract(masked_m, enc_m, "momentum_patch_extractor", "momentum", False)
                (qs, self.saved_proj), (ks, _) = self.mlp_forward_pair(pooled, "encoder_projection_layer", "online", True,
                                                                        pooled_m, "momentum_projection_layer", "momentum", False)
                qs, self.saved_pred = self.mlp_forward(qs, "predictor", "online", True)
                del enc_m, masked_m, pooled_m
        _mark("forward: both encoders + heads joined", dev)
        M._flat["bn_count"] += 1                                            # all 14 (Dis-only: 8) BatchNorm layers ran once
        # ---- InfoNCE (:444-461): q1 vs gathered k2, q2 vs gathered k1, labels = arange + n*rank
        n = B * M.n_patch                                                   # rows of q1 / q2
        dim = M.moco_dim
        qf = torch.empty((2 * n, dim), device=dev, dtype=F32)
        kf = torch.empty((2 * n, dim), device=dev, dtype=F32)
        ops.cast_bf16_to_f32(qs, qf)
        ops.cast_bf16_to_f32(ks, kf)
        qn, self.q_inv = ops.l2norm_fwd(qf)
        kn, _ = ops.l2norm_fwd(kf)
        self.qn = qn
        if dist_mode:
            kall = comm.all_gather_cat(kn.view(1, 2, n, dim))               # [W, 2, n, dim], rank order (:586-590)
            k1_all = kall[:, 0].reshape(comm.world * n, dim).contiguous()
            k2_all = kall[:, 1].reshape(comm.world * n, dim).contiguous()
        else:
            k1_all, k2_all = kn[:n], kn[n:]
        mk = comm.world * n
        stats = torch.zeros((2, 3), device=dev, dtype=F32)
        self.dqn = torch.empty((2 * n, dim), device=dev, dtype=F32)
        gs = 2.0 * M.T / n                                                  # d(mean CE * 2T)/dlogits scale
        for half, kk in ((0, k2_all), (1, k1_all)):
            logits = torch.empty((n, mk), device=dev, dtype=F32)
            ops.sgemm(qn[half * n:(half + 1) * n], kk, logits, n, mk, dim, False, 1.0 / M.T)
            ops.ce_rows(logits, n * comm.rank, gs, stats[half])
            ops.sgemm(logits, kk, self.dqn[half * n:(half + 1) * n], n, dim, mk, True, 1.0 / M.T)
        contra, accs = ops.infonce_finish(stats, 2.0 * M.T / n, 100.0 / n)   # accs: q1_acc1, q1_acc5, q2_acc1, q2_acc5
        if not M.use_pixel_target:
            vis_out = torch.empty((0, 0, M.dec_classes), device=dev, dtype=F32)     # (no 'vis_out' key for a Dis-only model: dig_forward)
            self.mim_views = 0
        elif vis_out is None:
            vis_out, _ = decoder()
        _mark("forward: InfoNCE + SimMIM decoder done", dev)
        return contra, accs, vis_out

    def _decoder_forward(self, mask_u8, images, mim_views):
        """pix_decoder on the masked rows of self.enc (modeling_pretrain_moco_mim_ori.py:560-570; the reference decodes all rows, then selects)."""
        M = self.m
        B, N, dev = images.shape[0], M.N, images.device
        per = M._mask_count(mask_u8, B)
        Mrows = mim_views * B * per
        Mp = (Mrows + 63) // 64 * 64
        idx, cnt = ops.mask_to_index(mask_u8[:mim_views * B], per)
        M._last_mask_counts, M._last_idx, M._last_images = cnt, idx[:B], images
        M._last_idx_views = [idx[:B]] + ([(idx[B:] - B * N).clamp_min_(0)] if mim_views == 2 else [])
        self.idx, self.Mrows, self.Mp, self.per, self.mim_views = idx, Mrows, Mp, per, mim_views
        w16, f32 = M._w("online"), M._f32
        gath = ops.gather_rows(self.enc, idx, Mrows, Mp)
        h0 = ops.linear_fwd(gath, w16["pix_decoder.0.weight"])
        h1 = ops.linear_fwd(h0, w16["pix_decoder.1.weight"])
        h2, mu, rs = ops.layernorm_fwd(h1, f32["pix_decoder.2.weight"], f32["pix_decoder.2.bias"], M.ln_eps, gelu=True)
        pred = torch.empty((Mp, 64), device=dev, dtype=F32)
        C = M.dec_classes
        ops.gemm(h2, w16["pix_decoder.4.weight"], Mp, C, M.dec_dim, out=pred, out_kind=ops.OUT_F32, bias=f32["pix_decoder.4.bias"], ldc=64)
        self.saved_dec = (gath, h0, h1, h2, mu, rs)
        return pred[:Mrows, :C].reshape(mim_views * B, per, C)

    def _forward_gen_only(self, ew_on, images, aug, mask_u8, mim_views, training, fresh):
        """MoCo_ViT.forward of a Gen-only model (use_moco_target=False: pretrain_simmim_ori_*, modeling_pretrain_moco_mim_ori.py:655-681):
        encoder -> its final LayerNorm (modeling_pretrain_vit.py:104; the moco branch is what replaces it by nn.Identity) -> pix_decoder on
        the masked rows.  The reference runs the augmented view through the encoder as well and reads none of its rows when
        only_mim_on_ori_img: the samples of a ViT batch are independent, so that view is not launched here (3 F per sample instead of 6)."""
        M = self.m
        dev = images.device
        views = mim_views
        self.gen_views = views
        self.wT = None
        if training and ((ops.mlp_chain_supported(M.D, M.F, views * self.B * M.N) and (ops.MLP_CHAIN_MASK & 4)) or
                         (ops.DGRAD_DIRECT and views * self.B * M.N >= 8192)):
            self.wT = self.mlp_weight_transposes(ew_on, fresh)
        enc_raw, self.saved_enc = self.encoder_forward(ew_on, images, aug, mask_u8, True, views=views, path=self.path_on)
        f32 = M._f32
        self.enc, mu, rs = ops.layernorm_fwd(enc_raw, f32["encoder.norm.weight"], f32["encoder.norm.bias"], M.ln_eps)
        self.saved_norm = (enc_raw, mu, rs)
        vis_out = self._decoder_forward(mask_u8, images, mim_views)
        # (no 'contra_loss' / accuracy keys for a Gen-only model: dig_forward drops these two; an operator's outputs must not alias each other)
        return torch.zeros((), device=dev, dtype=F32), torch.zeros(4, device=dev, dtype=F32), vis_out

    # ------------------------------------------------------------------ full backward
    def backward(self, g_contra, g_vis):
        M = self.m
        B, D, N, nw = self.B, M.D, M.N, M.num_windows
        dev = self.enc.device
        w16, f32, g32 = M._w("online"), M._f32, M._g32
        # ---- contrastive path.  g_contra is None when the loss did not use contra_loss (zero contrastive weight: epochs before
        # contrast_start_epoch, BASELINE config 2): every gradient of that branch -- predictor, projector, pix_projector and
        # the whole augmented view -- is exactly zero in the reference, so nothing is launched for it and the encoder
        # backward runs on view 0's rows only (the gradient arena was zero-filled by optimizer.zero_grad()).
        contrast = g_contra is not None and M.use_moco_target
        if g_vis is not None and not M.use_pixel_target:
            g_vis = None
        # the arena is known to be zero only right after optimizer.zero_grad(): a second backward() without it accumulates, as torch does
        self._assign, M._grads_fresh = bool(getattr(M, "_grads_fresh", False)), False
        _mark("backward: start (loss, MSE, autograd entry done)", dev)
        views = 2 if (contrast or (g_vis is not None and self.mim_views == 2)) else 1      # encoder rows that carry a gradient
        # (Dis-only: no pix_projector writes view 0's rows -- both halves come from the pooling gradient, which overwrites)
        d_enc = torch.empty_like(self.enc) if contrast else torch.zeros((views * B * N, D), device=dev, dtype=BF16)
        n = B * M.n_patch
        if contrast:
            dqn = self.dqn
            ops.scale_by_device_scalar(dqn, g_contra.reshape(1).float())
            dq = ops.l2norm_bwd(dqn, self.qn, self.q_inv)
            dq16 = torch.empty(dq.shape, device=dev, dtype=BF16)
            ops.cast_f32_to_bf16(dq, dq16)
            dproj = self.mlp_backward(dq16, "predictor", self.saved_pred)
            self._grad_ready(dev, "predictor")
            dpool = self.mlp_backward(dproj, "encoder_projection_layer", self.saved_proj)
            self._grad_ready(dev, "encoder_projection_layer")
            acc = False
            if M.patchnet == 'regular':
                # the patch transformer's backward: d(pooled windows) and the gradient w.r.t. the image tokens through both blocks' keys /
                # values -- [masked view | augmented view] rows, written where the pooling gradient is then ADDED
                from . import patchnet
                dpool, dfeat = patchnet.backward(self, dpool, "patch_extractor", self.saved_pnet, 2 * B)
                self._grad_ready(dev, "patch_extractor")
                d_enc, acc = dfeat, True
            if M.patchnet == 'conv':
                # ConvPatchNet's backward hands the gradient w.r.t. the token maps of [masked view | augmented view] straight back
                from . import convpatchnet
                d_enc = convpatchnet.backward(self, dpool, "patch_extractor", self.saved_pnet, 2 * B)
                self._grad_ready(dev, "patch_extractor")
                if M.has_pix_projector:
                    self.mlp_backward(d_enc[:B * N], "pix_projector", self.saved_pix, dx_out=d_enc[:B * N])
                    self._grad_ready(dev, "pix_projector")
            elif M.has_pix_projector:
                dmasked2 = d_enc[:B * N] if acc else torch.empty((B * N, D), device=dev, dtype=BF16)
                ops.window_pool_bwd(dpool[:n], dmasked2, B, M.gh, M.gw, nw, D, acc)
                ops.window_pool_bwd(dpool[n:], d_enc[B * N:], B, M.gh, M.gw, nw, D, acc)
                # (regular: the incoming gradient lives in d_enc's own first half; the stack's last data gradient overwrites it when it is done with it)
                self.mlp_backward(dmasked2, "pix_projector", self.saved_pix, dx_out=d_enc[:B * N])
                self._grad_ready(dev, "pix_projector")
            else:
                ops.window_pool_bwd(dpool[:n], d_enc[:B * N], B, M.gh, M.gw, nw, D, acc)
                ops.window_pool_bwd(dpool[n:], d_enc[B * N:], B, M.gh, M.gw, nw, D, acc)
        elif M.use_moco_target:
            for name in (("predictor", "encoder_projection_layer") + (("patch_extractor",) if M.patchnet != 'no_patchtrans' else ())
                         + (("pix_projector",) if M.has_pix_projector else ())):
                self.comm.grad_ready(M, name)
        # ---- SimMIM decoder path
        if g_vis is not None:
            gath, h0, h1, h2, mu, rs = self.saved_dec
            Mrows, Mp, C, Dd = self.Mrows, self.Mp, M.dec_classes, M.dec_dim
            dpred = torch.empty((Mp, 64), device=dev, dtype=BF16)
            ops.pad_cast_rows(g_vis.reshape(Mrows, C).contiguous().float(), dpred, Mrows, C)
            self._on_side(dev, lambda: (ops.wgrad(dpred, h2, g32["pix_decoder.4.weight"], C, Dd, Mp),
                                        ops.colsum(dpred, g32["pix_decoder.4.bias"], cols=C)), dpred, h2)
            dh2 = ops.gemm(dpred, w16["pix_decoder.4.weight"], Mp, Dd, 64, tb=True, b_rows=C)
            dh1 = ops.layernorm_bwd(dh2, h1, f32["pix_decoder.2.weight"], f32["pix_decoder.2.bias"], mu, rs, None,
                                    g32["pix_decoder.2.weight"], g32["pix_decoder.2.bias"], gelu=True)
            self._on_side(dev, lambda: ops.linear_wgrad(dh1, h0, g32["pix_decoder.1.weight"]), dh1, h0)
            dh0 = ops.linear_dgrad(dh1, w16["pix_decoder.1.weight"])
            self._on_side(dev, lambda: ops.linear_wgrad(dh0, gath, g32["pix_decoder.0.weight"]), dh0, gath)
            dgath = ops.linear_dgrad(dh0, w16["pix_decoder.0.weight"])
            ops.scatter_rows_add(dgath, self.idx, d_enc, Mrows)
        if M.use_pixel_target:
            self._grad_ready(dev, "pix_decoder")
        if M.has_final_norm:
            # Gen-only: the encoder's final LayerNorm between the blocks and the decoder (modeling_pretrain_vit.py:104)
            if g_vis is not None:
                enc_raw, mu, rs = self.saved_norm
                d_enc = ops.layernorm_bwd(d_enc, enc_raw, f32["encoder.norm.weight"], f32["encoder.norm.bias"], mu, rs, None,
                                          g32["encoder.norm.weight"], g32["encoder.norm.bias"])
            self._grad_ready(dev, "encoder.norm")
            self.saved_norm = None
        # ---- encoder
        ew_on, _ = _weights(M)
        _mark("backward: heads + decoder done", dev)
        if views == 2 or g_vis is not None:
            self.encoder_backward(ew_on, self.saved_enc, d_enc, self.images, self.aug, self.mask_u8, views=views)
        else:
            for i in reversed(range(M.depth)):
                self.comm.grad_ready(M, f"encoder.blocks.{i}")
            self.comm.grad_ready(M, "encoder.embed")
        main, side = self._streams(dev)
        if side is not main:
            main.wait_stream(side)                  # every gradient is final on the caller's stream (grad norm / AdamW follow)
        _mark("backward: encoder done, streams joined", dev)
        self._keep.clear()                          # (blocks go back to the caller's stream's pool: its later work is ordered behind the join)
        self._keep_marks.clear()
        self.saved_enc = self.saved_pix = self.saved_proj = self.saved_pred = self.saved_dec = self.saved_norm = self.saved_pnet = self.wT = None


class _DigFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, images, aug, mask, m, mim_views):
        step = _Step(model)
        contra, accs, vis_out = step.forward(images, aug, mask, m, mim_views, training=True)
        ctx.step = step
        ctx.set_materialize_grads(False)          # an unused output arrives as None, not as a zero tensor (host-visible)
        ctx.mark_non_differentiable(accs)
        return contra, accs, vis_out

    @staticmethod
    def backward(ctx, g_contra, g_accs, g_vis):
        step, ctx.step = ctx.step, None
        step.backward(g_contra, g_vis)
        return None, None, None, None, None, None, None


# ---- the step as registered PyTorch operators ---------------------------------------------------------------------------------------------
# `dig::pretrain_step_fwd` / `dig::pretrain_step_bwd` (torch.library): what MoCo_ViT.forward dispatches and what its autograd formula calls.
# The operators take the online parameter arena and the gradient arena as tensor arguments, the model object (and with it the state its
# forward updates: momentum parameters, BatchNorm running statistics) through a registry key, and hand the activations from forward to backward
# through a step handle.  They are
# the same code the autograd.Function form runs (DIG_STEP_OPS=0) -- one dispatcher hop per direction (~40 us of host time per step).
# No fake (meta) implementation: the SimMIM output's row count is the number of masked tokens, a value read from the mask on the host.
STEP_OPS = os.environ.get("DIG_STEP_OPS", "1") != "0"
_MODELS = weakref.WeakValueDictionary()
_LIVE_STEPS = {}
_handles = itertools.count(1)


@torch.library.custom_op("dig::pretrain_step_fwd", mutates_args=(), device_types="cuda")
def pretrain_step_fwd(anchor: torch.Tensor, online: torch.Tensor, images: torch.Tensor, aug: torch.Tensor, mask: torch.Tensor, m: float,
                      m_dev: Optional[torch.Tensor], mim_views: int, model_key: int, handle: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(contra_loss, [q1_acc1, q1_acc5, q2_acc1, q2_acc5], vis_out) of MoCo_ViT.forward (modeling_pretrain_moco_mim_ori.py:484-592).
    handle != 0: a backward will follow; the step's saved activations wait under that handle.
    Like the reference's forward (`_momentum_update_key_encoder` :526, BatchNorm running statistics) it updates MODULE STATE -- the momentum
    arena and the running statistics of the model behind `model_key` -- which is not part of the operator's argument list (torch.library
    registers autograd formulas for operators without mutated ARGUMENTS only)."""
    step = _Step(_MODELS[model_key])
    contra, accs, vis_out = step.forward(images, aug, mask, m_dev if m_dev is not None else m, mim_views, training=handle != 0)
    if handle:
        _LIVE_STEPS[handle] = step
    return contra, accs, vis_out


@torch.library.custom_op("dig::pretrain_step_bwd", mutates_args=("grads",), device_types="cuda")
def pretrain_step_bwd(grads: torch.Tensor, g_contra: Optional[torch.Tensor], g_vis: Optional[torch.Tensor], handle: int) -> None:
    """Accumulates the gradients of every online parameter into `grads` (the flat arena the parameters' .grad are views of)."""
    _LIVE_STEPS.pop(handle).backward(g_contra, g_vis)


def _step_setup_context(ctx, inputs, output):
    ctx.handle, ctx.model_key = inputs[-1], inputs[-2]
    ctx.step = _LIVE_STEPS.pop(ctx.handle, None)  # the saved activations live and die with the autograd graph, not with the table
    ctx.set_materialize_grads(False)              # an unused output arrives as None, not as a zero tensor (host-visible)
    ctx.mark_non_differentiable(output[1])


def _step_backward(ctx, g_contra, g_accs, g_vis):
    step, ctx.step = ctx.step, None
    if step is None:
        raise RuntimeError("dig::pretrain_step_fwd: backward called twice (the step's activations are released by the first backward)")
    _LIVE_STEPS[ctx.handle] = step
    torch.ops.dig.pretrain_step_bwd(_MODELS[ctx.model_key].flat_grads, g_contra, g_vis, ctx.handle)
    return (None,) * 10


pretrain_step_fwd.register_autograd(_step_backward, setup_context=_step_setup_context)


def dig_forward(model, image, aug_image, vis_mask_pos, m, only_mim_on_ori_img=True):
    if not image.is_cuda:
        raise RuntimeError("dig_amd.MoCo_ViT runs on an MI355X (cuda device) only; there is no CPU fallback")
    if model._flat["online"].device != image.device:
        raise RuntimeError("model and inputs are on different devices (call model.to(device))")
    mim_views = 1 if only_mim_on_ori_img else 2
    mask = vis_mask_pos
    if mask.dim() == 2 and not (mask.dtype == torch.uint8 and tuple(mask.shape) == (2 * image.shape[0], model.N)):
        mask = mask.view(image.shape[0], -1, model.N)           # ([2 B, N] uint8 = the engine's prepared view-major rows: taken as they are)
    m = m if isinstance(m, torch.Tensor) else float(m)          # a device [m, 1-m] pair under graph capture (step_graph.py)
    anchor = getattr(model, "_anchor", None)
    if anchor is None or anchor.device != image.device:
        anchor = model._anchor = torch.zeros(1, device=image.device, requires_grad=True)
    if STEP_OPS:
        # The operator is EAGER-ONLY: it updates module state that is not in its argument list (momentum arena, BatchNorm statistics, the
        # bf16 weight shadows) and its third output's shape depends on the mask, so a tracing compiler must neither dedupe / reorder it nor
        # see it at all.
        if torch.compiler.is_compiling():
            raise RuntimeError("dig::pretrain_step_fwd is an eager-only operator (it updates the momentum encoder and the BatchNorm statistics "
                               "of the module): call the model outside torch.compile")
        key = id(model)
        _MODELS[key] = model
        handle = next(_handles) if torch.is_grad_enabled() else 0
        try:
            contra, accs, vis_out = torch.ops.dig.pretrain_step_fwd(
                anchor, model._flat["online"], image, aug_image, mask,
                0.0 if isinstance(m, torch.Tensor) else m, m if isinstance(m, torch.Tensor) else None, mim_views, key, handle)
        finally:
            _LIVE_STEPS.pop(handle, None)           # (normally taken over by the autograd context in _step_setup_context: never left behind)
    elif torch.is_grad_enabled():
        contra, accs, vis_out = _DigFn.apply(anchor, model, image, aug_image, mask, m, mim_views)
    else:
        contra, accs, vis_out = _Step(model).forward(image, aug_image, mask, m, mim_views)
    B = image.shape[0]
    out = {}
    if model.use_moco_target:               # (the reference fills these keys only `if self.use_moco_target`, modeling_pretrain_moco_mim_ori.py:512-558)
        out.update({"contra_loss": contra, "q1_acc1": accs[0:1], "q1_acc5": accs[1:2], "q2_acc1": accs[2:3], "q2_acc5": accs[3:4]})
    if model.use_pixel_target:              # (:560-577)
        out["vis_out"] = [vis_out] if mim_views == 1 else [vis_out[:B], vis_out[B:]]
    return out
