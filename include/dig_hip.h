/* dig_hip.h -- C ABI of libdig_hip.so: the MI355X (gfx950) kernels behind the DiG pre-training hot path.
 *
 * The reference (ayumiymk/DiG) is pure PyTorch: it has no plugin / FFI boundary of its own.  The boundary this
 * library offers sits one level below the three Python surfaces the reference exposes (model factory, step engine,
 * optimizer -- SURVEY.md section 8b): each entry point replaces the ATen/cuDNN/cuBLAS work that ONE reference
 * expression launches, and the comment on each declaration cites that expression (paths relative to the reference
 * repository).  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller (the library never allocates, frees
 *     or keeps a pointer); scratch is caller-allocated (see the *_workspace_bytes queries);
 *   - `stream` is the HIP stream to launch on (pass the caller's current stream); launches are asynchronous, no
 *     host synchronisation happens inside; callable from any host thread (forward thread or autograd worker);
 *   - return value: 0 = launched; -1 bad argument, -2 misaligned pointer / leading dimension, -3 launch failed,
 *     -4 unsupported configuration.  Nothing throws across the ABI;
 *   - "bf16" buffers are raw bfloat16 bits (uint16), row-major; statistics / losses / optimizer state are fp32;
 *   - integer work (mask -> index, gathers) is bit-exact; floating point follows the tolerances in DESIGN.md.
 */
#ifndef DIG_HIP_H
#define DIG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Matrix-core GEMM.  Replaces F.linear / nn.Linear and its autograd (modeling_finetune.py:53-60,91-93,119;
 * modeling_pretrain_moco_mim_ori.py:414-426,463-482).
 *   C[I,J] = sum_r opA(i,r) * opB(j,r)
 *   trans_a = 0: A is [I, R] (r contiguous);  trans_a = 1: A is [R, I] (r is the row index)      (same for B / J)
 *   forward  y = x W^T   : trans_a 0, trans_b 0;   dgrad dx = dy W : trans_a 0, trans_b 1;   wgrad dW = dy^T x : 1, 1
 *   out_kind 0: bf16 C;  1: fp32 C;  2: fp32 partial slabs C[splits][I][J] (split over R, ldc must equal J) to be summed
 *               by dig_reduce_partials (bias/resid/act must be unset);  `splits` must come from dig_gemm_effective_splits
 *   epilogue (out_kind 0/1): v = (acc + bias[j]) * (j < alpha_cols ? alpha : 1);
 *               act 1: store v to pre_act (bf16, optional) then v = gelu_erf(v);  act 2: v *= gelu'(resid[i,j]);
 *               resid (bf16, act != 2): v += resid[i,j]
 *   requirements: J % 8 == 0; lda, ldb, ldc, ldr, ldp % 8 == 0; 16-byte aligned pointers; R % 64 == 0 for a non-transposed
 *   operand; a_rows / b_rows (0 = default) = rows of A / B that exist in memory (rows beyond read as zero).
 *   bk = tile variant, one of DIG_GEMM_TILE_* below (0 = default: 128x128 with K step 64 when both operands are read along
 *   their contiguous axis, 32 otherwise).
 *   colsum_partials (act 2 only): [ceil(I/64)][J] fp32 column sums of the result per 64-row group, or null.
 */
#define DIG_GEMM_TILE_DEFAULT 0
#define DIG_GEMM_TILE_128x128_K32 32   /* 4 waves; transposed-operand layers (dgrad / wgrad) */
#define DIG_GEMM_TILE_128x128_K64 64   /* 4 waves; forward layers */
#define DIG_GEMM_TILE_256x256 244      /* 16 waves, 128 KiB LDS: tall forward layers (half the operand traffic per FLOP) */
#define DIG_GEMM_TILE_256x192 264      /* 12 waves: tall layers whose width is 384 (a multiple of 192, not of 256) */
#define DIG_GEMM_TILE_64x128 212       /* 2 waves: few rows, long K (projection-head layers) */
#define DIG_GEMM_TILE_128x64 221       /* 2 waves: few rows (projection-head dgrads) */
/* Persistent forms of the two tall tiles: one workgroup per CU walks a contiguous range of tiles and keeps the next tile's first
 * operand stage in flight during the epilogue; same arithmetic order, bit-identical results.  Forward only (trans_a = trans_b = 0),
 * out_kind 0, act 0 / 1, no colsum_partials; ldr == ldc, ldp == ldc, I * ldc < 2^30, no residual together with pre_act, and no
 * residual at all for the 256x256 form.  A call outside those limits -- as well as transposed operands or fp32 / partial outputs with
 * these codes -- runs the one-tile-per-workgroup kernel of the same shape (244 / 264): same results, no error. */
#define DIG_GEMM_TILE_256x256_PERSISTENT 544
#define DIG_GEMM_TILE_256x192_PERSISTENT 564
int dig_gemm_bf16(const void* A, const void* B, void* C, int I, int J, int R, int lda, int ldb, int ldc, int trans_a,
                  int trans_b, int out_kind, const float* bias, const void* resid, int ldr, void* pre_act, int ldp, float alpha,
                  int alpha_cols, int act, int splits, int a_rows, int b_rows, int bk, float* colsum_partials,
                  hipStream_t stream);
int dig_gemm_effective_splits(int R, int splits);
/* out[e] (+)= sum_s partials[s][e], e < n  (deterministic split-R combine; accumulate=1 adds into the gradient arena) */
int dig_reduce_partials(const float* partials, int splits, long long n, float* out, int accumulate, hipStream_t stream);
/* out_bf16[e] = bf16(sum_s partials[s][e]), e < n (n % 4 == 0): the combine of a split-R FORWARD layer or DATA gradient (dig_gemm_bf16 with
 * out_kind 2 and non-transposed A: tile codes 212 / 221) -- the few-row, narrow-output, long-K layers of the BatchNorm-MLP heads
 * (1024 x 256 outputs over K = 4096: 32 workgroups walking 64 K-steps each become 256) */
int dig_reduce_partials_bf16(const float* partials, int splits, long long n, void* out, hipStream_t stream);
/* out += sum_s partials[s] for up to DIG_REDUCE_MAX_SEGS slab sets in one launch (the weight gradients of one encoder block:
 * 4 reduce launches -> 1; same fixed summation order per element as dig_reduce_partials with accumulate = 1).  `segs` is host memory. */
#define DIG_REDUCE_MAX_SEGS 8
typedef struct dig_reduce_seg { const float* partials; float* out; long long n; int splits; int reserved; } dig_reduce_seg_t;
int dig_reduce_partials_multi(const dig_reduce_seg_t* segs, int n_segs, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Grouped weight gradients (round 4).  Replaces the autograd of the four nn.Linear layers of a transformer Block
 * (modeling_finetune.py:53-60 Mlp.fc1 / fc2, :91-93 Attention.qkv, :119 Attention.proj): dW = dy^T x with the TOKEN rows as the
 * reduction dimension, for a LIST of problems in one launch:
 *     out_p[i, j] (+)= sum_r A_p[r, i] * B_p[r, j]        A_p [R, lda] bf16 = the wide operand  (I_p columns, I_p % 128 == 0)
 *                                                         B_p [R, ldb] bf16 = the narrow operand (J_p columns, J_p % (128 fn) == 0)
 *     trans_out 0: out_p is [I_p, ldo] fp32 (fc1, qkv, proj: A = dy, B = x);  1: out_p is [J_p, ldo] (fc2: A = x, B = dy)
 * fn = 3 (narrow width a multiple of 384: ViT-S) or 2 (a multiple of 256: D = 512), one value per launch; R % 64 == 0.
 * wa = 1: tiles of 128 x 128 fn, 4 waves (each 4 x fn MFMA accumulators), two workgroups per CU (max_wg 512 on MI355X);
 * wa = 2: tiles of 256 x 128 fn, 8 waves, one workgroup per CU (max_wg 256): 0.0065 instead of 0.0104 operand bytes per FLOP from L2,
 * a 7- / 8-slot LDS ring; the last tile of a problem whose I is not a multiple of 256 is computed in full and its surplus rows dropped.
 * One (fn, wa) pair per chain of launches (a fold reads the slab layout of the launch that wrote it).
 * Tiles x S R-splits = one round of workgroups; each
 * workgroup leaves its accumulators as one fp32 slab, and the slabs of launch n are summed in split order (deterministic: no
 * atomics) and added to out_p by launch n + 1 in its prologue -- the kernel boundary is the only synchronisation.  So:
 *     dig_wgrad_group(probs_n, n, probs_{n-1}, m, ...)  computes the partial products of probs_n into `slabs` and folds `fold_slabs`
 *     (written by the previous call with probs_{n-1}, fold_splits) into probs_{n-1}'s gradients;  n_probs = 0 folds only (the
 *     last pending set);  n_fold = 0: nothing pending.  `slabs` and `fold_slabs` must be different buffers (two alternate).
 * dig_wgrad_group_plan builds the workgroup table of a launch on the HOST (tiles of one (problem, split) pair on one XCD, the XCDs
 * loaded evenly) and chooses S; the caller copies map_out to device memory once per configuration and passes it as wg_map.
 * probs / fold_probs / tiles_per_prob / map_out / splits_out are host memory; out / slabs / wg_map device memory. */
#define DIG_WGRAD_MAX_PROBS 6
#include "dig_block_types.h"          /* dig_wgrad_prob_t { A, B, out, lda, ldb, ldo, I, J, trans_out } and the block-call tables */
int dig_wgrad_group_supported(int I, int J, int R);               /* 1: a problem of these sizes can join a group */
int dig_wgrad_group_fn(int J);                                    /* 3, 2, or 0 (narrow width not supported) */
int dig_wgrad_group_rows_per_split(int R, int splits);
int dig_wgrad_group_effective_splits(int R, int splits);
long long dig_wgrad_group_slab_bytes(int total_tiles, int splits, int fn, int wa);
int dig_wgrad_group_tiles(int I, int J, int fn, int wa);           /* tiles of one problem (0: sizes not supported) */
int dig_wgrad_group_plan(const int* tiles_per_prob, int n_probs, int R, int max_wg, int* splits_out, unsigned* map_out, int max_out);
int dig_wgrad_group(const dig_wgrad_prob_t* probs, int n_probs, const dig_wgrad_prob_t* fold_probs, int n_fold, int R, int splits,
                    const unsigned* wg_map, int n_wg, float* slabs, const float* fold_slabs, int fold_splits, int fn, int wa,
                    hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused two-layer MLP ("chain"): Mlp.forward of a transformer block and its data gradient (modeling_finetune.py:53-60 inside
 * Block.forward :150-158) in ONE launch each -- the [R, F] hidden tensor is never a GEMM operand in HBM.
 *   dig_mlp_chain_fwd:  out[R,D] = resid + b2 + gelu_erf(x[R,D] w1[F,D]^T + b1) w2[D,F]^T      (bf16 I/O, fp32 accumulation,
 *       the hidden values rounded to bf16 between the layers exactly as the two-GEMM path rounds them).
 *       pre_out / act_out ([R,F] bf16, both or neither): the pre-activation x w1^T + b1 and its GELU, saved for the backward
 *       (online branch); without them nothing of size [R,F] touches HBM (momentum branch, evaluation).
 *   dig_mlp_chain_bwd:  dpre_out[R,F] = (dy[R,D] w2) * gelu'(pre[R,F]);  dx_out[R,D] = dpre_out w1.  The weights are passed as
 *       K-contiguous copies w2t = w2^T [F,D] and w1t = w1^T [D,F] (dig_transpose_bf16, 1.2 MB each, once per step).
 *       colsum_partials (optional): [dig_mlp_chain_colsum_rows(R)][F] fp32 column sums of dpre_out over blocks of 32 rows -- the
 *       fc1 bias gradient after dig_colsum_partials.
 *   dig_mlp_chain_fwd_ln: the whole second half of Block.forward (modeling_finetune.py:156-158) with its LayerNorms:
 *       out = x + b2 + gelu_erf(LN(x; ln_g, ln_b) w1^T + b1) w2^T  and, when nln_g is given, nln_out = LN(out; nln_g, nln_b) -- the NEXT
 *       block's norm1 (:151), so that no LayerNorm launch and no second pass over the residual stream remain between two blocks.
 *       x holds the RAW residual rows and resid == x; or ln_g == NULL: x holds rows that are normalised already
 *       and resid the raw ones.  LayerNorm statistics are taken in fp32 over the bf16 values that are (or would be) stored,
 *       variance as E[x^2] - E[x]^2, eps inside the square root.  ln_out / ln_mean / ln_rstd (normalised rows [R,D] bf16, statistics
 *       [R] fp32: what the backward keeps) and nln_mean / nln_rstd may be null.  F <= 2048.
 *   dig_mlp_chain_bwd_ln: dig_mlp_chain_bwd with norm2's backward behind it (the first half of Block's gradient, modeling_finetune.py:157-158
 *       read backwards, in one launch): dx_mid_out[R,D] = dy + LN'(dpre_out w1; x_mid, ln_g, ln_mean, ln_rstd) -- the gradient w.r.t. the
 *       rows norm2 normalised, the data gradient of the MLP rounded to bf16 in between exactly as the two-launch path stores it -- and
 *       ln_partials = [dig_mlp_chain_ln_parts(R)][3][D] fp32 partial column sums over blocks of 128 rows: d(gamma), d(beta) of norm2 and the
 *       column sums of dy (= fc2's bias gradient), the layout dig_layernorm_bwd_finalize_parts reduces.  dx_mid_out must not alias dy / x_mid.
 *   Supported widths: dig_mlp_chain_supported(D, F) (D == 384, F a multiple of 128, F <= 6144); R is arbitrary (rows beyond R read as
 *   zero and are not written).  All pointers 16-byte aligned, dense row-major tensors.  Anything else: DIG_ERR_UNSUPPORTED (the
 *   caller runs the two dig_gemm_bf16 launches instead).
 */
int dig_mlp_chain_supported(int D, int F);
int dig_mlp_chain_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid, void* out,
                      void* pre_out, void* act_out, int R, int D, int F, hipStream_t stream);
int dig_mlp_chain_fwd_ln(const void* x, const void* resid, const float* ln_g, const float* ln_b, float eps, void* ln_out, float* ln_mean,
                         float* ln_rstd, const void* w1, const float* b1, const void* w2, const float* b2, void* out, void* pre_out, void* act_out,
                         const float* nln_g, const float* nln_b, void* nln_out, float* nln_mean, float* nln_rstd, int R, int D, int F,
                         hipStream_t stream);
int dig_mlp_chain_bwd(const void* dy, const void* w2t, const void* pre, const void* w1t, void* dpre_out, void* dx_out,
                      float* colsum_partials, int R, int D, int F, hipStream_t stream);
int dig_mlp_chain_bwd_ln(const void* dy, const void* w2t, const void* pre, const void* w1t, void* dpre_out, const void* x_mid,
                         const float* ln_g, const float* ln_mean, const float* ln_rstd, void* dx_mid_out, float* colsum_partials,
                         float* ln_partials, int R, int D, int F, hipStream_t stream);
/* ... and, behind norm2's backward, the attention projection's data gradient on the same rows (Attention.proj, modeling_finetune.py:113 read
 * backwards): dctx_out[R,D] = dx_mid_out proj_w, with projt = proj_w^T [D in, D out] (dig_transpose_bf16).  projt / dctx_out: both or neither
 * (neither: dig_mlp_chain_bwd_ln). */
int dig_mlp_chain_bwd_ln_proj(const void* dy, const void* w2t, const void* pre, const void* w1t, void* dpre_out, const void* x_mid,
                              const float* ln_g, const float* ln_mean, const float* ln_rstd, void* dx_mid_out, float* colsum_partials,
                              float* ln_partials, const void* projt, void* dctx_out, int R, int D, int F, hipStream_t stream);
int dig_mlp_chain_colsum_rows(int R);
int dig_mlp_chain_ln_parts(int R);
/* dst[cols, rows] = src[rows, cols]^T, bf16 */
int dig_transpose_bf16(const void* src, void* dst, int rows, int cols, hipStream_t stream);
/* the same for `count` (<= 32) equally shaped matrices in one launch; srcs / dsts are HOST arrays of device pointers */
int dig_transpose_bf16_multi(const void* const* srcs, void* const* dsts, int count, int rows, int cols, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused self-attention, 256 tokens x head_dim 64 (Attention.forward, modeling_finetune.py:97-118, and its gradient).
 * qkv: bf16 [n_img*256, 3*embed_dim] laid out (q | k | v) x (head, 64) with q already scaled by head_dim^-0.5.
 * ctx: bf16 [n_img*256, embed_dim];  lse: fp32 [n_img*heads, 256] (log-sum-exp of every score row, saved for backward).
 * bwd: dqkv = gradient w.r.t. qkv given dctx; the dq part is multiplied by `scale` (chain rule through the q scaling).
 */
int dig_attn_fwd(const void* qkv, void* ctx, float* lse, int n_img, int heads, int embed_dim, hipStream_t stream);
/* q_colsum / v_colsum (both or neither; fp32 [n_img][embed_dim]): per-image column sums of the dq (scaled) and dv parts,
 * i.e. the q_bias / v_bias gradient partials, finished by dig_colsum_partials(.., n_img, embed_dim, ..). */
int dig_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int n_img, int heads,
                 int embed_dim, float scale, float* q_colsum, float* v_colsum, hipStream_t stream);
/* Which kernel dig_attn_bwd launches for full self-attention (256 query rows) without dropout: 1 = the single-pass kernel (one 8-wave workgroup
 * per (image, head): q / k / v / dctx read once, five matrix products per tile pair, the dQ terms of the eight key blocks summed in LDS in a fixed
 * order), 0 = the two-phase kernel (two 4-wave workgroups per CU, seven products).  Same results up to fp32 summation order; both are
 * bit-reproducible run to run.  Returns the previous setting; any other argument only queries.  Process-wide. */
/* dig_attn_bwd with the attention projection's data gradient inside the launch: dy [R, embed_dim] bf16 = gradient of the projection's OUTPUT rows
 * (Attention.proj, modeling_finetune.py:117), projt = proj.weight^T [in][out] bf16.  Every (image, head) workgroup computes its own d(ctx) tile
 * dy Wproj[:, 64 h ..] on the matrix cores (fp32 accumulation, rounded to bf16 like the GEMM's output) before the backward proper: d(ctx) is
 * never written and the GEMM launch is gone.  embed_dim: a multiple of 128, at most 512.  Two-phase kernel, no dropout, 256 query rows. */
int dig_attn_bwd_proj(const void* qkv, const void* ctx, const void* dy, const void* projt, const float* lse, void* dqkv, int n_img, int heads,
                      int embed_dim, float scale, float* q_colsum, float* v_colsum, hipStream_t stream);
int dig_attn_bwd_mode(int single_pass);
/* How the two-phase kernel's results (dq | dk | dv rows of dqkv) leave: 0 = 16-byte row stores (a lane pair per row: 32 partial-line write
 * requests per instruction), 1 = the same stores, non-temporal, 3 = full 128-byte lines through 2 KiB of LDS per wave, non-temporal (default:
 * 143.5 -> 133.7 us per launch alone, 19.05 -> 19.00 ms in the step).  Bit-identical results.  Returns the previous setting; any other argument
 * only queries.  Process-wide. */
int dig_attn_bwd_store(int mode);

/* The attention sub-block of an encoder block in ONE launch (csrc/attn_block.hip; D = 384, 6 heads of 64, 256 tokens per image):
 *     x_mid = x + proj(softmax(q k^T) v) + proj_b,   (q | k | v) = ln1 qkv_w^T + qkv_b, q scaled by `scale`
 * = Attention.forward (modeling_finetune.py:87-120) + the first residual add of Block.forward (:156): what dig_gemm_bf16 (qkv) ->
 * dig_attn_fwd -> dig_gemm_bf16 (proj + residual) compute, without qkv / ctx as GEMM operands in HBM.  One workgroup per image.
 * ln1, x, x_mid: bf16 [n_img*256, embed_dim]; qkv_w bf16 [3*embed_dim, embed_dim], qkv_b fp32 [3*embed_dim] (q_bias | 0 | v_bias) or NULL;
 * proj_w bf16 [embed_dim, embed_dim], proj_b fp32 [embed_dim] or NULL.  ctx (bf16 [n_img*256, embed_dim]) is always written.
 * qkv (bf16 [n_img*256, 3*embed_dim]) and lse (fp32 [n_img*heads, 256]): both or neither -- what dig_attn_bwd and the weight gradients
 * read (online branch); the values of qkv are bit-identical to the dig_gemm_bf16 launch's.  The softmax runs over two halves of the keys
 * (running maximum, fp32), so ctx / x_mid agree with the three-launch path to bf16 rounding, not bit for bit.
 * dig_attn_block_supported: 1 if (heads, embed_dim) has a kernel (6, 384), else the caller keeps the three launches. */
int dig_attn_block_supported(int heads, int embed_dim);
int dig_attn_block_fwd(const void* ln1, const void* x, const void* qkv_w, const float* qkv_b, const void* proj_w, const float* proj_b,
                       void* qkv, void* ctx, float* lse, void* x_mid, int n_img, int heads, int embed_dim, float scale,
                       hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim D in {64,128,192,256,384,512} (nn.LayerNorm(eps=1e-6): modeling_finetune.py:134,140;
 * pix_decoder LN + GELU: modeling_pretrain_moco_mim_ori.py:424-425 when fuse_gelu=1).
 * fwd: y = [gelu](LN(x)); mean/rstd [rows] saved.  bwd: dx = [dres +] dLN(dy); dgamma/dbeta += ; dcolsum (optional) +=
 * column sums of dres (bias gradient of the layer feeding the residual); workspace: dig_layernorm_bwd_workspace_bytes.
 */
int dig_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int D,
                      float eps, int fuse_gelu, hipStream_t stream);
long long dig_layernorm_bwd_workspace_bytes(int rows, int D);
int dig_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* dgamma, float* dbeta, float* dcolsum, float* workspace, int rows, int D,
                      int fuse_gelu, hipStream_t stream);
/* The same in two launches: dig_layernorm_bwd_partials writes dx and leaves [dig_layernorm_bwd_parts(rows)][3][D] fp32 partial
 * sums (dgamma, dbeta, column sums of dres) in `workspace`; dig_layernorm_bwd_finalize accumulates them into dgamma / dbeta /
 * dcolsum.  Nothing on the data-gradient chain depends on the second launch, so a caller may put it on another stream. */
int dig_layernorm_bwd_parts(int rows);
int dig_layernorm_bwd_partials(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                               const float* rstd, const void* dres, void* dx, float* workspace, int rows, int D, int fuse_gelu,
                               hipStream_t stream);
int dig_layernorm_bwd_finalize(const float* workspace, int rows, int D, float* dgamma, float* dbeta, float* dcolsum,
                               hipStream_t stream);
/* dig_layernorm_bwd_finalize over a workspace of `parts` partial rows ([parts][3][D]) that another producer wrote (dig_mlp_chain_bwd_ln) */
int dig_layernorm_bwd_finalize_parts(const float* workspace, int parts, int D, float* dgamma, float* dbeta, float* dcolsum,
                                     hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * BatchNorm1d in training mode, split so that the [2,C] statistics vector can be all-reduced between the halves
 * (nn.BatchNorm1d in _build_mlp, modeling_pretrain_moco_mim_ori.py:463-482, under SyncBatchNorm,
 * run_mae_pretraining_moco.py:390).  x, y, dy, dx: bf16 [rows, C], C % 8 == 0.
 *   dig_bn_stats:      sums[0][c] = sum_r x, sums[1][c] = sum_r x^2   (two-stage, deterministic; workspace from
 *                      dig_bn_stats_workspace_bytes, also used by dig_bn_bwd_stats)
 *   dig_bn_fwd_apply:  mean/var from sums / n_total (biased var); y = [relu](gamma * xhat + beta); gamma == beta == NULL
 *                      for the affine=False last layer; writes mean_out / rstd_out [C]
 *   dig_bn_update_running: running_mean/var <- (1-momentum) * old + momentum * (mean, var * n/(n-1))
 *   dig_bn_bwd_stats:  g = dy * relu_mask; sums[0][c] = sum g (= dbeta), sums[1][c] = sum g*xhat (= dgamma)
 *   dig_bn_bwd_apply:  dx = gamma * rstd * (g - S0/n_total - xhat * S1/n_total) with S the all-rank sums
 */
long long dig_bn_stats_workspace_bytes(int rows, int C);
int dig_bn_stats(const void* x, float* sums, float* workspace, int rows, int C, hipStream_t stream);
int dig_bn_fwd_apply(const void* x, const float* sums, float n_total, float eps, const float* gamma, const float* beta, int relu,
                     void* y, float* mean_out, float* rstd_out, int rows, int C, hipStream_t stream);
int dig_bn_update_running(const float* sums, float n_total, float momentum, float* running_mean, float* running_var, int C,
                          hipStream_t stream);
/* dig_bn_fwd_apply + dig_bn_update_running in one launch (running_mean / running_var both or neither; n_total > 1 with them) */
int dig_bn_fwd_apply_running(const void* x, const float* sums, float n_total, float eps, const float* gamma, const float* beta, int relu,
                             void* y, float* mean_out, float* rstd_out, float momentum, float* running_mean, float* running_var, int rows,
                             int C, hipStream_t stream);
int dig_bn_bwd_stats(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                     int relu, float* sums, float* workspace, int rows, int C, hipStream_t stream);
/* dig_bn_bwd_stats that also accumulates the layer's affine gradients from the LOCAL sums (before any cross-rank reduction of `sums`):
 * dbeta_acc[c] += sums[0][c], dgamma_acc[c] += sums[1][c]  (both or neither; the two axpy launches per layer of the backward) */
int dig_bn_bwd_stats_acc(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         int relu, float* sums, float* dbeta_acc, float* dgamma_acc, float* workspace, int rows, int C, hipStream_t stream);
/* Few-row BatchNorm layers (the BN-MLP heads on 8 B pooled rows) in ONE launch each, for a SINGLE rank -- with a process group the cross-rank
 * reduction of the statistics sits between dig_bn_stats and dig_bn_fwd_apply, and those stay.  Same expressions as the three-launch path
 * (statistics over these rows, biased variance, running statistics with n / (n - 1)); the summation order over rows differs, so the two
 * paths agree to fp32 round-off.  dig_bn_fused_supported: 2 <= rows <= 4096, C a multiple of 32 and >= 256.
 * dig_bn_bwd_fused: dx = gamma rstd (g - mean_r(g) - xhat mean_r(g xhat)), g = dy under the ReLU mask; dbeta_acc / dgamma_acc (both or
 * neither) += sum g, sum g xhat. */
int dig_bn_fused_supported(int rows, int C);
int dig_bn_fwd_fused(const void* x, float eps, const float* gamma, const float* beta, int relu, void* y, float* mean_out, float* rstd_out,
                     float momentum, float* running_mean, float* running_var, int rows, int C, hipStream_t stream);
int dig_bn_bwd_fused(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int relu,
                     float* dbeta_acc, float* dgamma_acc, void* dx, int rows, int C, hipStream_t stream);
int dig_bn_bwd_apply(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                     int relu, const float* sums, float n_total, void* dx, int rows, int C, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Patch embedding + mask-token mix + position add (PatchEmbed conv k4 s4, modeling_finetune.py:188,195;
 * modeling_pretrain_vit.py:95-99).  img fp32 [n_img,3,4*gh,4*gw]; W fp32 [D,48] (conv weight flattened c,p1,p2);
 * mask uint8 [n_img, gh*gw] (1 = replaced by mask_token; may be NULL); pos fp32 [gh*gw, D]; out bf16 [n_img*gh*gw, D].
 * Backward: dig_patchify_bf16 builds the bf16 patch matrix P [n_tok, 64] (masked rows and pad columns zero) so that
 * dW = dy^T P runs as a wgrad dig_gemm_bf16; dig_colsum_masked gives the bias (unmasked rows) and mask_token (masked
 * rows) gradients in one pass.  dig_patch_embed_bwd is the scalar reference form of the same three gradients.
 */
int dig_patch_embed_fwd(const float* img, const float* W, const float* bias, const unsigned char* mask, const float* mask_token,
                        const float* pos, void* out, int n_img, int gh, int gw, int D, hipStream_t stream);
int dig_patch_embed_bwd(const void* dy, const float* img, const unsigned char* mask, float* dW, float* dbias, float* dmask_token,
                        int n_img, int gh, int gw, int D, hipStream_t stream);
int dig_patchify_bf16(const float* img, const unsigned char* mask, void* out, int n_img, int gh, int gw, hipStream_t stream);
int dig_colsum_masked(const void* x, const unsigned char* mask, float* out_unmasked, float* out_masked, float* workspace,
                      int rows, int C, hipStream_t stream);   /* workspace: 2 x dig_colsum_workspace_bytes(rows, C) */

/* PatchNet 'no_patchtrans' = adaptive_avg_pool2d of the gh x gw token grid to (1, nwin)
 * (modeling_pretrain_moco_mim_ori.py:189-193) and its gradient (accumulate=1 adds into dx).  Window `w` = all gh rows x columns
 * [floor(w gw / nwin), ceil((w + 1) gw / nwin)): equal windows when nwin divides gw (README: 4 on 32 columns), overlapping bins of
 * 7 / 7 / 8 / 7 / 7 columns for the argparse default --num_windows 5 (run_mae_pretraining_moco.py:143).  nwin > gw is allowed (ConvPatchNet pools
 * its 1 x 4 map to (1, num_windows), :254: a column then feeds several windows). */
int dig_window_pool_fwd(const void* x, void* out, int out_is_f32, int n_img, int gh, int gw, int nwin, int D, hipStream_t stream);
int dig_window_pool_bwd(const void* dpool, void* dx, int n_img, int gh, int gw, int nwin, int D, int accumulate, hipStream_t stream);

/* ConvPatchNet (`--patchnet_name conv`, modeling_pretrain_moco_mim_ori.py:207-260) on NHWC bf16 maps [n_img, H, W, C] (the encoder's token
 * matrix [n_img, 8 * 32, C] is the map of :251-253).  Its conv3x3_blocks (:239-248: nn.Conv2d(k 3, stride 1, pad 1) -> BatchNorm2d -> ReLU) run as
 * dig_gemm_bf16 over the im2col matrix in the reference's own weight layout, BatchNorm2d as dig_bn_* over the [n_img H W, C] rows:
 *   dig_im2col3x3:  col[(b, y, x), c * 9 + ky * 3 + kx] = map[b, y + ky - 1, x + kx - 1, c] (zero outside the map) -- the column order of
 *                   conv.weight.view(C_out, C_in * 9), so conv = col @ W^T (+ bias) and dW = dy^T @ col on the arena's own views.  ldc >= 9 C
 *                   (multiple of 8; a multiple of 64 for the GEMM's reduction granule): columns [9 C, ldc) are zero-filled.  C % 8 == 0.
 *   dig_conv3x3_weight_flip:  wt[c_in, c_out * 9 + t] = w[c_out, c_in * 9 + 8 - t]: the convolution's data gradient is the convolution of dy with
 *                   the flipped, transposed taps -- dx = im2col(dy) @ wt^T (what aten's conv backward-input computes).
 *   dig_maxpool2x2_fwd / _bwd:  nn.MaxPool2d(kernel_size=2, stride=2) (:219-223) and its gradient.  idx [n_img, H/2, W/2, C] bytes: the position
 *                   0..3 = (0,0), (0,1), (1,0), (1,1) of the maximum; the first one wins a tie (aten: `val > maxval` scan), NaN propagates.
 *                   The backward writes every dx element (the gradient at the arg-max, zero elsewhere).  H, W even, C % 8 == 0. */
int dig_im2col3x3(const void* x, void* col, int n_img, int H, int W, int C, int ldc, hipStream_t stream);
int dig_conv3x3_weight_flip(const void* w, void* wt, int c_out, int c_in, hipStream_t stream);
int dig_maxpool2x2_fwd(const void* x, void* y, unsigned char* idx, int n_img, int H, int W, int C, hipStream_t stream);
int dig_maxpool2x2_bwd(const void* dy, const unsigned char* idx, void* dx, int n_img, int H, int W, int C, hipStream_t stream);

/* The loader's mask [B, V, N] (elem_kind 0: 1-byte bool / uint8, 1: fp32, 2: fp64, 3: int32, 4: int64; non-zero = masked) as the view-major
 * uint8 rows [V * B, N] the encoder reads, views >= keep_views zeroed (only_mim_on_ori_img: keep_views = 1) -- the bool cast, fill, permute
 * copy and uint8 cast of engine_for_pretraining_moco.py:99-104 and modeling_pretrain_moco_mim_ori.py:497 in one launch. */
int dig_mask_views_u8(const void* mask, int elem_kind, int B, int V, int N, int keep_views, unsigned char* out, hipStream_t stream);
/* ------------------------------------------------------------------------------------------------------------------
 * SimMIM target / decoder plumbing (engine_for_pretraining_moco.py:85-111,141; modeling_pretrain_moco_mim_ori.py:560-570).
 *   dig_mask_to_index: idx[b][j] = b*N + (j-th set position of mask[b,:]) in ascending order -- the order boolean
 *                      indexing yields; count[b] = number of set positions (bit-exact)
 *   dig_gather_rows / dig_scatter_rows_add: dst[m,:] = src[idx[m],:] (rows M..M_pad-1 zero) and its transpose
 *   dig_mim_target:    target[m, (p1*4+p2)*3 + c] = img[b,c,ph*4+p1,pw*4+p2]*0.5+0.5 for token idx[m] ('(p1 p2 c)' order)
 *   dig_mse_fwd_bwd:   loss += mean((pred-target)^2); dpred (bf16, optional) = gscale * 2 (pred-target) / (M*C)
 */
int dig_mask_to_index(const unsigned char* mask, int* idx, int* count, int B, int N, int max_per_sample, hipStream_t stream);
int dig_gather_rows(const void* src, const int* idx, void* dst, int M, int M_pad, int D, hipStream_t stream);
int dig_scatter_rows_add(const void* src, const int* idx, void* dst, int M, int D, hipStream_t stream);
int dig_mim_target(const float* img, const int* idx, float* target, int M, int gh, int gw, int normalize /* normlize_target,
                   engine_for_pretraining_moco.py:88-93: per-patch, per-channel (x - mean) / (sqrt(unbiased var) + 1e-6) */, hipStream_t stream);
int dig_mse_fwd_bwd(const float* pred, int ld_pred, const float* target, int M, int C, float gscale, float* loss, void* dpred,
                    int ld_dpred, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * MoCo-v3 InfoNCE in fp32 (contrastive_loss / accuracy / label_smooth_loss, modeling_pretrain_moco_mim_ori.py:444-461,
 * 593-625).  dig_l2norm_* = F.normalize(dim=1, eps); dig_sgemm: C = alpha * A[I,R] * (trans_b ? B[R,J] : B[J,R]^T);
 * dig_ce_rows: per row i of logits [n,m] with label i+label_offset: out3[0] += lse - logit[label], out3[1] += top-1 hit,
 * out3[2] += top-5 hit, and logits is overwritten by gscale * (softmax - onehot).
 */
int dig_l2norm_fwd(const float* x, float* y, float* inv_norm, int n, int C, float eps, hipStream_t stream);
int dig_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int n, int C, hipStream_t stream);
int dig_sgemm(const float* A, const float* B, float* C, int I, int J, int R, int lda, int ldb, int ldc, int trans_b, float alpha,
              int r_splits /* >1: C is [r_splits][I][ldc] partial slabs over R, summed by dig_reduce_partials */, hipStream_t stream);
int dig_ce_rows(float* logits, int n, int m, int label_offset, float gscale, float* out3, hipStream_t stream);
/* The two loss reductions with a caller-owned workspace instead of floating-point atomics: the partial sums (one per workgroup / row) are
 * added in a fixed order by the last workgroup to finish, so the logged loss values are bit-reproducible (gradients never depended on
 * them).  workspace: 257 floats (dig_mse_fwd_bwd_ws) / 1 + 3 n floats (dig_ce_rows_ws), ZERO before the first use; a launch leaves
 * workspace[0] at zero again, so launches on one stream can share it.  NULL workspace = the atomic forms above. */
int dig_mse_fwd_bwd_ws(const float* pred, int ld_pred, const float* target, int M, int C, float gscale, float* loss, void* dpred,
                       int ld_dpred, float* workspace, hipStream_t stream);
int dig_ce_rows_ws(float* logits, int n, int m, int label_offset, float gscale, float* out3, float* workspace, hipStream_t stream);
/* The scalar tails, one launch each.  dig_infonce_finish: stats6 = the two out3 triples of the q1/k2 and q2/k1 dig_ce_rows launches ->
 * contra[0] = (stats6[0] + stats6[3]) * loss_scale (2 T / n: modeling_pretrain_moco_mim_ori.py:459-461), accs4 = (q1_acc1, q1_acc5, q2_acc1,
 * q2_acc5) = the hit counts * acc_scale.  dig_step_meters: the ten values a step logs (engine_for_pretraining_moco.py:146-183) as one
 * vector: loss, contra, pixel, accs4[0..3], min and max of counts[n_counts] (masked tokens per sample), grad_norm (NULL -> NaN). */
int dig_infonce_finish(const float* stats6, float loss_scale, float acc_scale, float* contra, float* accs4, hipStream_t stream);
int dig_step_meters(const float* loss, const float* contra, const float* pixel, const float* accs4, const int* counts, int n_counts,
                    const float* grad_norm, float* out10, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Flat-arena optimizer (custom_optim/adamw.py:55-121 + _functional.py:115-140 as one kernel; EMA
 * modeling_pretrain_moco_mim_ori.py:428-442; grad norm utils/utils.py:507-519).
 *   dig_adamw_step: group_flags[i / 256] in {0,1} selects (lr0, wd0) or (lr1, wd1) for element i (parameters are padded
 *                   to 256 elements), 2 = leave the granule untouched (a parameter without a gradient); step >= 1 is the Adam step count; grad_scale multiplies g (clipping / averaging);
 *                   bf16_shadow (optional) receives the updated parameters in bf16;  finite_gate (optional): a device float --
 *                   when it is not finite (the squared gradient norm of dig_sumsq after a NaN / Inf loss) the launch changes
 *                   nothing, as torch's GradScaler skips optimizer.step() on inf / nan gradients (utils/utils.py:498-504).
 *   dig_ema_update: pm = pm*m + p*(1-m); bf16_shadow (optional) receives pm in bf16.
 *   dig_sumsq:      out[0] = sum x^2 (two-stage, deterministic); workspace: dig_sumsq_workspace_bytes.
 */
int dig_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n, const unsigned char* group_flags,
                   float lr0, float wd0, float lr1, float wd1, float beta1, float beta2, float eps, int step, float grad_scale,
                   const float* finite_gate, hipStream_t stream);
/* dig_adamw_step that also leaves, for a list of 2-D weights, the updated values TRANSPOSED in bf16 (W^T [cols, rows], the K-contiguous
 * operand of the fused MLP backward, dig_mlp_chain_bwd*): what three dig_transpose_bf16_multi launches per step used to rebuild from the
 * shadow.  mats: device table of n_mats 32-byte records {int64 off (element offset of W [rows, cols] in the arena, a multiple of 256),
 * int64 dst_off (element offset of W^T in tr_out, a multiple of 4), int32 rows, int32 cols (multiples of 64), int32 tile0 (index of the
 * weight's first 64 x 64 tile: the running sum of rows / 64 * cols / 64), int32 0}; n_tiles = the sum over the table.  The granules of a
 * listed weight carry bit 7 in group_flags (0x80 | group).  group_flags value 2 (either form) = a parameter that never receives a gradient:
 * left untouched, as the reference's AdamW skips `p.grad is None` (custom_optim/adamw.py:78-79).  Same arithmetic as dig_adamw_step. */
int dig_adamw_step_tr(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n, const unsigned char* group_flags,
                      float lr0, float wd0, float lr1, float wd1, float beta1, float beta2, float eps, int step, float grad_scale,
                      const float* finite_gate, const void* mats, int n_mats, int n_tiles, void* tr_out, hipStream_t stream);
/* Same update with any number of parameter groups (layer-wise lr decay: optim_factory.py:33-100, run_class_finetuning.py:471-520):
 * group_idx holds one uint8 per 256-element granule indexing the device tables lr_tab / wd_tab; index 255 = granule without a
 * gradient, left untouched (the reference's AdamW skips p.grad is None). */
int dig_adamw_step_groups(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n, const unsigned char* group_idx,
                          const float* lr_tab, const float* wd_tab, float beta1, float beta2, float eps, int step, float grad_scale,
                          const float* finite_gate, hipStream_t stream);
int dig_ema_update(float* pm, const float* p, void* bf16_shadow, long long n, float m, hipStream_t stream);
/* The same two launches with their per-step scalars read from device memory, so that a captured HIP graph of the training step
 * (dig_amd/step_graph.py) replays with the current schedule values: scalars6 = {lr0, wd0, lr1, wd1, 1/(1-beta1^t), 1/sqrt(1-beta2^t)}
 * (the last two exactly as dig_adamw_step derives them from `step`: dig_adamw_bias_corrections, a host function);
 * m_and_one_minus_m = {m, (float)(1 - (double)m)}.  Bit-identical to the by-value forms for equal scalars. */
int dig_adamw_bias_corrections(float beta1, float beta2, int step, float* out2_host);
int dig_adamw_step_dev(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n, const unsigned char* group_flags,
                       const float* scalars6, float beta1, float beta2, float eps, float grad_scale, const float* finite_gate,
                       hipStream_t stream);
int dig_ema_update_dev(float* pm, const float* p, void* bf16_shadow, long long n, const float* m_and_one_minus_m, hipStream_t stream);
long long dig_sumsq_workspace_bytes(long long n);
int dig_sumsq(const float* x, long long n, float* workspace, float* out, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Small helpers: bias-gradient column sums (two-stage, deterministic), GELU backward, casts, fills, scaling.
 */
long long dig_colsum_workspace_bytes(int rows, int C);
int dig_colsum(const void* x, float* out, float* workspace, int rows, int C, int ld, hipStream_t stream);
/* out[c] += sum_b partials[b][c]: finishes the [ceil(I/64)][J] column sums that dig_gemm_bf16(act 2, colsum_partials) leaves
 * (the fc1 bias gradient fused into the fc2 data-gradient GEMM; reference: autograd of nn.Linear bias, modeling_pretrain_vit.py:60-76) */
int dig_colsum_partials(const float* partials, int n_parts, int C, float* out, hipStream_t stream);
/* out[c] += sum_b partials[b * stride + c], c < C, for up to DIG_COLSUM_MAX_SEGS partial sets in one launch: the bias and LayerNorm
 * parameter gradients an encoder block's backward leaves as partials (fc1 bias from the fc2 dgrad, q / v bias from dig_attn_bwd, and the
 * three interleaved vectors of each dig_layernorm_bwd_partials workspace: stride 3*D, bases workspace + {0, D, 2D}) -- 5 finalize
 * launches -> 1, same summation order per column as dig_colsum_partials.  C % 8 == 0, stride % 4 == 0; `segs` is host memory.
 * With more than 12 segments whose widths are all multiples of 32 (every encoder block's partial rows of a backward pass in one launch, round 6)
 * a workgroup owns 32 columns -- full 128-byte lines of a partial row, 32 row groups: another summation order (1e-6 relative). */
#define DIG_COLSUM_MAX_SEGS 112
typedef struct dig_colsum_seg { const float* partials; float* out; long long stride; int n_parts; int C; } dig_colsum_seg_t;
int dig_colsum_partials_multi(const dig_colsum_seg_t* segs, int n_segs, hipStream_t stream);
int dig_gelu_bwd(const void* dact, const void* pre, void* dpre, long long n, hipStream_t stream);
int dig_add_bf16(const void* a, const void* b, void* out, long long n, hipStream_t stream);
int dig_cast_f32_to_bf16(const float* x, void* y, long long n, hipStream_t stream);
int dig_cast_bf16_to_f32(const void* x, float* y, long long n, hipStream_t stream);
int dig_pad_cast_rows(const float* src, void* dst, int M, int C, int M_pad, int ld, hipStream_t stream);
int dig_fill_f32(float* x, long long n, float value, hipStream_t stream);
int dig_scale_f32(float* x, long long n, float s, hipStream_t stream);
int dig_scale_by_device_scalar(float* x, long long n, const float* scalar, float extra, hipStream_t stream);
int dig_axpy_f32(float* y, const float* x, long long n, float a, hipStream_t stream);

/* ---- input transform on the device (dataset/datasets.py:27-42, masking_generator.py:12-49)
 * dig_resize_bicubic_normalize_u8: n_img uint8 RGB crops (HWC, image i at packed + offsets[i], heights[i] x widths[i]; all
 * three arrays in device memory) -> out[n_img][3][out_h][out_w] fp32 = Normalize(ToTensor(Resize((out_h,out_w), BICUBIC))),
 * bit-exact with Pillow's ImagingResample.  max_h / max_w bound the crop sizes (they size the LDS coefficient tables). */
int dig_resize_bicubic_normalize_u8(const unsigned char* packed, const long long* offsets, const int* heights, const int* widths,
                                    int n_img, float* out, int out_h, int out_w, float mean, float std_, int max_h, int max_w,
                                    hipStream_t stream);
/* mask[n_rows][n_patches] uint8, exactly num_mask ones per row (RandomMaskingGenerator.__call__); row r, patch p gets the
 * Philox4x32-10 key of counter (r, p, step, 0) under key (seed), the num_mask smallest (key, p) are masked. */
int dig_random_masks(unsigned char* mask, int n_rows, int n_patches, int num_mask, unsigned long long seed, unsigned step,
                     hipStream_t stream);

/* ---- greedy decode with a K/V cache (models/decoder.py:173-252, models/transformer_layer.py:238-281)
 * One decode step of TFDecoder.forward_test per call sequence: dig_decode_embed (token embedding + position row t), then per
 * layer LayerNorm / dig_gemm_bf16 for the projections (the fused q|k|v GEMM writes row t of the [B, T, 3*heads*64] cache in
 * place), dig_decode_self_attn (row t against rows 0..t), dig_decode_cross_attn (against [B, n_mem, 2*heads*64] = k|v of the
 * encoder memory, projected once; `weights` optional [B, heads, n_mem] fp32), and dig_softmax_argmax on the classifier logits.
 * head_dim must be 64. */
int dig_decode_embed(const long long* tokens, const float* emb, const float* pe_row, void* x, int B, int d, int vocab,
                     hipStream_t stream);
int dig_decode_self_attn(const void* qkv_cache, void* out, int B, int T, int heads, int head_dim, int t, float scale,
                         hipStream_t stream);
int dig_decode_cross_attn(const void* q, const void* kv_mem, void* out, float* weights, int B, int n_mem, int heads, int head_dim,
                          float scale, int slots_per_mem /* query b reads memory b / slots_per_mem: the beam slots of one sample share
                          its projected memory (1 for greedy decoding) */, hipStream_t stream);
/* One step of TFDecoder.beam_search (models/decoder.py:283-307) over B samples x beam_width slots: log-softmax of the slots' logits,
 * + the running scores seq_scores [B*beam_width] (in / out: -inf for a slot that has just emitted `eos`), top-beam_width of the
 * beam_width*C candidates per sample -> symbols = candidate % C, predecessors = candidate / C + sample*beam_width (both int64
 * [B*beam_width]) and stored_scores (the un-masked top-k scores, what the final back-tracking reads). */
int dig_beam_step(const float* logits, int ld, float* seq_scores, int B, int beam_width, int C, int eos, long long* symbols,
                  long long* predecessors, float* stored_scores, hipStream_t stream);
int dig_softmax_argmax(const float* logits, int ld, float* probs, long long* tokens, int B, int C, hipStream_t stream);
/* Accuracy of evaluation_metric/metrics.py:19-81 on the device: rows of pred / target ([B][T] int64 class ids) are cut at `eos`;
 * canon[c] (uint8, n_classes entries) is the canonical code of class c -- 0 for classes the metric drops (UNKNOWN, anything that
 * is not a digit or letter), otherwise the same code for upper and lower case; match[b] = 1 when the normalised strings agree. */
int dig_string_match(const long long* pred, const long long* target, const unsigned char* canon, int n_classes, int eos, int B, int T,
                     unsigned char* match, hipStream_t stream);
/* SeqCrossEntropyLoss.forward (loss/seqCrossEntropyLoss.py:47-63, sample_normalize): loss[0] = -sum_{b, t < length[b]}
 * log_softmax(input[b,t,:])[target[b,t]] / B; row_workspace: B*T floats; fixed summation order. */
int dig_seq_cross_entropy(const float* input, const long long* target, const long long* length, int B, int T, int C,
                          float* row_workspace, float* loss, hipStream_t stream);
/* recognition_f_measure (evaluation_metric/metrics.py:83-100) per sample, double precision; same canon table as dig_string_match. */
int dig_char_fmeasure(const long long* pred, const long long* target, const unsigned char* canon, int n_classes, int eos, int B, int T,
                      double* f_per_sample, hipStream_t stream);

/* ---- fine-tune training step (drop rates 0)
 * Whole-sequence attention of the recognition decoder and its gradient (models/transformer_layer.py:238-281 under teacher forcing,
 * models/decoder.py:173-222): per (sample, head), Lq <= 32 queries against Lk <= 512 keys, head dim 64, logits = q.k * scale,
 * mask = (causal ? key <= query : 1) & (lens ? key < lens[sample] : 1).  q / k / v / out rows are (sample, position) with the
 * given leading dimensions (elements), head h at columns [64h, 64h+64); lse: fp32 [B][heads][Lq]. */
int dig_seq_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, float* lse, int B,
                     int heads, int Lq, int Lk, float scale, int causal, const long long* lens, hipStream_t stream);
int dig_seq_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo, const float* lse,
                     void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int heads, int Lq, int Lk, float scale, int causal,
                     const long long* lens, hipStream_t stream);
/* Token embedding + positional table for whole sequences (decoder.py:173-181) and its (deterministic) gradient into the fp32
 * embedding-gradient table; gradient of SeqCrossEntropyLoss w.r.t. the logits (bf16 rows of ldd >= C columns, pad columns zero),
 * scaled by the device scalar *gscalar (null = 1). */
int dig_seq_embed_fwd(const long long* tokens, const float* emb, const float* pos_table, void* x, int B, int T, int d, int vocab,
                      hipStream_t stream);
int dig_seq_embed_bwd(const long long* tokens, const void* dx, float* demb, int n_tok, int d, int vocab, hipStream_t stream);
/* the same, skipping positions t >= lens[sample] (rows are (sample, t), T positions per sample): under teacher forcing their dx
 * rows are exact zeros, and the padding token would otherwise dominate one vocabulary row's sum */
int dig_seq_embed_bwd_lens(const long long* tokens, const void* dx, float* demb, int n_tok, int d, int vocab, int T, const long long* lens,
                           hipStream_t stream);
int dig_seq_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar, int B,
                              int T, int C, void* dlogits, int ldd, hipStream_t stream);

/* SeqLabelSmoothingCrossEntropyLoss (loss/seqLabelSmoothingCrossEntropyLoss.py:48-70; `--smoothing` > 0, run_class_finetuning.py:538-541)
 * with the arithmetic the reference really performs: its `smooth_loss = -logprobs.mean(1) * mask` multiplies a [BT] vector with the
 * [BT,1] mask, which broadcasts to [BT,BT], so
 *   loss = ((1 - smoothing) * BT * sum_i mask_i nll_i + smoothing * (sum_i mask_i) * sum_j s_j) / B,  s_j = -mean_c log_softmax(x_j)_c
 * over ALL rows j.  row_workspace: 2*B*T floats.  The backward writes bf16 rows of ldd >= C columns (pad columns zero), scaled by the
 * device scalar *gscalar (null = 1). */
int dig_seq_ls_cross_entropy(const float* input, const long long* target, const long long* length, int B, int T, int C, float smoothing,
                             float* row_workspace, float* loss, hipStream_t stream);
int dig_seq_ls_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar, int B,
                                 int T, int C, float smoothing, void* dlogits, int ldd, hipStream_t stream);

/* ---- dropout / stochastic depth of the fine-tune step (README.md:100-118: --drop 0.1 --attn_drop_rate 0.1 --drop_path 0.1; the
 * recognition decoder's hard-wired dropout = 0.1, models/decoder.py:141).  Replaces nn.Dropout (modeling_finetune.py:51,83-85,271;
 * models/transformer_layer.py:236-237,394; models/decoder.py:160) and timm drop_path (modeling_finetune.py:37).
 * No mask is stored: an element is dropped when a keyed counter hash of its coordinates is below thr = floor(p * 2^32),
 *     x = a ^ k0;  x ^= x >> 16;  x *= 0x7feb352d;  x += k1 + b * 0x9e3779b9;  x ^= x >> 15;  x *= 0x846ca68b;  x ^= x >> 16   (uint32)
 * with (a, b) = (row * cols + col, 0) for [rows, cols] tensors, ((query << 16) | key, sample * heads + head) for attention
 * probabilities and (sample, 0) for drop-path; forward, backward and the CPU oracle regenerate it from the per-site key.  torch's
 * own Philox stream cannot be reproduced bit for bit (it depends on launch geometry); the distribution and the placement of every
 * mask are the reference's, which tests/test_finetune.py pins by feeding these masks to the unmodified reference. */
typedef struct dig_dropout {
  unsigned k0, k1;      /* element-dropout key of this site */
  unsigned thr;         /* floor(p * 2^32); 0 = no element dropout */
  float scale;          /* 1 / (1 - p) */
  unsigned pk0, pk1;    /* drop-path key */
  unsigned pthr;        /* floor(drop_path * 2^32); 0 = none */
  float pscale;         /* 1 / (1 - drop_path) */
  int rows_per_sample;  /* drop-path: sample = row / rows_per_sample */
} dig_dropout_t;
/* dig_gemm_bf16 with dropout and/or drop-path applied to the result after bias / activation and BEFORE the residual add
 * (x + drop_path(dropout(linear(h)))); with act 2 the mask multiplies the GELU' product (backward of dropout(gelu(.))) and the
 * fused column sums see the masked values.  Element index = i * J + j.  Separate kernel instantiations (the plain GEMMs carry
 * none of this code), built for out_kind 0 forward / dgrad on the 128x128 tiles (bk 0 / 64 / 32) and forward on bk 244 / 264; any
 * other combination returns -4. */
int dig_gemm_bf16_dropout(const void* A, const void* B, void* C, int I, int J, int R, int lda, int ldb, int ldc, int trans_a,
                          int trans_b, int out_kind, const float* bias, const void* resid, int ldr, void* pre_act, int ldp, float alpha,
                          int alpha_cols, int act, int splits, int a_rows, int b_rows, int bk, float* colsum_partials,
                          const dig_dropout_t* drop, hipStream_t stream);
/* dig_mlp_chain_fwd_ln with Mlp.drop behind fc2 and the block's drop_path on the MLP branch (modeling_finetune.py:59,158):
 * out = resid + drop_path(dropout(gelu_erf(LN(x) w1^T + b1) w2^T + b2)); the mask rule and element index (row * D + col) are
 * dig_gemm_bf16_dropout's, so the pattern is the two-GEMM path's bit for bit.  drop = NULL or both thresholds 0: dig_mlp_chain_fwd_ln.
 * (The backward takes the masked gradient dig_dropout_apply(dy) as its input: dig_mlp_chain_bwd / _bwd_ln as they are.) */
int dig_mlp_chain_fwd_ln_dropout(const void* x, const void* resid, const float* ln_g, const float* ln_b, float eps, void* ln_out, float* ln_mean,
                                 float* ln_rstd, const void* w1, const float* b1, const void* w2, const float* b2, void* out, void* pre_out,
                                 void* act_out, const float* nln_g, const float* nln_b, void* nln_out, float* nln_mean, float* nln_rstd, int R, int D,
                                 int F, const dig_dropout_t* drop, hipStream_t stream);
/* out = dropout/drop-path(in) on a [rows, cols] bf16 tensor (cols % 8 == 0, in-place allowed): pos_drop / embedding dropout in the
 * forward, and the gradient of any dropped branch in the backward (same mask, same scale). */
int dig_dropout_apply(const void* in, void* out, long long rows, int cols, const dig_dropout_t* drop, hipStream_t stream);
/* dig_attn_fwd / dig_attn_bwd / dig_seq_attn_fwd / dig_seq_attn_bwd with attention dropout: probabilities are normalised by the
 * full row sum, then masked and scaled (attn_drop after softmax); drop = NULL or thr = 0 is the plain kernel.
 * q_rows (1..256): only the first q_rows query rows of every image exist -- the recognition decoder's cross-attention (25 queries
 * against the 256 encoder tokens, models/transformer_layer.py:104-108) runs on these kernels with its queries in rows [0, 25) of a
 * fused q|k|v buffer; query blocks past ceil(q_rows / 32) are neither computed nor written (ctx / lse / dq rows there are left
 * untouched), dK / dV sum over the existing blocks only (rows between q_rows and the block end must carry dctx = 0). */
int dig_attn_fwd_dropout(const void* qkv, void* ctx, float* lse, int n_img, int heads, int embed_dim, const dig_dropout_t* drop,
                         int q_rows, hipStream_t stream);
int dig_attn_bwd_dropout(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int n_img, int heads,
                         int embed_dim, float scale, float* q_colsum, float* v_colsum, const dig_dropout_t* drop, int q_rows,
                         hipStream_t stream);
int dig_seq_attn_fwd_dropout(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, float* lse, int B,
                             int heads, int Lq, int Lk, float scale, int causal, const long long* lens, const dig_dropout_t* drop,
                             hipStream_t stream);
int dig_seq_attn_bwd_dropout(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo,
                             const float* lse, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int heads, int Lq, int Lk,
                             float scale, int causal, const long long* lens, const dig_dropout_t* drop, hipStream_t stream);


/* ---- GRU attention recognition head (`--decoder_type attention`: models/model_builder.py:40-72 AttnRecModel,
 * models/attn_decoder.py:11-78 AttentionRecognitionHead, :197-272 AttentionUnit / DecoderUnit).  One call per time step (the recurrence
 * orders the steps); the Linear layers run on dig_gemm_bf16.
 * dig_addattn_fwd: alpha[b,:] = softmax_n(w . tanh(sproj[b,:] + xproj[b,n,:])) (fp32 [B,N], kept for the backward) and
 *   context[b,:] = sum_n alpha[b,n] x[b,n,:] written to ctx + b*ldc (bf16: ldc lets it land inside the GRU input [yProj | context]).
 *   xproj [B,N,A], sproj [B,A], x [B,N,X] bf16; w [A] fp32 (wEmbed.weight; its bias cancels in the softmax).  N <= 512, A <= 1024.
 * dig_addattn_bwd (one step): dctx [B,X] (bf16 rows of stride ldd) -> dv [B,N] fp32 (gradient w.r.t. the scores, kept), dsproj [B,A]
 *   bf16, dw_acc [B,A] fp32 += per-sample partials of the wEmbed.weight gradient.
 * dig_addattn_bwd_tokens (after the last step): dxproj [B,N,A] and dx [B,N,X] (bf16) as sums over the T steps, from the kept dv_all
 *   [T,B,N], alpha_all [T,B,N] (fp32), sproj_all [T,B,A] and dctx_all rows (t,b) of stride ldd (bf16) -- the per-step [B,N,*] gradients
 *   are never materialised. */
int dig_addattn_fwd(const void* xproj, const void* sproj, const float* w, const void* x, float* alpha, void* ctx, int ldc, int B, int N, int A,
                    int X, hipStream_t stream);
int dig_addattn_bwd(const void* xproj, const void* sproj, const float* w, const void* x, const float* alpha, const void* dctx, int ldd, float* dv,
                    void* dsproj, float* dw_acc, int B, int N, int A, int X, hipStream_t stream);
int dig_addattn_bwd_tokens(const void* xproj, const void* sproj_all, const float* w, const float* dv_all, const float* alpha_all,
                           const void* dctx_all, int ldd, void* dxproj, void* dx, int T, int B, int N, int A, int X, hipStream_t stream);
/* torch.nn.GRU cell (gate order r | z | n): gi = W_ih [yProj|context] + b_ih, gh = W_hh s + b_hh as bf16 [B,3S] from the GEMMs;
 * s_prev fp32 [B,S] (NULL = zeros) -> s (fp32) and its bf16 copy; gates [B,4S] fp32 = r | z | n | gh_n kept for the backward.
 * Backward: ds = ds_a + ds_b + ds_c + ds_d (fp32, NULL terms skipped: classifier path, z-path of the later step and its two GEMM
 * paths) -> dgi, dgh [B,3S] bf16 and ds_prev = z * ds. */
int dig_gru_cell_fwd(const void* gi, const void* gh, const float* s_prev, float* s, void* s_bf16, float* gates, int B, int S, hipStream_t stream);
int dig_gru_cell_bwd(const float* ds_a, const float* ds_b, const float* ds_c, const float* ds_d, const float* gates, const float* s_prev, void* dgi,
                     void* dgh, float* ds_prev, int B, int S, hipStream_t stream);
/* out[r, :cols] (bf16, row stride ld) = table[clamp(tokens[r]), :cols] (fp32 table): tgt_embedding lookups (attn_decoder.py:264). */
int dig_embed_rows(const long long* tokens, const float* table, void* out, int ld, int rows, int cols, int vocab, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * One call per encoder block (Block.forward, modeling_finetune.py:150-158, and its gradient): the launch sequences of
 * dig_amd/engine_core.py's default plan issued from the library -- forward: dig_attn_block_fwd (fuse_attn; else dig_gemm_bf16 (qkv, q
 * scaled) -> dig_attn_fwd -> dig_gemm_bf16 (proj + residual)) -> dig_mlp_chain_fwd_ln;  backward: dig_mlp_chain_bwd_ln (fuse_ln2: the MLP's
 * data gradients and norm2's backward in one launch, with projt also the projection's data gradient; else dig_mlp_chain_bwd ->
 * dig_layernorm_bwd_partials (norm2)) -> dig_gemm_bf16 (proj data gradient) -> dig_attn_bwd (with the q / v bias sums) -> dig_wgrad_group (the
 * block's four weight gradients + the fold of the previous block's; not with wg_defer) -> dig_gemm_bf16 (qkv data gradient) ->
 * dig_layernorm_bwd_partials (norm1) on `stream`, then the five parameter-gradient reductions (dig_colsum_partials x 3,
 * dig_layernorm_bwd_finalize(_parts) x 2) on b->side behind one event.  Same kernels,
 * same arguments, same order as the per-entry-point path: results are bit-identical to it.  The tables are HOST memory.
 * Returns the first non-zero code of the sequence (nothing after it is launched). */
int dig_encoder_block_fwd(const dig_block_fwd_t* b, hipStream_t stream);
int dig_encoder_block_bwd(const dig_block_bwd_t* b, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Launch probe (measurement only; bench.py).  Between dig_probe_start() and dig_probe_stop() every launch of dig_gemm_bf16(_dropout),
 * dig_mlp_chain_fwd / _fwd_ln / _bwd and dig_wgrad_group carries the start / stop events of hipExtLaunchKernel; dig_probe_stop
 * synchronises the device and writes the launches' device-side durations (microseconds, in call order) to the HOST array us_out
 * (at most max_n entries); returns the number of launches recorded.  Off, the launchers are byte-for-byte what they were. */
int dig_probe_start(void);
int dig_probe_stop(float* us_out, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* DIG_HIP_H */
