/* dig_block_types.h -- argument tables of dig_encoder_block_fwd / dig_encoder_block_bwd (declared in dig_hip.h, which includes this file
 * after its hipStream_t typedef; the host-only build under cpu_abi/ includes it after its own).  Plain C, device pointers unless stated.
 *
 * One call = one encoder block of the DiG ViT (Block.forward, modeling_finetune.py:150-158, called from
 * PretrainVisionTransformerEncoder.forward_features :312-334) or its gradient: the launch sequence that dig_amd/engine_core.py otherwise
 * issues entry point by entry point, so that the host pays one FFI crossing per block instead of four (forward) or fifteen (backward).
 * The functions launch exactly the kernels the per-entry-point sequence launches, with the same arguments, in the same order. */
#ifndef DIG_BLOCK_TYPES_H
#define DIG_BLOCK_TYPES_H

typedef struct dig_wgrad_prob {
  const void* A; const void* B; float* out;
  int lda, ldb, ldo, I, J, trans_out;
} dig_wgrad_prob_t;

typedef struct dig_block_fwd {
  int n_img, heads, D, F, rows;        /* rows = n_img * 256 tokens */
  int save;                            /* 1: online branch (everything below is written); 0: momentum branch (ln2, mu2, rs2, pre, act, nmu, nrs may be null) */
  int tile_qkv, tile_proj;             /* DIG_GEMM_TILE_* of the two Linear layers */
  int fuse_attn;                       /* 1: qkv GEMM -> attention -> proj GEMM + residual as ONE launch (dig_attn_block_fwd) where dig_attn_block_supported(heads, D);
                                          qkv / lse are then written only when save = 1 (ctx always: it is the launch's scratch) */
  int reserved0;
  float eps, scale;                    /* LayerNorm eps; head_dim^-0.5 applied to the q columns in the qkv epilogue */
  /* parameters: bf16 weights [out, in], fp32 biases and LayerNorm parameters */
  const void* qkv_w; const float* qkv_b; const void* proj_w; const float* proj_b;
  const float* n2_g; const float* n2_b; const void* fc1_w; const float* fc1_b; const void* fc2_w; const float* fc2_b;
  const float* next_n1_g; const float* next_n1_b;     /* norm1 of the FOLLOWING block (null for the last block: nln, nmu, nrs unused) */
  /* in: the block's input rows and their norm1 (made by the previous block's call or dig_layernorm_fwd) */
  const void* x; const void* ln1;
  /* out */
  void* qkv; void* ctx; float* lse; void* x_mid; void* ln2; float* mu2; float* rs2; void* pre; void* act;
  void* out; void* nln; float* nmu; float* nrs;
} dig_block_fwd_t;

typedef struct dig_block_bwd {
  int n_img, heads, D, F, rows;
  int tile_dgrad;                      /* DIG_GEMM_TILE_* of the two data-gradient GEMMs (proj, qkv) */
  int tile_direct;                     /* DIG_GEMM_TILE_* of their DIRECT form on the K-contiguous weight copies proj_wt / qkv_wt (0: not used) */
  float scale;
  /* parameters; w2t = fc2.weight^T [F, D], w1t = fc1.weight^T [D, F] (dig_transpose_bf16) */
  const void* qkv_w; const void* proj_w; const void* w2t; const void* w1t;
  const void* projt;                   /* proj.weight^T [D, D] (dig_transpose_bf16), or NULL: with fuse_ln2, non-NULL puts the projection's data gradient
                                          into the fused MLP backward launch as well (dig_mlp_chain_bwd_ln_proj) instead of its own GEMM launch */
  /* K-contiguous copies proj.weight^T [D, D] and qkv.weight^T [D, 3D] (dig_adamw_step_tr leaves them; dig_transpose_bf16 otherwise), or
   * NULL: with tile_direct != 0 the two data gradients dctx = dx_mid Wproj and dln1 = dqkv Wqkv run as direct-form GEMMs on them (both
   * operands read along K: the 256 x 192 persistent tiles of the forward) instead of the transpose-read form on the [out, in] weights --
   * bit-identical results, 57.9 against 69.0 us (qkv) and 28.5 against 31.9 us (proj) per launch at 65 536 rows (tools/gpu_dgrad_form_probe.py) */
  const void* proj_wt; const void* qkv_wt;
  const float* n1_g; const float* n1_b; const float* n2_g; const float* n2_b;
  /* fp32 gradients, accumulated into (contiguous [out, in] matrices); g_q_b / g_v_b = the q and v thirds of the qkv bias gradient */
  float* g_n1_g; float* g_n1_b; float* g_qkv_w; float* g_q_b; float* g_v_b; float* g_proj_w; float* g_proj_b;
  float* g_n2_g; float* g_n2_b; float* g_fc1_w; float* g_fc1_b; float* g_fc2_w; float* g_fc2_b;
  /* what dig_encoder_block_fwd (save = 1) left */
  const void* x; const void* ln1; const float* mu1; const float* rs1; const void* qkv; const void* ctx; const float* lse;
  const void* x_mid; const void* ln2; const float* mu2; const float* rs2; const void* pre; const void* act;
  const void* dy;                      /* gradient w.r.t. the block's output rows, bf16 [rows, D] */
  /* temporaries ([rows, D], [rows, F], [rows, D], [rows, 3D] bf16); on return dctx holds the gradient w.r.t. the block's INPUT rows */
  void* dln2; void* dpre; void* dctx; void* dqkv;
  /* fp32 partial sums finished on the side stream: [dig_mlp_chain_colsum_rows(rows)][F], 2 x [dig_layernorm_bwd_parts(rows)][3][D] (ws2 with
   * fuse_ln2: [max(that, dig_mlp_chain_ln_parts(rows))][3][D]), 2 x [n_img][D] */
  float* bparts; float* ws1; float* ws2; float* qs; float* vs;
  /* grouped weight gradients (dig_wgrad_group): this block's four problems are written to wg_probs (HOST array of 4) in the order
   * fc2, fc1, proj, qkv and launched together with the fold of wg_fold_probs (the previous call's wg_probs; wg_fold_n = 0 for the first
   * block of a backward pass).  wg_trans[k] = 1: problem k puts the activation first and stores its result transposed. */
  int wg_fn, wg_wa, wg_splits, wg_n_wg, wg_fold_n, wg_fold_splits;
  int wg_trans[4];
  int wg_defer;                        /* 1: wg_probs is filled but dig_wgrad_group is NOT launched: the caller launches the blocks' weight gradients
                                          itself, behind the last data gradient (lab switch: the data-gradient chain then runs uninterrupted) */
  int fuse_ln2;                        /* 1: norm2's backward inside the fused MLP backward launch (dig_mlp_chain_bwd_ln) instead of its own launch; ws2 then
                                          holds [dig_mlp_chain_ln_parts(rows)][3][D] and is reduced by dig_layernorm_bwd_finalize_parts */
  int attn_proj;                       /* 1 (with proj_wt and D a multiple of 128, at most 512): the projection's data gradient inside the attention backward
                                          launch (dig_attn_bwd_proj: every (image, head) workgroup computes its own d(ctx) tile from dx_mid and
                                          proj.weight^T) -- no GEMM launch, d(ctx) is never written */
  int defer_red;                       /* 1: the block's parameter-gradient reductions (fc1 / q / v bias column sums, both LayerNorms' partials) are NOT launched:
                                          bparts / ws1 / ws2 / qs / vs keep their partial rows and the caller folds the blocks' partials together, in one
                                          dig_colsum_partials_multi launch behind the last data gradient (9 segments per block, same sums) */
  const unsigned* wg_map; float* wg_slabs; const float* wg_fold_slabs;
  dig_wgrad_prob_t* wg_probs; const dig_wgrad_prob_t* wg_fold_probs;
  hipStream_t side;                    /* stream of the parameter-gradient reductions (may equal the call's stream) */
} dig_block_bwd_t;

#endif /* DIG_BLOCK_TYPES_H */
