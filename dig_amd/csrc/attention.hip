// Fused self-attention for the DiG ViT encoder on gfx950: N = 256 tokens, head_dim = 64, dense softmax.
// Reference math: Attention.forward, modeling_finetune.py:87-120 (q is pre-scaled by head_dim^-0.5 in
// the QKV GEMM epilogue; K has no bias); backward is the analytic softmax-attention gradient.
//
// One workgroup owns one (image, head): its K and V ([256,64] bf16, 32 KiB each) live in LDS for the whole
// kernel, the 256x256 score matrix never leaves registers, nothing is written to HBM but the context
// rows and the per-row log-sum-exp (fp32) that the backward needs.
//
// LDS tile layout "U" (all four operands): addr(row, col) = row*128 + ((col>>3) ^ f(row))*16 + (col&7)*2
// with f(row) = ((row>>1)&1)<<2 | ((row>>2)&3).  The same image serves
//   * ds_read_b128 fragments along the contiguous (col) axis  -- 16 distinct rows hit 16 distinct 16-B slots;
//   * ds_read_b64_tr_b16 fragments along the row axis (4 consecutive rows x 16 cols per 16-lane group),
// both bank-conflict free, so Q/K/V/dO are staged once (buffer_load ... lds, swizzle on the source side).
//
// MFMAs are issued "swapped" (D' = X^T-side operand first) so that the lane that owns a query (forward,
// dQ phase) or a key (dK/dV phase) holds that row's scores in its own registers: the softmax row
// reductions are 127 in-lane ops + one cross-half exchange, and P / dS feed the next MFMA as the B operand
// without any data movement (the V / K / Q / dO operand is fetched with the matching row permutation by
// the transpose read).
#include "common.h"
#include <atomic>

// phase time stamps for tools/experiments/attn_bwd_lab.hip (empty in the product build)
#ifndef DIG_ATTN_SP_LAB
#define DIG_ATTN_SP_LAB 0                    // lab: 1 no barrier between the steps (wrong sums: timing only)
#endif
#ifndef DIG_ATTN_BWD_SP_DEFAULT
#define DIG_ATTN_BWD_SP_DEFAULT 0
#endif
#ifndef DIG_ATTN_TS
#define DIG_ATTN_TS(i)
#endif
#ifndef DIG_ATTN_B_PIPE
#define DIG_ATTN_B_PIPE 0                    // backward phase B: 1 = the tile's element arithmetic issued between its own MFMAs (see the loop)
#endif
#ifndef DIG_ATTN_BWD_STORE
#define DIG_ATTN_BWD_STORE 3                 // backward, how dq / dk / dv leave: 0 = 16-byte row stores, 1 = the same, non-temporal, 3 = full 128-byte lines
#endif                                       // through 2 KiB of LDS per wave, non-temporal (dig_attn_bwd_store() / environment DIG_ATTN_BWD_STORE: A/B in the step)
#ifndef DIG_ATTN_LIFT
#define DIG_ATTN_LIFT 3                      // backward restage: bit 0 both query blocks' Q / dO fragments from LDS, bit 1 odd K / V blocks from registers
#endif
#ifndef DIG_ATTN_B_ABL
#define DIG_ATTN_B_ABL 0                     // lab ablations of phase B (results wrong): 1 no element arithmetic, 2 no transposed fragment reads, 4 no seed
#endif                                       // reads, 8 no S / dP MFMAs, 16 no dV / dK MFMAs, 32 no direct fragment reads, 64 no phase A loop, 128 no result stores, 256 O rows all from one address
#ifndef DIG_ATTN_A_PIPE
#define DIG_ATTN_A_PIPE 0
#endif
#ifndef DIG_ATTN_SGB_V1
#define DIG_ATTN_SGB_V1 14                   // VALU instructions scheduled behind each of the four dP^T MFMAs (exp, pack P) ...
#define DIG_ATTN_SGB_V2 8                    // ... and behind each dV^T MFMA (P (dP - delta), pack dS)
#endif
#ifndef DIG_ATTN_SGB
#define DIG_ATTN_SGB 1                       // lab: 0 = the pipelined loops without sched_group_barrier directives (the compiler's own order)
#endif

#include "attn_tiles.h"

namespace {

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// DROP: attention dropout (Attention.attn_drop, modeling_finetune.py:116 / MultiHeadAttention.attn_drop,
// transformer_layer.py:271): the probabilities are normalised by the FULL row sum, then masked and scaled by 1/(1-p); the mask
// comes from dig_drop_keep(key, (query << 16) | key_index, image * H + head) and is regenerated in the backward.
// FULL: all eight 32-query blocks exist (self-attention over the 256 tokens): no per-block guards -- the guarded form costs the encoder's
// launches 3 us of 51 (tools/experiments/attn_fwd_lab.hip)
template <bool DROP, bool FULL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                          float* __restrict__ lse, int D, int H, unsigned qkv_bytes, dig_dropout_t drop,
                                                          int nqb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kt = smem;
  unsigned char* Vt = smem + TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / H, h = blockIdx.x - img * H;
  const int ld = 3 * D;
  const size_t tok0 = (size_t)img * N_TOK;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, qkv_bytes, 0x00020000);
  const unsigned base = (unsigned)((tok0 * ld + h * DH) * 2);
  stage_tile<256>(Kt, rs, base + (unsigned)(D * 2), ld, tid, wave);
  stage_tile<256>(Vt, rs, base + (unsigned)(2 * D * 2), ld, tid, wave);

  const int hi = lane >> 5;
  // Q fragments for both passes straight from global (each wave reads only its own 64 rows)
  bf16x8 qf[2][4];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int q = (wave * 2 + ps) * 32 + (lane & 31);
    const bf16_t* qp = qkv + (tok0 + q) * ld + h * DH + hi * 8;
    if (FULL || wave * 2 + ps < nqb) {                                   // nqb < 8: only the first nqb 32-query blocks exist (padded cross-attention)
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[ps][s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);
    }
  }
  __syncthreads();

#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qb = wave * 2 + ps;
    if (!FULL && qb >= nqb) continue;
    f32x16 sc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
    if (!DROP) {
      // k step outer, key tile inner: consecutive MFMAs go to eight independent accumulators (each accumulator still sums k in order)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct(Kt, kt * 32, s, lane), qf[ps][s], sc[kt], 0, 0, 0);
    } else {
      // (the dropout form keeps the key tile outer: with the mask arithmetic in the same registers the other order spills)
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct(Kt, kt * 32, s, lane), qf[ps][s], sc[kt], 0, 0, 0);
    }
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, sc[kt][e]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __expf(sc[kt][e] - m);
        sc[kt][e] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    if (DROP) {
      const unsigned qa = (unsigned)(qb * 32 + (lane & 31)) << 16;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned key = kt * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
          if (!dig_drop_keep(drop.k0, drop.k1, qa | key, blockIdx.x, drop.thr)) sc[kt][e] = 0.f;
        }
    }
    f32x16 oa[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oa[dt][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 pf = pack8(sc[kt], u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vt, kt * 32 + u * 16, dt * 32, lane), pf, oa[dt], 0, 0, 0);
      }
    }
    const float inv = (DROP ? drop.scale : 1.0f) / l;
    const int q = qb * 32 + (lane & 31);
    bf16_t* op = ctx + (tok0 + q) * D + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oa[dt][e] *= inv;
    store_rows(op, oa, hi);                                              // 16-byte stores: a lane pair trades column groups (v_permlane32_swap)
    if (hi == 0) lse[(size_t)blockIdx.x * N_TOK + q] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dqkv = d/d(qkv) given d(ctx); phase A (dQ, a wave owns query blocks), phase B (dK, dV, a wave owns key blocks)
// Column sums of a 32 x 64 block held as two 32x32 accumulators (lane & 31 = row; registers = columns): reduce over the 32
// rows of each lane half with DPP adds (quad xor 1, quad xor 2, half-row mirror, row mirror, row_bcast15 -- no LDS traffic;
// __shfl_xor lowers to ds_bpermute, 160 LDS round trips per block), then lanes 16 and 48 write their 2 x 16 columns into out[64].
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fold(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ void wave_colsum(const f32x16 (&acc)[2], float* out, int lane) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[dt][e];
      v = dpp_fold<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
      v = dpp_fold<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
      v = dpp_fold<0x141, 0xF>(v);       // row_half_mirror
      v = dpp_fold<0x140, 0xF>(v);       // row_mirror: every lane of a 16-lane row holds the row's sum
      v = dpp_fold<0x142, 0xA>(v);       // row_bcast15 into rows 1 and 3: lanes 16..31 / 48..63 hold their half's 32-row sum
      if ((lane & 31) == 16) out[dt * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// backward, two workgroups per CU: 4 waves per (image, head), 64 KiB of LDS time-shared between the two phases.
// Phase B (dK, dV) needs all of Q, dO in LDS and one key block's K / V rows in registers; phase A (dQ) needs all of K, V in
// LDS and one query block's Q / dO rows in registers.  So: stage Q, dO -> delta from the LDS copy of dO and a coalesced read
// of O -> phase B with K / V fragments fetched straight from global memory (each wave reads only its own 2 x 32 rows) ->
// restage the same 64 KiB with K, V (the Q / dO fragments of BOTH query blocks of the wave are lifted out of LDS first; the wave's
// second key block goes into the tiles from its registers, the LDS-DMA brings the even blocks) -> phase A.
// Two such workgroups fit one CU (2 x 78 KiB), so the staging bursts, LDS fragment reads and row stores of one
// (image, head) overlap the MFMA phases of another -- the 8-wave / 128 KiB form ran one workgroup per CU with nothing to
// overlap its serial phases (tools/experiments/attn_bwd_lab.hip: phase timeline and ablations).
// Round 6: with both tile loops EMPTY the launch took 95 of its 133 us -- 400 MB read + 151 MB written at the fabric's byte rate
// (profiles/r06_attn_bwd_lab.txt).  Hence the lifts above (192 instead of 256 KiB read per (image, head)), results that leave in full
// 128-byte lines, non-temporal (STORE = 3), and bias-gradient sums taken from rows the kernel handles anyway.
// ------------------------------------------------------------------------------------------------
constexpr int BWD_STG_OFF = 2 * TILE + 2 * N_TOK * 4 + 8 * 128 * 4;
constexpr int BWD_LDS = BWD_STG_OFF + 4 * 2048;                            // 78 KiB: two workgroups per CU
// PROJ (dig_attn_bwd_proj): `dctx` holds dy, the gradient of the projection's OUTPUT rows [R, D], and `projt` = Wproj^T [D in][D out]; the
// workgroup computes its own d(ctx) tile  dO[256 q, 64 d] = dy[256, D] Wproj[:, 64 h ..]  before anything else -- 96 (D = 384) MFMAs per wave,
// Wproj^T's 64 rows of this head through LDS (D / 64 sub-tiles [64 i][64 o] in layout U over T0 | T1), the dy rows as B fragments straight
// from global memory -- rounds it to bf16 as the projection's data-gradient GEMM does and parks it in T1 where the staging of d(ctx) would
// have put it.  d(ctx) never exists in HBM and the GEMM launch is gone.  The six heads of an image read the same dy rows: the block index is
// re-mapped so that they sit on ONE XCD (block b runs on XCD b mod 8) and share them through its L2.
template <bool DROP, int STORE, bool PROJ = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ ctx,
                                                           const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, int D, int H, float scale,
                                                           unsigned qkv_bytes, unsigned ctx_bytes, float* __restrict__ qsum, float* __restrict__ vsum,
                                                           dig_dropout_t drop, int nqb, const bf16_t* __restrict__ projt) {
  static_assert(!PROJ || (DIG_ATTN_LIFT & 1), "the fused projection gradient leaves no d(ctx) rows to re-read");
  constexpr int store_mode = STORE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* T0 = smem;                                              // Q, then K
  unsigned char* T1 = smem + TILE;                                       // dO, then V
  float* lse_s = reinterpret_cast<float*>(smem + 2 * TILE);              // [256]  -lse
  float* del_s = lse_s + N_TOK;                                          // [256]  -delta
  float* csum_s = del_s + N_TOK;                                         // [8 blocks][2][64]: column sums of dQ and dV (qsum only)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  if (PROJ && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);    // consecutive (image, head) pairs on one XCD
  const int img = bid / H, h = bid - img * H;
  const int ld = 3 * D;
  const size_t tok0 = (size_t)img * N_TOK;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, qkv_bytes, 0x00020000);
  const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)dctx, 0, ctx_bytes, 0x00020000);
  const unsigned base = (unsigned)((tok0 * ld + h * DH) * 2);
  const int hi = lane >> 5;
  const FragOff fo = frag_offsets(lane);
  DIG_ATTN_TS(0)
  // O rows for delta: 8 lanes cover one 128-byte row, 32 rows per pass (coalesced; a thread-per-row read of O and dO cost
  // a quarter of the kernel: every load instruction touched 64 different lines)
  bf16x8 orow[8];
  auto load_o = [&] {
#pragma unroll
    for (int ps = 0; ps < 8; ++ps)
      orow[ps] = *reinterpret_cast<const bf16x8*>(ctx + ((DIG_ATTN_B_ABL & 256) ? 0 : (tok0 + ps * 32 + (tid >> 3)) * D + h * DH + (tid & 7) * 8));
  };
  if constexpr (!PROJ) load_o();                                          // (PROJ: behind the projection -- its dy fragments need the registers)
  if constexpr (PROJ) {
    const int nk = D >> 6;                                                // 64-wide pieces of the reduction (even, <= 8: the host checks)
    const auto rp = __builtin_amdgcn_make_buffer_rsrc((void*)projt, 0, (unsigned)(D * D * 2), 0x00020000);
    for (int kk = 0; kk < nk; ++kk)
      stage_tile<256, 1, 512>(smem + kk * 8192, rp, (unsigned)(((size_t)h * DH * D + kk * 64) * 2), D, tid, wave);
    bf16x8 yf[4][2][4];                                                   // dy fragments [ring slot][query block][k step]: lane = query row, 8 contiguous o
    const bf16_t* yp = dctx + (tok0 + wave * 64 + (lane & 31)) * D + hi * 8;
    auto load_y = [&](bf16x8 (&y)[2][4], int kk) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int s = 0; s < 4; ++s) y[blk][s] = *reinterpret_cast<const bf16x8*>(yp + (size_t)blk * 32 * D + kk * 64 + s * 16);
    };
    f32x16 od[2][2];                                                      // dO^T [query block][dt]: lane = query, registers = d
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) od[blk][dt][e] = 0.f;
    auto mm = [&](const bf16x8 (&y)[2][4], int kk) {
      const unsigned char* wt = smem + kk * 8192;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 a = frag_direct_o(wt + dt * 4096, fo, s);         // Wproj^T rows 64 h + 32 dt + (lane & 31), o = 64 kk + 16 s + 8 hi ..
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) od[blk][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, y[blk][s], od[blk][dt], 0, 0, 0);
        }
    };
    // a ring of four 64-wide pieces of the dy rows in flight (128 registers): the row-per-lane loads are latency-bound, two pieces in flight
    // left three round trips exposed (18 k cycles per wave for 3 k cycles of MFMAs)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (kk < nk) load_y(yf[kk], kk);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk < nk) {
        mm(yf[kk & 3], kk);
        if (kk + 4 < nk) load_y(yf[kk & 3], kk + 4);
      }
    }
    __syncthreads();                                                      // every wave is done with Wproj^T: T0 | T1 are free
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int row = wave * 64 + blk * 32 + (lane & 31);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(T1 + row * 128 + (((4 * dt + g) ^ swz(row)) << 4) + hi * 8) =
              make_uint2(pack_bf2(od[blk][dt][4 * g], od[blk][dt][4 * g + 1]), pack_bf2(od[blk][dt][4 * g + 2], od[blk][dt][4 * g + 3]));
    }
    stage_tile<256>(T0, rs, base, ld, tid, wave);                                       // Q
    load_o();
  } else {
    stage_tile<256>(T0, rs, base, ld, tid, wave);                                       // Q
    stage_tile<256>(T1, rg, (unsigned)((tok0 * D + h * DH) * 2), D, tid, wave);         // dO
  }
  // K / V fragments of a key block, straight from global
  bf16x8 kf[4], vf[4];
  auto load_kv = [&](int kb) {
    const int key = kb * 32 + (lane & 31);
    const bf16_t* kp = qkv + (tok0 + key) * ld + D + h * DH + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + s * 16);
      vf[s] = *reinterpret_cast<const bf16x8*>(kp + D + s * 16);
    }
  };
  load_kv(wave * 2);
  lse_s[tid] = -lse[(size_t)bid * N_TOK + tid];              // negated: they seed the S / dP accumulators
  DIG_ATTN_TS(1)
  __syncthreads();
  DIG_ATTN_TS(2)
  // delta[q] = sum_d dO[q,d] * O[q,d]
  // ... and, without dropout, the v_bias gradient of this (image, head): the rows of P sum to one, so the column sums of dV = P^T dO over the keys
  // ARE the column sums of dO over the queries -- summed here from the rows this pass reads anyway (8 adds per row) instead of 160 DPP adds per
  // key block on the dV accumulators.  (The sums of the bf16 d(ctx) rows, in fp32: closer to the exact sum than the sums of the dV tiles were.)
  constexpr bool VSUM_DO = !DROP;
  float vs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    const int row = ps * 32 + (tid >> 3), c = tid & 7;
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(T1 + row * 128 + ((c ^ swz(row)) << 4));
    float acc = 0.f;
    const uint4 ow = __builtin_bit_cast(uint4, orow[ps]), gw = __builtin_bit_cast(uint4, gv);
    if (VSUM_DO && vsum) {
      const unsigned w4[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { vs8[2 * i] += __uint_as_float(w4[i] << 16); vs8[2 * i + 1] += __uint_as_float(w4[i] & 0xffff0000u); }
    }
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.x), __builtin_bit_cast(dig_bf16x2, gw.x), acc, false);   // v_dot2c_f32_bf16
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.y), __builtin_bit_cast(dig_bf16x2, gw.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.z), __builtin_bit_cast(dig_bf16x2, gw.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.w), __builtin_bit_cast(dig_bf16x2, gw.w), acc, false);
    // sum over the 8 lanes of the row with DPP moves (quad xor 1, quad xor 2, row_shl 4): __shfl_xor lowers to ds_bpermute,
    // an LDS round trip per step
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xF, 0xF, true));
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xF, 0xF, true));
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x104, 0xF, 0xF, true));
    if (c == 0) del_s[row] = -acc;
  }
  if (VSUM_DO && vsum) {                                                 // this wave's 64 queries -> slot `wave` of the dV sums; slots 4..7 stay zero
    lines_colsum_finish(vs8, csum_s + wave * 128 + 64, lane);
    if ((lane >> 3) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) csum_s[(wave + 4) * 128 + 64 + 8 * (lane & 7) + j] = 0.f;
    }
  }
  __syncthreads();
  DIG_ATTN_TS(3)

  // ---------------- phase B: dK, dV for key blocks 2*wave, 2*wave+1 ----------------
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int kb = wave * 2 + ps;
    const int k0 = kb * 32;
    const int key = k0 + (lane & 31);
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
    bf16x8 qfr[4], gfr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qfr[s] = frag_direct_o(T0, fo, s);
      gfr[s] = frag_direct_o(T1, fo, s);
    }
    // The accumulators of S^T and dP^T are seeded with -lse[q] and -delta[q] (rows = queries), so the MFMAs deliver
    // S - lse and dP - delta directly: 3 VALU ops per element (mul, exp, mul) instead of 5 and no [16] + [16] row constants
    // held in registers across the softmax arithmetic (the compiler had sunk half of them into the arithmetic as four
    // just-in-time LDS reads with a full lgkmcnt(0) stall each).  Dropout needs dP * mask - delta: seeded with 0 there.
    f32x16 st, dp;
    auto seed = [&](int qt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qr = qt * 32 + 8 * g + 4 * hi;
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
        st[g * 4] = l4.x; st[g * 4 + 1] = l4.y; st[g * 4 + 2] = l4.z; st[g * 4 + 3] = l4.w;
        if (!DROP) {
          const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
          dp[g * 4] = d4.x; dp[g * 4 + 1] = d4.y; dp[g * 4 + 2] = d4.z; dp[g * 4 + 3] = d4.w;
        } else {
          dp[g * 4] = 0.f; dp[g * 4 + 1] = 0.f; dp[g * 4 + 2] = 0.f; dp[g * 4 + 3] = 0.f;
        }
      }
    };
    seed(0);
    if constexpr (!DROP && DIG_ATTN_B_PIPE) {
      // One wave has nobody to hide behind while its partner on the SIMD waits for memory (two waves per SIMD, a third of a wave's life is
      // staging): the tile's serial chain  fragment reads -> 8 MFMAs -> 32 exp / mul + 16 converts -> 8 MFMAs  is re-ordered so that the matrix
      // pipe always has an instruction of THIS wave to run:  S^T (4 MFMAs) | dP^T (4) beside exp(S - lse) and the pack of P | dV^T (4, needs P
      // only) beside P (dP - delta) and the pack of dS | dK^T (4) beside the next tile's fragment and seed reads.  Same arithmetic per
      // element, same summation order in every accumulator: bit-identical to the fenced form.
      for (int qt = 0; qt < nqb; ++qt) {
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 gtr[2][2], qtr[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            gtr[u][dt] = frag_tr_o(T1 + qt * 4096 + u * 2048, fo, dt);
            qtr[u][dt] = frag_tr_o(T0 + qt * 4096 + u * 2048, fo, dt);
          }
#pragma unroll
        for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], st, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // dP^T | exp(S - lse), pack P
#pragma unroll
        for (int s = 0; s < 4; ++s) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[s], vf[s], dp, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = __expf(st[e]);
        bf16x8 pf[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) pf[u] = pack8(st, u);
#if DIG_ATTN_SGB
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, DIG_ATTN_SGB_V1, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // dV^T | P (dP - delta), pack dS
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtr[u][dt], pf[u], dv[dt], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = st[e] * dp[e];
#pragma unroll
        for (int u = 0; u < 2; ++u) ds[u] = pack8(dp, u);
#if DIG_ATTN_SGB
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, DIG_ATTN_SGB_V2, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // dK^T | the next tile's operand fragments and seeds
        const int qtn = (qt + 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qfr[s] = frag_direct_o(T0 + qtn * 4096, fo, s);
          gfr[s] = frag_direct_o(T1 + qtn * 4096, fo, s);
        }
        seed(qtn);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[u][dt], ds[u], dk[dt], 0, 0, 0);
#if DIG_ATTN_SGB
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
#endif
      }
    } else
    for (int qt = 0; qt < nqb; ++qt) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (DIG_ATTN_B_ABL & 8) break;
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[s], vf[s], dp, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 gtr[2][2], qtr[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          if (DIG_ATTN_B_ABL & 2) { gtr[u][dt] = kf[u * 2 + dt]; qtr[u][dt] = vf[u * 2 + dt]; continue; }
          gtr[u][dt] = frag_tr_o(T1 + qt * 4096 + u * 2048, fo, dt);
          qtr[u][dt] = frag_tr_o(T0 + qt * 4096 + u * 2048, fo, dt);
        }
      const int qtn = (qt + 1) & 7;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (DIG_ATTN_B_ABL & 32) break;
        qfr[s] = frag_direct_o(T0 + qtn * 4096, fo, s);
        gfr[s] = frag_direct_o(T1 + qtn * 4096, fo, s);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 pf[2], ds[2];
      if (!DROP) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (DIG_ATTN_B_ABL & 1) break;
          const float p = __expf(st[e]);
          st[e] = p;
          dp[e] = p * dp[e];
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 d4 = *reinterpret_cast<const float4*>(del_s + qt * 32 + 8 * g + 4 * hi);
          const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __expf(st[g * 4 + e]);
            const unsigned qi = qt * 32 + 8 * g + 4 * hi + e;
            const float m = dig_drop_keep(drop.k0, drop.k1, (qi << 16) | (unsigned)key, bid, drop.thr) ? drop.scale : 0.f;
            st[g * 4 + e] = p * m;                                          // dropped probabilities (for dV)
            dp[g * 4 + e] = p * (dp[g * 4 + e] * m + dl[e]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) { pf[u] = pack8(st, u); ds[u] = pack8(dp, u); }
      __builtin_amdgcn_sched_barrier(0);
      if (!(DIG_ATTN_B_ABL & 4)) seed(qtn);                                // next tile's seeds land while the output MFMAs run
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          if (DIG_ATTN_B_ABL & 16) { dv[dt][u] += __builtin_bit_cast(float, (int)gtr[u][dt][0] ^ (int)pf[u][1]); dk[dt][u] += __builtin_bit_cast(float, (int)qtr[u][dt][0] ^ (int)ds[u][1]); continue; }
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtr[u][dt], pf[u], dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[u][dt], ds[u], dk[dt], 0, 0, 0);
        }
    }
    if (ps == 0) { DIG_ATTN_TS(7) }
    if (ps == 0) load_kv(kb + 1);                                         // the next block's rows fly while this block's result is stored
    bf16_t* okp = dqkv + (tok0 + key) * ld + D + h * DH;
    if (!(DIG_ATTN_B_ABL & 128)) {
      if (store_mode == 3) {
        bf16_t* blk = dqkv + (tok0 + k0) * ld + D + h * DH;
        store_rows_lines<true>(blk, ld, dk, smem + BWD_STG_OFF + wave * 2048, lane);
        store_rows_lines<true>(blk + D, ld, dv, smem + BWD_STG_OFF + wave * 2048, lane);
      } else if (store_mode == 1) {
        store_rows<true>(okp, dk, hi);
        store_rows<true>(okp + D, dv, hi);
      } else {
        store_rows(okp, dk, hi);
        store_rows(okp + D, dv, hi);
      }
    } else if (dk[0][0] == 1.2345f && dv[1][3] == 5.4321f) okp[0] = 1;
    if (!VSUM_DO && vsum) wave_colsum(dv, csum_s + kb * 128 + 64, lane);
  }
  DIG_ATTN_TS(4)

  // ---------------- restage: K -> T0, V -> T1 (every wave is done with Q, dO) ----------------
  // What is already on the chip is not fetched again (the kernel runs at the fabric's byte rate: 95 us of its 133 are loads + stores with every
  // MFMA removed): the Q / dO fragments of BOTH query blocks of this wave are lifted out of LDS before it is overwritten (DIG_ATTN_LIFT bit 0;
  // without it the second block's come from global memory while the first block's result is stored), and the K / V rows of the wave's
  // second key block, still in its registers as phase B's fragments, are written into the tiles by the wave itself (bit 1): the LDS-DMA brings
  // only the even 32-row blocks.  256 -> 192 KiB read per (image, head).
  bf16x8 qf[4], gf[4], qf2[4], gf2[4];
  if (wave * 2 < nqb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = frag_direct_o(T0 + wave * 8192, fo, s);
      gf[s] = frag_direct_o(T1 + wave * 8192, fo, s);
    }
  }
  if ((DIG_ATTN_LIFT & 1) && wave * 2 + 1 < nqb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf2[s] = frag_direct_o(T0 + wave * 8192 + 4096, fo, s);
      gf2[s] = frag_direct_o(T1 + wave * 8192 + 4096, fo, s);
    }
  }
  auto load_qg = [&](int qb) {
    if (DIG_ATTN_LIFT & 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { qf[s] = qf2[s]; gf[s] = gf2[s]; }
      return;
    }
    const int q = qb * 32 + (lane & 31);
    const bf16_t* qp = qkv + (tok0 + q) * ld + h * DH + hi * 8;
    const bf16_t* gp = dctx + (tok0 + q) * D + h * DH + hi * 8;
    if (qb < nqb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);
        gf[s] = *reinterpret_cast<const bf16x8*>(gp + s * 16);
      }
    }
  };
  __syncthreads();
  if (DIG_ATTN_LIFT & 2) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {                                          // kf / vf hold key block 2 wave + 1 (rows 32 (2 wave + 1) + (lane & 31), chunk 2 s + hi)
      *reinterpret_cast<bf16x8*>(T0 + wave * 8192 + 4096 + fo.d[s]) = kf[s];     // (in front of the DMA: behind it the compiler would wait for vmcnt(0) first)
      *reinterpret_cast<bf16x8*>(T1 + wave * 8192 + 4096 + fo.d[s]) = vf[s];
    }
    stage_tile<256, 2>(T0, rs, base + (unsigned)(D * 2), ld, tid, wave);      // K, even 32-row blocks
    stage_tile<256, 2>(T1, rs, base + (unsigned)(2 * D * 2), ld, tid, wave);  // V
  } else {
    stage_tile<256>(T0, rs, base + (unsigned)(D * 2), ld, tid, wave);      // K
    stage_tile<256>(T1, rs, base + (unsigned)(2 * D * 2), ld, tid, wave);  // V
  }
  __syncthreads();
  DIG_ATTN_TS(5)

  // ---------------- phase A: dQ for query blocks 2*wave, 2*wave+1 ----------------
  float qs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qb = wave * 2 + ps;
    if (qb >= nqb) continue;
    const int q0 = qb * 32;
    const int q = q0 + (lane & 31);
    const float my_lse = lse_s[q], my_del = del_s[q];
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
    // Software pipeline over the 8 key tiles: operand fragments of tile kt+1 are requested from LDS right after the
    // S / dP MFMAs of tile kt have issued, and the transposed K fragments of tile kt while its softmax arithmetic runs
    bf16x8 kfr[4], vfr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kfr[s] = frag_direct_o(T0, fo, s);
      vfr[s] = frag_direct_o(T1, fo, s);
    }
#pragma unroll 2
    for (int kt = 0; kt < ((DIG_ATTN_B_ABL & 64) ? 0 : 8); ++kt) {
      // as in phase B the accumulators start from -lse[q] and -delta[q] (here one constant per lane: a lane owns a query), so the MFMAs
      // deliver S - lse and dP - delta and the element arithmetic is mul, exp, mul.  Dropout needs dP * mask - delta: seeded with 0 there.
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = DROP ? 0.f : my_lse; dp[e] = DROP ? 0.f : my_del; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], gf[s], dp, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 ktr[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ktr[u][dt] = frag_tr_o(T0 + kt * 4096 + u * 2048, fo, dt);
      const int ktn = (kt + 1) & 7;                                        // (the wrap-around load of the last tile is unused)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kfr[s] = frag_direct_o(T0 + ktn * 4096, fo, s);
        vfr[s] = frag_direct_o(T1 + ktn * 4096, fo, s);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (DROP) {                                                        // dP = mask * (dO V^T) / (1 - p)
          float g = dp[e];
          const unsigned key = kt * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
          g = dig_drop_keep(drop.k0, drop.k1, ((unsigned)q << 16) | key, bid, drop.thr) ? g * drop.scale : 0.f;
          st[e] = __expf(st[e] + my_lse) * (g + my_del);                   // dS^T (my_lse = -lse, my_del = -delta)
        } else {
          st[e] = __expf(st[e]) * dp[e];
        }
      }
      const bf16x8 ds0 = pack8(st, 0), ds1 = pack8(st, 1);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[0][dt], ds0, dq[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[1][dt], ds1, dq[dt], 0, 0, 0);
    }
    if (ps == 0) load_qg(qb + 1);                                         // next block's rows fly while this block's result is stored
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] *= scale;
    if (!(DIG_ATTN_B_ABL & 128)) {
      // (full-line form: the q_bias gradient = column sums of the rows as they are STORED, added up from the staging copy: 32 adds per block
      //  and one fold per wave instead of 160 DPP adds per block on the accumulators)
      if (store_mode == 3) store_rows_lines<true, true>(dqkv + (tok0 + q0) * ld + h * DH, ld, dq, smem + BWD_STG_OFF + wave * 2048, lane, qs8);
      else if (store_mode == 1) store_rows<true>(dqkv + (tok0 + q) * ld + h * DH, dq, hi);
      else store_rows(dqkv + (tok0 + q) * ld + h * DH, dq, hi);
    }
    else if (dq[0][0] == 1.2345f && dq[1][3] == 5.4321f) dqkv[0] = 1;
    if (qsum && store_mode != 3) wave_colsum(dq, csum_s + qb * 128, lane);
  }
  DIG_ATTN_TS(6)
  // fused q_bias / v_bias gradients: this (image, head)'s column sums of dQ and dV, one partial row per image
  if (qsum) {
    if (store_mode == 3) {                                                 // this wave's 64 query rows -> slot `wave`; slots 4..7 zero
      lines_colsum_finish(qs8, csum_s + wave * 128, lane);
      if ((lane >> 3) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) csum_s[(wave + 4) * 128 + 8 * (lane & 7) + j] = 0.f;
      }
    }
    __syncthreads();
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));                 // re-derived here: kept alive from the prologue the index cost the kernel a spilled register
    if (t < 128) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += csum_s[w * 128 + t];
      float* dst = t < 64 ? qsum : vsum;
      dst[(size_t)img * D + h * DH + (t & 63)] = a;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// backward, single pass: one 8-wave workgroup per (image, head), Q / K / V / dO / O read ONCE, five matrix products per (query tile, key tile)
// pair instead of the two-phase kernel's seven.
//   * wave w owns the 32 keys 32 w ..: their K / V rows sit in its registers (B operands: lane = key), dK^T / dV^T accumulate in registers
//     over the eight query tiles exactly as in phase B above;
//   * the same dS tile also feeds dQ (contraction over KEYS): a lane-owns-a-key accumulator cannot be multiplied along its lane dimension,
//     so the wave parks the bf16 dS tile in 2 KiB of its own LDS ([key][query] rows) and reads it back with the transposing read as the B
//     operand of dQ^T[d, q] += K^T[d, key] dS^T[key, q] (K^T fragments: the wave's K block, transposed once through LDS at the start);
//   * the eight waves' dQ^T contributions to one query tile are summed in LDS in fp32, in a FIXED order: at step s wave w works on query
//     tile (w + s) mod 8, so no two waves touch one tile within a step, a workgroup barrier separates the steps, and tile t receives wave
//     (t - s) mod 8's term at step s -- bit-reproducible, no atomics.  The partial tiles live in LDS in accumulator-register order
//     (lane-linear 16-byte pieces: conflict-free), the last step's owner scales, rounds and stores the rows.
// LDS: Q 32 KiB + dO 32 KiB + dQ 64 KiB + dS 16 KiB + lse / delta 2 KiB (+ 8 KiB column sums) = 146 (154) KiB: one workgroup per CU.
// ------------------------------------------------------------------------------------------------
constexpr int SP_DQ_OFF = 2 * TILE;
constexpr int SP_DS_OFF = SP_DQ_OFF + 65536;
constexpr int SP_VEC_OFF = SP_DS_OFF + 8 * 2048;
constexpr int SP_CS_OFF = SP_VEC_OFF + 2 * N_TOK * 4;
// the dS tile of a wave: [32 keys][32 queries] bf16, rows of 64 bytes in 8-byte pieces, piece index XORed with (row >> 1) & 7 (writes of a
// 16-lane group then cover all banks; a transposing read touches 4 rows x 4 pieces per 16 lanes: distinct banks with or without the XOR)
__device__ __forceinline__ int sp_ds_addr(int row, int col) { return row * 64 + ((((col >> 2) ^ (row >> 1)) & 7) << 3) + (col & 3) * 2; }

template <bool BIAS>
__global__ __launch_bounds__(512) void attn_bwd_sp_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ ctx,
                                                          const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int D, int H, float scale, unsigned qkv_bytes,
                                                          unsigned ctx_bytes, float* __restrict__ qsum, float* __restrict__ vsum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem;
  unsigned char* Gs = smem + TILE;
  float* lse_s = reinterpret_cast<float*>(smem + SP_VEC_OFF);            // [256]  -lse
  float* del_s = lse_s + N_TOK;                                          // [256]  -delta
  float* csum_s = reinterpret_cast<float*>(smem + SP_CS_OFF);            // BIAS: [8][2][64] column sums of dQ and dV per 32-row block
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / H, h = blockIdx.x - img * H;
  const int ld = 3 * D;
  const size_t tok0 = (size_t)img * N_TOK;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, qkv_bytes, 0x00020000);
  const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)dctx, 0, ctx_bytes, 0x00020000);
  const unsigned base = (unsigned)((tok0 * ld + h * DH) * 2);
  const int hi = lane >> 5, rr = lane & 31;
  const FragOff fo = frag_offsets(lane);
  // O rows for delta: 8 lanes cover one 128-byte row, 64 rows per pass
  bf16x8 orow[4];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps)
    orow[ps] = *reinterpret_cast<const bf16x8*>(ctx + (tok0 + ps * 64 + (tid >> 3)) * D + h * DH + (tid & 7) * 8);
  stage_tile<512>(Qs, rs, base, ld, tid, wave);                                           // Q
  stage_tile<512>(Gs, rg, (unsigned)((tok0 * D + h * DH) * 2), D, tid, wave);             // dO
  // this wave's K / V rows as B operands (lane = key)
  bf16x8 kf[4], vf[4];
  {
    const bf16_t* kp = qkv + (tok0 + wave * 32 + rr) * ld + D + h * DH + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + s * 16);
      vf[s] = *reinterpret_cast<const bf16x8*>(kp + D + s * 16);
    }
  }
  if (tid < N_TOK) lse_s[tid] = -lse[(size_t)blockIdx.x * N_TOK + tid];
  // the wave's K block once more, in layout U inside its (not yet used) dQ region: the source of the transposed fragments K^T[d, key]
  unsigned char* kst = smem + SP_DQ_OFF + wave * 8192;
#pragma unroll
  for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x8*>(kst + u_addr(rr, 16 * s + 8 * hi)) = kf[s];
  __syncthreads();
  // delta[q] = sum_d dO[q, d] O[q, d]
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 64 + (tid >> 3), c = tid & 7;
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(Gs + row * 128 + ((c ^ swz(row)) << 4));
    float acc = 0.f;
    const uint4 ow = __builtin_bit_cast(uint4, orow[ps]), gw = __builtin_bit_cast(uint4, gv);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.x), __builtin_bit_cast(dig_bf16x2, gw.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.y), __builtin_bit_cast(dig_bf16x2, gw.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.z), __builtin_bit_cast(dig_bf16x2, gw.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dig_bf16x2, ow.w), __builtin_bit_cast(dig_bf16x2, gw.w), acc, false);
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xF, 0xF, true));
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xF, 0xF, true));
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x104, 0xF, 0xF, true));
    if (c == 0) del_s[row] = -acc;
  }
  bf16x8 ktf[2][2];                                                      // K^T[d = 32 dt + (lane & 31), keys of 16-key half u] (frag_tr's row order)
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) ktf[u][dt] = frag_tr_o(kst + u * 2048, fo, dt);
  __syncthreads();

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
  unsigned char* dss = smem + SP_DS_OFF + wave * 2048;
  // transposed dS fragment addresses (rows = keys {16 u + 4 hi + 0..3 | + 8}, column = query lane & 31)
  int dsr[2][2];
  {
    const int i = lane & 15, col = ((lane >> 4) & 1) * 16 + (i & 3) * 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      dsr[u][0] = sp_ds_addr(16 * u + 4 * hi + (i >> 2), col);
      dsr[u][1] = sp_ds_addr(16 * u + 8 + 4 * hi + (i >> 2), col);
    }
  }
  typedef __attribute__((address_space(3))) bf16x4* lds4_p;
#pragma unroll 1
  for (int s = 0; s < 8; ++s) {
    const int qt = (wave + s) & 7;
    const unsigned char* Qt = Qs + qt * 4096;
    const unsigned char* Gt = Gs + qt * 4096;
    f32x16 st, dp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int qr = qt * 32 + 8 * g + 4 * hi;
      const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
      const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
      st[g * 4] = l4.x; st[g * 4 + 1] = l4.y; st[g * 4 + 2] = l4.z; st[g * 4 + 3] = l4.w;
      dp[g * 4] = d4.x; dp[g * 4 + 1] = d4.y; dp[g * 4 + 2] = d4.z; dp[g * 4 + 3] = d4.w;
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct_o(Qt, fo, k4), kf[k4], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct_o(Gt, fo, k4), vf[k4], dp, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __expf(st[e]);
      st[e] = pe;
      dp[e] = pe * dp[e];
    }
    bf16x8 pf[2], ds[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { pf[u] = pack8(st, u); ds[u] = pack8(dp, u); }
    // dS -> the wave's LDS tile: lane (key rr, half hi) holds queries 8 g + 4 hi + 0..3 of its key
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint4 w = __builtin_bit_cast(uint4, ds[g >> 1]);
      *reinterpret_cast<uint2*>(dss + sp_ds_addr(rr, 8 * g + 4 * hi)) = (g & 1) ? make_uint2(w.z, w.w) : make_uint2(w.x, w.y);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_o(Gt + u * 2048, fo, dt), pf[u], dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_o(Qt + u * 2048, fo, dt), ds[u], dk[dt], 0, 0, 0);
      }
    // dQ^T[d, q] of this (key block, query tile) pair
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_p)(dss + dsr[u][0]));
      const bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_p)(dss + dsr[u][1]));
      const bf16x8 dst = __builtin_shufflevector(lo, h4, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[u][dt], dst, dq[dt], 0, 0, 0);
    }
    // into the tile's fp32 sum (accumulator-register order: piece r = 4 dt + g of lane l at (r 64 + l) 16 bytes)
    float* acc = reinterpret_cast<float*>(smem + SP_DQ_OFF + qt * 8192) + lane * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {dq[dt][4 * g], dq[dt][4 * g + 1], dq[dt][4 * g + 2], dq[dt][4 * g + 3]};
        f32x4* a = reinterpret_cast<f32x4*>(acc + (dt * 4 + g) * 256);
        if (s > 0) v += *a;
        *a = v;
      }
#if !(DIG_ATTN_SP_LAB & 1)
    __syncthreads();
#endif
  }
  // ---- results: dK, dV rows of the wave's keys; dQ rows of query tile `wave`
  {
    const int key = wave * 32 + rr;
    bf16_t* okp = dqkv + (tok0 + key) * ld + D + h * DH;
    store_rows(okp, dk, hi);
    store_rows(okp + D, dv, hi);
    if (BIAS) wave_colsum(dv, csum_s + wave * 128 + 64, lane);
    f32x16 dq[2];
    const float* acc = reinterpret_cast<const float*>(smem + SP_DQ_OFF + wave * 8192) + lane * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(acc + (dt * 4 + g) * 256);
        dq[dt][4 * g] = v[0] * scale; dq[dt][4 * g + 1] = v[1] * scale; dq[dt][4 * g + 2] = v[2] * scale; dq[dt][4 * g + 3] = v[3] * scale;
      }
    store_rows(dqkv + (tok0 + key) * ld + h * DH, dq, hi);
    if (BIAS) wave_colsum(dq, csum_s + wave * 128, lane);
  }
  if (BIAS) {
    __syncthreads();
    if (tid < 128) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += csum_s[w * 128 + tid];
      float* dst = tid < 64 ? qsum : vsum;
      dst[(size_t)img * D + h * DH + (tid & 63)] = a;
    }
  }
}

}  // namespace

// which backward kernel dig_attn_bwd launches for full self-attention without dropout: 1 = single pass (attn_bwd_sp_kernel), 0 = two phases
// (read by dig_attn_bwd on autograd's thread while a caller's thread may set it: an atomic, read once per launch)
static std::atomic<int> g_attn_bwd_single_pass{DIG_ATTN_BWD_SP_DEFAULT};
// how the two-phase kernel's results leave (DIG_ATTN_BWD_STORE above); process-wide, read once per launch
static std::atomic<int> g_attn_bwd_store{DIG_ATTN_BWD_STORE};
extern "C" int dig_attn_bwd_store(int mode) {
  if (mode == 0 || mode == 1 || mode == 3) return g_attn_bwd_store.exchange(mode, std::memory_order_relaxed);
  return g_attn_bwd_store.load(std::memory_order_relaxed);
}
extern "C" int dig_attn_bwd_mode(int single_pass) {
  if (single_pass == 0 || single_pass == 1) return g_attn_bwd_single_pass.exchange(single_pass, std::memory_order_relaxed);
  return g_attn_bwd_single_pass.load(std::memory_order_relaxed);
}

extern "C" int dig_attn_fwd_dropout(const void* qkv, void* ctx, float* lse, int n_img, int heads, int embed_dim,
                                    const dig_dropout_t* drop, int q_rows, hipStream_t stream) {
  if (!qkv || !ctx || !lse || n_img <= 0 || heads <= 0 || embed_dim != heads * DH || q_rows < 1 || q_rows > N_TOK) return DIG_ERR_ARG;
  const int nqb = (q_rows + 31) / 32;
  if (!aligned16(qkv) || !aligned16(ctx)) return DIG_ERR_ALIGN;
  const size_t qb = (size_t)n_img * N_TOK * 3 * embed_dim * 2;
  if (qb >= (1ull << 32)) return DIG_ERR_ARG;
  static bool attr[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr[dev]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    attr[dev] = true;
  }
  const bool dropping = drop && drop->thr;
  const dig_dropout_t dr = dropping ? *drop : dig_dropout_t{};
#define DIG_ATTN_FWD_LAUNCH(DR, FU)                                                                                                       \
  dig_launch(attn_fwd_kernel<DR, FU>, dim3(n_img * heads), dim3(256), 2 * TILE, stream, (const bf16_t*)qkv, (bf16_t*)ctx, lse, embed_dim, \
             heads, (unsigned)qb, dr, nqb)
  if (dropping) DIG_ATTN_FWD_LAUNCH(true, false);                       // (the unguarded dropout form spills 36 registers: it keeps the guards)
  else { if (nqb == 8) DIG_ATTN_FWD_LAUNCH(false, true); else DIG_ATTN_FWD_LAUNCH(false, false); }
#undef DIG_ATTN_FWD_LAUNCH
  return dig_check_launch();
}

extern "C" int dig_attn_fwd(const void* qkv, void* ctx, float* lse, int n_img, int heads, int embed_dim, hipStream_t stream) {
  return dig_attn_fwd_dropout(qkv, ctx, lse, n_img, heads, embed_dim, nullptr, N_TOK, stream);
}

extern "C" int dig_attn_bwd_dropout(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int n_img,
                                    int heads, int embed_dim, float scale, float* q_colsum, float* v_colsum,
                                    const dig_dropout_t* drop, int q_rows, hipStream_t stream) {
  if (!qkv || !ctx || !dctx || !lse || !dqkv || n_img <= 0 || heads <= 0 || embed_dim != heads * DH || q_rows < 1 || q_rows > N_TOK) return DIG_ERR_ARG;
  const int nqb = (q_rows + 31) / 32;
  if (nqb < 8 && q_colsum) return DIG_ERR_UNSUPPORTED;                       // the fused dQ column sums assume all eight query blocks
  if ((q_colsum == nullptr) != (v_colsum == nullptr)) return DIG_ERR_ARG;
  if (!aligned16(qkv) || !aligned16(ctx) || !aligned16(dctx) || !aligned16(dqkv)) return DIG_ERR_ALIGN;
  const size_t qb = (size_t)n_img * N_TOK * 3 * embed_dim * 2;
  if (qb >= (1ull << 32)) return DIG_ERR_ARG;
  const int dev = dig_device();
  if (g_attn_bwd_single_pass.load(std::memory_order_relaxed) && !(drop && drop->thr) && nqb == 8) {
    static bool sp_attr[DIG_MAX_DEVICES] = {};
    if (!sp_attr[dev]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_sp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_CS_OFF);
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_sp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SP_CS_OFF + 8 * 128 * 4);
      sp_attr[dev] = true;
    }
    if (q_colsum)
      dig_launch(attn_bwd_sp_kernel<true>, dim3(n_img * heads), dim3(512), (unsigned)(SP_CS_OFF + 8 * 128 * 4), stream, (const bf16_t*)qkv, (const bf16_t*)ctx,
                 (const bf16_t*)dctx, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum, v_colsum);
    else
      dig_launch(attn_bwd_sp_kernel<false>, dim3(n_img * heads), dim3(512), (unsigned)SP_CS_OFF, stream, (const bf16_t*)qkv, (const bf16_t*)ctx,
                 (const bf16_t*)dctx, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum, v_colsum);
    return dig_check_launch();
  }
  const int lds = BWD_LDS;
  const int store_mode = g_attn_bwd_store.load(std::memory_order_relaxed);
  static bool attr[DIG_MAX_DEVICES] = {};
  if (!attr[dev]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<true, DIG_ATTN_BWD_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    attr[dev] = true;
  }
  if (drop && drop->thr)
    dig_launch(attn_bwd_kernel<true, DIG_ATTN_BWD_STORE>, dim3(n_img * heads), dim3(256), (unsigned)lds, stream, (const bf16_t*)qkv, (const bf16_t*)ctx,
               (const bf16_t*)dctx, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum, v_colsum, *drop, nqb, (const bf16_t*)nullptr);
  else {
    auto* k = store_mode == 3 ? attn_bwd_kernel<false, 3> : (store_mode == 1 ? attn_bwd_kernel<false, 1> : attn_bwd_kernel<false, 0>);
    dig_launch(k, dim3(n_img * heads), dim3(256), (unsigned)lds, stream, (const bf16_t*)qkv, (const bf16_t*)ctx,
               (const bf16_t*)dctx, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum, v_colsum,
               dig_dropout_t{}, nqb, (const bf16_t*)nullptr);
  }
  return dig_check_launch();
}

// The projection's data gradient inside the attention backward (attn_bwd_kernel<.., PROJ>): dy [R, D] = gradient of the projection's output
// rows, projt = Wproj^T [D in][D out] (bf16).  D a multiple of 128, at most 512 (the weight slice of a head fills the two operand tiles).
extern "C" int dig_attn_bwd_proj(const void* qkv, const void* ctx, const void* dy, const void* projt, const float* lse, void* dqkv, int n_img,
                                 int heads, int embed_dim, float scale, float* q_colsum, float* v_colsum, hipStream_t stream) {
  if (!qkv || !ctx || !dy || !projt || !lse || !dqkv || n_img <= 0 || heads <= 0) return DIG_ERR_ARG;
  if ((q_colsum == nullptr) != (v_colsum == nullptr)) return DIG_ERR_ARG;
  if (embed_dim != heads * DH || (embed_dim & 127) || embed_dim > 512) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(qkv) || !aligned16(ctx) || !aligned16(dy) || !aligned16(projt) || !aligned16(dqkv)) return DIG_ERR_ALIGN;
  const size_t qb = (size_t)n_img * N_TOK * 3 * embed_dim * 2;
  if (qb >= (1ull << 32)) return DIG_ERR_ARG;
  const int dev = dig_device();
  static bool attr[DIG_MAX_DEVICES] = {};
  if (!attr[dev]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false, DIG_ATTN_BWD_STORE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    attr[dev] = true;
  }
  dig_launch(attn_bwd_kernel<false, DIG_ATTN_BWD_STORE, true>, dim3(n_img * heads), dim3(256), (unsigned)BWD_LDS, stream, (const bf16_t*)qkv,
             (const bf16_t*)ctx, (const bf16_t*)dy, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum, v_colsum,
             dig_dropout_t{}, 8, (const bf16_t*)projt);
  return dig_check_launch();
}

extern "C" int dig_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int n_img,
                            int heads, int embed_dim, float scale, float* q_colsum, float* v_colsum, hipStream_t stream) {
  return dig_attn_bwd_dropout(qkv, ctx, dctx, lse, dqkv, n_img, heads, embed_dim, scale, q_colsum, v_colsum, nullptr, N_TOK, stream);
}
