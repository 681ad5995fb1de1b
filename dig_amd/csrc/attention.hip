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

namespace {

constexpr int N_TOK = 256;
constexpr int DH = 64;
constexpr int TILE = N_TOK * DH * 2;  // 32 KiB

__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int u_addr(int row, int col) { return row * 128 + ((((col >> 3) ^ swz(row))) << 4) + (col & 7) * 2; }

// Stage a [256 x 64] bf16 tile (row stride ld elements, starting at element offset base) into LDS layout U.
// 2048 16-B pieces; NT threads.
template <int NT>
__device__ __forceinline__ void stage_tile(unsigned char* lds, __amdgpu_buffer_rsrc_t rs, unsigned base_bytes, int ld,
                                           int tid, int wave) {
#pragma unroll
  for (int it = 0; it < 2048 / NT; ++it) {
    const int piece = it * NT + tid;
    const int row = piece >> 3, pc = piece & 7;
    const int c = pc ^ swz(row);
    const unsigned off = base_bytes + (unsigned)((row * ld + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + (it * NT + wave * 64) * 16), 16, off, 0, 0, 0);
  }
}

// 8 contiguous bf16 (cols 16*s + 8*hi ..) of row (rowoff + lane&31)
__device__ __forceinline__ bf16x8 frag_direct(const unsigned char* tile, int rowoff, int s, int lane) {
  const int row = rowoff + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ swz(row)) << 4));
}

// Transposed fragment: for column (coloff + lane&31) return rows {r0 + 4*hi + 0..3, r0 + 8 + 4*hi + 0..3}
// (the row permutation of a 32x32 MFMA accumulator quad pair).
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* tile, int r0, int coloff, int lane) {
  const int hi = lane >> 5;
  const int i = lane & 15;
  const int col = coloff + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
  const int ra = r0 + 4 * hi + (i >> 2);
  const int rb = ra + 8;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + u_addr(ra, col)));
  bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + u_addr(rb, col)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
  return r;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int u) {
  const uint4 w = make_uint4(pack_bf2(a[u * 8], a[u * 8 + 1]), pack_bf2(a[u * 8 + 2], a[u * 8 + 3]),
                             pack_bf2(a[u * 8 + 4], a[u * 8 + 5]), pack_bf2(a[u * 8 + 6], a[u * 8 + 7]));
  return __builtin_bit_cast(bf16x8, w);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// DROP: attention dropout (Attention.attn_drop, modeling_finetune.py:116 / MultiHeadAttention.attn_drop,
// transformer_layer.py:271): the probabilities are normalised by the FULL row sum, then masked and scaled by 1/(1-p); the mask
// comes from dig_drop_keep(key, (query << 16) | key_index, image * H + head) and is regenerated in the backward.
template <bool DROP>
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
    if (wave * 2 + ps < nqb) {                                           // nqb < 8: only the first nqb 32-query blocks exist (padded cross-attention)
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[ps][s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);
    }
  }
  __syncthreads();

#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qb = wave * 2 + ps;
    if (qb >= nqb) continue;
    f32x16 sc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
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
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * hi;
        *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf2(oa[dt][g * 4] * inv, oa[dt][g * 4 + 1] * inv),
                                                       pack_bf2(oa[dt][g * 4 + 2] * inv, oa[dt][g * 4 + 3] * inv));
      }
    if (hi == 0) lse[(size_t)blockIdx.x * N_TOK + q] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dqkv = d/d(qkv) given d(ctx); 8 waves, phase A (dQ, wave = query block), phase B (dK,dV, wave = key block)
// ------------------------------------------------------------------------------------------------
// Store one lane-row of a 32 x 64 result held as two 32x32 MFMA accumulators (lane = row, registers = 4-column groups
// interleaved between the two lane halves).  v_permlane32_swap trades column groups between lane l and l+32 so that each
// lane owns 16 contiguous columns per accumulator: 4 x 16-byte stores per row instead of 8 x 8-byte ones.
__device__ __forceinline__ void store_rows(bf16_t* row, const f32x16 (&acc)[2], int hi) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    unsigned P[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      P[g][0] = pack_bf2(acc[dt][g * 4], acc[dt][g * 4 + 1]);
      P[g][1] = pack_bf2(acc[dt][g * 4 + 2], acc[dt][g * 4 + 3]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(P[0][k], P[2][k], false, false);
      P[0][k] = r0[0]; P[2][k] = r0[1];
      const auto r1 = __builtin_amdgcn_permlane32_swap(P[1][k], P[3][k], false, false);
      P[1][k] = r1[0]; P[3][k] = r1[1];
    }
    bf16_t* o = row + dt * 32 + hi * 16;
    *reinterpret_cast<uint4*>(o) = make_uint4(P[0][0], P[0][1], P[2][0], P[2][1]);
    *reinterpret_cast<uint4*>(o + 8) = make_uint4(P[1][0], P[1][1], P[3][0], P[3][1]);
  }
}

// Column sums of a 32 x 64 block held as two 32x32 accumulators (lane & 31 = row; registers = columns): reduce over the 32
// rows of each lane half, then lanes 0 and 32 write their 2 x 16 columns into out[64].
__device__ __forceinline__ void wave_colsum(const f32x16 (&acc)[2], float* out, int lane) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[dt][e];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
      if ((lane & 31) == 0) out[dt * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3)] = v;
    }
}

template <bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ ctx,
                                                          const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int D, int H, float scale,
                                                          unsigned qkv_bytes, unsigned ctx_bytes, float* __restrict__ qsum, float* __restrict__ vsum,
                                                          dig_dropout_t drop, int nqb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qt = smem;
  unsigned char* Kt = smem + TILE;
  unsigned char* Vt = smem + 2 * TILE;
  unsigned char* Gt = smem + 3 * TILE;                                   // dO
  float* lse_s = reinterpret_cast<float*>(smem + 4 * TILE);              // [256]
  float* del_s = lse_s + N_TOK;                                          // [256]
  float* csum_s = del_s + N_TOK;                                         // [8 waves][2][64]: column sums of dQ and dV
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / H, h = blockIdx.x - img * H;
  const int ld = 3 * D;
  const size_t tok0 = (size_t)img * N_TOK;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, qkv_bytes, 0x00020000);
  const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)dctx, 0, ctx_bytes, 0x00020000);
  const unsigned base = (unsigned)((tok0 * ld + h * DH) * 2);
  stage_tile<512>(Qt, rs, base, ld, tid, wave);
  stage_tile<512>(Kt, rs, base + (unsigned)(D * 2), ld, tid, wave);
  stage_tile<512>(Vt, rs, base + (unsigned)(2 * D * 2), ld, tid, wave);
  stage_tile<512>(Gt, rg, (unsigned)((tok0 * D + h * DH) * 2), D, tid, wave);
  // delta[q] = sum_d dO[q,d] * O[q,d]; two threads per query (32 d each)
  {
    const int q = tid >> 1, half = tid & 1;
    const bf16_t* o = ctx + (tok0 + q) * D + h * DH + half * 32;
    const bf16_t* g = dctx + (tok0 + q) * D + h * DH + half * 32;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o + c * 8);
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += bf2f((bf16_t)ov[e]) * bf2f((bf16_t)gv[e]);
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) {
      del_s[q] = acc;
      lse_s[q] = lse[(size_t)blockIdx.x * N_TOK + q];
    }
  }
  __syncthreads();
  const int hi = lane >> 5;

  // ---------------- phase A: dQ for query block `wave` (only the first nqb blocks exist) ----------------
  if (wave < nqb) {
    const int q0 = wave * 32;
    const int q = q0 + (lane & 31);
    const float my_lse = lse_s[q], my_del = del_s[q];
    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = frag_direct(Qt, q0, s, lane);
      gf[s] = frag_direct(Gt, q0, s, lane);
    }
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
    // Software pipeline over the 8 key tiles: operand fragments of tile kt+1 are requested from LDS right after the
    // S / dP MFMAs of tile kt have issued, and the transposed K fragments of tile kt while its softmax arithmetic runs, so
    // no MFMA waits on an LDS round trip (the compiler's own schedule put every ds_read directly in front of its MFMA).
    bf16x8 kfr[4], vfr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kfr[s] = frag_direct(Kt, 0, s, lane);
      vfr[s] = frag_direct(Vt, 0, s, lane);
    }
#pragma unroll 2
    for (int kt = 0; kt < 8; ++kt) {
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
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
        for (int dt = 0; dt < 2; ++dt) ktr[u][dt] = frag_tr(Kt, kt * 32 + u * 16, dt * 32, lane);
      const int ktn = (kt + 1) & 7;                                        // (the wrap-around load of the last tile is unused)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kfr[s] = frag_direct(Kt, ktn * 32, s, lane);
        vfr[s] = frag_direct(Vt, ktn * 32, s, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float g = dp[e];
        if (DROP) {                                                        // dP = mask * (dO V^T) / (1 - p)
          const unsigned key = kt * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
          g = dig_drop_keep(drop.k0, drop.k1, ((unsigned)q << 16) | key, blockIdx.x, drop.thr) ? g * drop.scale : 0.f;
        }
        st[e] = __expf(st[e] - my_lse) * (g - my_del);                     // dS^T
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 ds = pack8(st, u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[u][dt], ds, dq[dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] *= scale;
    store_rows(dqkv + (tok0 + q) * ld + h * DH, dq, hi);
    if (qsum) wave_colsum(dq, csum_s + wave * 128, lane);
  }

  // ---------------- phase B: dK, dV for key block `wave` ----------------
  {
    const int k0 = wave * 32;
    const int key = k0 + (lane & 31);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = frag_direct(Kt, k0, s, lane);
      vf[s] = frag_direct(Vt, k0, s, lane);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
    bf16x8 qfr[4], gfr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qfr[s] = frag_direct(Qt, 0, s, lane);
      gfr[s] = frag_direct(Gt, 0, s, lane);
    }
#pragma unroll 2
    for (int qt = 0; qt < nqb; ++qt) {
      f32x16 st, dp;   // rows = queries qt*32 + (e&3) + 8*(e>>2) + 4*hi, col = key
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[s], vf[s], dp, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 gtr[2][2], qtr[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          gtr[u][dt] = frag_tr(Gt, qt * 32 + u * 16, dt * 32, lane);
          qtr[u][dt] = frag_tr(Qt, qt * 32 + u * 16, dt * 32, lane);
        }
      float ls[4][4], dl[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qr = qt * 32 + 8 * g + 4 * hi;
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
        ls[g][0] = l4.x; ls[g][1] = l4.y; ls[g][2] = l4.z; ls[g][3] = l4.w;
        dl[g][0] = d4.x; dl[g][1] = d4.y; dl[g][2] = d4.z; dl[g][3] = d4.w;
      }
      const int qtn = (qt + 1) & 7;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qfr[s] = frag_direct(Qt, qtn * 32, s, lane);
        gfr[s] = frag_direct(Gt, qtn * 32, s, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __expf(st[g * 4 + e] - ls[g][e]);
          float m = 1.f;
          if (DROP) {
            const unsigned qi = qt * 32 + 8 * g + 4 * hi + e;
            m = dig_drop_keep(drop.k0, drop.k1, (qi << 16) | (unsigned)key, blockIdx.x, drop.thr) ? drop.scale : 0.f;
          }
          st[g * 4 + e] = p * m;                                          // dropped probabilities (for dV)
          dp[g * 4 + e] = p * (dp[g * 4 + e] * m - dl[g][e]);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 pf = pack8(st, u);
        const bf16x8 ds = pack8(dp, u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtr[u][dt], pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[u][dt], ds, dk[dt], 0, 0, 0);
        }
      }
    }
    bf16_t* okp = dqkv + (tok0 + key) * ld + D + h * DH;
    store_rows(okp, dk, hi);
    store_rows(okp + D, dv, hi);
    if (vsum) wave_colsum(dv, csum_s + wave * 128 + 64, lane);
  }
  // fused q_bias / v_bias gradients: this (image, head)'s column sums of dQ and dV, one partial row per image
  if (qsum) {
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

extern "C" int dig_attn_fwd_dropout(const void* qkv, void* ctx, float* lse, int n_img, int heads, int embed_dim,
                                    const dig_dropout_t* drop, int q_rows, hipStream_t stream) {
  if (!qkv || !ctx || !lse || n_img <= 0 || heads <= 0 || embed_dim != heads * DH || q_rows < 1 || q_rows > N_TOK) return DIG_ERR_ARG;
  const int nqb = (q_rows + 31) / 32;
  if (!aligned16(qkv) || !aligned16(ctx)) return DIG_ERR_ALIGN;
  const size_t qb = (size_t)n_img * N_TOK * 3 * embed_dim * 2;
  if (qb >= (1ull << 32)) return DIG_ERR_ARG;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    attr = true;
  }
  if (drop && drop->thr)
    hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(n_img * heads), dim3(256), 2 * TILE, stream, (const bf16_t*)qkv, (bf16_t*)ctx,
                       lse, embed_dim, heads, (unsigned)qb, *drop, nqb);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(n_img * heads), dim3(256), 2 * TILE, stream, (const bf16_t*)qkv, (bf16_t*)ctx,
                       lse, embed_dim, heads, (unsigned)qb, dig_dropout_t{}, nqb);
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
  const int lds = 4 * TILE + 2 * N_TOK * 4 + 8 * 128 * 4;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  if (drop && drop->thr)
    hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(n_img * heads), dim3(512), lds, stream, (const bf16_t*)qkv, (const bf16_t*)ctx,
                       (const bf16_t*)dctx, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum,
                       v_colsum, *drop, nqb);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(n_img * heads), dim3(512), lds, stream, (const bf16_t*)qkv, (const bf16_t*)ctx,
                       (const bf16_t*)dctx, lse, (bf16_t*)dqkv, embed_dim, heads, scale, (unsigned)qb, (unsigned)(qb / 3), q_colsum,
                       v_colsum, dig_dropout_t{}, nqb);
  return dig_check_launch();
}

extern "C" int dig_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int n_img,
                            int heads, int embed_dim, float scale, float* q_colsum, float* v_colsum, hipStream_t stream) {
  return dig_attn_bwd_dropout(qkv, ctx, dctx, lse, dqkv, n_img, heads, embed_dim, scale, q_colsum, v_colsum, nullptr, N_TOK, stream);
}
