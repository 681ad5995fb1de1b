// bf16 MFMA GEMM for gfx950 (MI355X):  C[i,j] = sum_r opA(i,r) * opB(j,r)
//
// One kernel template covers the three contractions of a Linear layer (reference: nn.Linear inside
// modeling_finetune.py:43-125 and modeling_pretrain_moco_mim_ori.py:463-482, plus their autograd):
//   forward  y = x W^T     : A = x  [I=rows , R=in ] direct,      B = W  [J=out, R=in ] direct
//   dgrad    dx = dy W     : A = dy [I=rows , R=out] direct,      B = W  [R=out, J=in ] transposed storage
//   wgrad    dW = dy^T x   : A = dy [R=rows , I=out] transposed,  B = x  [R=rows, J=in ] transposed
// "direct" = the reduction index r is the contiguous one; "transposed storage" = r is the row index.
//
// Design (CDNA4): 128x128x64 tile per 256-thread workgroup (4 waves as 2x2, 64x64 each, 2x2
// v_mfma_f32_32x32x16_bf16 per 16-deep substep).  Operand tiles go HBM -> LDS with
// buffer_load_dwordx4 ... lds (no VGPR round trip; out-of-range rows are zero-filled by the buffer
// bounds check, which is what makes ragged I/J/R edges exact), double-buffered, one barrier per K-tile.
//   * direct tiles: [128 rows][64 r] with the 16-B chunk index XOR-swizzled by (row>>1)&7 on the SOURCE
//     address (LDS-DMA destinations are lane-linear), read back with ds_read_b128 conflict-free;
//   * transposed tiles: [64 r][128 c] stored as 128-B blocks of [4 r][16 c], read with
//     ds_read_b64_tr_b16 so that each lane receives 4 consecutive r of one column.
// The MFMA is issued with swapped operands (D' = B_frag x A_frag) so that every lane owns ONE output
// row and 4 consecutive output columns per accumulator quad: epilogue loads/stores are 8-16 B wide.
// Epilogue (runtime flags): + bias[j], * alpha on the first alpha_cols columns (q scaling,
// modeling_finetune.py:97), exact-erf GELU with optional pre-activation store, + residual, and either
// bf16 / fp32 store or fp32 atomic accumulate (split-R wgrad).
// Workgroup -> tile mapping is XCD-aware (block b runs on XCD b%8): each XCD walks a contiguous range of
// tiles with j fastest, so an A row-panel is re-read from that XCD's L2, not from HBM.
#include "common.h"

namespace {

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int I, J, R;
  int lda, ldb, ldc;
  unsigned a_bytes, b_bytes;
  const float* bias;
  const bf16_t* resid;
  int ldr;
  bf16_t* pre;
  int ldp;
  float alpha;
  int alpha_cols;
  int act;
  int r_per_split;
  int tiles_i, tiles_j;
};

constexpr int BI = 128, BJ = 128, BR = 64;
constexpr int TILE_BYTES = BI * BR * 2;  // 16 KiB per operand tile (both layouts)

template <bool T>
__device__ __forceinline__ unsigned stage_offset(int piece, int row0, int r0, int ld) {
  if (!T) {
    const int row = piece >> 3, pc = piece & 7;
    const int c = pc ^ ((row >> 1) & 7);
    return (unsigned)(((row0 + row) * ld + r0 + c * 8) * 2);
  } else {
    const int block = piece >> 3, w = piece & 7;
    const int rr = w >> 1, half = w & 1;
    const int rb = block >> 3, nb = block & 7;
    return (unsigned)(((r0 + rb * 4 + rr) * ld + row0 + nb * 16 + half * 8) * 2);
  }
}

// 8 bf16 of the reduction dim (substep s, k-slot group hi = lane>>5) for tile row/col (rowoff + (lane&31)).
template <bool T>
__device__ __forceinline__ bf16x8 load_frag(const unsigned char* tile, int rowoff, int s, int lane) {
  if (!T) {
    const int row = rowoff + (lane & 31);
    const int chunk = 2 * s + (lane >> 5);
    const int byte = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(tile + byte);
  } else {
    const int hi = lane >> 5;
    const int nb = (rowoff >> 4) + ((lane >> 4) & 1);
    const int rb = s * 4 + hi * 2;
    const unsigned char* p0 = tile + (rb * 8 + nb) * 128 + (lane & 15) * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p0));
    bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p0 + 8 * 128));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return r;
  }
}

template <bool TA, bool TB, int OUT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;

  const int nblk = p.tiles_i * p.tiles_j;
  const int bid = blockIdx.x;
  const int q = nblk >> 3, rm = nblk & 7, xcd = bid & 7;
  const int logical = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (bid >> 3);
  const int ti = logical / p.tiles_j, tj = logical - ti * p.tiles_j;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int rbeg = blockIdx.z * p.r_per_split;
  const int rend = min(p.R, rbeg + p.r_per_split);
  const int nt = (rend - rbeg + BR - 1) / BR;

  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  unsigned offA[4], offB[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = it * 256 + tid;
    offA[it] = stage_offset<TA>(piece, i0, rbeg, p.lda);
    offB[it] = stage_offset<TB>(piece, j0, rbeg, p.ldb);
  }
  const unsigned stepA = TA ? (unsigned)(BR * p.lda * 2) : (unsigned)(BR * 2);
  const unsigned stepB = TB ? (unsigned)(BR * p.ldb * 2) : (unsigned)(BR * 2);

  auto stage = [&](int buf) {
    unsigned char* a = smem + buf * 2 * TILE_BYTES + wave * 1024;
    unsigned char* b = a + TILE_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(a + it * 4096), 16, offA[it], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, LDS_PTR(b + it * 4096), 16, offB[it], 0, 0, 0);
      offA[it] += stepA;
      offB[it] += stepB;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  if (nt > 0) {
    stage(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) stage((t + 1) & 1);
      const unsigned char* at = smem + (t & 1) * 2 * TILE_BYTES;
      const unsigned char* bt = at + TILE_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          af[u] = load_frag<TA>(at, wi * 64 + u * 32, s, lane);
          bfr[u] = load_frag<TB>(bt, wj * 64 + u * 32, s, lane);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: lane owns output row i, columns j = jb + 8*g + 4*hi + (0..3) ----
  const int hi = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int i = i0 + wi * 64 + a * 32 + (lane & 31);
    if (i >= p.I) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = j0 + wj * 64 + b * 32 + 8 * g + 4 * hi;
        if (j >= p.J) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][g * 4 + e];
        if (OUT == 2) {
          float* c = reinterpret_cast<float*>(p.C) + (size_t)i * p.ldc + j;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < p.J) atomicAdd(c + e, v[e]);
          continue;
        }
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < p.J) v[e] += p.bias[j + e];
        }
        if (j < p.alpha_cols) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        }
        const bool full = (j + 4 <= p.J);
        if (p.act == 1) {
          if (p.pre) {
            bf16_t* pp = p.pre + (size_t)i * p.ldp + j;
            if (full) {
              *reinterpret_cast<uint2*>(pp) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else {
              for (int e = 0; e < 4; ++e)
                if (j + e < p.J) pp[e] = f2bf(v[e]);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
        }
        if (p.resid) {
          const bf16_t* rp = p.resid + (size_t)i * p.ldr + j;
          if (full) {
            const uint2 rr = *reinterpret_cast<const uint2*>(rp);
            v[0] += bf2f((bf16_t)(rr.x & 0xffff)); v[1] += bf2f((bf16_t)(rr.x >> 16));
            v[2] += bf2f((bf16_t)(rr.y & 0xffff)); v[3] += bf2f((bf16_t)(rr.y >> 16));
          } else {
            for (int e = 0; e < 4; ++e)
              if (j + e < p.J) v[e] += bf2f(rp[e]);
          }
        }
        if (OUT == 0) {
          bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)i * p.ldc + j;
          if (full) {
            *reinterpret_cast<uint2*>(c) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
          } else {
            for (int e = 0; e < 4; ++e)
              if (j + e < p.J) c[e] = f2bf(v[e]);
          }
        } else {
          float* c = reinterpret_cast<float*>(p.C) + (size_t)i * p.ldc + j;
          if (full) {
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            for (int e = 0; e < 4; ++e)
              if (j + e < p.J) c[e] = v[e];
          }
        }
      }
    }
  }
}

template <bool TA, bool TB, int OUT>
int launch(const GemmParams& p, int splits, hipStream_t stream) {
  static bool attr_set = false;
  auto k = gemm_kernel<TA, TB, OUT>;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    attr_set = true;
  }
  dim3 grid(p.tiles_i * p.tiles_j, 1, splits);
  hipLaunchKernelGGL(k, grid, dim3(256), 4 * TILE_BYTES, stream, p);
  return dig_check_launch();
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_gemm_bf16(const void* A, const void* B, void* C, int I, int J, int R, int lda, int ldb, int ldc,
                             int trans_a, int trans_b, int out_kind, const float* bias, const void* resid, int ldr,
                             void* pre_act, int ldp, float alpha, int alpha_cols, int act, int splits, int a_rows,
                             int b_rows, hipStream_t stream) {
  if (!A || !B || !C || I <= 0 || J <= 0 || R <= 0 || splits < 1) return DIG_ERR_ARG;
  if (out_kind < 0 || out_kind > 2 || act < 0 || act > 1) return DIG_ERR_ARG;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (lda & 7) || (ldb & 7)) return DIG_ERR_ALIGN;
  if ((ldc & 3) || (resid && ((ldr & 3) || (((uintptr_t)resid) & 7))) || (pre_act && (ldp & 3))) return DIG_ERR_ALIGN;
  if (!trans_a && (R % BR)) return DIG_ERR_ARG;   // direct operands need R % 64 == 0 (row-wrap would pollute)
  if (!trans_b && (R % BR)) return DIG_ERR_ARG;
  if (out_kind == 2 && (bias || resid || act)) return DIG_ERR_ARG;
  if (out_kind != 2 && splits != 1) return DIG_ERR_ARG;
  GemmParams p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.I = I; p.J = J; p.R = R; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  // a_rows / b_rows (0 = default) bound the rows that really exist in memory; rows past them read as zero.
  const size_t ab = (size_t)(a_rows > 0 ? a_rows : (trans_a ? R : I)) * lda * 2;
  const size_t bb = (size_t)(b_rows > 0 ? b_rows : (trans_b ? R : J)) * ldb * 2;
  if (ab >= (1ull << 32) || bb >= (1ull << 32)) return DIG_ERR_ARG;
  p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  p.bias = bias; p.resid = (const bf16_t*)resid; p.ldr = ldr; p.pre = (bf16_t*)pre_act; p.ldp = ldp;
  p.alpha = alpha; p.alpha_cols = alpha_cols; p.act = act;
  const int rtiles = (R + BR - 1) / BR;
  p.r_per_split = ((rtiles + splits - 1) / splits) * BR;
  splits = (R + p.r_per_split - 1) / p.r_per_split;
  p.tiles_i = (I + BI - 1) / BI; p.tiles_j = (J + BJ - 1) / BJ;
#define DIG_GEMM_CASE(ta, tb, o) \
  if ((trans_a != 0) == ta && (trans_b != 0) == tb && out_kind == o) return launch<ta, tb, o>(p, splits, stream);
  DIG_GEMM_CASE(false, false, 0)
  DIG_GEMM_CASE(false, false, 1)
  DIG_GEMM_CASE(false, true, 0)
  DIG_GEMM_CASE(false, true, 1)
  DIG_GEMM_CASE(true, true, 1)
  DIG_GEMM_CASE(true, true, 2)
#undef DIG_GEMM_CASE
  return DIG_ERR_UNSUPPORTED;
}
