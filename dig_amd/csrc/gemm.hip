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
// bf16 / fp32 store, or (split-R wgrad) an fp32 partial slab per R-slice that dig_reduce_partials then sums into
// the gradient arena -- device-scope fp32 atomics go to the memory side on a multi-XCD part and measured ~8x slower.
// Tile variants (the `bk` argument, include/dig_hip.h DIG_GEMM_TILE_*): 128x128 with K step 64 (forward default) or 32 (default
// when an operand is read transposed), 256x256 / 256x192 (16 / 12 waves: tall layers), 64x128 / 128x64 (layers with few rows).
// Everything else that was measured -- 3-/4-stage rings, persistent workgroups for both tile families, 4x2 / 2x4 / 4x4 MFMA
// blocks per wave, a 256x128 two-workgroups-per-CU form -- lost or tied on the shapes of this model and is not built any more
// (numbers: profiles/r01_gemm_variants.txt, profiles/r02_gemm_lab.txt).
// Workgroup -> tile mapping is XCD-aware (block b runs on XCD b%8): each XCD walks a contiguous range of
// tiles with j fastest, so an A row-panel is re-read from that XCD's L2, not from HBM.
#include "common.h"
#include <type_traits>

// phase time stamps for tools/experiments/gemm_lab.hip (empty in the product build)
#ifndef DIG_GEMM_TS
#define DIG_GEMM_TS(i)
#define DIG_GEMM_PW_BEGIN()
#define DIG_GEMM_PW_ACC(k)
#define DIG_GEMM_PW_END()
#endif

namespace {

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int I, J, R;
  int lda, ldb, ldc;
  unsigned a_bytes, b_bytes;
  unsigned c_bytes;                       // persistent forward tiles: bytes of C (= of pre_act / resid: same leading dimension required)
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
  float* colsum;                          // act 2 only: [ceil(I/64)][J] per-64-row column sums of the result (fp32), or null
  int splits_x;                           // >0: 1-D grid of tiles*splits blocks, every R-split pinned to one XCD
};
// dropout / drop-path of the result before the residual add: a second kernel argument of the DROP instantiations only (the plain
// kernels take a 4-byte dummy, their GemmParams and code stay what they were)
template <bool DROP> struct DropArg { int unused; };
template <> struct DropArg<true> { dig_dropout_t d; };
__device__ __forceinline__ void epilogue_drop(float (&v)[8], const DropArg<true>& a, int i, int j, int cols) { dig_drop_apply8(v, a.d, i, j, cols); }
__device__ __forceinline__ void epilogue_drop(float (&)[8], const DropArg<false>&, int, int, int) {}

constexpr int BI = 128, BJ = 128;
constexpr int BR = 64;                   // granule of the reduction dim (R % 64 rule for direct operands, split slabs)

// chunk swizzle of the direct layout: rows are BK*2 bytes; 16 consecutive rows must hit 16 distinct 16-B slots of a
// 256-B bank row for ds_read_b128
template <int BK>
__device__ __forceinline__ int dswz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

template <bool T, int BK>
__device__ __forceinline__ unsigned stage_offset(int piece, int row0, int r0, int ld) {
  if (!T) {
    constexpr int CPR = BK / 8;            // 16-B chunks per row
    const int row = piece / CPR, pc = piece % CPR;
    const int c = pc ^ dswz<BK>(row);
    return (unsigned)(((row0 + row) * ld + r0 + c * 8) * 2);
  } else {
    const int block = piece >> 3, w = piece & 7;
    const int rr = w >> 1, half = w & 1;
    const int rb = block >> 3, nb = block & 7;
    return (unsigned)(((r0 + rb * 4 + rr) * ld + row0 + nb * 16 + half * 8) * 2);
  }
}

// 8 bf16 of the reduction dim (substep s, k-slot group hi = lane>>5) for tile row/col (rowoff + (lane&31)).
template <bool T, int BK>
__device__ __forceinline__ bf16x8 load_frag(const unsigned char* tile, int rowoff, int s, int lane) {
  if (!T) {
    const int row = rowoff + (lane & 31);
    const int chunk = 2 * s + (lane >> 5);
    const int byte = row * (BK * 2) + ((chunk ^ dswz<BK>(row)) << 4);
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

template <int BK>
struct TileCfg {
  static constexpr int TILE_BYTES = BI * BK * 2;        // one operand tile (either layout)
  static constexpr int NIT = TILE_BYTES / 16 / 256;     // 16-B pieces per thread per operand tile
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NSTG-stage LDS ring: (NSTG-1) K-tiles of operand data are in flight while one is being multiplied; the hand-off is a
// counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads() would drain every LDS-DMA in flight).
// DROP: the dropout / drop-path epilogue (fine-tune step only) is a separate instantiation, so that the pre-training kernels
// carry none of its code or registers.
template <bool TA, bool TB, int OUT, int BK, bool RES, int NSTG, bool DROP = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p, DropArg<DROP> da) {
  constexpr int TILE_BYTES = TileCfg<BK>::TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;

  const int nblk = p.tiles_i * p.tiles_j;
  const int bid = blockIdx.x;
  int logical, split;
  if (p.splits_x > 0) {
    // split-R (weight-gradient) launches: all tiles of one R-split run back-to-back on ONE XCD (block b -> XCD b%8), so the
    // row window of both operands that the split streams is fetched from HBM once and re-read from that XCD's L2 by the
    // other tiles.  (Spreading a split's tiles over the XCDs fetched every operand ~2.3x: profiles/r01_pmc_traffic.json.)
    const int k = bid >> 3;
    const int round = k / nblk;
    split = (bid & 7) + 8 * round;
    logical = k - round * nblk;
  } else {
    const int q = nblk >> 3, rm = nblk & 7, xcd = bid & 7;
    logical = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (bid >> 3);
    split = blockIdx.z;
  }
  const int ti = logical / p.tiles_j, tj = logical - ti * p.tiles_j;
  const int i0 = ti * BI, j0 = tj * BJ;
  const int rbeg = split * p.r_per_split;
  const int rend = min(p.R, rbeg + p.r_per_split);
  const int nt = (rend - rbeg + BK - 1) / BK;

  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  unsigned offA[4], offB[4];   // (a dependent bound here makes hipcc 7.2 drop the host-side kernel stub)
#pragma unroll
  for (int it = 0; it < TileCfg<BK>::NIT; ++it) {
    const int piece = it * 256 + tid;
    offA[it] = stage_offset<TA, BK>(piece, i0, rbeg, p.lda);
    offB[it] = stage_offset<TB, BK>(piece, j0, rbeg, p.ldb);
  }
  const unsigned stepA = TA ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
  const unsigned stepB = TB ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);

  auto stage = [&](int buf) {
    unsigned char* a = smem + buf * 2 * TILE_BYTES + wave * 1024;
    unsigned char* b = a + TILE_BYTES;
#pragma unroll
    for (int it = 0; it < TileCfg<BK>::NIT; ++it) {
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

  {
    constexpr int LA = NSTG - 1;                                   // look-ahead (stages in flight)
    constexpr int LPS = 2 * TileCfg<BK>::NIT;                      // LDS-DMA instructions per thread per stage
#pragma unroll
    for (int q = 0; q < LA; ++q)
      if (q < nt) stage(q);
    for (int t = 0; t < nt; ++t) {
      if (t + LA - 1 < nt) wait_vmcnt<(LA - 1) * LPS>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                                // stage t landed for every wave; slot (t-1)%NSTG is free
      if (t + LA < nt) stage((t + LA) % NSTG);
      const unsigned char* at = smem + (t % NSTG) * 2 * TILE_BYTES;
      const unsigned char* bt = at + TILE_BYTES;
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          af[u] = load_frag<TA, BK>(at, wi * 64 + u * 32, s, lane);
          bfr[u] = load_frag<TB, BK>(bt, wj * 64 + u * 32, s, lane);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();                                               // every wave is done with the ring: reuse it as staging
  }

  // ---- epilogue: C-shuffle.  Each wave parks its 64x64 fp32 tile in its own 16 KiB of the (now idle) LDS with a
  // 16-B-chunk XOR swizzle, then re-reads it row-contiguously: 8 lanes cover one 64-column row segment, so every
  // global access of the epilogue (bias, residual, pre-activation, output) is a 16/32-B-per-lane, 128/256-B-per-row
  // coalesced transaction instead of 32 scattered 8-B pieces.
  const int cg = lane & 7;
  const int j = j0 + wj * 64 + cg * 8;
  const bool jok = j < p.J;                                        // J % 8 == 0 (host-checked)
  const int jc = jok ? j : 0;
  // issue every global read of the epilogue up front (rows clamped, so the loads are unconditional and overlap)
  uint4 rres[RES ? 8 : 1];
  if (RES) {
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int i = min(i0 + wi * 64 + ps * 8 + (lane >> 3), p.I - 1);
      rres[ps] = *reinterpret_cast<const uint4*>(p.resid + (size_t)i * p.ldr + jc);
    }
  }
  float bias8[8];
  if (OUT != 2 && p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + jc);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + jc + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
    bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  }
  float* stg = reinterpret_cast<float*>(smem + wave * 8192);      // 32 rows x 64 fp32 per wave, per half
  const float al = (j < p.alpha_cols) ? p.alpha : 1.0f;
  float* cpart = reinterpret_cast<float*>(p.C);
  if (OUT == 2) cpart += (size_t)split * p.I * p.ldc;         // split-R partial slab
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    {
      const int hi = lane >> 5, row = lane & 31;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int chunk = (b * 32 + 8 * g + 4 * hi) >> 2;          // 16-B chunk index within the 64-float row
          *reinterpret_cast<float4*>(stg + row * 64 + ((chunk ^ (row & 15)) << 2)) =
              make_float4(acc[a][b][g * 4], acc[a][b][g * 4 + 1], acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]);
        }
    }
    // wave-local LDS hand-off: only LDS ordering is needed (a memory fence here would also drain vmcnt, i.e. wait for
    // the previous half's global stores to land)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ps = a * 4 + q;
      const int row = q * 8 + (lane >> 3);
      const int i = i0 + wi * 64 + a * 32 + row;
      const float4 x0 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * cg) ^ (row & 15)) << 2));
      const float4 x1 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * cg + 1) ^ (row & 15)) << 2));
      const bool live = (i < p.I) && jok;
      float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if (OUT == 2) {
        if (live) {
          float* c = cpart + (size_t)i * p.ldc + j;
          *reinterpret_cast<float4*>(c) = x0;
          *reinterpret_cast<float4*>(c + 4) = x1;
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias8[e]) * al;
      if (p.act == 1) {
        if (p.pre && live)
          *reinterpret_cast<uint4*>(p.pre + (size_t)i * p.ldp + j) =
              make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        if (DROP) epilogue_drop(v, da, i, j, p.J);
      } else if (RES && p.act == 2) {                              // multiply by gelu'(pre): fused GELU backward (fc2 dgrad)
        const unsigned w[4] = {rres[ps].x, rres[ps].y, rres[ps].z, rres[ps].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] *= dgelu_f(bf2f((bf16_t)(w[e] & 0xffff))); v[2 * e + 1] *= dgelu_f(bf2f((bf16_t)(w[e] >> 16))); }
        if (DROP) epilogue_drop(v, da, i, j, p.J);
        if (live) {
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[e] += v[e];
        }
      } else if (DROP) {
        epilogue_drop(v, da, i, j, p.J);
      }
      if (RES && p.act != 2) {
        const unsigned w[4] = {rres[ps].x, rres[ps].y, rres[ps].z, rres[ps].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += bf2f((bf16_t)(w[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(w[e] >> 16)); }
      }
      if (!live) continue;
      if (OUT == 0) {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)i * p.ldc + j) =
            make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      } else {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)i * p.ldc + j;
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    // wave-local LDS hand-off: only LDS ordering is needed (a memory fence here would also drain vmcnt, i.e. wait for
    // the previous half's global stores to land)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  // fused bias gradient of the layer below (fc1): column sums of this wave's 64 rows, one partial row per 64 output rows
  if (RES && OUT == 0 && p.act == 2 && p.colsum) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      csum[e] += __shfl_xor(csum[e], 8, 64);
      csum[e] += __shfl_xor(csum[e], 16, 64);
      csum[e] += __shfl_xor(csum[e], 32, 64);
    }
    if (lane < 8 && jok && i0 + wi * 64 < p.I) {
      float* c = p.colsum + (size_t)(ti * 2 + wi) * p.J + j;
      *reinterpret_cast<float4*>(c) = make_float4(csum[0], csum[1], csum[2], csum[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// Wide-tile variant: WM x WN waves of 64x64 each (256x256, 256x128 or 128x256 outputs per workgroup, BK = 64, 2 stages).
// Per-wave code is the same as gemm_kernel; what changes is the operand traffic: a 256x256 tile moves half the L2->LDS
// bytes per FLOP of a 128x128 tile, and that traffic (measured ~55 GB/s per CU) is what bounds the K-loop on MI355X.
template <bool T, int COLS, int BK>
__device__ __forceinline__ unsigned wstage_offset(int piece, int row0, int r0, int ld) {
  if (!T) {
    constexpr int CPR = BK / 8;
    const int row = piece / CPR, pc = piece % CPR;
    const int c = pc ^ dswz<BK>(row);
    return (unsigned)(((row0 + row) * ld + r0 + c * 8) * 2);
  } else {
    constexpr int NBLK = COLS / 16;
    const int block = piece >> 3, w = piece & 7;
    const int rr = w >> 1, half = w & 1;
    const int rb = block / NBLK, nb = block % NBLK;
    return (unsigned)(((r0 + rb * 4 + rr) * ld + row0 + nb * 16 + half * 8) * 2);
  }
}

template <bool T, int COLS, int BK>
__device__ __forceinline__ bf16x8 wload_frag(const unsigned char* tile, int rowoff, int s, int lane) {
  if (!T) {
    const int row = rowoff + (lane & 31);
    const int chunk = 2 * s + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(tile + row * (BK * 2) + ((chunk ^ dswz<BK>(row)) << 4));
  } else {
    constexpr int NBLK = COLS / 16;
    const int hi = lane >> 5;
    const int nb = (rowoff >> 4) + ((lane >> 4) & 1);
    const int rb = s * 4 + hi * 2;
    const unsigned char* p0 = tile + (rb * NBLK + nb) * 128 + (lane & 15) * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p0));
    bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(p0 + NBLK * 128));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi4[0]; r[5] = hi4[1]; r[6] = hi4[2]; r[7] = hi4[3];
    return r;
  }
}

// WM x WN waves, each owning an (FM*32) x (FN*32) block of the tile as FM x FN MFMA accumulators.  Per 16-deep K step a wave
// reads FM + FN operand fragments from LDS for FM*FN MFMAs: 2x2 -> 1 read per MFMA, 4x2 -> 0.75, 4x4 -> 0.5 (the LDS pipe, not
// the matrix core, bounds the 2x2 shape: MI355X_MICROARCH.md LDS table, 256 B/clk).
template <int WM, int WN, int FM, int FN, int BK, int NSTG>
struct WideCfg {
  static constexpr int NT = 64 * WM * WN;
  static constexpr int TBI = 32 * FM * WM, TBJ = 32 * FN * WN;
  static constexpr int A_BYTES = TBI * BK * 2, B_BYTES = TBJ * BK * 2;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int PA = A_BYTES / 16, PB = B_BYTES / 16;                  // 16-byte pieces of the two operand tiles
  static constexpr int NITA = (PA + NT - 1) / NT, NITB = (PB + NT - 1) / NT;   // passes of the whole workgroup (the last one may be partial:
  static constexpr bool RAGGED = (PA % NT) || (PB % NT);                       //  whole waves skip it -- 12-wave tiles; NSTG must be 2 then)
  static constexpr int LDS = (NSTG * STAGE > (NT / 64) * 8192) ? NSTG * STAGE : (NT / 64) * 8192;
};

// BK = 64 / NSTG = 2: one stage in flight (wait-all hand-off).  BK = 32 / NSTG = 4: three half-depth stages in flight with a
// counted s_waitcnt vmcnt -- a 256x256x32 step is long enough (about 0.4 us of MFMA) for that look-ahead to cover the
// L2/HBM -> LDS latency, which a 128x128 tile's step is not.
template <bool TA, bool TB, int OUT, int WM, int WN, int FM, int FN, bool RES, int BK, int NSTG, bool DROP = false>
__global__ __launch_bounds__(64 * WM * WN, (BK == 32 && WM * WN == 8 && FM * FN == 4) ? 4 : ((BK == 32 && WM * WN == 4) ? 2 : 1)) void gemm_wide_kernel(GemmParams p, DropArg<DROP> da) {
  using Cfg = WideCfg<WM, WN, FM, FN, BK, NSTG>;
  static_assert(FN % 2 == 0 && Cfg::NITA <= 8 && Cfg::NITB <= 8, "tile shape");
  static_assert(!Cfg::RAGGED || NSTG == 2, "a partial staging pass changes the per-wave load count: only the wait-all hand-off is safe");
  static_assert(Cfg::PA % 64 == 0 && Cfg::PB % 64 == 0, "whole waves take part in a staging pass or skip it");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WN, wj = wave % WN;

  DIG_GEMM_TS(0)
  const int nblk = p.tiles_i * p.tiles_j;
  const int bid = blockIdx.x;
  int logical, split;
  if (p.splits_x > 0) {                        // split-R: every split pinned to one XCD (see gemm_kernel)
    const int k = bid >> 3, round = k / nblk;
    split = (bid & 7) + 8 * round;
    logical = k - round * nblk;
  } else {
    const int q = nblk >> 3, rm = nblk & 7, xcd = bid & 7;
    logical = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (bid >> 3);
    split = blockIdx.z;
  }
  const int ti = logical / p.tiles_j, tj = logical - ti * p.tiles_j;
  const int i0 = ti * Cfg::TBI, j0 = tj * Cfg::TBJ;
  const int rbeg = split * p.r_per_split;
  const int rend = min(p.R, rbeg + p.r_per_split);
  const int nt = (rend - rbeg + BK - 1) / BK;

  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  unsigned offA[8], offB[8];
#pragma unroll
  for (int it = 0; it < Cfg::NITA; ++it) offA[it] = wstage_offset<TA, Cfg::TBI, BK>(it * Cfg::NT + tid, i0, rbeg, p.lda);
#pragma unroll
  for (int it = 0; it < Cfg::NITB; ++it) offB[it] = wstage_offset<TB, Cfg::TBJ, BK>(it * Cfg::NT + tid, j0, rbeg, p.ldb);
  const unsigned stepA = TA ? (unsigned)(BK * p.lda * 2) : (unsigned)(BK * 2);
  const unsigned stepB = TB ? (unsigned)(BK * p.ldb * 2) : (unsigned)(BK * 2);
  auto stage = [&](int slot) {
    unsigned char* a = smem + slot * Cfg::STAGE + wave * 1024;
    unsigned char* b = a + Cfg::A_BYTES;
#pragma unroll
    for (int it = 0; it < Cfg::NITA; ++it) {
      if (!Cfg::RAGGED || it * Cfg::NT + wave * 64 < Cfg::PA)                 // (wave-uniform: piece counts are multiples of 64)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(a + it * Cfg::NT * 16), 16, offA[it], 0, 0, 0);
      offA[it] += stepA;
    }
#pragma unroll
    for (int it = 0; it < Cfg::NITB; ++it) {
      if (!Cfg::RAGGED || it * Cfg::NT + wave * 64 < Cfg::PB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, LDS_PTR(b + it * Cfg::NT * 16), 16, offB[it], 0, 0, 0);
      offB[it] += stepB;
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  constexpr int LA = NSTG - 1;
  constexpr int LPS = Cfg::NITA + Cfg::NITB;
#pragma unroll
  for (int q = 0; q < LA; ++q)
    if (q < nt) stage(q);
  for (int t = 0; t < nt; ++t) {
    if (t + LA - 1 < nt) wait_vmcnt<(LA - 1) * LPS>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t == 0) { DIG_GEMM_TS(1) }
    if (t + LA < nt) stage((t + LA) % NSTG);
    const unsigned char* at = smem + (t % NSTG) * Cfg::STAGE;
    const unsigned char* bt = at + Cfg::A_BYTES;
    // fragments of K-step s+1 are fetched from LDS while the MFMAs of step s run (register double buffer)
    bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
    for (int u = 0; u < FM; ++u) af[0][u] = wload_frag<TA, Cfg::TBI, BK>(at, wi * (32 * FM) + u * 32, 0, lane);
#pragma unroll
    for (int u = 0; u < FN; ++u) bfr[0][u] = wload_frag<TB, Cfg::TBJ, BK>(bt, wj * (32 * FN) + u * 32, 0, lane);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      if (s + 1 < BK / 16) {
#pragma unroll
        for (int u = 0; u < FM; ++u) af[(s + 1) & 1][u] = wload_frag<TA, Cfg::TBI, BK>(at, wi * (32 * FM) + u * 32, s + 1, lane);
#pragma unroll
        for (int u = 0; u < FN; ++u) bfr[(s + 1) & 1][u] = wload_frag<TB, Cfg::TBJ, BK>(bt, wj * (32 * FN) + u * 32, s + 1, lane);
      }
      __builtin_amdgcn_sched_barrier(0);             // keep the LDS reads ahead of the MFMAs they overlap with
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[s & 1][b], af[s & 1][a], acc[a][b], 0, 0, 0);
    }
  }
  DIG_GEMM_TS(2)
  __syncthreads();
  DIG_GEMM_TS(3)

  // ---- epilogue (C-shuffle per wave in 32-row x 64-column pieces; same contract as gemm_kernel)
  const int cg = lane & 7;
  float* stg = reinterpret_cast<float*>(smem + wave * 8192);
  float* cpart = reinterpret_cast<float*>(p.C);
  if (OUT == 2) cpart += (size_t)split * p.I * p.ldc;
#pragma unroll
  for (int bh = 0; bh < FN / 2; ++bh) {
  const int j = j0 + wj * (32 * FN) + bh * 64 + cg * 8;
  const bool jok = j < p.J;
  const int jc = jok ? j : 0;
  float bias8[8];
  if (OUT != 2 && p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + jc);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + jc + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
    bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  }
  const float al = (j < p.alpha_cols) ? p.alpha : 1.0f;
  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
  for (int a = 0; a < FM; ++a) {
    uint4 rres[RES ? 4 : 1];
    if (RES) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int i = min(i0 + wi * (32 * FM) + a * 32 + q4 * 8 + (lane >> 3), p.I - 1);
        rres[q4] = *reinterpret_cast<const uint4*>(p.resid + (size_t)i * p.ldr + jc);
      }
    }
    {
      const int hi = lane >> 5, row = lane & 31;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int chunk = (b * 32 + 8 * g + 4 * hi) >> 2;
          *reinterpret_cast<float4*>(stg + row * 64 + ((chunk ^ (row & 15)) << 2)) =
              make_float4(acc[a][2 * bh + b][g * 4], acc[a][2 * bh + b][g * 4 + 1], acc[a][2 * bh + b][g * 4 + 2], acc[a][2 * bh + b][g * 4 + 3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int row = q4 * 8 + (lane >> 3);
      const int i = i0 + wi * (32 * FM) + a * 32 + row;
      const float4 x0 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * cg) ^ (row & 15)) << 2));
      const float4 x1 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * cg + 1) ^ (row & 15)) << 2));
      const bool live = (i < p.I) && jok;
      float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if (OUT == 2) {
        if (live) {
          float* c = cpart + (size_t)i * p.ldc + j;
          *reinterpret_cast<float4*>(c) = x0;
          *reinterpret_cast<float4*>(c + 4) = x1;
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias8[e]) * al;
      if (p.act == 1) {
        if (p.pre && live)
          *reinterpret_cast<uint4*>(p.pre + (size_t)i * p.ldp + j) =
              make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        if (DROP) epilogue_drop(v, da, i, j, p.J);
      } else if (RES && p.act == 2) {
        const unsigned w[4] = {rres[q4].x, rres[q4].y, rres[q4].z, rres[q4].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] *= dgelu_f(bf2f((bf16_t)(w[e] & 0xffff))); v[2 * e + 1] *= dgelu_f(bf2f((bf16_t)(w[e] >> 16))); }
        if (DROP) epilogue_drop(v, da, i, j, p.J);
        if (live) {
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[e] += v[e];
        }
      } else if (DROP) {
        epilogue_drop(v, da, i, j, p.J);
      }
      if (RES && p.act != 2) {
        const unsigned w[4] = {rres[q4].x, rres[q4].y, rres[q4].z, rres[q4].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += bf2f((bf16_t)(w[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(w[e] >> 16)); }
      }
      if (!live) continue;
      if (OUT == 0) {
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)i * p.ldc + j) =
            make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      } else {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)i * p.ldc + j;
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  // fused bias gradient of the layer below (see gemm_kernel): column sums of this wave's 32*FM rows; FM == 2 only, so that
  // the partial rows have the same 64-row granularity as the 128x128 kernel's
  if (RES && OUT == 0 && FM == 2 && p.act == 2 && p.colsum) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      csum[e] += __shfl_xor(csum[e], 8, 64);
      csum[e] += __shfl_xor(csum[e], 16, 64);
      csum[e] += __shfl_xor(csum[e], 32, 64);
    }
    if (lane < 8 && jok && i0 + wi * 64 < p.I) {
      float* c = p.colsum + (size_t)(ti * WM + wi) * p.J + j;
      *reinterpret_cast<float4*>(c) = make_float4(csum[0], csum[1], csum[2], csum[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
    }
  }
  }
  DIG_GEMM_TS(4)
}

template <bool TA, bool TB, int OUT, int WM, int WN, int FM, int FN, bool RES, int BK, int NSTG, bool DROP = false>
int launch_wide(GemmParams p, int splits, hipStream_t stream, DropArg<DROP> da = DropArg<DROP>{}) {
  using Cfg = WideCfg<WM, WN, FM, FN, BK, NSTG>;
  static bool attr_set[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wide_kernel<TA, TB, OUT, WM, WN, FM, FN, RES, BK, NSTG, DROP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_set[dev] = true;
  }
  p.tiles_i = (p.I + Cfg::TBI - 1) / Cfg::TBI;
  p.tiles_j = (p.J + Cfg::TBJ - 1) / Cfg::TBJ;
  p.splits_x = (splits > 1 && splits % 8 == 0) ? splits : 0;
  dim3 grid(p.tiles_i * p.tiles_j * (p.splits_x ? splits : 1), 1, p.splits_x ? 1 : splits);
  dig_launch(gemm_wide_kernel<TA, TB, OUT, WM, WN, FM, FN, RES, BK, NSTG, DROP>, grid, dim3(Cfg::NT), Cfg::LDS, stream, p, da);
  return dig_check_launch();
}

// 16-byte LDS accesses the compiler does not see (see gemm_pwide_kernel); the caller orders them with explicit lgkmcnt waits
__device__ __forceinline__ void lds_write16(float* p, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)LDS_PTR(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read16x2(const float* p0, const float* p1, f32x4& a, f32x4& b) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(a), "=&v"(b)
               : "v"((unsigned)(uintptr_t)LDS_PTR(p0)), "v"((unsigned)(uintptr_t)LDS_PTR(p1))
               : "memory");
}

// ----------------------------------------------------------------------------------------------------------------
// Persistent form of the 256x256 / 256x192 forward tiles (bf16 output, 64x64 per wave, BK = 64, two LDS slots).  One workgroup
// per CU walks tiles vb = blockIdx.x, blockIdx.x + gridDim.x, ... (same tile <-> XCD assignment as gemm_wide_kernel) and the
// K-stage stream is flattened across tiles: while the last K-stage of a tile is multiplied, the FIRST stage of the next tile is
// already on its way into the other slot, and the epilogue runs out of the slot that was consumed last (16-row C-shuffle pieces,
// 4 KiB per wave).  Removes, per tile, the ~3.4 k cycles from workgroup entry to the first operand stage and the dispatch gap
// between two workgroups of a CU (tools/experiments/gemm_lab.hip: with one 128 KiB workgroup per CU and six K-stages per tile
// the sum of the wave lifetimes was only 80 % of the launch).
// Register discipline (16 waves = 128 VGPRs): everything derived from the lane id is re-derived per tile from a laundered copy,
// otherwise the compiler keeps K-loop and epilogue address sets alive across each other and spills ~30 registers -- and scratch
// accesses are VMEM operations, i.e. they would sit in the same in-order queue as the LDS-DMA and the result stores.
template <int WN, bool RES, bool PRE>
__global__ __launch_bounds__(256 * WN, 1) void gemm_pwide_kernel(GemmParams p, int n_items) {
  constexpr int WM = 4, FM = 2, FN = 2, BK = 64;
  using Cfg = WideCfg<WM, WN, FM, FN, BK, 2>;
  static_assert(Cfg::PA % 64 == 0 && Cfg::PB % 64 == 0, "whole waves take part in a staging pass or skip it");
  static_assert(Cfg::STAGE >= (Cfg::NT / 64) * 4096, "a slot holds the C-shuffle staging of every wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave / WN, wj = wave % WN;
  const int nblk = p.tiles_i * p.tiles_j;
  const int nt = p.R / BK;                                              // host-checked: R % 64 == 0
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  // outputs / residual through buffer descriptors too: 32-bit offsets instead of 64-bit per-lane pointers (registers), and rows
  // beyond I are dropped by the bounds check, so every lane issues the same stores (the counted hand-off below needs that)
  const auto rc_ = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.c_bytes, 0x00020000);
  const auto rp_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre, 0, PRE ? p.c_bytes : 0, 0x00020000);
  const auto rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, RES ? p.c_bytes : 0, 0x00020000);
  const auto rbias_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.J * 4 : 0, 0x00020000);   // no bias: reads as zeros
  const int q8 = nblk >> 3, rm8 = nblk & 7;
  auto tile_of = [&](int vb, int& ti, int& tj) {                        // virtual block -> tile: each XCD walks a contiguous tile range
    const int xcd = vb & 7;
    const int logical = (xcd < rm8 ? xcd * (q8 + 1) : rm8 * (q8 + 1) + (xcd - rm8) * q8) + (vb >> 3);
    ti = logical / p.tiles_j; tj = logical - ti * p.tiles_j;
  };
  // ---- issue cursor (runs one K-stage ahead of the compute cursor, across tile boundaries)
  unsigned offA[8], offB[8];   // (a dependent bound here makes hipcc 7.2 drop the host-side kernel stub)
  int is_vb = blockIdx.x, is_t = 0;
  auto issue_setup = [&]() {
    int ti, tj;
    tile_of(is_vb, ti, tj);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int it = 0; it < Cfg::NITA; ++it) offA[it] = wstage_offset<false, Cfg::TBI, BK>(it * Cfg::NT + tid, ti * Cfg::TBI, 0, p.lda);
#pragma unroll
    for (int it = 0; it < Cfg::NITB; ++it) offB[it] = wstage_offset<false, Cfg::TBJ, BK>(it * Cfg::NT + tid, tj * Cfg::TBJ, 0, p.ldb);
    is_t = 0;
  };
  auto issue_stage = [&](int slot) {
    unsigned char* a = smem + slot * Cfg::STAGE + wave * 1024;
    unsigned char* b = a + Cfg::A_BYTES;
#pragma unroll
    for (int it = 0; it < Cfg::NITA; ++it) {
      if (!Cfg::RAGGED || it * Cfg::NT + wave * 64 < Cfg::PA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(a + it * Cfg::NT * 16), 16, offA[it], 0, 0, 0);
      offA[it] += (unsigned)(BK * 2);
    }
#pragma unroll
    for (int it = 0; it < Cfg::NITB; ++it) {
      if (!Cfg::RAGGED || it * Cfg::NT + wave * 64 < Cfg::PB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, LDS_PTR(b + it * Cfg::NT * 16), 16, offB[it], 0, 0, 0);
      offB[it] += (unsigned)(BK * 2);
    }
    ++is_t;
  };
  if (is_vb >= n_items) return;
  DIG_GEMM_PW_BEGIN()
  issue_setup();
  issue_stage(0);
  int slot = 0;
  bool counted = false;                                                 // the last thing this wave issued: exactly NST stores behind the DMA
  constexpr int NST = 8 * (PRE ? 2 : 1);                                // 16-byte stores per thread per full tile
  for (int vb = blockIdx.x; vb < n_items; vb += gridDim.x) {
    int ti, tj;
    tile_of(vb, ti, tj);
    const int i0 = ti * Cfg::TBI, j0 = tj * Cfg::TBJ;
    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    float4 bias_lo = make_float4(0.f, 0.f, 0.f, 0.f), bias_hi = bias_lo;
    uint4 rres[RES ? 8 : 1];
    {
      int lane = threadIdx.x & 63;
      asm volatile("" : "+v"(lane));                                     // K-loop addresses are rebuilt per tile (see the header)
      auto kstep = [&](auto last_tag, auto first_tag) {
        constexpr bool LAST = decltype(last_tag)::value;                   // the tile's last K-stage also requests the epilogue operands
        constexpr bool FIRST = decltype(first_tag)::value;
        if (counted) wait_vmcnt<NST>(); else wait_vmcnt<0>();
        counted = false;
        __builtin_amdgcn_s_barrier();                                    // this stage has landed for every wave; the other slot is free
        if (FIRST) { DIG_GEMM_PW_ACC(0) }
        if (is_t >= nt) {
          is_vb += gridDim.x;
          if (is_vb < n_items) issue_setup();
        }
        if (LAST) {
          // bias: requested BEFORE the next tile's DMA, so that using it in the epilogue does not wait for that DMA (VMEM results
          // return in issue order)
          const int jb = min(j0 + wj * 64 + (lane & 7) * 8, p.J - 8);
          bias_lo = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rbias_, jb * 4, 0, 0));
          bias_hi = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rbias_, jb * 4 + 16, 0, 0));
        }
        if (is_vb < n_items) issue_stage(slot ^ 1);
        if (RES && LAST) {
          // residual rows: requested right behind the DMA (32 registers for one K-stage); their first use comes a whole K-stage
          // later, when the DMA in front of them has landed anyway
          const int jr = min(j0 + wj * 64 + (lane & 7) * 8, p.J - 8);
#pragma unroll
          for (int ps = 0; ps < 8; ++ps) {
            const int i = i0 + wi * 64 + ps * 8 + (lane >> 3);                                   // rows beyond I read as zero
            rres[ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rr_, (i * p.ldr + jr) * 2, 0, 0));
          }
        }
        const unsigned char* at = smem + slot * Cfg::STAGE;
        const unsigned char* bt = at + Cfg::A_BYTES;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
          bf16x8 af[FM], bfr[FN];
#pragma unroll
          for (int u = 0; u < FM; ++u) af[u] = wload_frag<false, Cfg::TBI, BK>(at, wi * (32 * FM) + u * 32, s, lane);
#pragma unroll
          for (int u = 0; u < FN; ++u) bfr[u] = wload_frag<false, Cfg::TBJ, BK>(bt, wj * (32 * FN) + u * 32, s, lane);
#pragma unroll
          for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
        }
        slot ^= 1;
      };
      if (nt > 1) kstep(std::false_type{}, std::true_type{});
      for (int t = 1; t + 1 < nt; ++t) kstep(std::false_type{}, std::false_type{});
      kstep(std::true_type{}, std::false_type{});            // (nt == 1: the 'first stage' stamp of the lab build is skipped)
    }
    // ---------------- epilogue of this tile: staging = the slot consumed last (slot ^ 1 after the toggle) ----------------
    DIG_GEMM_PW_ACC(1)
    __builtin_amdgcn_s_barrier();                                        // every wave is done reading that slot
    DIG_GEMM_PW_ACC(2)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int cg = lane & 7, hi = lane >> 5;
    float* stg = reinterpret_cast<float*>(smem + (slot ^ 1) * Cfg::STAGE + wave * 4096);     // 16 rows x 64 fp32
    const int j = j0 + wj * 64 + cg * 8;
    const bool jok = j < p.J;
    const float bias8[8] = {bias_lo.x, bias_lo.y, bias_lo.z, bias_lo.w, bias_hi.x, bias_hi.y, bias_hi.z, bias_hi.w};
    const float al = (j < p.alpha_cols) ? p.alpha : 1.0f;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {                                     // 16-row pieces: rows qt*16 .. qt*16+15 of the wave's 64
      const int a = qt >> 1;
      if (((lane >> 4) & 1) == (qt & 1)) {
        const int row = lane & 15;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int chunk = (b * 32 + 8 * g + 4 * hi) >> 2;
            const f32x4 q4 = {acc[a][b][g * 4], acc[a][b][g * 4 + 1], acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]};
            lds_write16(stg + row * 64 + ((chunk ^ row) << 2), q4);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int row = hh * 8 + (lane >> 3);
        const int ps = qt * 2 + hh;
        const int i = i0 + wi * 64 + qt * 16 + row;
        f32x4 x0, x1;
        lds_read16x2(stg + row * 64 + (((2 * cg) ^ row) << 2), stg + row * 64 + (((2 * cg + 1) ^ row) << 2), x0, x1);
        const int ooff = jok ? (i * p.ldc + j) * 2 : -1;                  // byte offset of this lane's 8 outputs (-1: out of range, dropped)
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias8[e]) * al;
        if (p.act == 1) {
          if (PRE)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dig_u32x4, make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]))),
                                                   rp_, ooff, 0, 0);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        }
        if (RES) {
          const unsigned w[4] = {rres[ps].x, rres[ps].y, rres[ps].z, rres[ps].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += bf2f((bf16_t)(w[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(w[e] >> 16)); }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dig_u32x4, make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]))),
                                               rc_, ooff, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    DIG_GEMM_PW_ACC(3)
    counted = !PRE || p.act == 1;                                        // every wave has issued exactly NST stores behind the DMA
  }
  DIG_GEMM_PW_END()
}

template <int WN, bool RES, bool PRE>
int launch_pwide(GemmParams p, hipStream_t stream) {
  using Cfg = WideCfg<4, WN, 2, 2, 64, 2>;
  constexpr int LDS = 2 * Cfg::STAGE;
  static bool attr_set[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pwide_kernel<WN, RES, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set[dev] = true;
  }
  p.tiles_i = (p.I + Cfg::TBI - 1) / Cfg::TBI;
  p.tiles_j = (p.J + Cfg::TBJ - 1) / Cfg::TBJ;
  const int n_items = p.tiles_i * p.tiles_j;
  static int n_cu[DIG_MAX_DEVICES] = {};
  if (!n_cu[dev]) {
    hipDeviceProp_t prop;
    n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int grid = std::min(n_items, n_cu[dev]);                         // one 16- / 12-wave workgroup per CU
  dig_launch(gemm_pwide_kernel<WN, RES, PRE>, dim3(grid), dim3(Cfg::NT), LDS, stream, p, n_items);
  return dig_check_launch();
}

// out[e] (+)= sum_s part[s][e]   (deterministic split-R combine; also the "+=" into the gradient arena)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int splits, long long n4,
                                                              float* __restrict__ out, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(part)[(long long)s * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

// out_bf16[e] = bf16(sum_s part[s][e]): the combine of a split-R forward layer / data gradient (fixed order)
__global__ __launch_bounds__(256) void reduce_partials_bf16_kernel(const float* __restrict__ part, int splits, long long n4, bf16_t* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(part)[(long long)s * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
  }
}

// the same for up to DIG_REDUCE_MAX_SEGS slab sets in ONE launch (the four weight gradients of an encoder block): blocks
// [first[k], first[k+1]) walk segment k
struct ReduceSegs {
  const float* part[DIG_REDUCE_MAX_SEGS];
  float* out[DIG_REDUCE_MAX_SEGS];
  long long n4[DIG_REDUCE_MAX_SEGS];
  int splits[DIG_REDUCE_MAX_SEGS];
  int first[DIG_REDUCE_MAX_SEGS + 1];
  int n_segs;
};
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceSegs a) {
  int k = 0;
  while (k + 1 < a.n_segs && (int)blockIdx.x >= a.first[k + 1]) ++k;
  const float* __restrict__ part = a.part[k];
  float* __restrict__ out = a.out[k];
  const long long n4 = a.n4[k];
  const int splits = a.splits[k], nblk = a.first[k + 1] - a.first[k];
  for (long long i = (long long)((int)blockIdx.x - a.first[k]) * blockDim.x + threadIdx.x; i < n4; i += (long long)nblk * blockDim.x) {
    float4 acc = reinterpret_cast<const float4*>(out)[i];
    for (int s = 0; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(part)[(long long)s * n4 + i];
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}

template <bool TA, bool TB, int OUT, int BK, bool RES, int NSTG, bool DROP = false>
int launch(const GemmParams& p, int splits, hipStream_t stream, DropArg<DROP> da = DropArg<DROP>{}) {
  constexpr int LDS = (NSTG * 2 * BI * BK * 2) > 32768 ? (NSTG * 2 * BI * BK * 2) : 32768;   // ring; >= epilogue staging
  static bool attr_set[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<TA, TB, OUT, BK, RES, NSTG, DROP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set[dev] = true;
  }
  GemmParams q = p;
  q.splits_x = (splits > 1 && splits % 8 == 0) ? splits : 0;
  dim3 grid(p.tiles_i * p.tiles_j * (q.splits_x ? splits : 1), 1, q.splits_x ? 1 : splits);
  dig_launch(gemm_kernel<TA, TB, OUT, BK, RES, NSTG, DROP>, grid, dim3(256), LDS, stream, q, da);
  return dig_check_launch();
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_gemm_bf16_dropout(const void* A, const void* B, void* C, int I, int J, int R, int lda, int ldb, int ldc,
                                     int trans_a, int trans_b, int out_kind, const float* bias, const void* resid, int ldr,
                                     void* pre_act, int ldp, float alpha, int alpha_cols, int act, int splits, int a_rows,
                                     int b_rows, int bk, float* colsum_partials, const dig_dropout_t* drop,
                                     hipStream_t stream) {
  if (!A || !B || !C || I <= 0 || J <= 0 || R <= 0 || splits < 1) return DIG_ERR_ARG;
  const bool dropping = drop && (drop->thr || drop->pthr);
  if (dropping) {
    // instantiated for the tiles the fine-tune step uses: 128x128 (bk 0 / 64 / 32) forward and dgrad, 256x256 / 256x192 (bk 244 / 264) forward
    if (out_kind != 0 || trans_a || !(bk == 0 || bk == 32 || bk == 64 || ((bk == 244 || bk == 264) && !trans_b))) return DIG_ERR_UNSUPPORTED;
    if ((size_t)I * (size_t)J >= (1ull << 32) || (drop->pthr && drop->rows_per_sample <= 0)) return DIG_ERR_ARG;
  }
  if (bias && !aligned16(bias)) return DIG_ERR_ALIGN;
  if (out_kind < 0 || out_kind > 2 || act < 0 || act > 2 || (bk != 0 && bk != 32 && bk != 64 && bk != 244 && bk != 264 && bk != 212 && bk != 221 && bk != 544 && bk != 564)) return DIG_ERR_ARG;
  if (act == 2 && !resid) return DIG_ERR_ARG;                   // act 2: resid carries the saved pre-activation
  if (bk == 0) bk = 64;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (lda & 7) || (ldb & 7)) return DIG_ERR_ALIGN;
  if ((J & 7) || (ldc & 7) || (resid && ((ldr & 7) || !aligned16(resid))) || (pre_act && ((ldp & 7) || !aligned16(pre_act)))) return DIG_ERR_ALIGN;
  if (!trans_a && (R % BR)) return DIG_ERR_ARG;   // direct operands need R % 64 == 0 (row-wrap would pollute)
  if (!trans_b && (R % BR)) return DIG_ERR_ARG;
  if (out_kind == 2 && (bias || resid || act)) return DIG_ERR_ARG;
  if (out_kind != 2 && splits != 1) return DIG_ERR_ARG;
  if (out_kind == 2 && ldc != J) return DIG_ERR_ARG;            // partial slabs are dense [splits][I][J]
  if (colsum_partials && !(act == 2 && out_kind == 0 && trans_b && !trans_a && (bk < 100 || bk == 244)))
    return DIG_ERR_UNSUPPORTED;                                  // 128x128 kernel and the 64x64-per-wave wide tiles
  if (colsum_partials && !aligned16(colsum_partials)) return DIG_ERR_ALIGN;
  GemmParams p;
  p.splits_x = 0;
  p.colsum = colsum_partials;
  p.c_bytes = 0;
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
  if ((R + p.r_per_split - 1) / p.r_per_split != splits) return DIG_ERR_ARG;   // use dig_gemm_effective_splits()
  p.tiles_i = (I + BI - 1) / BI; p.tiles_j = (J + BJ - 1) / BJ;
  if (dropping) {
    const DropArg<true> da{*drop};
    if (bk == 244) return resid ? launch_wide<false, false, 0, 4, 4, 2, 2, true, 64, 2, true>(p, splits, stream, da)
                                : launch_wide<false, false, 0, 4, 4, 2, 2, false, 64, 2, true>(p, splits, stream, da);
    if (bk == 264) return resid ? launch_wide<false, false, 0, 4, 3, 2, 2, true, 64, 2, true>(p, splits, stream, da)
                                : launch_wide<false, false, 0, 4, 3, 2, 2, false, 64, 2, true>(p, splits, stream, da);
    if (!trans_b)
      return bk == 32 ? (resid ? launch<false, false, 0, 32, true, 2, true>(p, splits, stream, da) : launch<false, false, 0, 32, false, 2, true>(p, splits, stream, da))
                      : (resid ? launch<false, false, 0, 64, true, 2, true>(p, splits, stream, da) : launch<false, false, 0, 64, false, 2, true>(p, splits, stream, da));
    return bk == 32 ? (resid ? launch<false, true, 0, 32, true, 2, true>(p, splits, stream, da) : launch<false, true, 0, 32, false, 2, true>(p, splits, stream, da))
                    : (resid ? launch<false, true, 0, 64, true, 2, true>(p, splits, stream, da) : launch<false, true, 0, 64, false, 2, true>(p, splits, stream, da));
  }
  if ((bk == 544 || bk == 564) && !trans_a && !trans_b && out_kind == 0 && act != 2 && !colsum_partials) {   // persistent forward tiles
    const bool pre_ = pre_act != nullptr && act == 1;
    // Outside the persistent kernel's limits (32-bit byte offsets of C, one leading dimension for C / residual / pre-activation, no
    // residual on the 16-wave tile or together with a saved pre-activation) the call runs on the one-tile-per-workgroup kernel of the
    // same shape below -- same results, no row-count cliff for a very large per-GPU batch.
    const bool fits = (size_t)I * ldc * 2 < (1ull << 31) && !(resid && ldr != ldc) && !(pre_ && ldp != ldc) && !(pre_ && resid) &&
                      !(bk == 544 && resid);
    if (fits) {
      p.c_bytes = (unsigned)((size_t)I * ldc * 2);
      if (bk == 544) return (pre_ ? launch_pwide<4, false, true>(p, stream) : launch_pwide<4, false, false>(p, stream));
      return resid ? launch_pwide<3, true, false>(p, stream) : (pre_ ? launch_pwide<3, false, true>(p, stream) : launch_pwide<3, false, false>(p, stream));
    }
  }
  if (bk == 544 || bk == 564) bk -= 300;                               // other operand forms: the one-tile-per-workgroup kernel of that shape
#define DIG_GEMM_WCASE(ta, tb, o)                                                                                   \
  if ((trans_a != 0) == ta && (trans_b != 0) == tb && out_kind == o && bk >= 200) {                                  \
    if (bk == 212) return resid ? launch_wide<ta, tb, o, 1, 2, 2, 2, true, 64, 2>(p, splits, stream) : launch_wide<ta, tb, o, 1, 2, 2, 2, false, 64, 2>(p, splits, stream); \
    if (bk == 221) return resid ? launch_wide<ta, tb, o, 2, 1, 2, 2, true, 64, 2>(p, splits, stream) : launch_wide<ta, tb, o, 2, 1, 2, 2, false, 64, 2>(p, splits, stream); \
    if (bk == 264) return resid ? launch_wide<ta, tb, o, 4, 3, 2, 2, true, 64, 2>(p, splits, stream) : launch_wide<ta, tb, o, 4, 3, 2, 2, false, 64, 2>(p, splits, stream); \
    return resid ? launch_wide<ta, tb, o, 4, 4, 2, 2, true, 64, 2>(p, splits, stream) : launch_wide<ta, tb, o, 4, 4, 2, 2, false, 64, 2>(p, splits, stream);               \
  }
  DIG_GEMM_WCASE(false, false, 0)
  DIG_GEMM_WCASE(false, false, 1)
  DIG_GEMM_WCASE(false, false, 2)                                      // split-R slabs of a forward layer / a data gradient: the few-row,
  DIG_GEMM_WCASE(false, true, 0)                                       // narrow-output, long-K layers of the BatchNorm-MLP heads
  DIG_GEMM_WCASE(false, true, 2)                                       // (dig_reduce_partials_bf16 combines them)
  DIG_GEMM_WCASE(true, true, 2)
#undef DIG_GEMM_WCASE
#define DIG_GEMM_CASE(ta, tb, o)                                                \
  if ((trans_a != 0) == ta && (trans_b != 0) == tb && out_kind == o)           \
    return bk == 64 ? (resid ? launch<ta, tb, o, 64, true, 2>(p, splits, stream) : launch<ta, tb, o, 64, false, 2>(p, splits, stream)) \
                    : (resid ? launch<ta, tb, o, 32, true, 2>(p, splits, stream) : launch<ta, tb, o, 32, false, 2>(p, splits, stream));
  DIG_GEMM_CASE(false, false, 0)
  DIG_GEMM_CASE(false, false, 1)
  DIG_GEMM_CASE(false, true, 0)
  DIG_GEMM_CASE(false, true, 1)
  DIG_GEMM_CASE(true, true, 2)
#undef DIG_GEMM_CASE
  return DIG_ERR_UNSUPPORTED;
}

extern "C" int dig_gemm_bf16(const void* A, const void* B, void* C, int I, int J, int R, int lda, int ldb, int ldc,
                             int trans_a, int trans_b, int out_kind, const float* bias, const void* resid, int ldr,
                             void* pre_act, int ldp, float alpha, int alpha_cols, int act, int splits, int a_rows,
                             int b_rows, int bk, float* colsum_partials, hipStream_t stream) {
  return dig_gemm_bf16_dropout(A, B, C, I, J, R, lda, ldb, ldc, trans_a, trans_b, out_kind, bias, resid, ldr, pre_act, ldp, alpha,
                               alpha_cols, act, splits, a_rows, b_rows, bk, colsum_partials, nullptr, stream);
}

// Number of R-splits dig_gemm_bf16 will really use for a requested split count (slabs are whole 64-row K-tiles).
extern "C" int dig_gemm_effective_splits(int R, int splits) {
  if (R <= 0 || splits < 1) return 0;
  const int rtiles = (R + BR - 1) / BR;
  const int per = ((rtiles + splits - 1) / splits) * BR;
  return (R + per - 1) / per;
}

extern "C" int dig_reduce_partials_multi(const dig_reduce_seg_t* segs, int n_segs, hipStream_t stream) {
  if (!segs || n_segs < 1 || n_segs > DIG_REDUCE_MAX_SEGS) return DIG_ERR_ARG;
  ReduceSegs a;
  a.n_segs = n_segs;
  int blocks = 0;
  for (int k = 0; k < n_segs; ++k) {
    const dig_reduce_seg_t& g = segs[k];
    if (!g.partials || !g.out || g.splits < 1 || g.n <= 0 || (g.n & 3)) return DIG_ERR_ARG;
    if (!aligned16(g.partials) || !aligned16(g.out)) return DIG_ERR_ALIGN;
    a.part[k] = g.partials; a.out[k] = g.out; a.n4[k] = g.n / 4; a.splits[k] = g.splits;
    a.first[k] = blocks;
    blocks += (int)std::min<long long>(1024, (g.n / 4 + 255) / 256);
  }
  a.first[n_segs] = blocks;
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return dig_check_launch();
}

extern "C" int dig_reduce_partials_bf16(const float* partials, int splits, long long n, void* out, hipStream_t stream) {
  if (!partials || !out || splits < 1 || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(partials) || (((uintptr_t)out) & 7u)) return DIG_ERR_ALIGN;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(reduce_partials_bf16_kernel, dim3((unsigned)std::min<long long>(2048, (n4 + 255) / 256)), dim3(256), 0, stream, partials,
                     splits, n4, (bf16_t*)out);
  return dig_check_launch();
}

extern "C" int dig_reduce_partials(const float* partials, int splits, long long n, float* out, int accumulate,
                                   hipStream_t stream) {
  if (!partials || !out || splits < 1 || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(partials) || !aligned16(out)) return DIG_ERR_ALIGN;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<long long>(2048, (n4 + 255) / 256)), dim3(256), 0, stream,
                     partials, splits, n4, out, accumulate);
  return dig_check_launch();
}
