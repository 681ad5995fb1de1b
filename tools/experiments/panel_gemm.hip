// Row-panel GEMM for 384-wide outputs on gfx950: a workgroup owns WHOLE output rows, so row-wise epilogues run out of the accumulators.
//
//     out[R, 384] = A[R, K] B[384, K]^T (+ bias) (+ resid)          A, B with K contiguous, bf16; fp32 accumulation
//   EPI_STORE:   out is rounded to bf16 and stored;
//   EPI_LN_FWD:  the same, plus LayerNorm of the stored rows: ln_out = LN(out; g, b), mean / rstd per row.  Reference: the attention output
//                projection + residual + norm2 of Block.forward (modeling_finetune.py:152-156): x = x + proj(attn); norm2(x);
//   EPI_LN_BWD:  the GEMM result is the gradient d of a LayerNorm OUTPUT (data gradient of the Linear that consumed it: qkv or fc1); the
//                epilogue is that LayerNorm's backward, dx = dres + rstd (g d - mean_k(g d) - xhat mean_k(g d xhat)), xhat = (x - mean) rstd,
//                with the per-workgroup column sums of (d xhat, d, dres) = partials of (dgamma, dbeta, bias gradient of the layer that fed the
//                residual) in the [parts][3][384] layout of dig_layernorm_bwd_finalize.
//
// Design (CDNA4).  256 rows per workgroup = 8 waves x 32 rows; a lane owns one row (swapped-operand MFMA: the A tile is the B operand) and
// its 192 accumulator registers run over 12 column blocks of 32 -- the O-wave of csrc/mlp_chain.hip with the left operand coming from HBM
// instead of from a partner wave.  K is walked in chunks of 64; a tick = (chunk, third of the 384 outputs) = one 16 KiB slot [128 j][64 k]
// of B (3-slot LDS-DMA ring, XOR swizzle on the source side) x the chunk's A tile [256 rows][64 k] (32 KiB, two buffers, fetched one chunk
// ahead in two halves) = 16 MFMAs per wave; one raw s_barrier and one counted s_waitcnt vmcnt per tick.  One workgroup per CU (112 KiB of
// LDS, 2 waves per SIMD): 65 536 rows are exactly one round on 256 CUs, the weights are re-read from L2 once per 256 rows.
#include "common.h"
#include <type_traits>

namespace {

constexpr int KD = 384;                  // output width
constexpr int PBM = 256;                 // rows per workgroup
constexpr int NJB = KD / 32;
constexpr int SLOT = 16384;
constexpr int B_RING = 0;
constexpr int A_OFF = 3 * SLOT;          // two A tiles of 2 slots each
constexpr int V_OFF = 7 * SLOT;          // bias, gamma, beta [KD] fp32 each
enum { EPI_STORE = 0, EPI_LN_FWD = 1, EPI_LN_BWD = 2 };

struct PanelParams {
  const bf16_t* A;       // [R, K]
  const bf16_t* B;       // [KD, K]
  const float* bias;     // [KD] or null
  const bf16_t* resid;   // [R, KD] or null        (EPI_LN_BWD: dres, the gradient that by-passes the LayerNorm)
  bf16_t* out;           // [R, KD]                (EPI_LN_BWD: dx)
  const float* g; const float* b;                 // LayerNorm weight / bias [KD]
  bf16_t* ln_out; float* mean; float* rstd;       // EPI_LN_FWD outputs (mean / rstd may be null);  EPI_LN_BWD: mean / rstd are INPUTS
  const bf16_t* x;       // EPI_LN_BWD: the LayerNorm input rows [R, KD]
  float* parts;          // EPI_LN_BWD: [gridDim.x][3][KD] column sums of (d xhat, d, dres)
  int R, K;
  unsigned a_bytes, b_bytes, o_bytes;
  float eps;
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }

template <int EPI>
__global__ __launch_bounds__(512) void panel_gemm_kernel(PanelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, rr = lane & 31;
  const int m0 = blockIdx.x * PBM;
  const int K = p.K, NC = K / 64;

  const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  // per-thread source offsets (the swizzle lives on the source side: LDS-DMA destinations are lane-linear)
  unsigned vb[2], va[4];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pc = it * 512 + tid, r2 = pc >> 3, c2 = (pc & 7) ^ ((r2 >> 1) & 7);                  // B slot: [128 j][64 k]
    vb[it] = (unsigned)((r2 * K + 8 * c2) * 2);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int pc = it * 512 + tid, r = pc >> 3, c2 = (pc & 7) ^ ((r >> 1) & 7);                    // A tile: [256 rows][64 k]
    va[it] = (unsigned)(((size_t)(m0 + r) * K + 8 * c2) * 2);                                       // rows beyond R read as zero
  }
  auto dma_b = [&](int slot, int c, int third, int it) {
    const int cc = c < NC ? c : 0;                                                                  // past the end: a valid piece into a free slot
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_PTR(smem + B_RING + slot * SLOT + wave * 1024 + it * 8192), 16, vb[it],
                                             (unsigned)((third * 128 * K + cc * 64) * 2), 0, 0);
  };
  auto dma_a = [&](int c, int it) {
    const int cc = c < NC ? c : 0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(smem + A_OFF + (c & 1) * 2 * SLOT + wave * 1024 + it * 8192), 16, va[it],
                                             (unsigned)(cc * 64 * 2), 0, 0);
  };
  // prologue: A tile of chunk 0, B slots of ticks (0, 0) and (0, 1); then the epilogue's vectors -> LDS (inline-asm stores: a visible LDS
  // store would be ordered behind the DMA in flight with s_waitcnt vmcnt(0); the first barrier publishes them)
#pragma unroll
  for (int it = 0; it < 4; ++it) dma_a(0, it);
#pragma unroll
  for (int it = 0; it < 2; ++it) dma_b(0, 0, 0, it);
#pragma unroll
  for (int it = 0; it < 2; ++it) dma_b(1, 0, 1, it);
  {
    const float* const src[3] = {p.bias, p.g, p.b};
    float v[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const auto rv = __builtin_amdgcn_make_buffer_rsrc((void*)src[u], 0, src[u] ? KD * 4 : 0, 0x00020000);   // null: zeros
      v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, (unsigned)(tid * 4), 0, 0));
    }
    if (tid < KD) {
#pragma unroll
      for (int u = 0; u < 3; ++u) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(smem + V_OFF) + (unsigned)((u * KD + tid) * 4)), "v"(v[u]) : "memory");
    }
  }

  f32x16 D2[NJB];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int e = 0; e < 16; ++e) D2[jb][e] = 0.f;
  const int psw = (rr >> 1) & 7;
  int aoff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) aoff[s] = rr * 128 + (((2 * s + hi) ^ psw) << 4);
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;

  bf16x8 pf[4];
  // Tick (c, TAU).  NW = VMEM operations this wave has issued behind the DMA the tick needs (see the header of the loop below).
  auto tick = [&](auto tau_tag, auto nw_tag, int c) {
    constexpr int TAU = decltype(tau_tag)::value;
    wait_vm<decltype(nw_tag)::value>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // DMA for the tick after next (B) and, during ticks 0 and 1, the two halves of the NEXT chunk's A tile
    constexpr int tn = (TAU + 2) % 3;
    const int cn = TAU == 0 ? c : c + 1;
    const unsigned char* w = smem + B_RING + TAU * SLOT;
    if (TAU == 0) {
      const unsigned char* at = smem + A_OFF + (c & 1) * 2 * SLOT + wave * 4096;
#pragma unroll
      for (int s = 0; s < 4; ++s) pf[s] = *reinterpret_cast<const bf16x8*>(at + aoff[s]);
    }
    bf16x8 wfa[4], wfb[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) wfa[jb] = *reinterpret_cast<const bf16x8*>(w + jb * 4096 + aoff[0]);
    __builtin_amdgcn_sched_barrier(0);
    auto kstep = [&](auto s_tag, bf16x8 (&cur)[4], bf16x8 (&nxt)[4]) {
      constexpr int S = decltype(s_tag)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (S < 3) {
          nxt[2 * h] = *reinterpret_cast<const bf16x8*>(w + (2 * h) * 4096 + aoff[S + 1]);
          nxt[2 * h + 1] = *reinterpret_cast<const bf16x8*>(w + (2 * h + 1) * 4096 + aoff[S + 1]);
        }
#pragma unroll
        for (int jb = 2 * h; jb < 2 * h + 2; ++jb)
          D2[TAU * 4 + jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[jb], pf[S], D2[TAU * 4 + jb], 0, 0, 0);
        if (h == 0) __builtin_amdgcn_sched_barrier(0);
      }
      // one DMA piece per k-step: B pieces in steps 0 and 1, A pieces (ticks 0 and 1 only) in steps 2 and 3
      if (S < 2) dma_b(tn, cn, tn, S);
      else if (TAU < 2) dma_a(c + 1, 2 * TAU + S - 2);
      __builtin_amdgcn_sched_barrier(0);
    };
    kstep(std::integral_constant<int, 0>{}, wfa, wfb); kstep(std::integral_constant<int, 1>{}, wfb, wfa);
    kstep(std::integral_constant<int, 2>{}, wfa, wfb); kstep(std::integral_constant<int, 3>{}, wfb, wfa);
  };
  // vmcnt bookkeeping (in-order counter).  Issue order: tick (c, 0): B x2, A x2;  (c, 1): B x2, A x2;  (c, 2): B x2.
  //   tick (c, 0) needs the B slot issued at (c-1, 1) and the A tile issued at (c-1, 0..1): behind them only (c-1, 2)'s 2 operations;
  //   tick (c, 1) needs the B slot of (c-1, 2): behind it the 4 of (c, 0);   tick (c, 2) needs the B slot of (c, 0): behind it 2 + 4.
  for (int c = 0; c < NC; ++c) {
    tick(T0{}, std::integral_constant<int, 2>{}, c);
    tick(T1{}, std::integral_constant<int, 4>{}, c);
    tick(T2{}, std::integral_constant<int, 6>{}, c);
  }

  // ------------------------------------------------------------------ epilogue (the lane's row: columns 32 jb + 8 g + 4 hi + e)
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));                                      // re-derive the row here: kept live across the loop it would be spilled
  const int row = m0 + (tid2 >> 6) * 32 + (tid2 & 31), hi2 = (tid2 >> 5) & 1;
  const float* vs = reinterpret_cast<const float*>(smem + V_OFF) + 4 * hi2;
  const auto rRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, p.resid ? p.o_bytes : 0, 0x00020000);   // null: zeros; rows beyond R: zeros
  const unsigned ro = (unsigned)(((size_t)row * KD + 4 * hi2) * 2);
  auto store_block = [&](bf16_t* dst, unsigned (&Pk)[4][2]) {         // one 32-column block: v_permlane32_swap -> 16 contiguous columns per lane
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(Pk[0][k], Pk[2][k], false, false);
      Pk[0][k] = r0[0]; Pk[2][k] = r0[1];
      const auto r1 = __builtin_amdgcn_permlane32_swap(Pk[1][k], Pk[3][k], false, false);
      Pk[1][k] = r1[0]; Pk[3][k] = r1[1];
    }
    *reinterpret_cast<uint4*>(dst) = make_uint4(Pk[0][0], Pk[0][1], Pk[2][0], Pk[2][1]);
    *reinterpret_cast<uint4*>(dst + 8) = make_uint4(Pk[1][0], Pk[1][1], Pk[3][0], Pk[3][1]);
  };
  const dig_bf16x2 ones = __builtin_bit_cast(dig_bf16x2, 0x3F803F80u);

  if (EPI != EPI_LN_BWD) {
    // + bias + residual, rounded to bf16 in place (what is stored is what the LayerNorm sees), row sums by v_dot2 on the packed pairs
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
      uint2 rq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) rq[g] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rRes, ro + (jb * 32 + g * 8) * 2, 0, 0));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(vs + jb * 32 + g * 8);
        const unsigned w[2] = {rq[g].x, rq[g].y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float a0 = D2[jb][4 * g + 2 * q] + bv[2 * q] + __uint_as_float(w[q] << 16);
          const float a1 = D2[jb][4 * g + 2 * q + 1] + bv[2 * q + 1] + __uint_as_float(w[q] & 0xffff0000u);
          const unsigned u = pack_bf2(a0, a1);
          if (EPI == EPI_LN_FWD) {
            const dig_bf16x2 pr = __builtin_bit_cast(dig_bf16x2, u);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, s1, false);
            s2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2, false);
          }
          D2[jb][4 * g + 2 * q] = __uint_as_float(u << 16);
          D2[jb][4 * g + 2 * q + 1] = __uint_as_float(u & 0xffff0000u);
        }
      }
    }
    if (row < p.R) {
      bf16_t* orow = p.out + (size_t)row * KD + hi2 * 16;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        unsigned Pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Pk[g][0] = pack_bf2(D2[jb][g * 4], D2[jb][g * 4 + 1]);
          Pk[g][1] = pack_bf2(D2[jb][g * 4 + 2], D2[jb][g * 4 + 3]);
        }
        store_block(orow + jb * 32, Pk);
      }
      if (EPI == EPI_LN_FWD) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float mean = s1 * (1.0f / KD);
        const float rs = rsqrtf(fmaxf(s2 * (1.0f / KD) - mean * mean, 0.f) + p.eps);
        bf16_t* nrow = p.ln_out + (size_t)row * KD + hi2 * 16;
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          unsigned Pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(vs + KD + jb * 32 + g * 8);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(vs + 2 * KD + jb * 32 + g * 8);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf(D2[jb][g * 4 + e] - mean, rs * g4[e], b4[e]);
            Pk[g][0] = pack_bf2(y[0], y[1]);
            Pk[g][1] = pack_bf2(y[2], y[3]);
          }
          store_block(nrow + jb * 32, Pk);
        }
        if (p.mean && hi2 == 0) { p.mean[row] = mean; p.rstd[row] = rs; }
      }
    }
  }
}

template <int EPI>
int launch_panel(const PanelParams& p, hipStream_t stream) {
  constexpr int LDS = V_OFF + 3 * KD * 4;
  static bool attr[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_gemm_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DIG_ERR_LAUNCH;
    attr[dev] = true;
  }
  hipLaunchKernelGGL((panel_gemm_kernel<EPI>), dim3((p.R + PBM - 1) / PBM), dim3(512), LDS, stream, p);
  return dig_check_launch();
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_panel_gemm_supported(int J, int K) { return J == KD && K >= 64 && K % 64 == 0 && K <= 8192; }

extern "C" int dig_panel_gemm_ln_fwd(const void* a, const void* w, const float* bias, const void* resid, void* out, const float* ln_g,
                                     const float* ln_b, float eps, void* ln_out, float* ln_mean, float* ln_rstd, int R, int J, int K,
                                     hipStream_t stream) {
  if (!a || !w || !out || R <= 0) return DIG_ERR_ARG;
  if (!dig_panel_gemm_supported(J, K)) return DIG_ERR_UNSUPPORTED;
  if ((ln_g == nullptr) != (ln_b == nullptr) || (ln_g == nullptr) != (ln_out == nullptr) || (ln_mean == nullptr) != (ln_rstd == nullptr)) return DIG_ERR_ARG;
  if (!ln_g && ln_mean) return DIG_ERR_ARG;
  if (!aligned16(a) || !aligned16(w) || !aligned16(out) || (resid && !aligned16(resid)) || (ln_out && !aligned16(ln_out)) || (bias && !aligned16(bias)))
    return DIG_ERR_ALIGN;
  if ((size_t)R * K * 2 >= (1ull << 32) || (size_t)R * J * 2 >= (1ull << 32)) return DIG_ERR_UNSUPPORTED;
  PanelParams p;
  p.A = (const bf16_t*)a; p.B = (const bf16_t*)w; p.bias = bias; p.resid = (const bf16_t*)resid; p.out = (bf16_t*)out;
  p.g = ln_g; p.b = ln_b; p.ln_out = (bf16_t*)ln_out; p.mean = ln_mean; p.rstd = ln_rstd; p.x = nullptr; p.parts = nullptr;
  p.R = R; p.K = K; p.eps = eps;
  p.a_bytes = (unsigned)((size_t)R * K * 2); p.b_bytes = (unsigned)((size_t)J * K * 2); p.o_bytes = (unsigned)((size_t)R * J * 2);
  return ln_g ? launch_panel<EPI_LN_FWD>(p, stream) : launch_panel<EPI_STORE>(p, stream);
}
