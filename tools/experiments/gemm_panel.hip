// A-resident ("panel") bf16 MFMA GEMM for the short-K layers of the ViT encoder:  C[I,J] = A[I,K] * B[J,K]^T, K <= 384.
//
// Why a second GEMM kernel: at D = 384 the QKV / proj / fc1 GEMMs are 6 K-tiles deep.  A tile-per-workgroup kernel
// (gemm.hip) spends most of its life in the first-load latency and the epilogue, and re-reads each 96 KiB A row-panel
// from L2 once per N-tile (9-12x).  Here one workgroup owns 128 rows of A for their whole life:
//   * the [128 x K] A panel is brought into LDS ONCE (96 KiB at K = 384) and stays resident;
//   * the weight tiles [128 x 64] (L2-resident, <= 1.2 MiB per layer) stream through a 3-stage LDS ring filled by
//     buffer_load ... lds, flattened across N-tiles, so the DMA never drains at a tile boundary and the epilogue of
//     tile j overlaps the loads of tile j+1;
//   * ring hand-off uses counted s_waitcnt vmcnt(N) + raw s_barrier (a __syncthreads() would drain every DMA in flight);
//     gfx9 retires a wave's VMEM operations (LDS-DMA loads and stores alike) in order, so N counts the younger ones;
//   * MFMA / fragment / C-shuffle code is the same as gemm.hip (v_mfma_f32_32x32x16_bf16, swapped operands, XOR-swizzled
//     ds_read_b128); the epilogue stages 16 rows per wave (4 KiB) at a time because the panel leaves 16 KiB of LDS.
// LDS: A panel 128*K*2 | B ring 3 x 16 KiB | staging 16 KiB  (= 160 KiB at K = 384: one workgroup per CU).
//
// STATUS (round 1): parity-green but NOT on the default path (ops.USE_PANEL = False).  Measured on MI355X it is ~1.8x
// slower than gemm.hip on the fc1 shape: with one 4-wave workgroup per CU there is a single wave per SIMD, and
// compiler-scheduled code then exposes every ds_read / VALU / MFMA dependency (probe: 0.44 us per EMPTY ring step,
// 10 us per 128x128x384 tile against 1.3 us of MFMA time).  Kept as the starting point for an 8-wave or hand-scheduled
// version; the DIG_PANEL_DBG switches are the probes used for that measurement.
#include "common.h"
#include <cstdlib>

namespace {

struct PanelParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  int I, J, K;
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
  int dbg;   // probe switches (DIG_PANEL_DBG): 1 skip epilogue, 2 skip MFMA, 4 skip B re-loads
};

constexpr int PM = 128, PN = 128, PK = 64;
constexpr int NSTAGE = 3;
constexpr int BSTAGE_BYTES = PN * PK * 2;   // 16 KiB

__device__ __forceinline__ int bswz(int row) { return (row >> 1) & 7; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// OUT: 0 bf16, 1 fp32.  RES: residual / gelu' operand present.  NST = global stores per thread per tile (for the counted wait).
template <int OUT, bool RES, bool PRE>
__global__ __launch_bounds__(256, 1) void gemm_panel_kernel(PanelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int i0 = blockIdx.x * PM;
  const int K = p.K;
  const int nk = K / PK;                       // K-steps per N-tile
  const int row_bytes = K * 2;
  const int cpr = K / 8;                       // 16-B chunks per panel row
  unsigned char* Ap = smem;
  unsigned char* Bring = smem + PM * row_bytes;
  float* stg = reinterpret_cast<float*>(Bring + NSTAGE * BSTAGE_BYTES) + wave * 1024;   // 16 rows x 64 fp32 per wave

  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);

  // ---- A panel: PM rows x cpr chunks, lane-linear destination, swizzle (chunk ^ row) & 15 inside each 16-chunk group
  const int a_pieces = PM * cpr;
  for (int pc0 = 0; pc0 < a_pieces; pc0 += 256) {
    const int piece = pc0 + tid;
    const int row = piece / cpr, pcx = piece - row * cpr;
    const int c = (pcx & ~15) | ((pcx ^ row) & 15);
    const unsigned off = (unsigned)(((i0 + row) * p.lda + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(Ap + (pc0 + wave * 64) * 16), 16, off, 0, 0, 0);
  }
  // ---- B ring: step s = jt * nk + ks covers B rows [jt*128, +128), k-range [ks*64, +64)
  const int tiles_j = (p.J + PN - 1) / PN;
  const int S = tiles_j * nk;
  unsigned boff[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = it * 256 + tid;
    const int row = piece >> 3, pc = piece & 7;
    boff[it] = (unsigned)((row * p.ldb + (pc ^ bswz(row)) * 8) * 2);
  }
  auto issue_b = [&](int s) {
    const int jt = s / nk, ks = s - jt * nk;
    const unsigned base = (unsigned)((jt * PN * p.ldb + ks * PK) * 2);
    unsigned char* dst = Bring + (s % NSTAGE) * BSTAGE_BYTES + wave * 1024;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(dst + it * 4096), 16, base + boff[it], 0, 0, 0);
  };
  issue_b(0);
  if (S > 1) issue_b(1);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int hi = lane >> 5;
  const int cg = lane & 7;
  constexpr int NST = (PRE ? 16 : 8) * (OUT == 1 ? 2 : 1);       // stores per thread per tile
  int after_epilogue = 0;      // 1: previous step ended a fully-live tile (every thread issued exactly NST stores), 2: ragged

  for (int s = 0; s < S; ++s) {
    // wait for step s: younger operations that may stay in flight = B(s+1) [4] (+ the previous tile's stores)
    if (s + 1 < S) {
      if (after_epilogue == 1) wait_vmcnt<4 + NST>(); else wait_vmcnt<4>();
    } else {
      wait_vmcnt<0>();
    }
    after_epilogue = 0;
    __builtin_amdgcn_s_barrier();
    if (s + 2 < S && !(p.dbg & 4)) issue_b(s + 2);              // slot (s+2)%3 was last read in step s-1: every wave is past it
    const int jt = s / nk, ks = s - jt * nk;
    const unsigned char* bt = Bring + (s % NSTAGE) * BSTAGE_BYTES;
    if (!(p.dbg & 2))
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = wi * 64 + u * 32 + (lane & 31);
        const int c = ks * 8 + 2 * s4 + hi;
        const int pc = (c & ~15) | ((c ^ row) & 15);
        af[u] = *reinterpret_cast<const bf16x8*>(Ap + row * row_bytes + pc * 16);
        const int brow = wj * 64 + u * 32 + (lane & 31);
        bfr[u] = *reinterpret_cast<const bf16x8*>(bt + brow * 128 + (((2 * s4 + hi) ^ bswz(brow)) << 4));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
    }
    if (ks != nk - 1 || (p.dbg & 1)) continue;

    // ---------------- epilogue of N-tile jt (wave-local staging, 16 rows at a time) ----------------
    const int j = jt * PN + wj * 64 + cg * 8;
    const bool jok = j < p.J;
    const int jc = jok ? j : 0;
    float bias8[8];
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + jc);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + jc + 4);
      bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
      bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    }
    const float al = (j < p.alpha_cols) ? p.alpha : 1.0f;
    uint4 rres[RES ? 8 : 1];
    if (RES) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = min(i0 + wi * 64 + q * 8 + (lane >> 3), p.I - 1);
        rres[q] = *reinterpret_cast<const uint4*>(p.resid + (size_t)i * p.ldr + jc);
      }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {              // quarter: rows qt*16 .. +15 of the wave's 64
      const int a = qt >> 1;
      if (((lane >> 4) & 1) == (qt & 1)) {
        const int row = lane & 15;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int chunk = (b * 32 + 8 * g + 4 * hi) >> 2;
            *reinterpret_cast<float4*>(stg + row * 64 + ((chunk ^ row) << 2)) =
                make_float4(acc[a][b][g * 4], acc[a][b][g * 4 + 1], acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 8 + (lane >> 3);
        const int q = qt * 2 + h;                 // == (row within wave tile) / 8
        const int i = i0 + wi * 64 + qt * 16 + row;
        const float4 x0 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * cg) ^ row) << 2));
        const float4 x1 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * cg + 1) ^ row) << 2));
        const bool live = (i < p.I) && jok;
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias8[e]) * al;
        if (p.act == 1) {
          if (PRE && live)
            *reinterpret_cast<uint4*>(p.pre + (size_t)i * p.ldp + j) =
                make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        } else if (RES && p.act == 2) {
          const unsigned w[4] = {rres[q].x, rres[q].y, rres[q].z, rres[q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] *= dgelu_f(bf2f((bf16_t)(w[e] & 0xffff))); v[2 * e + 1] *= dgelu_f(bf2f((bf16_t)(w[e] >> 16))); }
        }
        if (RES && p.act != 2) {
          const unsigned w[4] = {rres[q].x, rres[q].y, rres[q].z, rres[q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += bf2f((bf16_t)(w[e] & 0xffff)); v[2 * e + 1] += bf2f((bf16_t)(w[e] >> 16)); }
        }
        if (live) {
          if (OUT == 0) {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)i * p.ldc + j) =
                make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
          } else {
            float* c = reinterpret_cast<float*>(p.C) + (size_t)i * p.ldc + j;
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    after_epilogue = (i0 + PM <= p.I && (jt + 1) * PN <= p.J) ? 1 : 2;
  }
}

template <int OUT, bool RES, bool PRE>
int launch_panel(const PanelParams& p, hipStream_t stream) {
  const int lds = PM * p.K * 2 + NSTAGE * BSTAGE_BYTES + 16384;
  static int attr_set = 0;
  if (attr_set < lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_kernel<OUT, RES, PRE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr_set = 163840;
  }
  hipLaunchKernelGGL((gemm_panel_kernel<OUT, RES, PRE>), dim3((p.I + PM - 1) / PM), dim3(256), lds, stream, p);
  return dig_check_launch();
}

}  // namespace

// C-ABI (include/dig_hip.h): forward-layout GEMM with the A row-panel resident in LDS.  Same epilogue contract as
// dig_gemm_bf16 (out_kind 0/1 only).  K must be a multiple of 128 and <= 384.
extern "C" int dig_gemm_panel_bf16(const void* A, const void* B, void* C, int I, int J, int K, int lda, int ldb, int ldc,
                                   int out_kind, const float* bias, const void* resid, int ldr, void* pre_act, int ldp,
                                   float alpha, int alpha_cols, int act, hipStream_t stream) {
  if (!A || !B || !C || I <= 0 || J <= 0 || K <= 0) return DIG_ERR_ARG;
  if ((K & 127) || K > 384) return DIG_ERR_UNSUPPORTED;
  if (out_kind < 0 || out_kind > 1 || act < 0 || act > 2 || (act == 2 && !resid) || (pre_act && act != 1)) return DIG_ERR_ARG;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (lda & 7) || (ldb & 7) || (J & 7) || (ldc & 7)) return DIG_ERR_ALIGN;
  if ((bias && !aligned16(bias)) || (resid && ((ldr & 7) || !aligned16(resid))) || (pre_act && ((ldp & 7) || !aligned16(pre_act)))) return DIG_ERR_ALIGN;
  PanelParams p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.I = I; p.J = J; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  const size_t ab = (size_t)I * lda * 2, bb = (size_t)J * ldb * 2;
  if (ab >= (1ull << 32) || bb >= (1ull << 32)) return DIG_ERR_ARG;
  p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  p.bias = bias; p.resid = (const bf16_t*)resid; p.ldr = ldr; p.pre = (bf16_t*)pre_act; p.ldp = ldp;
  p.alpha = alpha; p.alpha_cols = alpha_cols; p.act = act;
  { const char* e = getenv("DIG_PANEL_DBG"); p.dbg = e ? atoi(e) : 0; }
  const bool res = resid != nullptr, pre = pre_act != nullptr;
  if (out_kind == 0) {
    if (res) return launch_panel<0, true, false>(p, stream);
    if (pre) return launch_panel<0, false, true>(p, stream);
    return launch_panel<0, false, false>(p, stream);
  }
  if (res) return launch_panel<1, true, false>(p, stream);
  return launch_panel<1, false, false>(p, stream);
}
