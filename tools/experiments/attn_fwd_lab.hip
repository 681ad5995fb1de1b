// Where the attention FORWARD kernel spends its 51-55 us (256 images x 6 heads, 256 tokens, head dim 64): a copy of attn_fwd_kernel
// (dig_amd/csrc/attention.hip) with compile-time ablations, timed against the product kernel.
//   ABL bit 0: no softmax arithmetic (p = score, no max / exp / sum)      bit 1: no P V MFMAs      bit 2: no Q K^T MFMAs
//       bit 3: no context / lse stores                                     bit 4: no K / V staging (LDS left as it is), no Q loads
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=fast -w -I include -I dig_amd/csrc tools/experiments/attn_fwd_lab.hip -o build/lab/attn_fwd_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../dig_amd/csrc/attention.hip"
#ifndef LAB_PABL
#define LAB_PABL 0
#endif
#ifndef LAB_SLEEP
#define LAB_SLEEP 0
#endif

namespace {
template <int ABL>
__global__ __launch_bounds__(256, 2) void attn_fwd_abl(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, float* __restrict__ lse, int D, int H,
                                                       unsigned qkv_bytes) {
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
  if (!(ABL & 16)) {
    stage_tile<256>(Kt, rs, base + (unsigned)(D * 2), ld, tid, wave);
    stage_tile<256>(Vt, rs, base + (unsigned)(2 * D * 2), ld, tid, wave);
  }
  const int hi = lane >> 5;
  bf16x8 qf[2][4];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int q = (wave * 2 + ps) * 32 + (lane & 31);
    const bf16_t* qp = qkv + (tok0 + q) * ld + h * DH + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (!(ABL & 16)) qf[ps][s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);
      else { for (int e = 0; e < 8; ++e) qf[ps][s][e] = (short)(0x3c00 + lane + s); }
    }
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qb = wave * 2 + ps;
    f32x16 sc[8];
    if (ABL & 64) {                                                      // k step outer, key tile inner: eight independent accumulators in a row
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct(Kt, kt * 32, s, lane), qf[ps][s], sc[kt], 0, 0, 0);
    } else {
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[kt][e] = (ABL & 4) ? (float)(lane + e + kt) * 0.001f : 0.f;
      if (!(ABL & 4)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct(Kt, kt * 32, s, lane), qf[ps][s], sc[kt], 0, 0, 0);
      }
    }
    }
    float m = 0.f, l = 1.f;
    if (!(ABL & 1)) {
      m = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, sc[kt][e]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      l = 0.f;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __expf(sc[kt][e] - m);
          sc[kt][e] = p;
          l += p;
        }
      l += __shfl_xor(l, 32, 64);
    }
    f32x16 oa[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oa[dt][e] = (ABL & 2) ? sc[dt][e] + sc[dt + 2][e] + sc[dt + 4][e] + sc[dt + 6][e] : 0.f;
    if (!(ABL & 2)) {
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
    }
    const float inv = 1.0f / l;
    const int q = qb * 32 + (lane & 31);
    bf16_t* op = ctx + (tok0 + q) * D + h * DH;
    if (ABL & 32) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oa[dt][e] *= inv;
      store_rows(op, oa, hi);
      if (hi == 0) lse[(size_t)blockIdx.x * N_TOK + q] = m + __logf(l);
    } else if (!(ABL & 8) || oa[0][0] == 123.456f) {
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
}

// ---- K before V: asm LDS-DMA (unseen by the compiler), Q K^T starts when Q and K have landed, V is awaited in front of P V
__device__ __forceinline__ void lab_dma16b(unsigned lds_dst, unsigned voff, dig_u32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__global__ __launch_bounds__(256, 2) void attn_fwd_kfirst(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, float* __restrict__ lse, int D, int H,
                                                          unsigned qkv_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kt = smem;
  unsigned char* Vt = smem + TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / H, h = blockIdx.x - img * H;
  const int ld = 3 * D;
  const size_t tok0 = (size_t)img * N_TOK;
  const int hi = lane >> 5;
  dig_u32x4 rs;
  {
    const unsigned long long a = (unsigned long long)qkv;
    rs[0] = __builtin_amdgcn_readfirstlane((unsigned)a); rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    rs[2] = __builtin_amdgcn_readfirstlane(qkv_bytes); rs[3] = 0x00020000u;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  const unsigned base = (unsigned)((tok0 * ld + h * DH) * 2);
  // Q fragments first (needed first), as ONE asm block of 8 loads so that the compiler's own waits do not order them behind the DMA
  bf16x8 qf[2][4];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int q = (wave * 2 + ps) * 32 + (lane & 31);
    const bf16_t* qp = qkv + (tok0 + q) * ld + h * DH + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[ps][s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);
  }
  const int prow = tid >> 3;
  const unsigned pvoff = (unsigned)((prow * ld + (((tid & 7) ^ swz(prow)) * 8)) * 2);
#pragma unroll
  for (int t = 1; t < 3; ++t)
#pragma unroll
    for (int it = 0; it < 8; ++it)
      lab_dma16b(lds0 + (unsigned)((t - 1) * TILE) + (unsigned)((it * 256 + wave * 64) * 16), pvoff, rs, base + (unsigned)(t * D * 2) + (unsigned)(it * 32 * ld * 2));
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  // the Q loads and the K tile (8 + 8 of the 24 requests) have landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qb = wave * 2 + ps;
    f32x16 sc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct(Kt, kt * 32, s, lane), qf[ps][s], sc[kt], 0, 0, 0);
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
    if (ps == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // V
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
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
    const float inv = 1.0f / l;
    const int q = qb * 32 + (lane & 31);
    bf16_t* op = ctx + (tok0 + q) * D + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oa[dt][e] *= inv;
    store_rows(op, oa, hi);
    if (hi == 0) lse[(size_t)blockIdx.x * N_TOK + q] = m + __logf(l);
  }
}

// ---- the persistent form: 8 waves (a wave = one block of 32 queries), K / V double-buffered in LDS, Q through LDS, the next (image, head)'s
// three tiles on their way (LDS-DMA from inline asm, unseen by the compiler) while the current one is multiplied
__device__ __forceinline__ void lab_dma16(unsigned lds_dst, unsigned voff, dig_u32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__global__ __launch_bounds__(512, 2) void attn_fwd_persist(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, float* __restrict__ lse, int D, int H,
                                                           unsigned qkv_bytes, int n_items) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int ld = 3 * D;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  unsigned char* Qt = smem + 4 * TILE;
  dig_u32x4 rs;
  {
    const unsigned long long a = (unsigned long long)qkv;
    rs[0] = __builtin_amdgcn_readfirstlane((unsigned)a); rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    rs[2] = __builtin_amdgcn_readfirstlane(qkv_bytes); rs[3] = 0x00020000u;
  }
  // a tile = 2048 16-byte pieces; 512 threads bring it in 4 rounds of 64 rows: piece row = 64 it + (tid >> 3), position tid & 7 holds column
  // chunk (tid & 7) ^ swz(row) (the swizzle depends on the low row bits only)
  const int prow = tid >> 3;
  const unsigned pvoff = (unsigned)((prow * ld + (((tid & 7) ^ swz(prow)) * 8)) * 2);
  const unsigned pdst = (unsigned)(wave * 64 * 16);
  auto stage_item = [&](int item, int buf) {
    const int img = item / H, h = item - img * H;
    const unsigned base = (unsigned)((((size_t)img * N_TOK) * ld + h * DH) * 2);
#pragma unroll
    for (int t = 0; t < 3; ++t) {                                       // q, k, v thirds of the row
      const unsigned dst = lds0 + (t == 0 ? 4u * TILE : (unsigned)((2 * buf + (t - 1)) * TILE)) + pdst;
#pragma unroll
      for (int it = 0; it < 4; ++it)
        lab_dma16(dst + it * 512 * 16, pvoff, rs, base + (unsigned)(t * D * 2) + (unsigned)(it * 64 * ld * 2));
    }
  };
  int item = blockIdx.x, cur = 0;
  if (item < n_items) stage_item(item, 0);
  for (; item < n_items; item += gridDim.x, cur ^= 1) {
    const unsigned char* Kt = smem + (2 * cur) * TILE;
    const unsigned char* Vt = smem + (2 * cur + 1) * TILE;
    // the tiles of this item were requested BEFORE the previous item's 17 stores: let those stay in flight
    if (item == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = frag_direct(Qt, wave * 32, s, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // every wave has its queries: the Q tile may be overwritten
    asm volatile("" ::: "memory");
#if !(LAB_PABL & 1)
    if (item + (int)gridDim.x < n_items) stage_item(item + gridDim.x, cur ^ 1);
#endif
#if LAB_PABL & 2
    if (qf[0][0] != 12345) continue;
#endif
#if LAB_SLEEP > 0
    if (wave >= 4) __builtin_amdgcn_s_sleep(LAB_SLEEP);               // the two waves of a SIMD out of phase: one multiplies while the other exponentiates
#endif
    const int img = item / H, h = item - img * H;
    const size_t tok0 = (size_t)img * N_TOK;
    f32x16 sc[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct(Kt, kt * 32, s, lane), qf[s], sc[kt], 0, 0, 0);
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
    const float inv = 1.0f / l;
    const int q = wave * 32 + (lane & 31);
    bf16_t* op = ctx + (tok0 + q) * D + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * hi;
        *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_bf2(oa[dt][g * 4] * inv, oa[dt][g * 4 + 1] * inv),
                                                       pack_bf2(oa[dt][g * 4 + 2] * inv, oa[dt][g * 4 + 3] * inv));
      }
    if (hi == 0) lse[(size_t)item * N_TOK + q] = m + __logf(l);
  }
}
}  // namespace

static const int Bn = 256, H = 6, D = H * 64;

template <int ABL>
static void run(const char* name, const unsigned short* qkv, unsigned short* ctx, float* lse) {
  const size_t qb = (size_t)Bn * 256 * 3 * D * 2;
  hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_abl<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(attn_fwd_abl<ABL>, dim3(Bn * H), dim3(256), 2 * TILE, 0, qkv, ctx, lse, D, H, (unsigned)qb);
  hipEventRecord(e0, 0);
  const int n = 100;
  for (int it = 0; it < n; ++it) hipLaunchKernelGGL(attn_fwd_abl<ABL>, dim3(Bn * H), dim3(256), 2 * TILE, 0, qkv, ctx, lse, D, H, (unsigned)qb);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %7.1f us\n", name, ms / n * 1e3);
}

int main() {
  const size_t nq = (size_t)Bn * 256 * 3 * D, nc = (size_t)Bn * 256 * D;
  std::vector<unsigned short> hq(nq);
  srand(1);
  for (auto& v : hq) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  unsigned short *qkv, *ctx; float* lse;
  hipMalloc(&qkv, nq * 2); hipMalloc(&ctx, nc * 2); hipMalloc(&lse, (size_t)Bn * H * 256 * 4);
  hipMemcpy(qkv, hq.data(), nq * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 200; ++it) dig_attn_fwd(qkv, ctx, lse, Bn, H, D, 0);
  hipEventRecord(e0, 0);
  for (int it = 0; it < 100; ++it) dig_attn_fwd(qkv, ctx, lse, Bn, H, D, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %7.1f us\n", "product kernel (dig_attn_fwd)", ms / 100 * 1e3);
  run<0>("copy, nothing removed", qkv, ctx, lse);
  run<1>("no softmax arithmetic", qkv, ctx, lse);
  run<2>("no P V MFMAs", qkv, ctx, lse);
  run<4>("no Q K^T MFMAs", qkv, ctx, lse);
  run<6>("no MFMAs at all", qkv, ctx, lse);
  run<7>("no MFMAs, no softmax arithmetic (staging + stores only)", qkv, ctx, lse);
  run<32>("16-byte stores (v_permlane32_swap, store_rows)", qkv, ctx, lse);
  run<64>("Q K^T with the k step outer (independent accumulators in a row)", qkv, ctx, lse);
  run<96>("... and 16-byte stores", qkv, ctx, lse);
  run<8>("no stores", qkv, ctx, lse);
  run<16>("no K / V staging, no Q loads", qkv, ctx, lse);
  run<24>("no staging / loads, no stores (compute only)", qkv, ctx, lse);
  run<25>("... and no softmax arithmetic (MFMAs only)", qkv, ctx, lse);
  run<30>("... no MFMAs instead (softmax arithmetic only)", qkv, ctx, lse);
  {
    unsigned short* ctx3; float* lse3;
    hipMalloc(&ctx3, nc * 2); hipMalloc(&lse3, (size_t)Bn * H * 256 * 4);
    hipMemset(ctx3, 0xff, nc * 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kfirst), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
    dig_attn_fwd(qkv, ctx, lse, Bn, H, D, 0);
    hipLaunchKernelGGL(attn_fwd_kfirst, dim3(Bn * H), dim3(256), 2 * TILE, 0, qkv, ctx3, lse3, D, H, (unsigned)(nq * 2));
    hipDeviceSynchronize();
    std::vector<unsigned short> a(nc), b(nc);
    hipMemcpy(a.data(), ctx, nc * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ctx3, nc * 2, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < nc; ++i) bad += a[i] != b[i];
    printf("K-first form vs product: %zu of %zu context values differ\n", bad, nc);
    for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(attn_fwd_kfirst, dim3(Bn * H), dim3(256), 2 * TILE, 0, qkv, ctx3, lse3, D, H, (unsigned)(nq * 2));
    hipEventRecord(e0, 0);
    for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(attn_fwd_kfirst, dim3(Bn * H), dim3(256), 2 * TILE, 0, qkv, ctx3, lse3, D, H, (unsigned)(nq * 2));
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("K-first form (asm LDS-DMA, Q K^T starts before V has landed): %7.1f us\n", ms / 100 * 1e3);
  }
  {
    // the persistent form: results against the product kernel, then time
    unsigned short* ctx2; float* lse2;
    hipMalloc(&ctx2, nc * 2); hipMalloc(&lse2, (size_t)Bn * H * 256 * 4);
    hipMemset(ctx2, 0xff, nc * 2);
    const size_t qb = nq * 2;
    const int LDS = 5 * TILE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_persist), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    dig_attn_fwd(qkv, ctx, lse, Bn, H, D, 0);
    hipLaunchKernelGGL(attn_fwd_persist, dim3(256), dim3(512), LDS, 0, qkv, ctx2, lse2, D, H, (unsigned)qb, Bn * H);
    hipError_t e = hipDeviceSynchronize();
    printf("persistent launch: %s\n", hipGetErrorString(e));
    std::vector<unsigned short> a(nc), b(nc); std::vector<float> la((size_t)Bn * H * 256), lb((size_t)Bn * H * 256);
    hipMemcpy(a.data(), ctx, nc * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ctx2, nc * 2, hipMemcpyDeviceToHost);
    hipMemcpy(la.data(), lse, la.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(lb.data(), lse2, lb.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, badl = 0;
    for (size_t i = 0; i < nc; ++i) bad += a[i] != b[i];
    for (size_t i = 0; i < la.size(); ++i) badl += la[i] != lb[i];
    printf("persistent vs product: %zu of %zu context values differ, %zu of %zu lse values differ\n", bad, nc, badl, la.size());
    for (int grid : {256, 512}) {
      for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(attn_fwd_persist, dim3(grid), dim3(512), LDS, 0, qkv, ctx2, lse2, D, H, (unsigned)qb, Bn * H);
      hipEventRecord(e0, 0);
      for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(attn_fwd_persist, dim3(grid), dim3(512), LDS, 0, qkv, ctx2, lse2, D, H, (unsigned)qb, Bn * H);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("persistent form, grid %d: %7.1f us\n", grid, ms / 100 * 1e3);
    }
  }
  return 0;
}
