// Where the time of the fused MLP chain kernel (dig_amd/csrc/mlp_chain.hip) goes: per-wave s_memtime accounting of the tick protocol
// (waiting for the ring DMA | waiting at the workgroup barrier | issuing the next DMA | the tick's own work) for S-waves and O-waves,
// plus the wall time per launch, for the product kernel and for compile-time ablations (-DDIG_CHAIN_ABL=bits, see mlp_chain.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -I include -I dig_amd/csrc [-DDIG_CHAIN_ABL=n] tools/experiments/chain_lab.hip -o build/lab/chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
// Stamps go to a spare LDS region (12 KiB at the top of the 160 KiB; the product kernel uses at most 144 KiB) with one ds_write_b32 each
// -- accumulators or global stores would cost registers (the kernel sits at 245-256 VGPRs; an earlier version of this lab spilled 768
// bytes per lane and measured its own scratch traffic) -- and workgroup TL_BLOCK copies them out at its end.
__device__ unsigned* g_tl;                   // [8 waves][80 ticks][4 stamps]: 3 = arrive (end of the previous tick's work), 0 = DMA landed, 1 = barrier released, 2 = tick set up
#define TL_BLOCK 300
#define TL_OFF (160 * 1024 - 12 * 1024)
#define DIG_CHAIN_LDS_ALL 1
#define DIG_CHAIN_T_BEGIN() int ct_n = 0;
#define DIG_CHAIN_T(k) { if (ct_n < 80) { const unsigned ct_now = (unsigned)__builtin_amdgcn_s_memtime(); const unsigned ct_a = TL_OFF + (((threadIdx.x >> 6) * 80 + ct_n) * 4 + (k)) * 4; \
  asm volatile("ds_write_b32 %0, %1" ::"v"(ct_a), "v"(ct_now) : "memory"); } if ((k) == 2) ++ct_n; }
#define DIG_CHAIN_T_END() if (blockIdx.x == TL_BLOCK) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const int ct_w = threadIdx.x >> 6, ct_l = threadIdx.x & 63; \
  for (int i = ct_l; i < 320; i += 64) g_tl[ct_w * 320 + i] = *reinterpret_cast<const unsigned*>(smem + TL_OFF + (ct_w * 320 + i) * 4); }
#include "../../dig_amd/csrc/mlp_chain.hip"

static void fill_bf16(unsigned short* d, size_t n, unsigned seed, float scale) {
  std::vector<unsigned short> h(n);
  srand(seed);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f * scale; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  const int R = 65536, D = 384, F = 1536;
  unsigned short *x, *w1, *w2, *res, *out, *pre, *act, *dpre;
  float *b1, *b2, *cs;
  hipMalloc(&x, (size_t)R * D * 2); hipMalloc(&res, (size_t)R * D * 2); hipMalloc(&out, (size_t)R * D * 2);
  hipMalloc(&w1, (size_t)F * D * 2); hipMalloc(&w2, (size_t)F * D * 2);
  hipMalloc(&pre, (size_t)R * F * 2); hipMalloc(&act, (size_t)R * F * 2); hipMalloc(&dpre, (size_t)R * F * 2);
  hipMalloc(&b1, F * 4); hipMalloc(&b2, D * 4); hipMalloc(&cs, (size_t)(R / 32) * F * 4);
  hipMemset(b1, 0, F * 4); hipMemset(b2, 0, D * 4);
  fill_bf16(x, (size_t)R * D, 1, 1.f); fill_bf16(res, (size_t)R * D, 2, 1.f);
  fill_bf16(w1, (size_t)F * D, 3, 0.1f); fill_bf16(w2, (size_t)F * D, 4, 0.07f);
  fill_bf16(pre, (size_t)R * F, 5, 2.f);
  unsigned* tl; hipMalloc(&tl, 8 * 320 * 4); hipMemset(tl, 0, 8 * 320 * 4);
  hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
  // modes 3 / 4: the forward forms with the LayerNorms fused at both ends (dig_mlp_chain_fwd_ln; 4 = with everything the backward keeps)
  unsigned short *lno, *nlno; float *lg, *lb, *st;
  hipMalloc(&lno, (size_t)R * D * 2); hipMalloc(&nlno, (size_t)R * D * 2); hipMalloc(&lg, D * 4); hipMalloc(&lb, D * 4); hipMalloc(&st, (size_t)R * 4 * 4);
  { std::vector<float> one(D, 1.f); hipMemcpy(lg, one.data(), D * 4, hipMemcpyHostToDevice); hipMemset(lb, 0, D * 4); }
  for (int mode = 0; mode < 5; ++mode) {
    auto fn = [&]() {
      if (mode == 0) dig_mlp_chain_fwd(x, w1, b1, w2, b2, res, out, nullptr, nullptr, R, D, F, 0);
      else if (mode == 1) dig_mlp_chain_fwd(x, w1, b1, w2, b2, res, out, pre, act, R, D, F, 0);
      else if (mode == 2) dig_mlp_chain_bwd(x, w1, pre, w2, dpre, out, cs, R, D, F, 0);
      else if (mode == 3) dig_mlp_chain_fwd_ln(x, x, lg, lb, 1e-6f, nullptr, nullptr, nullptr, w1, b1, w2, b2, out, nullptr, nullptr, lg, lb, nlno, nullptr, nullptr, R, D, F, 0);
      else dig_mlp_chain_fwd_ln(x, x, lg, lb, 1e-6f, lno, st, st + R, w1, b1, w2, b2, out, pre, act, lg, lb, nlno, st + 2 * R, st + 3 * R, R, D, F, 0);
    };
    for (int it = 0; it < 200; ++it) fn();
    hipDeviceSynchronize();
    hipMemset(tl, 0, 8 * 320 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    const int n = 30;
    for (int it = 0; it < n; ++it) fn();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / n * 1e3;
    printf("ABL %d mode %d: %7.1f us  %6.0f TFLOP/s\n", DIG_CHAIN_ABL, mode, us, 4.0 * R * D * F / us / 1e6);
    std::vector<unsigned> L(8 * 320);
    hipMemcpy(L.data(), tl, L.size() * 4, hipMemcpyDeviceToHost);
    // aggregates of workgroup TL_BLOCK (last launch): per role, mean over its 4 waves of the 78 protocol ticks
    unsigned t0 = L[3];
    for (int w = 0; w < 8; ++w) if ((int)(L[w * 320 + 3] - t0) < 0) t0 = L[w * 320 + 3];
    for (int role = 0; role < 2; ++role) {
      double dw = 0, bw = 0, su = 0, wk = 0, life = 0;
      for (int w = role * 4; w < role * 4 + 4; ++w) {
        for (int k = 0; k < 78; ++k) {
          const unsigned* q = &L[(w * 80 + k) * 4];
          dw += (int)(q[0] - q[3]); bw += (int)(q[1] - q[0]); su += (int)(q[2] - q[1]);
          wk += (int)(L[(w * 80 + k + 1) * 4 + 3] - q[2]);
        }
        life += (int)(L[(w * 80 + 78) * 4 + 3] - L[w * 320 + 3]);
      }
      printf("   %s-wave (workgroup %d): 78 ticks = %7.0f cycles = dma wait %6.0f + barrier %6.0f + set-up %5.0f + work %7.0f   (%5.0f per tick)\n",
             role ? "O" : "S", TL_BLOCK, life / 4, dw / 4, bw / 4, su / 4, wk / 4, life / 4 / 78.0);
    }
    if (getenv("CHAIN_TL")) {
      printf("   timeline (per wave and tick: arrival, +dma wait, +barrier wait), cycles since the workgroup's first stamp\n");
      for (int k = 0; k < 79; ++k) {
        printf("   t%02d", k);
        for (int w = 0; w < 8; ++w) { const unsigned* q = &L[(w * 80 + k) * 4]; printf(" | %c%d %6d %5d %5d", w < 4 ? 'S' : 'O', w & 3, (int)(q[3] - t0), (int)(q[0] - q[3]), (int)(q[1] - q[0])); }
        printf("\n");
      }
    }
  }
  return 0;
}
