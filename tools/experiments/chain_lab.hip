// Where the time of the fused MLP chain kernel (dig_amd/csrc/mlp_chain.hip) goes: per-wave s_memtime accounting of the tick protocol
// (waiting for the ring DMA | waiting at the workgroup barrier | issuing the next DMA | the tick's own work) for S-waves and O-waves,
// plus the wall time per launch, for the product kernel and for compile-time ablations (-DDIG_CHAIN_ABL=bits, see mlp_chain.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -I include -I dig_amd/csrc [-DDIG_CHAIN_ABL=n] tools/experiments/chain_lab.hip -o build/lab/chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ long long* g_ts;
#define NTS 8
#define DIG_CHAIN_T_BEGIN() long long ct_last = (long long)__builtin_amdgcn_s_memtime(), ct_acc[4] = {0, 0, 0, 0}; const long long ct_t0 = ct_last;
#define DIG_CHAIN_T(k) { const long long ct_now = (long long)__builtin_amdgcn_s_memtime(); ct_acc[k] += ct_now - ct_last; ct_last = ct_now; }
#define DIG_CHAIN_T_END() if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) { long long* q = g_ts + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * NTS; for (int k = 0; k < 4; ++k) q[k] = ct_acc[k]; q[4] = ct_last - ct_t0; q[5] = ct_t0; }
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
  long long* ts; hipMalloc(&ts, (size_t)1024 * 8 * NTS * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts));
  for (int mode = 0; mode < 3; ++mode) {
    auto fn = [&]() {
      if (mode == 0) dig_mlp_chain_fwd(x, w1, b1, w2, b2, res, out, nullptr, nullptr, R, D, F, 0);
      else if (mode == 1) dig_mlp_chain_fwd(x, w1, b1, w2, b2, res, out, pre, act, R, D, F, 0);
      else dig_mlp_chain_bwd(x, w1, pre, w2, dpre, out, cs, R, D, F, 0);
    };
    for (int it = 0; it < 200; ++it) fn();
    hipDeviceSynchronize();
    hipMemset(ts, 0, (size_t)1024 * 8 * NTS * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    const int n = 30;
    for (int it = 0; it < n; ++it) fn();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t((size_t)1024 * 8 * NTS);
    hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost);
    const double us = ms / n * 1e3;
    printf("ABL %d mode %d: %7.1f us  %6.0f TFLOP/s\n", DIG_CHAIN_ABL, mode, us, 4.0 * R * D * F / us / 1e6);
    for (int role = 0; role < 2; ++role) {
      double acc[5] = {0}; long cnt = 0;
      for (int b = 0; b < 512; ++b)
        for (int w = role * 4; w < role * 4 + 4; ++w) {
          const long long* q = &t[((size_t)b * 8 + w) * NTS];
          if (!q[4]) continue;
          for (int k = 0; k < 5; ++k) acc[k] += (double)q[k];
          ++cnt;
        }
      if (cnt) printf("   %s-wave: wave life %8.0f ticks = dma wait %7.0f + barrier %7.0f + dma issue %6.0f + work %7.0f   (78 protocol ticks: %5.0f per tick)\n",
                      role ? "O" : "S", acc[4] / cnt, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt / 78.0);
    }
  }
  return 0;
}
