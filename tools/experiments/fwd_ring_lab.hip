// Stand-alone harness of the ring forward GEMM (dig_amd/csrc/gemm_ring.hip): results against the persistent 256-row tiles of
// csrc/gemm.hip (dig_gemm_bf16, tile codes 544 / 564) on the same operands -- random (bf16 round-off of the two summation orders) and
// small integers (exact: must be equal bit for bit) -- and wall time of both for the encoder's two short-K layers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -I include -I dig_amd/csrc tools/experiments/fwd_ring_lab.hip -o build/lab/fwd_ring_lab
//   build/lab/fwd_ring_lab [R=65536]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../dig_amd/csrc/probe.hip"
#include "../../dig_amd/csrc/gemm.hip"
#include "gemm_ring.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float b2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

static void fill(unsigned short* d, size_t n, unsigned seed, bool integers, float scale) {
  std::vector<unsigned short> h(n);
  unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (auto& v : h) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const int r = (int)((s >> 40) & 0xffff) - 32768;
    v = integers ? f2b((float)(r % 3)) : f2b(r / 32768.0f * scale);
  }
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 65536;
  const int D = 384;
  hipStream_t st; CK(hipStreamCreate(&st));
  struct Case { const char* name; int N, K; bool bias, resid; float alpha; int alpha_cols; int tile; };
  const Case cases[] = {{"qkv  (N 1152, K 384, bias, q scale)", 3 * D, D, true, false, 0.125f, D, 544},
                        {"proj (N 384, K 384, bias, residual)", D, D, true, true, 1.0f, 0, 564},
                        {"fc2-like (N 384, K 1536, bias, residual)", D, 4 * D, true, true, 1.0f, 0, 564},
                        {"D=512 fc2-like (N 512, K 2048, bias, residual)", 512, 2048, true, true, 1.0f, 0, 244}};
  for (const Case& c : cases) {
    unsigned short *A, *W, *C0, *C1, *res;
    float* bias;
    CK(hipMalloc(&A, (size_t)R * c.K * 2)); CK(hipMalloc(&W, (size_t)c.N * c.K * 2));
    CK(hipMalloc(&C0, (size_t)R * c.N * 2)); CK(hipMalloc(&C1, (size_t)R * c.N * 2)); CK(hipMalloc(&res, (size_t)R * c.N * 2));
    CK(hipMalloc(&bias, c.N * 4));
    for (int integers = 0; integers < 2; ++integers) {
      fill(A, (size_t)R * c.K, 1 + integers, integers, 1.0f); fill(W, (size_t)c.N * c.K, 3 + integers, integers, 0.05f);
      fill(res, (size_t)R * c.N, 5 + integers, integers, 1.0f);
      std::vector<float> hb(c.N);
      for (int j = 0; j < c.N; ++j) hb[j] = integers ? (float)(j % 5 - 2) : 0.01f * (j % 17 - 8);
      CK(hipMemcpy(bias, hb.data(), c.N * 4, hipMemcpyHostToDevice));
      CK(hipMemset(C0, 0xff, (size_t)R * c.N * 2)); CK(hipMemset(C1, 0xff, (size_t)R * c.N * 2));
      int rc = dig_gemm_bf16(A, W, C0, R, c.N, c.K, c.K, c.K, c.N, 0, 0, 0, c.bias ? bias : nullptr, c.resid ? res : nullptr, c.N, nullptr, 0, c.alpha,
                             c.alpha_cols, 0, 1, 0, 0, c.tile, nullptr, st);
      if (rc) { printf("dig_gemm_bf16 rc %d\n", rc); return 1; }
      rc = dig_gemm_ring_fwd(A, W, C1, R, c.N, c.K, c.K, c.K, c.N, c.bias ? bias : nullptr, c.resid ? res : nullptr, c.N, c.alpha, c.alpha_cols, st);
      if (rc) { printf("dig_gemm_ring_fwd rc %d\n", rc); return 1; }
      CK(hipStreamSynchronize(st));
      std::vector<unsigned short> h0((size_t)R * c.N), h1((size_t)R * c.N);
      CK(hipMemcpy(h0.data(), C0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, h1.size() * 2, hipMemcpyDeviceToHost));
      double num = 0, den = 0; size_t neq = 0; double worst = 0; size_t wi = 0;
      for (size_t e = 0; e < h0.size(); ++e) {
        const double a = b2f(h0[e]), b = b2f(h1[e]);
        num += (a - b) * (a - b); den += a * a; neq += h0[e] != h1[e];
        if (fabs(a - b) > worst) { worst = fabs(a - b); wi = e; }
      }
      printf("%-44s %s: rel Frobenius diff %.3e, %zu of %zu values differ, worst |d| %.4g at (%zu, %zu): %g vs %g\n", c.name,
             integers ? "integers" : "random  ", sqrt(num / (den + 1e-30)), neq, h0.size(), worst, wi / c.N, wi % c.N, b2f(h0[wi]), b2f(h1[wi]));
    }
    // timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    for (int which = 0; which < 2; ++which) {
      for (int w = 0; w < 3 + reps; ++w) {
        if (w == 3) CK(hipEventRecord(e0, st));
        if (which == 0) dig_gemm_bf16(A, W, C0, R, c.N, c.K, c.K, c.K, c.N, 0, 0, 0, c.bias ? bias : nullptr, c.resid ? res : nullptr, c.N, nullptr, 0, c.alpha,
                                      c.alpha_cols, 0, 1, 0, 0, c.tile, nullptr, st);
        else dig_gemm_ring_fwd(A, W, C1, R, c.N, c.K, c.K, c.K, c.N, c.bias ? bias : nullptr, c.resid ? res : nullptr, c.N, c.alpha, c.alpha_cols, st);
      }
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("    %-28s %8.1f us  %7.1f TFLOP/s\n", which ? "ring (this kernel)" : "persistent tile (gemm.hip)", us, 2.0 * R * c.N * c.K / us * 1e-6);
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C0)); CK(hipFree(C1)); CK(hipFree(res)); CK(hipFree(bias));
  }
  return 0;
}
