// Phase timeline of the attention-backward kernel on one MI355X (warm clocks): s_memtime stamps at the phase boundaries of every wave
// (hooks: DIG_ATTN_TS in dig_amd/csrc/attention.hip), averaged over all workgroups, the s_memrealtime life of a wave and the wall time
// per launch.  (Round 2 used copies of the kernel with compile-time ablations here -- no softmax arithmetic / no LDS re-reads / no
// stores / no delta loads: the per-thread-row delta loads and the result stores were 45-50 us each of a 200 us launch.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I include -I dig_amd/csrc tools/experiments/attn_bwd_lab.hip -o build/lab/attn_bwd_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ long long* g_ts;
__device__ long long* g_rt;
#define NTS 8
#define DIG_ATTN_TS(i) if ((threadIdx.x & 63) == 0) { g_ts[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * NTS + (i)] = (long long)__builtin_amdgcn_s_memtime(); if ((i) == 0 || (i) == 6) g_rt[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + ((i) ? 1 : 0)] = (long long)__builtin_amdgcn_s_memrealtime(); }
#include "../../dig_amd/csrc/attention.hip"

static const int Bn = 256, H = 6, D = H * 64;
static unsigned short *qkv, *ctx, *dctx, *dqkv; static float* lse; static long long* ts; static long long* rt;

template <typename F> void timeline(const char* name, F fn) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(ts, 0, (size_t)Bn * H * 8 * NTS * 8);
  for (int it = 0; it < 400; ++it) fn();                     // ~70 ms of warm-up: the clocks ramp up over milliseconds
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int n = 50;
  for (int it = 0; it < n; ++it) fn();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t((size_t)Bn * H * 8 * NTS);
  hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost);
  double acc[NTS] = {0}; long cnt = 0; long long tmin = -1, tmax = 0;
  for (int b = 0; b < Bn * H; ++b)
    for (int w = 0; w < 8; ++w) {
      const long long* p = &t[((size_t)b * 8 + w) * NTS];
      if (!p[0] || !p[6]) continue;
      for (int i = 1; i <= 6; ++i) acc[i] += (double)(p[i] - p[i - 1]);
      if (tmin < 0 || p[0] < tmin) tmin = p[0];
      if (p[6] > tmax) tmax = p[6];
      ++cnt;
    }
  printf("%-34s %7.1f us |", name, ms / n * 1e3);
  double tot = 0;
  for (int i = 1; i <= 6; ++i) { printf(" %7.0f", cnt ? acc[i] / cnt : 0.0); tot += cnt ? acc[i] / cnt : 0.0; }
  { double a7 = 0; long c7 = 0; for (int b = 0; b < Bn * H; ++b) for (int w = 0; w < 4; ++w) { const long long* p = &t[((size_t)b * 8 + w) * NTS]; if (p[7] && p[3]) { a7 += (double)(p[7] - p[3]); ++c7; } }
    printf(" | wave %7.0f ticks; B block 0 loop %7.0f", tot, c7 ? a7 / c7 : 0.0); }
  { std::vector<long long> r((size_t)Bn * H * 8 * 2); hipMemcpy(r.data(), rt, r.size() * 8, hipMemcpyDeviceToHost);
    double life = 0; long c = 0; long long mn = -1, mx = 0;
    for (int b = 0; b < Bn * H; ++b) for (int w = 0; w < 4; ++w) { const long long* q = &r[((size_t)b * 8 + w) * 2]; if (q[0] && q[1]) { life += (double)(q[1] - q[0]); ++c; if (mn < 0 || q[0] < mn) mn = q[0]; if (q[1] > mx) mx = q[1]; } }
    printf("; realtime: wave life %.2f us, launch span %.1f us\n", c ? life / c / 100.0 : 0.0, (mx - mn) / 100.0); }
}

int main() {
  const size_t nq = (size_t)Bn * 256 * 3 * D, nc = (size_t)Bn * 256 * D;
  std::vector<unsigned short> h(nq);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&ctx, nc * 2); hipMalloc(&dctx, nc * 2); hipMalloc(&lse, (size_t)Bn * H * 256 * 4);
  hipMalloc(&ts, (size_t)Bn * H * 8 * NTS * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts));
  hipMalloc(&rt, (size_t)Bn * H * 8 * 2 * 8); hipMemset(rt, 0, (size_t)Bn * H * 8 * 2 * 8); hipMemcpyToSymbol(HIP_SYMBOL(g_rt), &rt, sizeof(rt));
  hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice);
  hipMemcpy(dctx, h.data(), nc * 2, hipMemcpyHostToDevice);
  dig_attn_fwd(qkv, ctx, lse, Bn, H, D, 0);
  printf("columns: issue loads | wait staging | delta | phase B (dK, dV) | restage K, V | phase A (dQ)  (s_memtime ticks per wave)\n");
  timeline("product dig_attn_bwd", [] { dig_attn_bwd(qkv, ctx, dctx, lse, dqkv, Bn, H, D, 0.125f, nullptr, nullptr, 0); });
  float *qs, *vs; hipMalloc(&qs, (size_t)Bn * D * 4); hipMalloc(&vs, (size_t)Bn * D * 4);
  timeline("product dig_attn_bwd + q/v bias sums", [=] { dig_attn_bwd(qkv, ctx, dctx, lse, dqkv, Bn, H, D, 0.125f, qs, vs, 0); });
  // the projection's data gradient inside the launch (dctx's buffer holds dy; the weight is random data)
  unsigned short* projt; hipMalloc(&projt, (size_t)D * D * 2); hipMemcpy(projt, h.data(), (size_t)D * D * 2, hipMemcpyHostToDevice);
  timeline("dig_attn_bwd_proj + q/v bias sums", [=] { dig_attn_bwd_proj(qkv, ctx, dctx, projt, lse, dqkv, Bn, H, D, 0.125f, qs, vs, 0); });
  return 0;
}
