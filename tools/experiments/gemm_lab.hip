// Phase timeline of the forward GEMM tiles on one MI355X (warm clocks): per-wave s_memtime stamps at kernel entry, first K-step
// ready, K-loop end, barrier before the epilogue, kernel exit (hooks: DIG_GEMM_TS in dig_amd/csrc/gemm.hip), and the wall time per
// launch through the C-ABI for a list of (shape, tile variant, epilogue) cases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I include -I dig_amd/csrc tools/experiments/gemm_lab.hip -o build/lab/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ long long* g_ts;
__device__ long long* g_rt;
#define NTS 8
#define MAXW 16
#define DIG_GEMM_TS(i) if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) { g_ts[((size_t)blockIdx.x * MAXW + (threadIdx.x >> 6)) * NTS + (i)] = (long long)__builtin_amdgcn_s_memtime(); if ((i) == 0 || (i) == 4) g_rt[((size_t)blockIdx.x * MAXW + (threadIdx.x >> 6)) * 2 + ((i) ? 1 : 0)] = (long long)__builtin_amdgcn_s_memrealtime(); }
// persistent kernel: per-wave sums over its tiles of (wait for the tile's first stage | K loop | barrier | epilogue), in slots 0..3, tile count in 5
#define DIG_GEMM_PW_BEGIN() long long pw_last = (long long)__builtin_amdgcn_s_memtime(), pw_acc[4] = {0, 0, 0, 0}; int pw_n = 0; const long long pw_rt0 = (long long)__builtin_amdgcn_s_memrealtime();
#define DIG_GEMM_PW_ACC(k) { const long long pw_now = (long long)__builtin_amdgcn_s_memtime(); pw_acc[k] += pw_now - pw_last; pw_last = pw_now; if ((k) == 3) ++pw_n; }
#define DIG_GEMM_PW_END() if ((threadIdx.x & 63) == 0) { long long* q = g_ts + ((size_t)blockIdx.x * MAXW + (threadIdx.x >> 6)) * NTS; q[0] = 1; q[1] = 1 + pw_acc[0] / pw_n; q[2] = q[1] + pw_acc[1] / pw_n; q[3] = q[2] + pw_acc[2] / pw_n; q[4] = q[3] + pw_acc[3] / pw_n; long long* r = g_rt + ((size_t)blockIdx.x * MAXW + (threadIdx.x >> 6)) * 2; r[0] = pw_rt0; r[1] = (long long)__builtin_amdgcn_s_memrealtime(); }
#include "../../dig_amd/csrc/gemm.hip"

static long long *ts, *rt;
static const size_t NSLOT = (size_t)4096 * MAXW;

static void fill_bf16(unsigned short* d, size_t n, unsigned seed) {
  std::vector<unsigned short> h(n);
  srand(seed);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

template <typename F> void timeline(const char* name, double flops, double bytes, F fn) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 300; ++it) fn();                       // warm clocks
  hipDeviceSynchronize();
  hipMemset(ts, 0, NSLOT * NTS * 8); hipMemset(rt, 0, NSLOT * 2 * 8);
  hipEventRecord(e0, 0);
  const int n = 50;
  for (int it = 0; it < n; ++it) fn();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(NSLOT * NTS), r(NSLOT * 2);
  hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), rt, r.size() * 8, hipMemcpyDeviceToHost);
  double acc[NTS] = {0}, life = 0; long cnt = 0;
  for (size_t s = 0; s < NSLOT; ++s) {
    const long long* p = &t[s * NTS];
    if (!p[0] || !p[4]) continue;
    for (int i = 1; i <= 4; ++i) acc[i] += (double)(p[i] - p[i - 1]);
    life += (double)(r[s * 2 + 1] - r[s * 2]);
    ++cnt;
  }
  const double us = ms / n * 1e3;
  printf("%-44s %7.1f us %5.0f TF/s %5.2f TB/s |", name, us, flops / us / 1e6, bytes / us / 1e6);
  if (cnt) printf(" first stage %6.0f  K loop %6.0f  barrier %5.0f  epilogue %6.0f ticks | wave life %.2f us", acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, life / cnt / 100.0);
  printf("\n");
}

int main() {
  const int I = 65536;
  unsigned short *x384, *x1536, *w, *y, *pre, *res; float* bias;
  hipMalloc(&x384, (size_t)I * 384 * 2); hipMalloc(&x1536, (size_t)I * 1536 * 2); hipMalloc(&w, (size_t)1536 * 1536 * 2);
  hipMalloc(&y, (size_t)I * 1536 * 2); hipMalloc(&pre, (size_t)I * 1536 * 2); hipMalloc(&res, (size_t)I * 1536 * 2); hipMalloc(&bias, 1536 * 4);
  hipMalloc(&ts, NSLOT * NTS * 8); hipMalloc(&rt, NSLOT * 2 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)); hipMemcpyToSymbol(HIP_SYMBOL(g_rt), &rt, sizeof(rt));
  fill_bf16(x384, (size_t)I * 384, 1); fill_bf16(x1536, (size_t)I * 1536, 2); fill_bf16(w, (size_t)1536 * 1536, 3); fill_bf16(res, (size_t)I * 1536, 4);
  hipMemset(bias, 0, 1536 * 4);
  struct Case { const char* name; int J, R; int act; bool pre, res; int bk; };
  const Case cases[] = {
      {"fc1 384->1536 gelu      256x256 (244)", 1536, 384, 1, false, false, 244},
      {"fc1 384->1536 gelu+pre  256x256 (244)", 1536, 384, 1, true, false, 244},
      {"fc1 384->1536 plain     256x256 (244)", 1536, 384, 0, false, false, 244},
      {"fc1 384->1536 gelu      persistent 256x256 (544)", 1536, 384, 1, false, false, 544},
      {"fc1 384->1536 gelu+pre  persistent 256x256 (544)", 1536, 384, 1, true, false, 544},
      {"fc1 384->1536 plain     persistent 256x256 (544)", 1536, 384, 0, false, false, 544},
      {"qkv 384->1152 bias      persistent 256x256 (544)", 1152, 384, 0, false, false, 544},
      {"proj 384->384 +res      persistent 256x192 (564)", 384, 384, 0, false, true, 564},
      {"fc2 1536->384 +res      persistent 256x192 (564)", 384, 1536, 0, false, true, 564},
      {"qkv 384->1152 bias      256x256 (244)", 1152, 384, 0, false, false, 244},
      {"proj 384->384 +res      256x192 (264)", 384, 384, 0, false, true, 264},
      {"fc2 1536->384 +res      256x192 (264)", 384, 1536, 0, false, true, 264},
  };
  printf("ticks = s_memtime (shader clock); first stage = entry -> first operand stage landed\n");
  for (const Case& c : cases) {
    const unsigned short* x = c.R == 384 ? x384 : x1536;
    const double fl = 2.0 * I * c.J * c.R;
    const double by = 2.0 * I * c.R + 2.0 * c.J * c.R + 2.0 * I * c.J * (1 + (c.pre ? 1 : 0) + (c.res ? 1 : 0));
    timeline(c.name, fl, by, [&] {
      dig_gemm_bf16(x, w, y, I, c.J, c.R, c.R, c.R, c.J, 0, 0, 0, bias, c.res ? res : nullptr, c.J, c.pre ? pre : nullptr, c.J, 1.0f, 0, c.act, 1, 0, 0, c.bk,
                    nullptr, 0);
    });
  }
  // 8-wave forms of the 256x256 tile (not in the library): 2x4 waves of 128x64 and 4x2 waves of 64x128
  {
    GemmParams q{};
    q.A = x384; q.B = w; q.C = y; q.I = I; q.J = 1536; q.R = 384; q.lda = 384; q.ldb = 384; q.ldc = 1536;
    q.a_bytes = (unsigned)((size_t)I * 384 * 2); q.b_bytes = 1536u * 384 * 2; q.bias = bias; q.resid = nullptr; q.ldr = 0; q.pre = nullptr; q.ldp = 0;
    q.alpha = 1.f; q.alpha_cols = 0; q.act = 0; q.r_per_split = 384; q.colsum = nullptr; q.splits_x = 0;
    const double fl = 2.0 * I * 1536 * 384, by = 2.0 * I * 384 + 2.0 * 1536 * 384 + 2.0 * I * 1536;
    timeline("fc1 plain 256x256, 8 waves 2x4 of 128x64", fl, by, [&] { launch_wide<false, false, 0, 2, 4, 4, 2, false, 64, 2>(q, 1, 0); });
    timeline("fc1 plain 256x256, 8 waves 4x2 of 64x128", fl, by, [&] { launch_wide<false, false, 0, 4, 2, 2, 4, false, 64, 2>(q, 1, 0); });
    q.act = 1;
    timeline("fc1 gelu  256x256, 8 waves 2x4 of 128x64", fl, by, [&] { launch_wide<false, false, 0, 2, 4, 4, 2, false, 64, 2>(q, 1, 0); });
  }
  // ---- backward shapes: ring depth of the 128x128 tile (bytes in flight per CU) for HBM-streamed operands
  {
    unsigned short* dact = x1536;                                        // [I, 1536]
    float* slabs; hipMalloc(&slabs, (size_t)24 * 1536 * 384 * 4);
    auto dgrad_params = [&](int J, int R) {
      GemmParams q{};
      q.A = dact; q.B = w; q.C = y; q.I = I; q.J = J; q.R = R; q.lda = R; q.ldb = J; q.ldc = J;
      q.a_bytes = (unsigned)((size_t)I * R * 2); q.b_bytes = (unsigned)((size_t)R * J * 2); q.bias = nullptr; q.resid = nullptr; q.pre = nullptr;
      q.alpha = 1.f; q.alpha_cols = 0; q.act = 0; q.r_per_split = R; q.colsum = nullptr; q.splits_x = 0;
      q.tiles_i = (I + 127) / 128; q.tiles_j = (J + 127) / 128;
      return q;
    };
    {
      GemmParams q = dgrad_params(384, 1536);
      const double fl = 2.0 * I * 384 * 1536, by = 2.0 * I * 1536 + 2.0 * I * 384;
      timeline("dgrad fc1 (K 1536 -> 384) 128x128 BK32 2-stage", fl, by, [&] { launch<false, true, 0, 32, false, 2>(q, 1, 0); });
      timeline("dgrad fc1 (K 1536 -> 384) 128x128 BK32 3-stage", fl, by, [&] { launch<false, true, 0, 32, false, 3>(q, 1, 0); });
      timeline("dgrad fc1 (K 1536 -> 384) 128x128 BK32 4-stage", fl, by, [&] { launch<false, true, 0, 32, false, 4>(q, 1, 0); });
      timeline("dgrad fc1 (K 1536 -> 384) 128x128 BK64 2-stage", fl, by, [&] { launch<false, true, 0, 64, false, 2>(q, 1, 0); });
      timeline("dgrad fc1 (K 1536 -> 384) 256x192 BK64 2-stage", fl, by, [&] { launch_wide<false, true, 0, 4, 3, 2, 2, false, 64, 2>(q, 1, 0); });
    }
    {
      GemmParams q = dgrad_params(384, 1152);
      const double fl = 2.0 * I * 384 * 1152, by = 2.0 * I * 1152 + 2.0 * I * 384;
      timeline("dgrad qkv (K 1152 -> 384) 128x128 BK32 2-stage", fl, by, [&] { launch<false, true, 0, 32, false, 2>(q, 1, 0); });
      timeline("dgrad qkv (K 1152 -> 384) 128x128 BK32 4-stage", fl, by, [&] { launch<false, true, 0, 32, false, 4>(q, 1, 0); });
    }
    {
      // wgrad fc1: dW[1536, 384] = dact^T [1536, R] x ln2 [R, 384], 24 R-splits into fp32 slabs
      GemmParams q{};
      q.A = dact; q.B = x384; q.C = slabs; q.I = 1536; q.J = 384; q.R = I; q.lda = 1536; q.ldb = 384; q.ldc = 384;
      q.a_bytes = (unsigned)((size_t)I * 1536 * 2); q.b_bytes = (unsigned)((size_t)I * 384 * 2); q.bias = nullptr; q.resid = nullptr; q.pre = nullptr;
      q.alpha = 1.f; q.alpha_cols = 0; q.act = 0; q.colsum = nullptr; q.splits_x = 0;
      q.tiles_i = 12; q.tiles_j = 3;
      const int splits = dig_gemm_effective_splits(I, 24);
      q.r_per_split = ((I / 64 + splits - 1) / splits) * 64;
      const double fl = 2.0 * I * 384 * 1536, by = 2.0 * I * 1536 + 2.0 * I * 384;
      timeline("wgrad fc1 (1536 x 384, R 65536) 128x128 BK32 2-stage", fl, by, [&] { launch<true, true, 2, 32, false, 2>(q, splits, 0); });
      timeline("wgrad fc1 (1536 x 384, R 65536) 128x128 BK32 3-stage", fl, by, [&] { launch<true, true, 2, 32, false, 3>(q, splits, 0); });
      timeline("wgrad fc1 (1536 x 384, R 65536) 128x128 BK32 4-stage", fl, by, [&] { launch<true, true, 2, 32, false, 4>(q, splits, 0); });
    }
  }
  return 0;
}
