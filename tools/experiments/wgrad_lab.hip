// Stand-alone harness of the grouped weight-gradient kernel (dig_amd/csrc/wgrad.hip): correctness against the tiled split-R path of
// csrc/gemm.hip (dig_gemm_bf16 + dig_reduce_partials) on the same random operands, then wall time per encoder block's four weight
// gradients for the old launch sequence and for the grouped kernel in its launch groupings.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -I include -I dig_amd/csrc tools/experiments/wgrad_lab.hip -o build/lab/wgrad_lab
//   build/lab/wgrad_lab [R=65536] [D=384]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../dig_amd/csrc/probe.hip"
#include "../../dig_amd/csrc/gemm.hip"
#ifdef WG_LAB_TS
// per-wave phase accounting of the wide kernel's stage loop: [0] wait for own fragment reads, [1] wait for own LDS-DMA pieces, [2] barrier,
// [3] DMA issue + fragment requests + MFMAs, [4] final drain; [5] loop total
__device__ long long* g_wts;
#define DIG_WG_TS_BEGIN() const long long wts_k0 = (long long)__builtin_amdgcn_s_memtime(); const long long wts_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#define DIG_WG_TS_DECL() long long wts_acc[5] = {0, 0, 0, 0, 0}; long long wts_last = (long long)__builtin_amdgcn_s_memtime(); const long long wts_begin = wts_last;
#define DIG_WG_TS(k) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); wts_acc[k] += n_ - wts_last; wts_last = n_; }
#define DIG_WG_TS_END() if ((threadIdx.x & 63) == 0) { long long* q_ = g_wts + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; for (int k_ = 0; k_ < 5; ++k_) q_[k_] = wts_acc[k_]; q_[5] = wts_acc[0] + wts_acc[1] + wts_acc[2] + wts_acc[3]; q_[6] = wts_begin - wts_k0; q_[7] = (long long)__builtin_amdgcn_s_memtime() - wts_k0; q_[4] = (long long)__builtin_amdgcn_s_memrealtime() - wts_r0; }
#endif
#include "../../dig_amd/csrc/wgrad.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static void fill_bf16(unsigned short* d, size_t n, unsigned seed) {
  std::vector<unsigned short> h(n);
  unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (auto& v : h) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const float f = ((int)((s >> 40) & 0xffff) - 32768) / 32768.0f;
    unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
  }
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
}

struct Prob { const char* name; unsigned short* A; unsigned short* B; float* out; float* ref; int I, J, lda, ldb, ldo, trans; };

static int old_wgrad(const Prob& q, int R, float* ws, hipStream_t st) {
  // the product path of rounds 1-3: dW[Iout, Jout] += dy^T x as 128x128 tiles, 16 / 40 R-splits, then the slab sum.  For a transposed
  // problem the roles of the operands are swapped (out is [J, I]).
  const unsigned short* dy = q.trans ? q.B : q.A;
  const unsigned short* x = q.trans ? q.A : q.B;
  const int I = q.trans ? q.J : q.I, J = q.trans ? q.I : q.J;
  const int lda = q.trans ? q.ldb : q.lda, ldb = q.trans ? q.lda : q.ldb;
  const int tiles = ((I + 127) / 128) * ((J + 127) / 128);
  int want = tiles >= 24 ? 16 : std::min(40, 8 * std::max(1, (int)lround(360.0 / tiles / 8)));
  const int bk = tiles >= 24 ? 32 : 64;
  const int sp = dig_gemm_effective_splits(R, want);
  int rc = dig_gemm_bf16(dy, x, ws, I, J, R, lda, ldb, J, 1, 1, 2, nullptr, nullptr, 0, nullptr, 0, 1.f, 0, 0, sp, 0, 0, bk, nullptr, st);
  if (rc) return rc;
  return dig_reduce_partials(ws, sp, (long long)I * J, q.ref, 1, st);
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 65536;
  const int D = argc > 2 ? atoi(argv[2]) : 384;
  const int F = 4 * D, Q = 3 * D;
  const int fn = dig_wgrad_group_fn(D);
  printf("wgrad lab: R = %d, D = %d (fn %d)\n", R, D, fn);
  unsigned short *dact, *act, *ln2, *dx, *dqkv, *ln1, *ctx, *dxm;
  CK(hipMalloc(&dact, (size_t)R * F * 2)); CK(hipMalloc(&act, (size_t)R * F * 2)); CK(hipMalloc(&ln2, (size_t)R * D * 2)); CK(hipMalloc(&dx, (size_t)R * D * 2));
  CK(hipMalloc(&dqkv, (size_t)R * Q * 2)); CK(hipMalloc(&ln1, (size_t)R * D * 2)); CK(hipMalloc(&ctx, (size_t)R * D * 2)); CK(hipMalloc(&dxm, (size_t)R * D * 2));
  fill_bf16(dact, (size_t)R * F, 1); fill_bf16(act, (size_t)R * F, 2); fill_bf16(ln2, (size_t)R * D, 3); fill_bf16(dx, (size_t)R * D, 4);
  fill_bf16(dqkv, (size_t)R * Q, 5); fill_bf16(ln1, (size_t)R * D, 6); fill_bf16(ctx, (size_t)R * D, 7); fill_bf16(dxm, (size_t)R * D, 8);
  auto falloc = [](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
  Prob probs[4] = {
      {"fc1  dW[F,D] = dact^T ln2", dact, ln2, falloc((size_t)F * D), falloc((size_t)F * D), F, D, F, D, D, 0},
      {"fc2  dW[D,F] = dx^T act  ", act, dx, falloc((size_t)F * D), falloc((size_t)F * D), F, D, F, D, F, 1},
      {"qkv  dW[3D,D] = dqkv^T ln1", dqkv, ln1, falloc((size_t)Q * D), falloc((size_t)Q * D), Q, D, Q, D, D, 0},
      {"proj dW[D,D] = dxm^T ctx ", dxm, ctx, falloc((size_t)D * D), falloc((size_t)D * D), D, D, D, D, D, 0},
  };
  // (proj: out[i, j] = sum_r dxm[r, i] ctx[r, j]: the "wide" operand is dxm)
  float* ws; CK(hipMalloc(&ws, (size_t)40 * F * D * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
#ifdef WG_LAB_TS
  long long* d_ts; CK(hipMalloc(&d_ts, 1024 * 8 * 8 * 8)); CK(hipMemset(d_ts, 0, 1024 * 8 * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wts), &d_ts, sizeof(d_ts)));
#endif

  // ---- grouped-kernel plumbing
  const long long slab_cap = 2ll * 640 * 128 * 128 * fn * 4;            // two slab sets of up to 640 (tile, split) slabs
  float* slabs; CK(hipMalloc(&slabs, slab_cap));
  float* slab_set[2] = {slabs, slabs + slab_cap / 8};
  unsigned* d_map; CK(hipMalloc(&d_map, 8 * 4096 * 4));
  struct Group { std::vector<int> idx; int splits, eff, n_wg; unsigned* map; int tiles; };
  int wa = 1;
  auto make_group = [&](std::vector<int> idx, int slots, unsigned* dmap) {
    Group g; g.idx = idx; g.map = dmap;
    std::vector<int> tp; g.tiles = 0;
    for (int k : idx) { tp.push_back(dig_wgrad_group_tiles(probs[k].I, probs[k].J, fn, wa)); g.tiles += tp.back(); }
    std::vector<unsigned> hm(8 * 4096);
    g.n_wg = dig_wgrad_group_plan(tp.data(), (int)tp.size(), R, slots, &g.eff, hm.data(), (int)hm.size());
    if (g.n_wg <= 0) { printf("plan failed %d\n", g.n_wg); exit(3); }
    CK(hipMemcpy(dmap, hm.data(), (size_t)g.n_wg * 4, hipMemcpyHostToDevice));
    g.splits = g.eff;
    return g;
  };
  auto to_abi = [&](const Group& g, dig_wgrad_prob_t* o) {
    for (size_t k = 0; k < g.idx.size(); ++k) {
      const Prob& q = probs[g.idx[k]];
      o[k] = {q.A, q.B, q.out, q.lda, q.ldb, q.ldo, q.I, q.J, q.trans};
    }
  };
  int set = 0;
  const Group* pending = nullptr;
  auto launch_group = [&](const Group* g) {               // g == nullptr: flush
    dig_wgrad_prob_t cur[6], prev[6];
    if (g) to_abi(*g, cur);
    if (pending) to_abi(*pending, prev);
    const int rc = dig_wgrad_group(g ? cur : nullptr, g ? (int)g->idx.size() : 0, pending ? prev : nullptr, pending ? (int)pending->idx.size() : 0, R,
                                   g ? g->splits : 1, g ? g->map : nullptr, g ? g->n_wg : 512 / wa, slab_set[set], pending ? slab_set[set ^ 1] : nullptr,
                                   pending ? pending->splits : 1, fn, wa, st);
    if (rc) { printf("dig_wgrad_group rc %d\n", rc); exit(4); }
    pending = g;
    set ^= 1;
  };

  // ---- correctness: one pass of each grouping against the tiled path
  for (int k = 0; k < 4; ++k) { int rc = old_wgrad(probs[k], R, ws, st); if (rc) { printf("old path rc %d\n", rc); return 5; } }
  CK(hipStreamSynchronize(st));
  std::vector<std::vector<std::vector<int>>> groupings = {{{0, 1, 2, 3}}, {{1, 0}, {3, 2}}, {{1}, {0}, {3}, {2}}};
  const char* gname[3] = {"one launch per block", "two launches (MLP pair, attention pair)", "four launches"};
  int bad_total = 0;
  for (wa = 1; wa <= 2; ++wa)
  for (size_t gi = 0; gi < groupings.size(); ++gi) {
    const int slots = 512 / wa;
    for (int k = 0; k < 4; ++k) CK(hipMemset(probs[k].out, 0, (size_t)probs[k].I * probs[k].J * 4));
    std::vector<Group> gs;
    for (size_t j = 0; j < groupings[gi].size(); ++j) gs.push_back(make_group(groupings[gi][j], slots, d_map + j * 4096));
    for (auto& g : gs) launch_group(&g);
    launch_group(nullptr);
    CK(hipStreamSynchronize(st));
    for (int k = 0; k < 4; ++k) {
      const size_t n = (size_t)probs[k].I * probs[k].J;
      std::vector<float> a(n), b(n);
      CK(hipMemcpy(a.data(), probs[k].out, n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), probs[k].ref, n * 4, hipMemcpyDeviceToHost));
      double num = 0, den = 0, mx = 0; size_t nbad = 0;
      for (size_t e = 0; e < n; ++e) { const double d = (double)a[e] - b[e]; num += d * d; den += (double)b[e] * b[e]; mx = std::max(mx, fabs(d)); if (!(fabs(d) <= 1e-3 * (1.0 + fabs(b[e])))) ++nbad; }
      printf("  [wa %d, %s] %s: rel err %.3e  max abs %.3e  bad %zu / %zu\n", wa, gname[gi], probs[k].name, sqrt(num / (den + 1e-30)), mx, nbad, n);
      bad_total += nbad != 0;
    }
    printf("  [%s] splits:", gname[gi]); for (auto& g : gs) printf(" %d (%d tiles, %d workgroups)", g.splits, g.tiles, g.n_wg); printf("\n");
  }
  printf("WGRAD_LAB_CORRECT=%d\n", bad_total == 0);

  // ---- timing (warm clocks: 200 blocks first)
  const double flops = 2.0 * R * ((double)F * D * 2 + (double)Q * D + (double)D * D);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto fn_) {
    for (int it = 0; it < 100; ++it) fn_();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int n = 48;
    for (int it = 0; it < n; ++it) fn_();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / n * 1e3;
    printf("%-64s %8.1f us per block  %6.0f TFLOP/s\n", name, us, flops / us / 1e6);
  };
  time_it("tiled 128x128 split-R + reduce_partials (4 + 4 launches)", [&]() { for (int k = 0; k < 4; ++k) old_wgrad(probs[k], R, ws, st); });
  for (wa = 1; wa <= 2; ++wa)
  for (int sl0 : {512, 448, 384}) {
    const int sl = sl0 / wa;
    for (size_t gi = 0; gi < 2; ++gi) {
      std::vector<Group> gs;
      for (size_t j = 0; j < groupings[gi].size(); ++j) gs.push_back(make_group(groupings[gi][j], sl, d_map + j * 4096));
      pending = nullptr;
      char nm[160]; snprintf(nm, sizeof nm, "grouped wa %d, %s, %d slots (splits %d..)", wa, gname[gi], sl, gs[0].splits);
      time_it(nm, [&]() { for (auto& g : gs) launch_group(&g); });
      launch_group(nullptr);
      CK(hipStreamSynchronize(st));
    }
  }
#ifdef WG_LAB_TS
  {
    CK(hipMemset(d_ts, 0, 1024 * 8 * 8 * 8));
    wa = 2;
    std::vector<Group> gs; gs.push_back(make_group(groupings[0][0], 256, d_map));
    pending = nullptr;
    for (int it = 0; it < 50; ++it) launch_group(&gs[0]);
    launch_group(nullptr);
    CK(hipStreamSynchronize(st));
    std::vector<long long> h(256 * 8 * 8);
    CK(hipMemcpy(h.data(), d_ts, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0}; int n = 0;
    for (int w = 0; w < 256 * 8; ++w) if (h[w * 8 + 5]) { for (int k = 0; k < 8; ++k) acc[k] += h[w * 8 + k]; ++n; }
    const int stages = dig_wgrad_group_rows_per_split(R, gs[0].splits) / 16;
    printf("wide kernel, per wave and stage (%d waves, %d stages; s_memtime ticks = shader cycles): own LDS reads %.0f | own DMA %.0f | barrier %.0f | issue+MFMA %.0f | loop %.0f ; drain %.0f total\n",
           n, stages, acc[0] / n / stages, acc[1] / n / stages, acc[2] / n / stages, acc[3] / n / stages, acc[5] / n / stages, acc[4] / n);
    printf("  whole kernel per wave: %.0f ticks, of which before the loop (operand-stream start + fold of the previous launch) %.0f, loop %.0f, after (drain + slab stores) %.0f\n",
           acc[7] / n, acc[6] / n, acc[5] / n, (acc[7] - acc[6] - acc[5]) / n);
    printf("  wave lifetime %.1f us (s_memrealtime, 100 MHz) -> s_memtime runs at %.3f GHz\n", acc[4] / n / 100.0, acc[7] / (acc[4] * 10.0));
  }
#endif
  printf("status=%s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
