// Where the fused attention sub-block (dig_amd/csrc/attn_block.hip) spends its time: the product kernel with compile-time ablations
// (-DDIG_AB_ABL=<bits>: 1 no attention MFMAs, 2 no tick MFMAs, 4 no HBM stores, 8 no softmax arithmetic, 16 no ring DMA traffic) and, with
// -DLAB_TS, per-wave time accounting by phase (s_memtime between the DIG_AB_TS hooks: 0 wait + barrier of a tick, 1 tick MFMAs, 2 block
// epilogue, 3 attention, 4 row loads), averaged over all waves of the grid.  256 images (one workgroup per CU), both forms.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=fast -w -I include -I dig_amd/csrc [-DDIG_AB_ABL=n] [-DLAB_TS]
//         tools/experiments/attn_block_lab.hip -o build/lab/attn_block_lab          (tools/experiments/run_attn_block_lab.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef LAB_TS
__device__ unsigned long long g_ts[8];
#define DIG_AB_TS_BEGIN() unsigned long long ts_acc[5] = {0, 0, 0, 0, 0}; unsigned long long ts_last = __builtin_amdgcn_s_memtime(); int ts_cur = 4; const unsigned long long ts_t0 = ts_last;
#define DIG_AB_TS(k) { const unsigned long long ts_now = __builtin_amdgcn_s_memtime(); ts_acc[ts_cur] += ts_now - ts_last; ts_last = ts_now; ts_cur = (k); }
#define DIG_AB_TS_END() { const unsigned long long ts_now = __builtin_amdgcn_s_memtime(); ts_acc[ts_cur] += ts_now - ts_last; \
  if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 5; ++i) atomicAdd(&g_ts[i], ts_acc[i]); atomicAdd(&g_ts[5], ts_now - ts_t0); } }
#endif
#include "../../dig_amd/csrc/attn_block.hip"
// the launch probe of the library (csrc/probe.hip) is not linked here
bool dig_probe_on() { return false; }
void dig_probe_events(hipEvent_t*, hipEvent_t*) {}

static void fill_bf16(unsigned short* d, size_t n, unsigned seed, float scale) {
  std::vector<unsigned short> h(n);
  srand(seed);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f * scale; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

int main() {
  const int n_img = 256, D = 384, H = 6, R = n_img * 256;
  unsigned short *ln1, *x, *wq, *wp, *qkv, *ctx, *xm;
  float *bq, *bp, *lse;
  hipMalloc(&ln1, (size_t)R * D * 2); hipMalloc(&x, (size_t)R * D * 2); hipMalloc(&wq, (size_t)3 * D * D * 2); hipMalloc(&wp, (size_t)D * D * 2);
  hipMalloc(&qkv, (size_t)R * 3 * D * 2); hipMalloc(&ctx, (size_t)R * D * 2); hipMalloc(&xm, (size_t)R * D * 2);
  hipMalloc(&bq, 3 * D * 4); hipMalloc(&bp, D * 4); hipMalloc(&lse, (size_t)n_img * H * 256 * 4);
  fill_bf16(ln1, (size_t)R * D, 1, 1.0f); fill_bf16(x, (size_t)R * D, 2, 1.0f); fill_bf16(wq, (size_t)3 * D * D, 3, 0.08f); fill_bf16(wp, (size_t)D * D, 4, 0.08f);
  hipMemset(bq, 0, 3 * D * 4); hipMemset(bp, 0, D * 4);
  const double flop = 2.0 * R * D * 3 * D + 4.0 * R * 256 * D + 2.0 * R * D * D;
  for (int save = 0; save < 2; ++save) {
    auto run = [&]() { return dig_attn_block_fwd(ln1, x, wq, bq, wp, bp, save ? qkv : nullptr, ctx, save ? lse : nullptr, xm, n_img, H, D, 0.125f, 0); };
    for (int i = 0; i < 20; ++i) if (run()) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
#ifdef LAB_TS
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_ts), z, sizeof z);
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 100;
    hipEventRecord(e0, 0);
    for (int i = 0; i < N; ++i) run();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("ABL %2d NSPLIT %d %s form: %7.1f us  %7.1f TFLOP/s", DIG_AB_ABL, DIG_AB_NSPLIT, save ? "online  " : "momentum", ms * 1e3 / N, flop / (ms * 1e-3 / N) * 1e-12);
#ifdef LAB_TS
    unsigned long long t[8];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ts), sizeof t);
    const double nw = (double)N * n_img * 8;
    printf("   wave life %8.0f ticks = wait+barrier %7.0f + tick %7.0f + epilogue %7.0f + attention %7.0f + row loads %6.0f", t[5] / nw, t[0] / nw, t[1] / nw,
           t[2] / nw, t[3] / nw, t[4] / nw);
#endif
    printf("\n");
  }
  return 0;
}
