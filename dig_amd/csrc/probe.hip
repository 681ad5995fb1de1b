// Launch probe: per-launch kernel durations of the matrix-core entry points (dig_gemm_bf16, dig_mlp_chain_*, dig_wgrad_group), taken
// with the start / stop events of hipExtLaunchKernel -- the kernel's own begin and end on the device, i.e. what rocprofv3's kernel
// trace reports -- so that bench.py can measure, live and inside the step with both streams running, how long the dominant kernel
// family really executes (an event recorded on the stream in front of a launch would also count the time the launch waits for CUs the
// other stream's kernel still holds).  Off by default: the launchers then go through hipLaunchKernelGGL as before.
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>

namespace {
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_ev;
}  // namespace

bool dig_probe_on() { return g_on.load(std::memory_order_relaxed) != 0; }
void dig_probe_events(hipEvent_t* e0, hipEvent_t* e1) {
  (void)hipEventCreate(e0);
  (void)hipEventCreate(e1);
  std::lock_guard<std::mutex> lk(g_mu);
  g_ev.emplace_back(*e0, *e1);
}

// C-ABI: see include/dig_hip.h
extern "C" int dig_probe_start() {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& p : g_ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  g_ev.clear();
  g_on.store(1);
  return DIG_OK;
}
extern "C" int dig_probe_stop(float* us_out, int max_n) {
  g_on.store(0);
  if (hipDeviceSynchronize() != hipSuccess) return DIG_ERR_LAUNCH;
  std::lock_guard<std::mutex> lk(g_mu);
  const int n = (int)g_ev.size();
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    if (us_out && i < max_n && hipEventElapsedTime(&ms, g_ev[i].first, g_ev[i].second) == hipSuccess) us_out[i] = ms * 1e3f;
    (void)hipEventDestroy(g_ev[i].first);
    (void)hipEventDestroy(g_ev[i].second);
  }
  g_ev.clear();
  return n;
}
