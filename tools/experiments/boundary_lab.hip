// What does a kernel boundary cost on a stream of dependent launches, and does the flavour of the producer's stores change it?
// A streaming kernel (16-byte loads, 16-byte stores, MB bytes each way) is launched N times back to back on one stream; the wall time per
// launch is compared with the device-side duration of the same launch taken from hipExtLaunchKernel's start/stop events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/experiments/boundary_lab.hip -o build/lab/boundary_lab
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint4 v = in[i];
    v.x ^= 1u; v.y += 3u;
    if (MODE == 0) out[i] = v;
    else if (MODE == 1) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<u4*>(out + i)); }
    else {
      uint4* p = out + i;
      typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 w = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory");
    }
  }
}

template <int MODE>
static void run(const char* name, const uint4* in, uint4* a, uint4* b, size_t n, int N, hipStream_t st) {
  const int grid = 256 * 8;
  std::vector<hipEvent_t> ev(2 * N);
  for (auto& e : ev) CK(hipEventCreate(&e));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(grid), dim3(256), 0, st, in, a, n);
  CK(hipStreamSynchronize(st));
  // (1) plain back-to-back launches, each reading the previous one's output
  CK(hipEventRecord(t0, st));
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(grid), dim3(256), 0, st, (k & 1) ? (const uint4*)a : (const uint4*)b, (k & 1) ? b : a, n);
  CK(hipEventRecord(t1, st)); CK(hipStreamSynchronize(st));
  float wall; CK(hipEventElapsedTime(&wall, t0, t1));
  // (2) the same with start/stop events per launch
  for (int k = 0; k < N; ++k)
    hipExtLaunchKernelGGL(stream_kernel<MODE>, dim3(grid), dim3(256), 0, st, ev[2 * k], ev[2 * k + 1], 0, (k & 1) ? (const uint4*)a : (const uint4*)b, (k & 1) ? b : a, n);
  CK(hipStreamSynchronize(st));
  double dev = 0, span = 0;
  for (int k = 0; k < N; ++k) { float d; CK(hipEventElapsedTime(&d, ev[2 * k], ev[2 * k + 1])); dev += d; }
  { float d; CK(hipEventElapsedTime(&d, ev[0], ev[2 * N - 1])); span = d; }
  printf("%-28s %6.1f MB each way: wall/launch %7.2f us, device duration %7.2f us, span/launch with events %7.2f us -> boundary %5.2f us\n",
         name, n * 16 / 1e6, wall * 1e3 / N, dev * 1e3 / N, span * 1e3 / N, wall * 1e3 / N - dev * 1e3 / N);
}

// (b) what an event record / a satisfied event wait between two dependent launches costs the stream, and whether the any-order flag of
// hipExtLaunchKernel lets a small kernel run beside its predecessor on the same stream
template <int KIND>
static void between(const char* name, const uint4* in, uint4* a, uint4* b, uint4* c, size_t n, int N, hipStream_t st, hipStream_t other) {
  const int grid = 256 * 8;
  hipEvent_t t0, t1, e; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t done; CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  CK(hipEventRecord(done, other)); CK(hipStreamSynchronize(other));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(t0, st));
    for (int k = 0; k < N; ++k) {
      hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, st, (k & 1) ? (const uint4*)a : (const uint4*)b, (k & 1) ? b : a, n);
      if (KIND == 1) CK(hipEventRecord(e, st));
      if (KIND == 2) CK(hipStreamWaitEvent(st, done, 0));
      if (KIND == 3) { CK(hipEventRecord(e, st)); CK(hipStreamWaitEvent(other, e, 0)); }
      if (KIND == 4) hipLaunchKernelGGL(stream_kernel<0>, dim3(8), dim3(256), 0, st, in, c, (size_t)65536);
      if (KIND == 5) hipExtLaunchKernelGGL(stream_kernel<0>, dim3(8), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, in, c, (size_t)65536);
      if (KIND == 6) { CK(hipEventRecord(e, st)); CK(hipStreamWaitEvent(other, e, 0)); hipLaunchKernelGGL(stream_kernel<0>, dim3(8), dim3(256), 0, other, in, c, (size_t)65536); }
    }
    CK(hipEventRecord(t1, st)); CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(other));
  }
  float wall; CK(hipEventElapsedTime(&wall, t0, t1));
  printf("  %-64s wall per launch %7.2f us\n", name, wall * 1e3 / N);
}

int main(int argc, char** argv) {
  const int N = 40;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipStream_t other; CK(hipStreamCreate(&other));
  {
    const size_t n = 50 * 1000000 / 16;
    uint4 *in, *a, *b, *c;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16));
    CK(hipMemset(in, 1, n * 16)); CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 1, n * 16));
    printf("between two dependent 50 MB launches on one stream:\n");
    between<0>("nothing", in, a, b, c, n, N, st, other);
    between<1>("hipEventRecord", in, a, b, c, n, N, st, other);
    between<2>("hipStreamWaitEvent on an event that is already complete", in, a, b, c, n, N, st, other);
    between<3>("hipEventRecord + the other stream waits for it", in, a, b, c, n, N, st, other);
    between<4>("a small kernel (1 MB, 8 workgroups), in order", in, a, b, c, n, N, st, other);
    between<5>("the same small kernel with hipExtAnyOrderLaunch", in, a, b, c, n, N, st, other);
    between<6>("the same small kernel on the other stream after an event", in, a, b, c, n, N, st, other);
    between<0>("nothing", in, a, b, c, n, N, st, other);
    CK(hipFree(in)); CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c));
  }
  for (size_t mb : {1, 8, 50, 150}) {
    const size_t n = mb * 1000000 / 16;
    uint4 *in, *a, *b;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(in, 1, n * 16)); CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 1, n * 16));
    run<0>("plain stores", in, a, b, n, N, st);
    run<1>("nontemporal stores", in, a, b, n, N, st);
    run<2>("sc0 sc1 (write-through)", in, a, b, n, N, st);
    CK(hipFree(in)); CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
