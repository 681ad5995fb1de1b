// Probe: how fast can 6144 workgroups of 256 threads write a [65536 x 1536] bf16 matrix as 128x128 tiles,
// as a function of per-lane pattern and LDS footprint (occupancy)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PATTERN>
__global__ __launch_bounds__(256) void k_store(unsigned short* C, int ldc, int tiles_j) {
  extern __shared__ unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
  const int wi = wave >> 1, wj = wave & 1;
  if (smem[0] == 77 && tid == 999) C[0] = 1;   // keep smem alive
  uint4 v = make_uint4(tid, ti, tj, 7);
  if (PATTERN == 0) {   // C-shuffle pattern: 8 lanes x 16 B per row, 8 rows per instruction, 8 passes
    const int cg = lane & 7;
    for (int ps = 0; ps < 8; ++ps) {
      const int row = ti * 128 + wi * 64 + ps * 8 + (lane >> 3);
      const int col = tj * 128 + wj * 64 + cg * 8;
      *reinterpret_cast<uint4*>(C + (size_t)row * ldc + col) = v;
    }
  } else if (PATTERN == 1) {  // whole-WG rows: 16 lanes x 16 B = 256 B per row, 16 rows per pass (4 waves x 4 rows), 8 passes
    for (int ps = 0; ps < 8; ++ps) {
      const int row = ti * 128 + ps * 16 + (tid >> 4);
      const int col = tj * 128 + (tid & 15) * 8;
      *reinterpret_cast<uint4*>(C + (size_t)row * ldc + col) = v;
    }
  }
}

__global__ void k_fill(uint4* C, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) C[i] = make_uint4(1, 2, 3, 4);
}

int main() {
  const int M = 65536, N = 1536;
  unsigned short* C; hipMalloc(&C, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto fn, const char* name) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) fn();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
  };
  const int tiles = (M / 128) * (N / 128);
  for (int lds : {0, 32768, 65536}) {
    hipFuncSetAttribute((const void*)k_store<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k_store<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    char nm[64];
    snprintf(nm, 64, "cshuffle pattern, lds %d", lds);
    timeit([&] { hipLaunchKernelGGL(k_store<0>, dim3(tiles), dim3(256), lds, 0, C, N, N / 128); }, nm);
    snprintf(nm, 64, "wg-row pattern, lds %d", lds);
    timeit([&] { hipLaunchKernelGGL(k_store<1>, dim3(tiles), dim3(256), lds, 0, C, N, N / 128); }, nm);
  }
  timeit([&] { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint4*)C, (size_t)M * N * 2 / 16); }, "linear fill");
  return 0;
}
