// How fast can one CU pull operand tiles into LDS with buffer_load_dwordx4 ... lds?  The GEMM K loops of this model move
// 52-55 GB/s per CU; is that the path's ceiling or the kernels' doing?  One 1024-thread workgroup per CU streams 64 KiB
// "stages" (the 256x256xBK64 operand pair of gemm_wide_kernel, same row-swizzled source pattern) from an L2-resident window
// with 1 to 3 stages (32 KiB each) in flight, no MFMA, no LDS reads; optionally every wave also stores 16 B per lane per stage (the
// epilogue's share of the same vector-memory path).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/experiments/dma_rate_lab.hip -o build/lab/dma_rate_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int DEPTH, int ROW_BYTES, bool STORES>
__global__ __launch_bounds__(1024, 1) void dma_kernel(const unsigned char* src, unsigned bytes, unsigned char* dst, int stages, int window_stages) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // a stage = 512 rows x 128 B taken from rows of ROW_BYTES (768 = K 384 bf16): piece = 16 B, 8 pieces per row, chunk XOR-swizzled
  unsigned off[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int piece = it * 1024 + tid, row = piece >> 3, pc = piece & 7;
    off[it] = (unsigned)((blockIdx.x * 256 + row) * ROW_BYTES + ((pc ^ ((row >> 1) & 7)) << 4));
  }
  auto issue = [&](int slot, int s) {
    const unsigned koff = (unsigned)((s % window_stages) * 128);          // walk along K inside the row, wrap inside the window
    unsigned char* a = smem + slot * 32768 + wave * 1024;
#pragma unroll
    for (int it = 0; it < 2; ++it)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(a + it * 16384), 16, off[it] + koff, 0, 0, 0);
  };
#pragma unroll
  for (int q = 0; q < DEPTH - 1; ++q) issue(q, q);
  for (int s = 0; s < stages; ++s) {
    // in flight behind the stage being waited for: (DEPTH - 2) younger stages of 2 DMA instructions (+ their stores)
    constexpr int YOUNGER = (DEPTH - 2) * (2 + (STORES ? 1 : 0));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    __builtin_amdgcn_s_barrier();
    if (s + DEPTH - 1 < stages) issue((s + DEPTH - 1) % DEPTH, s + DEPTH - 1);
    if (STORES) {
      const uint4 v = make_uint4(s, tid, 0, 0);
      *reinterpret_cast<uint4*>(dst + ((size_t)blockIdx.x * stages + s) * 16384 + tid * 16) = v;
    }
  }
}

template <int DEPTH, int ROW_BYTES, bool STORES>
void run(const char* name, const unsigned char* src, unsigned bytes, unsigned char* dst, int window_stages) {
  const int lds = DEPTH * 32768, stages = 192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<DEPTH, ROW_BYTES, STORES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((dma_kernel<DEPTH, ROW_BYTES, STORES>), dim3(256), dim3(1024), lds, 0, src, bytes, dst, stages, window_stages);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int n = 50;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((dma_kernel<DEPTH, ROW_BYTES, STORES>), dim3(256), dim3(1024), lds, 0, src, bytes, dst, stages, window_stages);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / n * 1e3;
  const double gb = 256.0 * stages * 32768 / 1e9;
  printf("%-64s %8.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", name, us, gb / us * 1e6 / 256, gb / us * 1e6 / 1e3);
}

int main() {
  const size_t bytes = (size_t)256 * 512 * 768;                          // 100 MB: 131072 rows of 768 B
  unsigned char *src, *dst;
  hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 1, bytes + (1 << 20));
  hipMalloc(&dst, (size_t)256 * 192 * 16384);
  // stage = 32 KiB (256 rows x 128 B).  Window of 6 stages = the whole 768-byte row (K = 384): a stage re-reads rows the CU touched
  // 6 stages ago (L2 / MALL hits); window 1 = the same 32 KiB again and again
  run<2, 768, false>("2 slots (1 stage = 32 KiB in flight), 6-stage window", src, (unsigned)bytes, dst, 6);
  run<3, 768, false>("3 slots (64 KiB in flight), 6-stage window", src, (unsigned)bytes, dst, 6);
  run<4, 768, false>("4 slots (96 KiB in flight), 6-stage window", src, (unsigned)bytes, dst, 6);
  run<2, 768, false>("2 slots, 1-stage window", src, (unsigned)bytes, dst, 1);
  run<4, 768, false>("4 slots, 1-stage window", src, (unsigned)bytes, dst, 1);
  run<3, 768, true>("3 slots, 6-stage window + 16 KiB of stores per stage", src, (unsigned)bytes, dst, 6);
  run<4, 768, true>("4 slots, 6-stage window + 16 KiB of stores per stage", src, (unsigned)bytes, dst, 6);
  return 0;
}
