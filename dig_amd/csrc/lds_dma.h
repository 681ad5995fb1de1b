// LDS-DMA issued from inline asm, counted waits and buffer resources: the pieces of the kernels that stream operands HBM / L2 -> LDS behind the
// compiler's back (csrc/wgrad.hip, csrc/attn_block.hip).  hipcc orders every LDS access it cannot analyse behind a VISIBLE LDS-DMA with
// s_waitcnt vmcnt(0); a buffer_load ... lds inside an asm statement is invisible to that bookkeeping, so the kernel places its own counted
// s_waitcnt vmcnt(N) + s_barrier where a ring slot is first read (vmcnt counts loads, stores and LDS-DMA of a wave in issue order).
#pragma once
#include "common.h"

namespace {

// M0 = LDS destination of the wave (lane l lands at +16 l), written in the same statement that reads it
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, dig_u32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ dig_u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  dig_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

}  // namespace
