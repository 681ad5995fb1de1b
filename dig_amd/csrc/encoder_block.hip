// One C-ABI call per encoder block: dig_encoder_block_fwd / dig_encoder_block_bwd (include/dig_hip.h, tables in include/dig_block_types.h).
// Host code only -- the launch sequences themselves are in encoder_block.inc (shared with the host-only build under cpu_abi/); this file
// supplies the one thing the HIP build adds: the hand-over of the backward's parameter-gradient reductions to the side stream.
//
// Why it exists: a step of the ViT-S model is 36 encoder blocks (12 online + 12 momentum forward, 12 backward); issued entry point by entry
// point from Python a block costs 4 (forward) or 15 (backward) FFI crossings, a dozen output allocations and a stream switch.  With one
// crossing per block the host side of a step drops below the GPU time of its smallest configuration.
#include <hip/hip_runtime.h>
#include <mutex>
#include "dig_hip.h"                       // the library's own entry points: this file is a caller of the C ABI like any other

namespace {

// The side stream waits for everything the main stream has queued so far.  Events come from a small per-device ring: hipStreamWaitEvent
// captures the event's state at the time of the call, so a slot may be re-recorded as soon as its wait has been enqueued.  The entry point is
// callable from any host thread (the forward runs on the caller's thread, the backward on autograd's): slot creation and the record + wait pair
// of a call happen under the device's mutex, so two threads driving one device neither create a slot twice nor re-record a slot between
// another thread's record and its wait.
int handover(hipStream_t main, hipStream_t side) {
  if (main == side) return 0;
  constexpr int RING = 8, MAX_DEV = 64, ERR_ARG = -1, ERR_LAUNCH = -3;
  static hipEvent_t ring[MAX_DEV][RING] = {};
  static unsigned next[MAX_DEV] = {};
  static std::mutex lock[MAX_DEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ERR_LAUNCH;
  if (dev < 0 || dev >= MAX_DEV) return ERR_ARG;                 // (an ordinal beyond the table must not alias device 0's events)
  std::lock_guard<std::mutex> guard(lock[dev]);
  hipEvent_t& ev = ring[dev][next[dev]++ % RING];
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ERR_LAUNCH;
  if (hipEventRecord(ev, main) != hipSuccess) return ERR_LAUNCH;
  if (hipStreamWaitEvent(side, ev, 0) != hipSuccess) return ERR_LAUNCH;
  return 0;
}

}  // namespace

#define DIG_BLOCK_HANDOVER(main, side) handover(main, side)
#include "encoder_block.inc"
