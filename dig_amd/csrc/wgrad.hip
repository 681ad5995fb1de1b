// Grouped weight-gradient GEMM for gfx950 (MI355X):  out_p[i, j] += sum_r A_p[r, i] * B_p[r, j]  for a list of problems p.
//
// Reference math: the autograd of nn.Linear inside a transformer block (modeling_finetune.py:53-60 Mlp, :91-93,119 Attention):
// dW = dy^T x for the four Linear layers of a Block, i.e. the contraction runs over the TOKEN rows (R = 2 B 256 = 65 536 at the
// BASELINE batch) and the outputs are small ([1536, 384], [384, 1536], [1152, 384], [384, 384] for ViT-S).  Both operands are
// stored with the reduction index as the row index ("transposed storage" of csrc/gemm.hip).
//
// Why a second kernel beside gemm_kernel<true, true, 2, ...>: that path runs 128x128 tiles (64x64 per wave: one LDS fragment read
// per MFMA, 0.0156 operand bytes from L2 per FLOP), a 2-stage ring that the compiler drains in front of every transpose read, 16
// R-splits x 4 launches per block each followed by its own slab-sum launch (59 reduce_partials launches per step).  Here:
//   * tile = 128 (wide operand) x 128 FN (narrow operand: 384 for ViT-S, FN = 3; 256 for D = 512, FN = 2), 4 waves as 1 x 4, a wave
//     owns 128 x 32 FN = 4 x FN MFMA accumulators (192 registers at FN = 3): (4 + FN) fragment reads per 4 FN MFMAs = 0.58 per MFMA,
//     0.0104 operand bytes per FLOP; two workgroups per CU (one wave of each per SIMD) cover each other's barrier / issue stalls;
//   * operands HBM/L2 -> LDS by buffer_load_dwordx4 ... lds issued from INLINE ASM (the compiler must not see them: it orders every
//     transpose read it cannot analyse behind a visible LDS-DMA with s_waitcnt vmcnt(0)), ring of 4 stages x 16 token rows, three
//     stages in flight, counted s_waitcnt vmcnt + raw s_barrier; the transpose reads stay compiler-visible builtins, so hipcc
//     places its own lgkmcnt ladders between them and the MFMAs;
//   * ONE launch serves several problems (all four weight gradients of a block, or any subset): tiles of all problems x R-splits
//     = one round of workgroups (<= 2 per CU), so a block dumps its accumulators once (~100 MB of fp32 slabs) instead of four times;
//   * the slabs are raw accumulator dumps (48 coalesced 16-byte stores per lane, no C-shuffle) and they are summed -- in split
//     order, deterministic -- by the NEXT launch of this kernel in its prologue (the kernel boundary is the release / acquire;
//     no counters, no spinning), which also performs the += into the fp32 gradient arena, transposed where the problem's output is
//     [narrow, wide] (fc2).  A launch with no problems folds the last pending set (dig_wgrad_group with n_probs = 0).
// Workgroup -> (tile, split) comes from a host-built table (dig_wgrad_group_plan): all tiles of one (problem, split) pair sit on
// ONE XCD (block b runs on XCD b % 8), so the narrow operand's row window is fetched from HBM once and re-read from that XCD's L2.
#include "common.h"
#include "lds_dma.h"
#include <type_traits>
#include <vector>

// phase time stamps for tools/experiments/wgrad_lab.hip (empty in the product build)
#ifndef DIG_WG_TS
#define DIG_WG_TS_BEGIN()
#define DIG_WG_TS_DECL()
#define DIG_WG_TS(k)
#define DIG_WG_TS_END()
#endif

#define DIG_WGRAD_MAX_PROBS 6
struct dig_wgrad_prob_t {
  const void* A;      // [R, lda] bf16: the wide operand (I columns, I % 128 == 0)
  const void* B;      // [R, ldb] bf16: the narrow operand (J columns, J % (128 FN) == 0)
  float* out;         // fp32 gradient: [I, ldo] (trans_out 0) or [J, ldo] (trans_out 1), accumulated into
  int lda, ldb, ldo;
  int I, J;
  int trans_out;
};

namespace {

constexpr int WG_BK = 16;                 // token rows per ring stage
constexpr int WG_NSTG = 4;
constexpr unsigned WG_NONE = 0xffffffffu;

struct WgProb {
  const bf16_t* A; const bf16_t* B; float* out;
  int lda, ldb, ldo, I, J, trans_out, tile0;
};
struct WgParams {
  WgProb prob[DIG_WGRAD_MAX_PROBS];
  WgProb fold[DIG_WGRAD_MAX_PROBS];
  int n_prob, n_fold;
  int R, r_per_split, splits;
  int fold_splits, fold_tiles;
  float* slabs;
  const float* fold_slabs;
  const unsigned* wg_map;
};

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_p;
__device__ __forceinline__ bf16x8 tr_frag(unsigned base, int off_lo, int off_hi) {
  // 8 consecutive token rows of one column for this lane: two [4 r][16 c] blocks
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_p)(uintptr_t)(base + off_lo));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_p)(uintptr_t)(base + off_hi));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Fold of the previous launch's slabs into the gradient arena (fixed split order: deterministic), spread over all waves of the grid.
// A unit = one accumulator quad of one wave of one tile (64 lanes x 16 bytes per split); a wave takes four units at a time and
// requests four splits of each before it sums them: 16 independent 16-byte loads per lane in flight.  WA = 1: tiles of 128 x 128 FN
// written by 4 waves; WA = 2: tiles of 256 x 128 FN written by 8 waves (rows beyond the problem's I are never folded).
template <int FN, int WA>
__device__ __forceinline__ void fold_prev(const WgParams& p, int wave, int lane) {
  constexpr int TJ = 128 * FN, TI = 128 * WA, NWV = 4 * WA;
  constexpr int NQ = 16 * FN;
  constexpr int SLAB = NWV * NQ * 256;
  const int hi = lane >> 5;
  const int gw = blockIdx.x * NWV + wave, nw = gridDim.x * NWV;
  const int units = p.n_fold > 0 ? p.fold_tiles * (NWV * NQ) : 0;
  const int S = p.fold_splits;
  for (int base = gw; base < units; base += 4 * nw) {
    const float* src[4];
    f32x4 sum[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int un = min(base + k * nw, units - 1);
      const int T = un / (NWV * NQ), rem = un - T * (NWV * NQ);
      src[k] = p.fold_slabs + (size_t)T * S * SLAB + rem * 256 + lane * 4;
      sum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int s0 = 0; s0 < S; s0 += 4) {
      f32x4 v[4][4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const size_t so = (size_t)min(s0 + d, S - 1) * SLAB;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k][d] = *reinterpret_cast<const f32x4*>(src[k] + so);
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (s0 + d < S) {
#pragma unroll
          for (int k = 0; k < 4; ++k) sum[k] += v[k][d];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int un = base + k * nw;
      if (un >= units) break;
      const int T = un / (NWV * NQ), rem = un - T * (NWV * NQ);
      const int w = rem / NQ, q = rem - w * NQ;
      int fi = 0;
#pragma unroll
      for (int kk = 1; kk < DIG_WGRAD_MAX_PROBS; ++kk)
        if (kk < p.n_fold && T >= p.fold[kk].tile0) fi = kk;
      const WgProb Fp = p.fold[fi];
      const int flt = T - Fp.tile0, ftj_n = Fp.J / TJ;
      const int fti = flt / ftj_n, ftj = flt - fti * ftj_n;
      const int u = q / (4 * FN), v = (q >> 2) % FN, g = q & 3;
      const int i = fti * TI + 128 * (w >> 2) + 32 * u + (lane & 31);
      const int j = ftj * TJ + (w & 3) * (32 * FN) + 32 * v + 8 * g + 4 * hi;
      if (i >= Fp.I) continue;
      if (!Fp.trans_out) {
        f32x4* o = reinterpret_cast<f32x4*>(Fp.out + (size_t)i * Fp.ldo + j);
        *o = *o + sum[k];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) Fp.out[(size_t)(j + e) * Fp.ldo + i] += sum[k][e];
      }
    }
  }
}

template <int FN>
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(WgParams p) {
  constexpr int TJ = 128 * FN;
  constexpr int NBA = 8, NBB = TJ / 16;                          // 16-column blocks per block row of the two operand tiles
  constexpr int A_BYTES = 4 * NBA * 128;                         // [16 r][128 c] as 4 block rows of [4 r][16 c] blocks
  constexpr int B_BYTES = 4 * NBB * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;                       // 16 KiB (FN 3), 12 KiB (FN 2)
  constexpr int NP = 1 + FN;                                     // LDS-DMA instructions per wave per stage
  constexpr int NQ = 16 * FN;                                    // accumulator quads per lane
  constexpr int SLAB = 4 * NQ * 256;                             // floats per (tile, split) slab = 128 x TJ
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;

  // ---- this workgroup's item
  const unsigned item = __builtin_amdgcn_readfirstlane(p.n_prob > 0 ? p.wg_map[blockIdx.x] : WG_NONE);
  const bool has_work = item != WG_NONE;
  const int tile = (int)(item & 0xffffu), split = (int)(item >> 16);
  int pi = 0;
#pragma unroll
  for (int k = 1; k < DIG_WGRAD_MAX_PROBS; ++k)
    if (k < p.n_prob && tile >= p.prob[k].tile0) pi = k;
  const WgProb P = p.prob[pi];
  const int lt = tile - P.tile0;
  const int tiles_j = P.J / TJ;
  const int ti = lt / tiles_j, tj = lt - ti * tiles_j;
  const int i0 = ti * 128, j0 = tj * TJ;
  const int rbeg = split * p.r_per_split;
  const int rend = min(p.R, rbeg + p.r_per_split);
  const int nt = has_work ? (rend - rbeg) / WG_BK : 0;           // host-checked: a multiple of 4, >= 4

  // ---- operand stream: wave w brings token rows 4w .. 4w+3 of every stage (one 1-KiB piece of A, FN pieces of B)
  const dig_u32x4 rA = make_rsrc(P.A, (unsigned)((size_t)p.R * P.lda * 2));
  const dig_u32x4 rB = make_rsrc(P.B, (unsigned)((size_t)p.R * P.ldb * 2));
  const int drr = (lane & 7) >> 1, dhalf = lane & 1, dnb = lane >> 3;
  const unsigned voffA = (unsigned)(((4 * wave + drr) * P.lda + i0 + dnb * 16 + dhalf * 8) * 2);
  unsigned voffB[FN];
#pragma unroll
  for (int g = 0; g < FN; ++g) voffB[g] = (unsigned)(((4 * wave + drr) * P.ldb + j0 + g * 128 + dnb * 16 + dhalf * 8) * 2);
  const unsigned stepA = (unsigned)(WG_BK * P.lda * 2), stepB = (unsigned)(WG_BK * P.ldb * 2);
  unsigned soffA = (unsigned)rbeg * (unsigned)(P.lda * 2), soffB = (unsigned)rbeg * (unsigned)(P.ldb * 2);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  auto issue_piece = [&](auto slot_tag, int k) {                  // piece 0: A, pieces 1..FN: B
    constexpr int SLOT = decltype(slot_tag)::value;
    if (k == 0) { dma16(lds0 + SLOT * STAGE + wave * 1024, voffA, rA, soffA); soffA += stepA; }
    else { dma16(lds0 + SLOT * STAGE + A_BYTES + (wave * NBB + (k - 1) * 8) * 128, voffB[k - 1], rB, soffB); if (k == FN) soffB += stepB; }
  };
  auto issue = [&](auto slot_tag) {
#pragma unroll
    for (int k = 0; k < NP; ++k) issue_piece(slot_tag, k);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  if (has_work) { issue(S0{}); issue(S1{}); issue(S2{}); issue(S3{}); }

  fold_prev<FN, 1>(p, wave, lane);
  if (!has_work) return;

  // ---- main loop
  f32x16 acc[4][FN];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < FN; ++v)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[u][v][e] = 0.f;
  // fragment addresses inside a stage: A frag u = columns 32u.. of the tile, B frag v = columns 32 (FN w + v)..
  const unsigned fa = lds0 + (unsigned)(((hi * 2) * NBA + ((lane >> 4) & 1)) * 128 + (lane & 15) * 8);
  const unsigned fb = lds0 + (unsigned)(A_BYTES + ((hi * 2) * NBB + wave * 2 * FN + ((lane >> 4) & 1)) * 128 + (lane & 15) * 8);
  bf16x8 af[4], bfr[2][FN];
  auto read_a = [&](auto slot_tag, int u) {
    constexpr int SLOT = decltype(slot_tag)::value;
    af[u] = tr_frag(fa, SLOT * STAGE + u * 256, SLOT * STAGE + u * 256 + NBA * 128);
  };
  auto read_b = [&](auto slot_tag, auto buf_tag) {
    constexpr int SLOT = decltype(slot_tag)::value, BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int v = 0; v < FN; ++v) bfr[BUF][v] = tr_frag(fb, SLOT * STAGE + v * 256, SLOT * STAGE + v * 256 + NBB * 128);
  };
  // One stage = 16 token rows: 4 FN MFMAs per wave.  On entry the fragments of stage t are in registers (requested during stage
  // t - 1); the barrier publishes stage t + 1 (every wave has waited for its own pieces) and retires stage t's slot (every wave's
  // fragment reads of it have been consumed by an MFMA or waited for by the compiler before the barrier is reached: they were
  // issued a whole stage earlier and the first MFMA group below uses all B fragments and the first A fragment... (see NOTE)).
  auto stage = [&](auto slot_tag, auto vm_tag, auto issue_tag, auto next_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool ISSUE = decltype(issue_tag)::value, NEXT = decltype(next_tag)::value;
    using NS = std::integral_constant<int, (SLOT + 1) & 3>;
    using CB = std::integral_constant<int, SLOT & 1>;
    using NB = std::integral_constant<int, (SLOT + 1) & 1>;
    // NOTE: all fragment reads of stage t must have LANDED before the barrier that frees its slot for the DMA of stage t + 4
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NEXT) wg_wait_vm<decltype(vm_tag)::value>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (NEXT) read_b(NS{}, NB{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                  // (the LDS-DMA pieces go out between the MFMA groups: see wgrad_wide_kernel)
#pragma unroll
      for (int v = 0; v < FN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[CB::value][v], af[u], acc[u][v], 0, 0, 0);
      if (NEXT) read_a(NS{}, u);
      if (ISSUE && u < NP) issue_piece(slot_tag, u);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using V2 = std::integral_constant<int, 2 * NP>;
  using V1 = std::integral_constant<int, NP>;
  using V0 = std::integral_constant<int, 0>;
  using TRUE_ = std::true_type;
  using FALSE_ = std::false_type;
  // stage 0's fragments
  wg_wait_vm<3 * NP>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_b(S0{}, S0{});
#pragma unroll
  for (int u = 0; u < 4; ++u) read_a(S0{}, u);
  for (int t = 0; t + 4 < nt; t += 4) {
    stage(S0{}, V2{}, TRUE_{}, TRUE_{});
    stage(S1{}, V2{}, TRUE_{}, TRUE_{});
    stage(S2{}, V2{}, TRUE_{}, TRUE_{});
    stage(S3{}, V2{}, TRUE_{}, TRUE_{});
  }
  stage(S0{}, V2{}, FALSE_{}, TRUE_{});                          // last four stages: nothing left to request
  stage(S1{}, V1{}, FALSE_{}, TRUE_{});
  stage(S2{}, V0{}, FALSE_{}, TRUE_{});
  stage(S3{}, V0{}, FALSE_{}, FALSE_{});

  // ---- slab = the accumulators as they are: 16-byte stores, 1 KiB per wave instruction
  float* slab = p.slabs + ((size_t)tile * p.splits + split) * SLAB + (size_t)wave * NQ * 256 + lane * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < FN; ++v)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {acc[u][v][4 * g], acc[u][v][4 * g + 1], acc[u][v][4 * g + 2], acc[u][v][4 * g + 3]};
        *reinterpret_cast<f32x4*>(slab + ((u * FN + v) * 4 + g) * 256) = o;
      }
}

// LDS-DMA of the first 32 lanes only (a 512-byte piece): exec is narrowed inside the statement
__device__ __forceinline__ void dma16_half(unsigned lds_dst, unsigned voff, dig_u32x4 rsrc, unsigned soff) {
  unsigned long long keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_hi, 0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// ----------------------------------------------------------------------------------------------------------------
// Wide form: tile = 256 (wide operand) x 128 FN, 8 waves as 2 x 4 (the same 128 x 32 FN block per wave), ONE workgroup per CU.  Per
// FLOP it moves 0.0065 operand bytes from L2 into LDS (128-row tiles: 0.0104) -- the rate that bounds the 4-wave form (2.4 GB per ViT-S
// block at ~9.4 TB/s) -- and its ring is NSTG slots of 20 KiB addressed at run time (slot = stage mod NSTG as a scalar, fragment
// addresses = base + slot offset), so the depth is not tied to an unroll factor: NSTG - 1 stages = 120 KiB in flight at NSTG = 7.
// Operand stream: a block row (4 token rows) of a stage = 2 KiB of A + 1 KiB FN of B, brought by a PAIR of waves in 1-KiB pieces;
// where the pair's share is 2.5 KiB (FN = 3) each wave's last piece is half a wave wide (dma16_half).  The stream never stops: stages
// past the split's end are requested as well (rows of the next split, or zeros past R) and never read, which keeps every s_waitcnt vmcnt
// count a compile-time constant.  The wide dimension need not be a multiple of 256: the last tile's surplus columns multiply
// whatever the rows hold beyond I (finite activations; zeros past the buffer) and are dropped by the fold.
template <int FN, int NSTG>
__global__ __launch_bounds__(512, 2) void wgrad_wide_kernel(WgParams p) {
  constexpr int TJ = 128 * FN;
  constexpr int NBA = 16, NBB = TJ / 16;
  constexpr int A_BYTES = 4 * NBA * 128;                         // 8 KiB
  constexpr int B_BYTES = 4 * NBB * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;                       // 20 KiB (FN 3), 16 KiB (FN 2)
  constexpr int ROWB = (NBA + NBB) * 128;                        // bytes of one block row (A part, then B part)
  constexpr int SHARE = ROWB / 2;                                // per wave of the pair
  constexpr int NPW = (SHARE + 1023) / 1024;                     // LDS-DMA instructions per wave per stage
  constexpr bool HALF = (SHARE % 1024) != 0;                     // the last piece of every wave is 512 bytes
  static_assert(SHARE % 512 == 0 && (NBA * 128) % 1024 == 0 && (!HALF || ((NBA * 128) % SHARE) % 1024 == 0 || true), "piece layout");
  constexpr int NQ = 16 * FN;
  constexpr int SLAB = 8 * NQ * 256;
  DIG_WG_TS_BEGIN()
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int wa = wave >> 2, wb = wave & 3;

  const unsigned item = __builtin_amdgcn_readfirstlane(p.n_prob > 0 ? p.wg_map[blockIdx.x] : WG_NONE);
  const bool has_work = item != WG_NONE;
  const int tile = (int)(item & 0xffffu), split = (int)(item >> 16);
  int pi = 0;
#pragma unroll
  for (int k = 1; k < DIG_WGRAD_MAX_PROBS; ++k)
    if (k < p.n_prob && tile >= p.prob[k].tile0) pi = k;
  const WgProb P = p.prob[pi];
  const int lt = tile - P.tile0;
  const int tiles_j = P.J / TJ;
  const int ti = lt / tiles_j, tj = lt - ti * tiles_j;
  const int i0 = ti * 256, j0 = tj * TJ;
  const int rbeg = split * p.r_per_split;
  const int rend = min(p.R, rbeg + p.r_per_split);
  const int nt = has_work ? (rend - rbeg) / WG_BK : 0;           // host-checked: a multiple of 4, >= 4

  // ---- operand stream.  Wave pair rb = wave >> 1 owns block row rb (token rows 4 rb .. 4 rb + 3); the pair's bytes in stage order
  // [A blocks 0..15 | B blocks 0..NBB-1]; wave `sub` takes bytes [sub SHARE, (sub + 1) SHARE) in 1-KiB pieces.
  const dig_u32x4 rA = make_rsrc(P.A, (unsigned)((size_t)p.R * P.lda * 2));
  const dig_u32x4 rB = make_rsrc(P.B, (unsigned)((size_t)p.R * P.ldb * 2));
  const int rb = wave >> 1, sub = wave & 1;
  const int drr = (lane & 7) >> 1, dhalf = lane & 1, dnb = lane >> 3;
  unsigned pv[NPW], pdst[NPW], psoff[NPW], pstep[NPW];
  dig_u32x4 prs[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int vb = sub * SHARE + k * 1024;                       // first byte of the piece within the block row (wave-uniform)
    const bool isB = vb >= NBA * 128;
    const int blk0 = (isB ? vb - NBA * 128 : vb) / 128;           // first 16-column block of the piece within its operand's block row
    const int ld = isB ? P.ldb : P.lda;
    const int col0 = (isB ? j0 : i0) + (blk0 + dnb) * 16 + dhalf * 8;
    pv[k] = (unsigned)(((4 * rb + drr) * ld + col0) * 2);
    pdst[k] = (unsigned)((isB ? A_BYTES + rb * NBB * 128 : rb * NBA * 128) + blk0 * 128);
    psoff[k] = (unsigned)rbeg * (unsigned)(ld * 2);
    pstep[k] = (unsigned)(WG_BK * ld * 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) prs[k][e] = isB ? rB[e] : rA[e];
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  auto issue_piece = [&](unsigned slot_off, int k) {
    if (HALF && k == NPW - 1) dma16_half(lds0 + slot_off + pdst[k], pv[k], prs[k], psoff[k]);
    else dma16(lds0 + slot_off + pdst[k], pv[k], prs[k], psoff[k]);
    psoff[k] += pstep[k];
  };
  auto issue = [&](unsigned slot_off) {
#pragma unroll
    for (int k = 0; k < NPW; ++k) issue_piece(slot_off, k);
  };
  if (has_work) {
#pragma unroll
    for (int q = 0; q < NSTG - 1; ++q) issue((unsigned)(q * STAGE));
  }

  fold_prev<FN, 2>(p, wave, lane);
  if (!has_work) return;

  f32x16 acc[4][FN];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < FN; ++v)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[u][v][e] = 0.f;
  const unsigned fa = lds0 + (unsigned)(((hi * 2) * NBA + wa * 8 + ((lane >> 4) & 1)) * 128 + (lane & 15) * 8);
  const unsigned fb = lds0 + (unsigned)(A_BYTES + ((hi * 2) * NBB + wb * 2 * FN + ((lane >> 4) & 1)) * 128 + (lane & 15) * 8);
  bf16x8 af[4], bfr[2][FN];
  auto read_a = [&](unsigned so, int u) { af[u] = tr_frag(fa + so, u * 256, u * 256 + NBA * 128); };
  auto read_b = [&](unsigned so, auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int v = 0; v < FN; ++v) bfr[BUF][v] = tr_frag(fb + so, v * 256, v * 256 + NBB * 128);
  };
  // slot offsets (scalars): `sp` = the slot of the stage multiplied LAST (free: refilled during this stage), `sn` = the slot of the next stage
  unsigned sp = (unsigned)((NSTG - 1) * STAGE), sn = STAGE;
  DIG_WG_TS_DECL()
  // One barrier per TWO stages (24 MFMAs per wave).  Stage t multiplies fragments that are in registers, requests the fragments of stage
  // t + 1 from LDS and refills the slot of stage t - 1 with stage t - 1 + NSTG.  At the barrier of an even stage t every wave has waited
  // for its own pieces of stages <= t + 2 and for its own fragment reads of stage t: behind it stages t + 1 and t + 2 are complete in
  // LDS and the slots of stages t - 1 and t are free -- which is all that stage t and the barrier-less stage t + 1 need.
  auto stage = [&](auto cb_tag, auto sync_tag, auto next_tag) {
    constexpr int CB = decltype(cb_tag)::value;
    constexpr bool SYNC = decltype(sync_tag)::value, NEXT = decltype(next_tag)::value;
    using NB = std::integral_constant<int, CB ^ 1>;
    DIG_WG_TS(3)
    if (SYNC) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's fragment reads of stage t have landed
      DIG_WG_TS(0)
      wg_wait_vm<(NSTG - 4) * NPW>();                             // this wave's pieces of stages t + 1, t + 2 have landed
      DIG_WG_TS(1)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      DIG_WG_TS(2)
    }
    // The MFMAs start right away (their fragments are in registers); the LDS-DMA pieces go out BETWEEN the MFMA groups: an LDS-DMA costs
    // its wave 60-100 issue cycles, during which the other wave of the SIMD keeps the matrix pipe busy
    if (NEXT) read_b(sn, NB{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int v = 0; v < FN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[CB][v], af[u], acc[u][v], 0, 0, 0);
      if (NEXT) read_a(sn, u);
      if (u < NPW) issue_piece(sp, u);
      __builtin_amdgcn_sched_barrier(0);
    }
    sp = (sp == (unsigned)((NSTG - 1) * STAGE)) ? 0u : sp + STAGE;
    sn = (sn == (unsigned)((NSTG - 1) * STAGE)) ? 0u : sn + STAGE;
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  DIG_WG_TS(4)
  wg_wait_vm<(NSTG - 2) * NPW>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_b(0u, B0{});
#pragma unroll
  for (int u = 0; u < 4; ++u) read_a(0u, u);
  for (int t = 0; t + 2 < nt; t += 2) {
    stage(B0{}, std::true_type{}, std::true_type{});
    stage(B1{}, std::false_type{}, std::true_type{});
  }
  stage(B0{}, std::true_type{}, std::true_type{});
  stage(B1{}, std::false_type{}, std::false_type{});
  DIG_WG_TS(3)
  wg_wait_vm<0>();                                                // the run-ahead pieces still on their way into LDS
  DIG_WG_TS(4)

  float* slab = p.slabs + ((size_t)tile * p.splits + split) * SLAB + (size_t)wave * NQ * 256 + lane * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < FN; ++v)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {acc[u][v][4 * g], acc[u][v][4 * g + 1], acc[u][v][4 * g + 2], acc[u][v][4 * g + 3]};
        *reinterpret_cast<f32x4*>(slab + ((u * FN + v) * 4 + g) * 256) = o;
      }
  DIG_WG_TS_END()
}

template <int FN, int NSTG>
int launch_wgrad_wide(const WgParams& p, int n_wg, hipStream_t stream) {
  constexpr int LDS = NSTG * (4 * 16 * 128 + 4 * (128 * FN / 16) * 128);
  static bool attr_set[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_kernel<FN, NSTG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set[dev] = true;
  }
  dig_launch(wgrad_wide_kernel<FN, NSTG>, dim3(n_wg), dim3(512), LDS, stream, p);
  return dig_check_launch();
}

template <int FN>
int launch_wgrad(const WgParams& p, int n_wg, hipStream_t stream) {
  constexpr int LDS = WG_NSTG * (4 * 8 * 128 + 4 * (128 * FN / 16) * 128);
  static bool attr_set[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel<FN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set[dev] = true;
  }
  dig_launch(wgrad_group_kernel<FN>, dim3(n_wg), dim3(256), LDS, stream, p);
  return dig_check_launch();
}

int fill_probs(const dig_wgrad_prob_t* in, int n, int fn, int wa, WgProb* out, int* tiles_out) {
  int tiles = 0;
  const int TJ = 128 * fn, TI = 128 * wa;
  for (int k = 0; k < n; ++k) {
    const dig_wgrad_prob_t& q = in[k];
    if (!q.out || q.I <= 0 || q.J <= 0 || (q.I % 128) || (q.J % TJ)) return DIG_ERR_ARG;
    if ((q.ldo & 3) || !aligned16(q.out)) return DIG_ERR_ALIGN;
    out[k].A = (const bf16_t*)q.A; out[k].B = (const bf16_t*)q.B; out[k].out = q.out;
    out[k].lda = q.lda; out[k].ldb = q.ldb; out[k].ldo = q.ldo; out[k].I = q.I; out[k].J = q.J; out[k].trans_out = q.trans_out;
    out[k].tile0 = tiles;
    tiles += ((q.I + TI - 1) / TI) * (q.J / TJ);
  }
  *tiles_out = tiles;
  return DIG_OK;
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_wgrad_group_supported(int I, int J, int R) {
  // (operands are addressed through 32-bit buffer offsets: R rows of at least I / J columns each must stay below 4 GiB -- the launcher
  //  checks the real leading dimensions again)
  return (I > 0 && J > 0 && R >= 64 && I % 128 == 0 && (J % 384 == 0 || J % 256 == 0) && R % 64 == 0 &&
          (unsigned long long)R * (unsigned)I * 2ull < (1ull << 32) && (unsigned long long)R * (unsigned)J * 2ull < (1ull << 32)) ? 1 : 0;
}
extern "C" int dig_wgrad_group_fn(int J) { return J % 384 == 0 ? 3 : (J % 256 == 0 ? 2 : 0); }

// rows per R-split for a requested split count (whole groups of four 16-row stages), and the split count that results
extern "C" int dig_wgrad_group_rows_per_split(int R, int splits) {
  if (R <= 0 || splits < 1 || R % 64) return 0;
  return ((R / 64 + splits - 1) / splits) * 64;
}
extern "C" int dig_wgrad_group_effective_splits(int R, int splits) {
  const int per = dig_wgrad_group_rows_per_split(R, splits);
  return per ? (R + per - 1) / per : 0;
}
extern "C" long long dig_wgrad_group_slab_bytes(int total_tiles, int splits, int fn, int wa) {
  return (long long)total_tiles * splits * 128 * wa * 128 * fn * 4;
}
// tiles of one problem: wa = 1: 128 x 128 fn (4 waves, two workgroups per CU);  wa = 2: 256 x 128 fn (8 waves, one workgroup per CU)
extern "C" int dig_wgrad_group_tiles(int I, int J, int fn, int wa) {
  if (I <= 0 || J <= 0 || (fn != 2 && fn != 3) || (wa != 1 && wa != 2) || (I % 128) || (J % (128 * fn))) return 0;
  return ((I + 128 * wa - 1) / (128 * wa)) * (J / (128 * fn));
}

// Workgroup table of one launch (host memory in, host memory out): tiles_per_prob[n_probs] tiles x S R-splits, the tiles of one
// (problem, split) pair on one XCD, the eight XCDs loaded evenly.  S = the largest split count <= max_wg / tiles (whole groups of
// four 16-row stages per split) whose table fits max_wg workgroups (one round at two workgroups per CU: 512 on MI355X); *splits_out
// receives it.  map_out[n_wg] entries: tile | split << 16, or 0xffffffff (a workgroup that only takes part in the fold).
// Returns n_wg (a multiple of 8, <= max_out) or a negative error.
extern "C" int dig_wgrad_group_plan(const int* tiles_per_prob, int n_probs, int R, int max_wg, int* splits_out, unsigned* map_out,
                                    int max_out) {
  if (!tiles_per_prob || !map_out || !splits_out || n_probs < 1 || n_probs > DIG_WGRAD_MAX_PROBS || R < 64 || (R % 64) || max_wg < 8) return DIG_ERR_ARG;
  int tiles = 0;
  for (int k = 0; k < n_probs; ++k) {
    if (tiles_per_prob[k] < 1) return DIG_ERR_ARG;
    tiles += tiles_per_prob[k];
  }
  if (tiles > 65535) return DIG_ERR_ARG;
  struct Grp { int tile0, n, split; };
  int last_eff = -1;
  for (int want = std::max(1, std::min(max_wg / tiles, R / 64));; --want) {
    const int S = dig_wgrad_group_effective_splits(R, want);
    if (S == last_eff && want > 1) continue;
    last_eff = S;
    std::vector<Grp> groups;
    int tile0 = 0;
    for (int k = 0; k < n_probs; ++k) {
      for (int s = 0; s < S; ++s) groups.push_back({tile0, tiles_per_prob[k], s});
      tile0 += tiles_per_prob[k];
    }
    std::stable_sort(groups.begin(), groups.end(), [](const Grp& a, const Grp& b) { return a.n > b.n; });
    std::vector<unsigned> bins[8];
    for (const Grp& g : groups) {
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (bins[x].size() < bins[best].size()) best = x;
      for (int t = 0; t < g.n; ++t) bins[best].push_back((unsigned)(g.tile0 + t) | ((unsigned)g.split << 16));
    }
    size_t len = 0;
    for (int x = 0; x < 8; ++x) len = std::max(len, bins[x].size());
    if ((long long)len * 8 > max_wg && want > 1) continue;
    if ((long long)len * 8 > max_out || S > 65535) return DIG_ERR_ARG;
    for (size_t k = 0; k < len; ++k)
      for (int x = 0; x < 8; ++x) map_out[k * 8 + x] = k < bins[x].size() ? bins[x][k] : WG_NONE;
    *splits_out = S;
    return (int)(len * 8);
  }
}

// One launch: fold the previous launch's slabs (fold_probs / fold_slabs / fold_splits; n_fold = 0: nothing pending) into their
// gradients, then the partial products of `probs` into `slabs` ([total_tiles][splits] slabs of 128 wa x 128 fn floats).  n_probs = 0:
// fold only (wg_map may be null; n_wg workgroups share the fold).  All problems of a launch share R, splits and fn.
extern "C" int dig_wgrad_group(const dig_wgrad_prob_t* probs, int n_probs, const dig_wgrad_prob_t* fold_probs, int n_fold, int R,
                               int splits, const unsigned* wg_map, int n_wg, float* slabs, const float* fold_slabs, int fold_splits,
                               int fn, int wa, hipStream_t stream) {
  if (n_probs < 0 || n_probs > DIG_WGRAD_MAX_PROBS || n_fold < 0 || n_fold > DIG_WGRAD_MAX_PROBS || (n_probs == 0 && n_fold == 0)) return DIG_ERR_ARG;
  if ((fn != 2 && fn != 3) || (wa != 1 && wa != 2)) return DIG_ERR_UNSUPPORTED;
  if (n_wg < 1 || (n_probs > 0 && (!probs || !wg_map || !slabs || R < 64 || (R % 64) || splits < 1))) return DIG_ERR_ARG;
  if (n_fold > 0 && (!fold_probs || !fold_slabs || fold_splits < 1)) return DIG_ERR_ARG;
  WgParams p{};
  int tiles = 0, ftiles = 0;
  int rc = n_probs ? fill_probs(probs, n_probs, fn, wa, p.prob, &tiles) : DIG_OK;
  if (rc) return rc;
  rc = n_fold ? fill_probs(fold_probs, n_fold, fn, wa, p.fold, &ftiles) : DIG_OK;
  if (rc) return rc;
  for (int k = 0; k < n_probs; ++k) {
    const WgProb& q = p.prob[k];
    if (!q.A || !q.B) return DIG_ERR_ARG;
    if (!aligned16(q.A) || !aligned16(q.B) || (q.lda & 7) || (q.ldb & 7) || q.lda < q.I || q.ldb < q.J) return DIG_ERR_ALIGN;
    if ((size_t)R * q.lda * 2 >= (1ull << 32) || (size_t)R * q.ldb * 2 >= (1ull << 32)) return DIG_ERR_UNSUPPORTED;
  }
  p.n_prob = n_probs; p.n_fold = n_fold;
  p.R = R; p.splits = splits;
  p.r_per_split = n_probs ? dig_wgrad_group_rows_per_split(R, splits) : 64;
  if (n_probs && dig_wgrad_group_effective_splits(R, splits) != splits) return DIG_ERR_ARG;
  p.fold_splits = fold_splits; p.fold_tiles = ftiles;
  p.slabs = slabs; p.fold_slabs = fold_slabs; p.wg_map = wg_map;
  if (n_probs && !aligned16(slabs)) return DIG_ERR_ALIGN;
  if (n_fold && !aligned16(fold_slabs)) return DIG_ERR_ALIGN;
  if (wa == 2) return fn == 3 ? launch_wgrad_wide<3, 7>(p, n_wg, stream) : launch_wgrad_wide<2, 8>(p, n_wg, stream);
  return fn == 3 ? launch_wgrad<3>(p, n_wg, stream) : launch_wgrad<2>(p, n_wg, stream);
}
