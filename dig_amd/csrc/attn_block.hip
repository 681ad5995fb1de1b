// The attention sub-block of a DiG ViT encoder block in ONE launch (gfx950, D = 384, 6 heads of 64, 256 tokens per image):
//     x_mid = x + proj( softmax( (ln1 Wq^T + bq) s (ln1 Wk^T)^T ) (ln1 Wv^T + bv) ) + bp
// Reference math: Attention.forward, modeling_finetune.py:87-120 (qkv Linear with the (q_bias | 0 | v_bias) vector :103-109, q scaled by
// head_dim^-0.5, softmax over the 256 keys, proj :117-120) and the first residual add of Block.forward :156.  It replaces three launches
// (qkv GEMM -> dig_attn_fwd -> proj GEMM + residual) whose intermediates (qkv [R, 1152], ctx [R, 384]) made 573 MB of HBM traffic per
// block against 100 MB (momentum branch: read ln1 and x, write x_mid) or ~350 MB (online branch: qkv, ctx, lse are what the backward
// reads) of necessary bytes.
//
// Design (CDNA4).  One workgroup = one image (2 B = 256 images at the BASELINE batch = one per CU), 8 waves, wave w owns the 32 token rows
// 32 w .. 32 w + 31 for EVERYTHING: their ln1 rows stay in its registers as MFMA operand fragments (96 VGPRs) for the whole kernel.
//   * The weights are ONE stream of 24 row blocks of 64 output rows x 384 (q_h, k_h, v_h for h = 0..5, then the six 64-row blocks of
//     the projection), HBM/L2 -> LDS by buffer_load ... lds from inline asm into a ring of three 24-KiB slots (a "tick" = [64 rows][192 k] =
//     24 MFMAs per wave, one workgroup barrier per tick, the DMA of tick t + 2 issued between the MFMAs of tick t, counted s_waitcnt vmcnt).
//     A slot is two XOR-swizzled images ([64][128] + [64][64], swizzle on the source side: LDS-DMA destinations are lane-linear).
//   * MFMAs are issued swapped: D^T[64 rows of W, 32 tokens] = Wblock X^T, so a lane owns a TOKEN and its registers run over the output
//     features.  The finished block is scaled / biased / rounded in registers.  q_h stays in registers (it is the B operand of
//     S^T = K Q^T); k_h goes to LDS as ready-made A fragments (lane-linear 16-byte pieces: the producing lane of wave kt IS the consuming
//     lane of key tile kt, so the hidden-dimension permutation of the accumulator layout cancels between Q and K); v_h goes to LDS in
//     layout "U" of attention.hip and is read back with ds_read_b64_tr_b16.  Online branch: the same packed values are written to qkv.
//   * attention for head h: the wave's 32 queries against the 256 keys in two halves of 128 keys (online softmax: the second half
//     rescales the context accumulator by exp(m_1 - m)): 64 score registers instead of 128, which is what lets the ln1 rows stay
//     resident.  The normalised context rows go to ctx (HBM: the backward reads them; the momentum branch reads them back below).
//   * projection: the wave re-reads ITS OWN 32 context rows (L2-warm, written by itself) as operand fragments in place of the ln1
//     rows and runs six more blocks of the same tick loop over Wproj; the residual rows arrive by LDS-DMA into the (now free) K region,
//     bias + residual are added in registers and x_mid is stored with 16-byte row stores.
// LDS: ring 72 KiB + K 32 KiB + V 32 KiB + bias vectors 6 KiB + store staging 16 KiB = 158 KiB.  MFMAs per wave: 24 blocks x 48 + 6 heads x 64 = 1 536.
#include "common.h"
#include "lds_dma.h"
#include "attn_tiles.h"
#include <type_traits>

// hooks of tools/experiments/attn_block_lab.hip (empty in the product build)
#ifndef DIG_AB_SAFE_WAITS
#define DIG_AB_SAFE_WAITS 0               // 1: every counted wait becomes vmcnt(0) (lab: the counts must not matter for the result)
#endif
#ifndef DIG_AB_NSPLIT
#define DIG_AB_NSPLIT 2                   // key parts of the online softmax (2: 64 score registers live, 4: 32)
#endif
#ifndef DIG_AB_P0
#define DIG_AB_P0 2                        // k-steps of a tick behind which the three ring pieces of tick t + 2 are issued
#define DIG_AB_P1 5
#define DIG_AB_P2 8
#endif
#ifndef DIG_AB_PRIO
#define DIG_AB_PRIO 0                      // 1: waves 4-7 at s_setprio 1 for the whole kernel
#endif
#ifndef DIG_AB_NT
#define DIG_AB_NT 1                       // non-temporal stores: bit 0 qkv (read again only by the backward: with the default policy the 150 MB of
                                          // write-allocated lines cost the online form 15 us of 147; lab), 1 x_mid, 2 ctx (both re-read at once: default policy)
#endif
#ifndef DIG_AB_SKEW
#define DIG_AB_SKEW 0
#endif
#ifndef DIG_AB_ABL
#define DIG_AB_ABL 0                      // lab ablations: 1 no attention MFMAs, 2 no tick MFMAs, 4 no HBM stores, 8 no softmax arithmetic, 16 no ring DMA
#endif
#ifndef DIG_AB_TS                         // lab: per-wave time accounting (phase k starts here): 0 wait + barrier, 1 tick, 2 block epilogue, 3 attention, 4 row loads
#define DIG_AB_TS(k)
#define DIG_AB_TS_BEGIN()
#define DIG_AB_TS_END()
#endif

namespace {

constexpr int KD = 384;                   // model width
constexpr int NH = 6;                     // heads (of DH = 64)
constexpr int SLOTB = 24576;              // ring slot: [64 rows][128 k] (16 KiB, 16 pieces per row) + [64 rows][64 k] (8 KiB, 8 pieces per row)
constexpr int RING_OFF = 0;
constexpr int K_OFF = 3 * SLOTB;          // K fragments [key tile 8][k-step 4][lane 64] x 16 B;  projection phase: residual staging, 4 KiB per wave
constexpr int V_OFF = K_OFF + 32768;      // V [256 keys][64] in layout U
constexpr int VEC_OFF = V_OFF + 32768;    // fp32: qkv bias [1152], proj bias [384]
constexpr int STG_OFF = VEC_OFF + (3 * KD + KD) * 4;   // store staging: 2 KiB per wave = [16 rows][64 columns]
constexpr int LDS_BYTES = STG_OFF + 8 * 2048;
constexpr int NTICK = 48;                 // 24 blocks x 2 ticks

struct AbParams {
  const bf16_t* X;        // ln1 rows [R, KD]
  const bf16_t* resid;    // x rows [R, KD]
  const bf16_t* Wqkv;     // [3 KD, KD]
  const float* bqkv;      // [3 KD] or null
  const bf16_t* Wp;       // [KD, KD]
  const float* bp;        // [KD] or null
  bf16_t* qkv;            // [R, 3 KD]   (SAVE)
  bf16_t* ctx;            // [R, KD]
  float* lse;             // [n_img NH, 256]  (SAVE)
  bf16_t* out;            // x_mid [R, KD]
  float scale;
  unsigned x_bytes;       // R KD 2
};

template <int N>
__device__ __forceinline__ void ab_wait_vm() {
  if (DIG_AB_SAFE_WAITS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ab_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Row stores of a wave's finished block: 32 token rows x 64 columns, lane (row rr, half hi) holding its row's packed pairs
// P[b][g][k] = columns 32 b + 8 g + 4 hi + 2 k + 0..1.  Written row-per-lane (each lane 16 bytes of its own row) a store instruction makes 64
// separate 16-byte write requests, and at 256 workgroups x 8 waves the L2's request rate, not its bandwidth, prices the epilogue (lab: the
// stores cost the momentum form 44 of 136 us).  So the block goes through 2 KiB of wave-private LDS, 16 rows at a time, and leaves in FULL
// 128-byte lines: 8 adjacent lanes per row.  Still four 16-byte store instructions per block (the vmcnt arithmetic does not change).
template <bool NT = false>
__device__ __forceinline__ void store_block_lines(bf16_t* blk /* &T[wave's first row][first column] */, int ld, unsigned (&P)[2][4][2],
                                                  unsigned char* stg, int lane) {
  const int rr = lane & 31, hi = lane >> 5;
  const int wr = (rr & 15) * 128 + hi * 8, ws = rr & 7;                   // write: row rr & 15, 16-byte position (4 b + g) ^ (row & 7)
  const int r8 = lane >> 3, c = lane & 7;                                  // read: rows r8 / 8 + r8 of the pass, chunk c
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if ((rr >> 4) == pass) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(stg + wr + (((4 * b + g) ^ ws) << 4)) = make_uint2(P[b][g][0], P[b][g][1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // (rows r8 and 8 + r8 of the pass: (8 + r8) & 7 == r8 & 7, one position for both)
    const dig_u32x4 v0 = *reinterpret_cast<const dig_u32x4*>(stg + r8 * 128 + ((c ^ (r8 & 7)) << 4));
    const dig_u32x4 v1 = *reinterpret_cast<const dig_u32x4*>(stg + (8 + r8) * 128 + ((c ^ (r8 & 7)) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (!(DIG_AB_ABL & 4)) {
      dig_u32x4* d0 = reinterpret_cast<dig_u32x4*>(blk + (size_t)(16 * pass + r8) * ld + c * 8);
      dig_u32x4* d1 = reinterpret_cast<dig_u32x4*>(blk + (size_t)(16 * pass + 8 + r8) * ld + c * 8);
      if (NT) { __builtin_nontemporal_store(v0, d0); __builtin_nontemporal_store(v1, d1); }
      else { *d0 = v0; *d1 = v1; }
    }
  }
}

// SAVE: online branch (qkv and lse are written).  Stores a wave issues behind a block's last ring pieces (they enter the vmcnt arithmetic):
template <bool SAVE> struct AbCounts {
  static constexpr int E_BLK = (DIG_AB_ABL & 4) ? 0 : (SAVE ? 4 : 0);           // q / k / v rows of a block -> qkv
  static constexpr int E_ATT = (DIG_AB_ABL & 4) ? 0 : 4 + (SAVE ? 1 : 0);       // context rows (+ lse)
  static constexpr int E_OUT = (DIG_AB_ABL & 4) ? 0 : 4;                        // x_mid rows of a projection block
  static constexpr int E_RES = 4;                            // residual pieces of a projection block (LDS-DMA, in front of the tick's ring pieces)
};

// Everything a lane derives from its thread id (LDS addresses, DMA source offsets, row pointers) is re-derived where it is used from a copy of
// the id the optimiser cannot see through: hoisted out of the head loop as ~40 loop-invariant registers they were spilled around the attention
// phase (the ln1 rows already take 96 of the 256 registers for the whole kernel).  A handful of integer operations per block instead.
__device__ __forceinline__ int ab_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <bool SAVE, int NSPLIT>
__global__ __launch_bounds__(512) void attn_block_kernel(AbParams p) {
  using C = AbCounts<SAVE>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int img = blockIdx.x;
  DIG_AB_TS_BEGIN()
#if DIG_AB_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#if DIG_AB_SKEW
  // start skew: the workgroups of a launch run their 24 blocks in lock-step (they start together and do identical work), so every block
  // epilogue is a chip-wide burst of stores; a start delay spread over one block period de-phases the CUs
  for (int i = 0; i < (int)((blockIdx.x >> 3) & 15); ++i) __builtin_amdgcn_s_sleep(DIG_AB_SKEW);
#endif

  const dig_u32x4 rWq = make_rsrc(p.Wqkv, 3u * KD * KD * 2u);
  const dig_u32x4 rWp = make_rsrc(p.Wp, (unsigned)(KD * KD * 2));
  const dig_u32x4 rRes = make_rsrc(p.resid, p.x_bytes);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);

  // ---- ring: source offset of the 16-byte piece k (0..2) a thread brings per tick (lane-linear destinations: the swizzle is on the source side)
  auto piece_voff = [&](int tid, int k) -> unsigned {
    if (k < 2) {                                                                     // image A [64][128]: row tid >> 4 (+ 32), position tid & 15
      const int r0 = tid >> 4, c0 = (tid & 15) ^ (r0 & 15);
      return (unsigned)(((r0 + 32 * k) * KD + 8 * c0) * 2);
    }
    const int r2 = tid >> 3, c2 = (tid & 7) ^ ((r2 >> 1) & 7);                      // image B [64][64]: k 128..191
    return (unsigned)((r2 * KD + 128 + 8 * c2) * 2);
  };
  // Source of the ticks a block requests: its two ticks bring the two halves of the NEXT block of the stream (tick t requests tick t + 2).
  // Stream order: q_h, k_h, v_h for h = 0..5 (rows part KD + 64 h of Wqkv), then the six 64-row blocks of Wproj; past the end: block 0 again.
  dig_u32x4 src_rs = rWq;
  unsigned src_off = 0;
  auto select_source = [&](bool proj, int row0) {                                   // wave-uniform: scalar selects, no branch
#pragma unroll
    for (int e = 0; e < 4; ++e) src_rs[e] = proj ? rWp[e] : rWq[e];
    src_off = (unsigned)(row0 * KD * 2);
  };
  auto issue_piece = [&](int slot, int tau, int k, unsigned voff) {
    if (DIG_AB_ABL & 16) voff = 0x7fffff00u;                                        // (lab: out-of-range source offsets -- the DMA is issued, nothing is fetched)
    dma16(lds0 + (unsigned)(RING_OFF + slot * SLOTB + k * 8192 + wave * 1024), voff, src_rs, src_off + (unsigned)(tau * 192 * 2));
  };
#pragma unroll
  for (int tau = 0; tau < 2; ++tau)
#pragma unroll
    for (int k = 0; k < 3; ++k) issue_piece(tau, tau, k, piece_voff(tid0, k));      // block 0 (q_0) -> slots 0, 1

  // ---- the wave's 32 ln1 rows as MFMA operand fragments: lane (token rr, half hi) holds k = 16 s + 8 hi .. + 7 of step s
  dig_u32x4 xf[KD / 16];
  auto load_rows = [&](const bf16_t* base) {
    const int tid = ab_opaque(tid0);
    const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.x_bytes, 0x00020000);
    const unsigned xo = (unsigned)((((unsigned)img * 256u + (unsigned)(wave * 32 + (tid & 31))) * KD + ((tid >> 5) & 1) * 8) * 2);
#pragma unroll
    for (int s = 0; s < KD / 16; ++s) xf[s] = __builtin_amdgcn_raw_buffer_load_b128(rX, xo, s * 32, 0);
  };
  load_rows(p.X);
  // bias vectors -> LDS (published by the first tick barrier)
  {
    float* vec = reinterpret_cast<float*>(smem + VEC_OFF);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = u * 512 + tid0;
      float v;
      if (i < 3 * KD) v = p.bqkv ? p.bqkv[i] : 0.f;
      else v = p.bp ? p.bp[i - 3 * KD] : 0.f;
      vec[i] = v;
    }
  }

  // One tick: barrier (this wave's pieces of the tick have landed: NW = VMEM operations it has issued behind them), 24 MFMAs into acc
  // (rows 0..31 / 32..63 of the block), the three pieces of tick t + 2 between them.  SLOT is the tick's ring slot = its position in the
  // triple of blocks (compile-time), TAU its half of the block's reduction.
  // Fragment read offsets within a slot: k-step s of the tick (0..11), row rr of a 32-row half (second half: + 8192 / + 4096).  The XOR
  // swizzle commutes with the step: offset(s) = base ^ (s << 5), so two registers stand for the twelve offsets.
  auto tick = [&](auto slot_tag, auto tau_tag, auto nw_tag, auto pre_tag, f32x16 (&acc)[2], auto&& pre_fn) {
    constexpr int SLOT = decltype(slot_tag)::value, TAU = decltype(tau_tag)::value, NW = decltype(nw_tag)::value;
    constexpr int FREE = (SLOT + 2) % 3;
    DIG_AB_TS(0)
    ab_wait_vm<NW>();
    ab_wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DIG_AB_TS(1)
    if (decltype(pre_tag)::value) pre_fn();                                         // (projection: the block's residual pieces, in front of the ring pieces)
    const int tid = ab_opaque(tid0);
    const int rr = tid & 31, hi = (tid >> 5) & 1;
    const int bA = rr * 256 + ((hi ^ (rr & 15)) << 4) + RING_OFF + SLOT * SLOTB;
    const int bB = 16384 + rr * 128 + ((hi ^ ((rr >> 1) & 7)) << 4) + RING_OFF + SLOT * SLOTB;
    auto frag = [&](int s, int half) {
      const int a = s < 8 ? (bA ^ (s << 5)) + half * 8192 : (bB ^ ((s - 8) << 5)) + half * 4096;
      return *reinterpret_cast<const bf16x8*>(smem + a);
    };
    bf16x8 fa[2], fb[2];
    fa[0] = frag(0, 0);
    fa[1] = frag(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    auto kstep = [&](auto s_tag, bf16x8 (&cur)[2], bf16x8 (&nxt)[2]) {
      constexpr int S = decltype(s_tag)::value;
      if constexpr (S < 11) {
        nxt[0] = frag(S + 1, 0);
        nxt[1] = frag(S + 1, 1);
      }
      if (!(DIG_AB_ABL & 2)) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], __builtin_bit_cast(bf16x8, xf[TAU * 12 + S]), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[1], __builtin_bit_cast(bf16x8, xf[TAU * 12 + S]), acc[1], 0, 0, 0);
      } else {
        asm volatile("" ::"v"(cur[0]), "v"(cur[1]));
      }
      if (S == DIG_AB_P0) issue_piece(FREE, TAU, 0, piece_voff(tid, 0));
      if (S == DIG_AB_P1) issue_piece(FREE, TAU, 1, piece_voff(tid, 1));
      if (S == DIG_AB_P2) issue_piece(FREE, TAU, 2, piece_voff(tid, 2));
      __builtin_amdgcn_sched_barrier(0);
    };
    kstep(std::integral_constant<int, 0>{}, fa, fb); kstep(std::integral_constant<int, 1>{}, fb, fa);
    kstep(std::integral_constant<int, 2>{}, fa, fb); kstep(std::integral_constant<int, 3>{}, fb, fa);
    kstep(std::integral_constant<int, 4>{}, fa, fb); kstep(std::integral_constant<int, 5>{}, fb, fa);
    kstep(std::integral_constant<int, 6>{}, fa, fb); kstep(std::integral_constant<int, 7>{}, fb, fa);
    kstep(std::integral_constant<int, 8>{}, fa, fb); kstep(std::integral_constant<int, 9>{}, fb, fa);
    kstep(std::integral_constant<int, 10>{}, fa, fb); kstep(std::integral_constant<int, 11>{}, fb, fa);
    DIG_AB_TS(2)
  };
  auto no_pre = []() {};
  using FALSE_ = std::false_type;
  using TRUE_ = std::true_type;

  // finished block -> packed pairs: P[b][g][k] = bf16 pair of ((acc[b][4 g + 2 k + 0..1] + bias[col]) * al); col = col0 + 32 b + 8 g + 4 hi + e
  auto finish_block = [&](const f32x16 (&acc)[2], int vec_col0, float al, unsigned (&P)[2][4][2], int hi) {
    const float* bv = reinterpret_cast<const float*>(smem + VEC_OFF + (vec_col0 + 4 * hi) * 4);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + 32 * b + 8 * g);
        P[b][g][0] = pack_bf2((acc[b][4 * g] + b4[0]) * al, (acc[b][4 * g + 1] + b4[1]) * al);
        P[b][g][1] = pack_bf2((acc[b][4 * g + 2] + b4[2]) * al, (acc[b][4 * g + 3] + b4[3]) * al);
      }
  };
  const size_t wave_row0 = (DIG_AB_ABL & 32) ? (size_t)wave * 32 : (size_t)img * 256 + wave * 32;   // first of the wave's 32 token rows (lab bit 32: every image writes image 0's rows: no HBM write volume)

  bf16x8 Qp[4];                                                                     // q_h of the wave's 32 tokens: B fragments of S^T = K Q^T, k-step (b, u)

  // ---- one head: blocks q_h, k_h, v_h (ring slots 0 1 | 2 0 | 1 2), then the attention of the wave's 32 queries.
  // EPREV = stores the wave issued behind the last ring pieces of the block before q_h (the previous head's v rows + context rows)
  auto head = [&](auto first_tag, int h) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int EPREV = FIRST ? 0 : C::E_BLK + C::E_ATT;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    f32x16 acc[2];
    unsigned P[2][4][2];
    auto zero = [&]() {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    };
    // q_h (its ticks request k_h)
    select_source(false, KD + h * 64);
    zero();
    tick(I0{}, I0{}, std::integral_constant<int, 3 + EPREV>{}, FALSE_{}, acc, no_pre);
    tick(I1{}, I1{}, std::integral_constant<int, 3 + EPREV>{}, FALSE_{}, acc, no_pre);
    {
      const int tid = ab_opaque(tid0), hi = (tid >> 5) & 1;
      finish_block(acc, h * 64, p.scale, P, hi);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          Qp[2 * b + u] = __builtin_bit_cast(bf16x8, make_uint4(P[b][2 * u][0], P[b][2 * u][1], P[b][2 * u + 1][0], P[b][2 * u + 1][1]));
      if (SAVE) store_block_lines<(DIG_AB_NT & 1) != 0>(p.qkv + wave_row0 * (3 * KD) + h * 64, 3 * KD, P, smem + STG_OFF + wave * 2048, tid & 63);
    }
    __builtin_amdgcn_sched_barrier(0);
    // k_h: A fragments of S^T for key tile `wave`, k-step (b, u): the lane's own 16 bytes
    select_source(false, 2 * KD + h * 64);
    zero();
    tick(I2{}, I0{}, std::integral_constant<int, 3 + C::E_BLK>{}, FALSE_{}, acc, no_pre);
    tick(I0{}, I1{}, std::integral_constant<int, 3 + C::E_BLK>{}, FALSE_{}, acc, no_pre);
    {
      const int tid = ab_opaque(tid0), hi = (tid >> 5) & 1;
      finish_block(acc, KD + h * 64, 1.0f, P, hi);
      unsigned char* kw = smem + K_OFF + wave * 4096 + (tid & 63) * 16;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          *reinterpret_cast<uint4*>(kw + (2 * b + u) * 1024) = make_uint4(P[b][2 * u][0], P[b][2 * u][1], P[b][2 * u + 1][0], P[b][2 * u + 1][1]);
      if (SAVE) store_block_lines<(DIG_AB_NT & 1) != 0>(p.qkv + wave_row0 * (3 * KD) + KD + h * 64, 3 * KD, P, smem + STG_OFF + wave * 2048, tid & 63);
    }
    __builtin_amdgcn_sched_barrier(0);
    // v_h: layout U, row = key (its ticks request the next head's q rows, or the first projection block)
    select_source(h + 1 == NH, h + 1 == NH ? 0 : (h + 1) * 64);
    zero();
    tick(I1{}, I0{}, std::integral_constant<int, 3 + C::E_BLK>{}, FALSE_{}, acc, no_pre);
    tick(I2{}, I1{}, std::integral_constant<int, 3 + C::E_BLK>{}, FALSE_{}, acc, no_pre);
    {
      const int tid = ab_opaque(tid0), hi = (tid >> 5) & 1;
      finish_block(acc, 2 * KD + h * 64, 1.0f, P, hi);
      // u_addr(key, 32 b + 8 g + 4 hi) = key 128 + 8 hi + (((4 b + g) ^ swz(key)) << 4): one base, the column group is an XOR
      const int key = wave * 32 + (tid & 31);
      const int vw = V_OFF + key * 128 + 8 * hi + (swz(key) << 4);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(smem + (vw ^ ((4 * b + g) << 4))) = make_uint2(P[b][g][0], P[b][g][1]);
      if (SAVE) store_block_lines<(DIG_AB_NT & 1) != 0>(p.qkv + wave_row0 * (3 * KD) + 2 * KD + h * 64, 3 * KD, P, smem + STG_OFF + wave * 2048, tid & 63);
    }
    ab_wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- attention: S^T[key, query] per 32-key tile (lane = query, registers = keys)
    DIG_AB_TS(3)
    const int tid = ab_opaque(tid0);
    const int lane = tid & 63, hi = lane >> 5;
    f32x16 O[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
    float m = 0.f, l = 0.f;
    const unsigned char* Kl = smem + K_OFF + lane * 16;                              // every tile offset below is an instruction immediate
    // transposed V fragment of 16-key block blk16, 32-column half dt: rows {4 hi + 0..3, 8 + 4 hi + 0..3} of column 32 dt + (lane & 31)
    // (frag_tr of attn_tiles.h; u_addr(ra + 8, col) = (u_addr(ra, col) ^ 32) + 1024 and the column half is an XOR with 64)
    const int i16 = lane & 15;
    const int vt0 = V_OFF + u_addr(4 * hi + (i16 >> 2), ((lane >> 4) & 1) * 16 + (i16 & 3) * 4);
    auto vfrag = [&](int blk16, int dt) {
      typedef __attribute__((address_space(3))) bf16x4* lp;
      const int a = vt0 ^ (dt * 64);
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(smem + a + blk16 * 2048));
      const bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(smem + (a ^ 32) + 1024 + blk16 * 2048));
      return __builtin_shufflevector(lo, h4, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // the value of the other lane half beside the lane's own (v_permlane32_swap: no LDS round trip).  (The results are taken out of the
    // returned pair as integers first: hipcc 7.2 folds __builtin_bit_cast of a vector ELEMENT to element 0 -- see dig_as_bf16x2 in common.h.)
    auto xhalf = [&](float v, float& lo, float& up) {
      const unsigned a = __float_as_uint(v);
      const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
      const unsigned r0 = r[0], r1 = r[1];
      lo = __uint_as_float(r0);                                                      // lanes 0..31's value, on both halves
      up = __uint_as_float(r1);                                                      // lanes 32..63's
    };
    auto xhalf_max = [&](float v) {
      float lo, up;
      xhalf(v, lo, up);
      return fmaxf(lo, up);
    };
    // NSPLIT parts of TPP key tiles: part 0 sets the running maximum, every later part rescales what has been accumulated so far
    constexpr int TPP = 8 / NSPLIT;
#pragma unroll
    for (int part = 0; part < NSPLIT; ++part) {
      f32x16 S[TPP];
#pragma unroll
      for (int kt = 0; kt < TPP; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) S[kt][e] = 0.f;
      __builtin_amdgcn_sched_barrier(0);
      if (!(DIG_AB_ABL & 1)) {
#pragma unroll
        for (int bu = 0; bu < 4; ++bu)
#pragma unroll
          for (int kt = 0; kt < TPP; ++kt)
            S[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Kl + ((TPP * part + kt) * 4 + bu) * 1024), Qp[bu],
                                                            S[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      float mx = S[0][0];
#pragma unroll
      for (int kt = 0; kt < TPP; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, S[kt][e]);
      mx = xhalf_max(mx);
      if (part == 0) {
        m = mx;
      } else {
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) O[dt][e] *= alpha;
      }
      __builtin_amdgcn_sched_barrier(0);
      // tile by tile, in place: the tile's 16 scores become probabilities, then the four P V MFMAs (V fragments requested per 16-key step)
#pragma unroll
      for (int kt = 0; kt < TPP; ++kt) {
        float la = 0.f, lb = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const float pa = (DIG_AB_ABL & 8) ? S[kt][e] : __expf(S[kt][e] - m), pb = (DIG_AB_ABL & 8) ? S[kt][e + 1] : __expf(S[kt][e + 1] - m);
          S[kt][e] = pa; S[kt][e + 1] = pb;
          la += pa; lb += pb;
        }
        l += la + lb;
        asm volatile("" : "+v"(l));                                                   // (summed HERE: sunk to the end of the phase, the 16 probabilities were kept alive in scratch)
        if (!(DIG_AB_ABL & 1)) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bf16x8 v0 = vfrag((TPP * part + kt) * 2 + u, 0), v1 = vfrag((TPP * part + kt) * 2 + u, 1);
            const bf16x8 pf = pack8(S[kt], u);
            O[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf, O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf, O[1], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(S[kt][e]));             // (one register per statement: a 64-byte operand fails the HOST pass silently)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      float lo, up;
      xhalf(l, lo, up);
      l = lo + up;
    }
    const float inv = 1.0f / l;
    {
      unsigned Pc[2][4][2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Pc[dt][g][0] = pack_bf2(O[dt][4 * g] * inv, O[dt][4 * g + 1] * inv);
          Pc[dt][g][1] = pack_bf2(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv);
        }
      store_block_lines<(DIG_AB_NT & 4) != 0>(p.ctx + wave_row0 * KD + h * 64, KD, Pc, smem + STG_OFF + wave * 2048, lane);
    }
    if (SAVE && hi == 0 && !(DIG_AB_ABL & 4)) p.lse[((size_t)img * NH + h) * 256 + wave * 32 + (lane & 31)] = m + __logf(l);
    __builtin_amdgcn_sched_barrier(0);
  };

  head(TRUE_{}, 0);
  for (int h = 1; h < NH; ++h) head(FALSE_{}, h);

  // ---- projection: the wave's own context rows take the place of the ln1 rows
  DIG_AB_TS(4)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // the wave's context stores are complete (same wave: visible to its loads)
  load_rows(p.ctx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // Residual pieces of block jb (LDS-DMA, wave-private 4 KiB = [32 rows][64 columns] in the K region): piece k brings rows 8 k + (lane >> 3);
  // position lane & 7 of a row holds the 16-byte chunk (lane & 7) ^ ((row >> 1) & 7) of the block's 64 columns ((row >> 1) & 7 = 4 (k & 1) + (lane >> 4))
  auto triple = [&](auto first_tag, int jb0) {
    constexpr bool FIRST = decltype(first_tag)::value;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto block = [&](auto s0_tag, auto s1_tag, auto eprev_tag, int jb) {
      constexpr int EPREV = decltype(eprev_tag)::value;
      select_source(true, jb < 5 ? (jb + 1) * 64 : 0);                               // (the last block's requests are never read)
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
      auto res_pre = [&]() {
        const int tid = ab_opaque(tid0), lane = tid & 63;
        const unsigned rbase = ((unsigned)img * 256u + (unsigned)(wave * 32 + (lane >> 3))) * KD;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          dma16(lds0 + (unsigned)(K_OFF + wave * 4096 + k * 1024),
                (unsigned)((rbase + 8 * ((lane & 7) ^ (4 * (k & 1) + (lane >> 4)))) * 2) + (unsigned)(k * 8 * KD * 2), rRes, (unsigned)(jb * 64 * 2));
      };
      tick(s0_tag, I0{}, std::integral_constant<int, 3 + EPREV>{}, TRUE_{}, acc, res_pre);
      tick(s1_tag, I1{}, std::integral_constant<int, 3 + EPREV + C::E_RES>{}, FALSE_{}, acc, no_pre);
      // + bias + residual -> x_mid (the wave's residual pieces were issued in front of the ring pieces of the block's two ticks)
      ab_wait_vm<6>();
      const int tid = ab_opaque(tid0), rr = tid & 31, hi = (tid >> 5) & 1;
      unsigned P[2][4][2];
      const float* bv = reinterpret_cast<const float*>(smem + VEC_OFF + (3 * KD + jb * 64 + 4 * hi) * 4);
      const int ro = K_OFF + wave * 4096 + rr * 128 + hi * 8 + (((rr >> 1) & 7) << 4);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + 32 * b + 8 * g);
          const uint2 rw = *reinterpret_cast<const uint2*>(smem + (ro ^ ((4 * b + g) << 4)));
          const float v0 = (acc[b][4 * g] + b4[0]) + bf2f((bf16_t)(rw.x & 0xffff));
          const float v1 = (acc[b][4 * g + 1] + b4[1]) + bf2f((bf16_t)(rw.x >> 16));
          const float v2 = (acc[b][4 * g + 2] + b4[2]) + bf2f((bf16_t)(rw.y & 0xffff));
          const float v3 = (acc[b][4 * g + 3] + b4[3]) + bf2f((bf16_t)(rw.y >> 16));
          P[b][g][0] = pack_bf2(v0, v1);
          P[b][g][1] = pack_bf2(v2, v3);
        }
      store_block_lines<(DIG_AB_NT & 2) != 0>(p.out + wave_row0 * KD + jb * 64, KD, P, smem + STG_OFF + wave * 2048, tid & 63);
      __builtin_amdgcn_sched_barrier(0);
    };
    block(I0{}, I1{}, std::integral_constant<int, FIRST ? 0 : C::E_OUT>{}, jb0);
    block(I2{}, I0{}, std::integral_constant<int, C::E_OUT>{}, jb0 + 1);
    block(I1{}, I2{}, std::integral_constant<int, C::E_OUT>{}, jb0 + 2);
  };
  triple(TRUE_{}, 0);
  triple(FALSE_{}, 3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // the run-ahead pieces still on their way into LDS
  DIG_AB_TS_END()
}

}  // namespace

extern "C" int dig_attn_block_supported(int heads, int embed_dim) { return (heads == NH && embed_dim == KD) ? 1 : 0; }

// x_mid = x + proj(attention(ln1)) for n_img images of 256 tokens; qkv / lse null: nothing is kept for a backward (ctx is always written: scratch)
extern "C" int dig_attn_block_fwd(const void* ln1, const void* x, const void* qkv_w, const float* qkv_b, const void* proj_w, const float* proj_b,
                                  void* qkv, void* ctx, float* lse, void* x_mid, int n_img, int heads, int embed_dim, float scale,
                                  hipStream_t stream) {
  if (!ln1 || !x || !qkv_w || !proj_w || !ctx || !x_mid || n_img <= 0) return DIG_ERR_ARG;
  if ((qkv == nullptr) != (lse == nullptr)) return DIG_ERR_ARG;
  if (!dig_attn_block_supported(heads, embed_dim)) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(ln1) || !aligned16(x) || !aligned16(qkv_w) || !aligned16(proj_w) || !aligned16(ctx) || !aligned16(x_mid) || (qkv && !aligned16(qkv)))
    return DIG_ERR_ALIGN;
  const size_t qb = (size_t)n_img * 256 * 3 * KD * 2;
  if (qb >= (1ull << 32)) return DIG_ERR_ARG;
  AbParams p;
  p.X = (const bf16_t*)ln1; p.resid = (const bf16_t*)x; p.Wqkv = (const bf16_t*)qkv_w; p.bqkv = qkv_b; p.Wp = (const bf16_t*)proj_w; p.bp = proj_b;
  p.qkv = (bf16_t*)qkv; p.ctx = (bf16_t*)ctx; p.lse = lse; p.out = (bf16_t*)x_mid; p.scale = scale;
  p.x_bytes = (unsigned)((size_t)n_img * 256 * KD * 2);
  static bool attr[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block_kernel<false, DIG_AB_NSPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block_kernel<true, DIG_AB_NSPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return DIG_ERR_LAUNCH;
    attr[dev] = true;
  }
  if (qkv) dig_launch(attn_block_kernel<true, DIG_AB_NSPLIT>, dim3(n_img), dim3(512), LDS_BYTES, stream, p);
  else dig_launch(attn_block_kernel<false, DIG_AB_NSPLIT>, dim3(n_img), dim3(512), LDS_BYTES, stream, p);
  return dig_check_launch();
}
