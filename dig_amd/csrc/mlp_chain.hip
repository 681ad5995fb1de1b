// Fused two-layer MLP ("chain") kernels for gfx950: the hidden tensor of a transformer FFN never goes through HBM as a GEMM operand.
//
// Reference math: Mlp.forward, modeling_finetune.py:53-60 -- x + fc2(gelu(fc1(LN2(x)))) inside Block.forward :150-158 -- and its
// autograd.  One launch replaces the fc1 GEMM (+ bias + exact-erf GELU) and the fc2 GEMM (+ bias + residual) of a block
// (forward), or the fc2 data gradient (x GELU') and the fc1 data gradient (backward):
//     forward :  out[R,D] = resid + b2 + gelu(X[R,D] W1[F,D]^T + b1) W2[D,F]^T
//     backward:  dX[R,D]  = ((dY[R,D] W2[D,F]) * gelu'(pre[R,F])) W1[F,D]          (given W2^T [F,D] and W1^T [D,F] copies)
// Both are the same chain   out = epi( ew( X B1^T ) B2^T )   with B1 [F,D] and B2 [D,F] read along their contiguous axis.
//
// Design (CDNA4).  A workgroup owns 128 token rows and walks the hidden dimension F in chunks of 64; 8 waves = 4 pairs, a pair
// owns 32 tokens, the two waves of a pair sit on the same SIMD and have different ROLES:
//   * the S-wave keeps its 32 x D slice of X in registers (MFMA B fragments, 96 VGPRs) and computes S^T[64 f, 32 tokens] =
//     B1chunk X^T (swapped operands: a lane owns a token, its registers run over f), applies the elementwise op (bias + GELU, or
//     x GELU'(pre)) in fp32 and parks the bf16 result P[32 tokens][64 f] in LDS (4 KiB per pair);
//   * the O-wave keeps the 32 x D output accumulators in registers (192 VGPRs, initialised with bias + residual, so the epilogue
//     is a convert-and-store) and computes out^T[D, 32 tokens] += B2chunk P^T for the chunk the S-wave finished a period earlier;
//     it also drains the side outputs (pre-activation / GELU output for the online forward, d(pre-activation) for the backward)
//     from LDS to HBM with row-contiguous 16-byte stores.
// The matrix pipe of a SIMD is shared by the pair, and the S-wave's VALU work (14 ops per hidden element) runs in the shadow of
// the O-wave's MFMAs -- the hardware's own wave arbitration does the overlap that a single 16-wave lock-step GEMM cannot
// (its GELU epilogue is as long as its K loop).
// Weights stream HBM/L2 -> LDS by buffer_load ... lds into two 3-slot rings of 16 KiB (B1: [64 f][128 k], B2: [128 j][64 f],
// XOR swizzle on the source side); a "tick" = one slot of each ring = 16 MFMAs per wave, one workgroup barrier per tick, the
// DMA for tick n+2 is issued right behind the barrier of tick n (counted s_waitcnt vmcnt, raw s_barrier).  Three ticks = one chunk,
// so the ring slot of a tick is its position in the chunk (compile-time).
// LDS: rings 96 KiB + P 16 KiB + (forward) b1 [F] fp32 + (online forward) pre tile 16 KiB / (backward) pre tiles 2 x 16 KiB.
// The per-chunk pipeline is  S-wave: MFMA(c) | elementwise(c-1)   O-wave: MFMA(c-2)   => two chunk periods of fill/drain in 26.
//
// Backward extras: the fc1 bias gradient = column sums of d(pre-activation) over tokens.  In the S-wave's layout tokens run
// across lanes; instead of 160 DPP adds per chunk the packed bf16 block is multiplied by a constant selection matrix on the
// matrix core (D[m, n] = P[m, n]: an MFMA used as a transposer, 4 MFMAs per chunk), after which tokens run over a lane's
// registers: 15 in-lane adds + one cross-half add per 32 columns.  Partials go to colsum[ceil(R/32)][F] (fp32), summed by
// dig_colsum_partials.
#include "common.h"
#include <type_traits>

// hooks of tools/experiments/chain_lab.hip (empty in the product build): per-wave time accounting of the tick protocol and
// compile-time ablations (bit 0: no elementwise work, 1: no ring DMA, 2: no O-wave MFMAs, 3: no S-wave MFMAs)
#ifndef DIG_CHAIN_T
#define DIG_CHAIN_T(k)
#define DIG_CHAIN_T_BEGIN()
#define DIG_CHAIN_T_END()
#endif
#ifndef DIG_CHAIN_LNB_ABL             // lab (tools/gpu_chain_ln_lab.py), norm2's backward phase: 1 dy rows read from x_mid's addresses (no dy traffic),
#define DIG_CHAIN_LNB_ABL 0           // 2 no row loads at all, 4 no dx_mid stores; projection phase: 8 no MFMAs / fragment reads, 16 no weight DMA,
                                      // 32 no dctx stores (results are wrong)
#endif
#ifndef DIG_CHAIN_ABL
#define DIG_CHAIN_ABL 0
#endif
#ifndef DIG_CHAIN_SIDE_AUX
#define DIG_CHAIN_SIDE_AUX 2                // cache policy of the online forward's side-output stores (2 = nt)
#endif
#ifndef DIG_CHAIN_BWD_AUX
#define DIG_CHAIN_BWD_AUX 0                 // cache policy of the backward's d(pre-activation) stores (2 = nt measured: no difference in the step, 19.14 vs 19.14 ms)
#endif
#ifndef DIG_CHAIN_SDMA
#define DIG_CHAIN_SDMA 0                  // lab (round 6): 1 = online forward (MODE 1) with the S-waves bringing ALL ring pieces and the O-waves -- which issue the
                                          // side-output stores -- none.  Measured (tools/experiments/r06_sdma_ab.sh, profiles/r06_chain_sdma_lab.txt): 209.0 us
                                          // against 201.9 (bare form), 231.7 against 223.7 (product form) -- the O-waves' counted waits were never the cost
                                          // (11 k of 172 k cycles): the side outputs cost what they add to BOTH roles' work (O: 8 LDS reads + 8 KiB-stores per
                                          // period on a 64 B/clk path shared by the CU; S: the second tile's packs and LDS writes) under one barrier per tick
#endif
#ifndef DIG_CHAIN_PRIO
#define DIG_CHAIN_PRIO 0                  // 1: S-waves at s_setprio 1, 2: O-waves (static, for the whole kernel)
#endif

namespace {

constexpr int KD = 384;                 // model width: reduction dim of stage 1, output width of stage 2
constexpr int FC = 64;                  // hidden units per chunk
constexpr int BM = 128;                 // token rows per workgroup
constexpr int NJB = KD / 32;            // 12 output column blocks per token block
constexpr int SLOT = 16384;
constexpr int W1_RING = 0;
constexpr int W2_RING = 3 * SLOT;
constexpr int P_OFF = 6 * SLOT;         // 4 pairs x [32 tokens][64 f] bf16
constexpr int X_OFF = P_OFF + SLOT;     // MODE 1: pre-activation tile (output staging); MODE 2: two pre tiles (DMA input); MODE 0: b1
constexpr int KS_PER_TICK = 8;          // 128 k per B1 slot
constexpr int JB_PER_TICK = 4;          // 128 output columns per B2 slot

struct ChainParams {
  const bf16_t* X;        // [R, KD]
  const bf16_t* B1;       // [F, KD]
  const bf16_t* B2;       // [KD, F]
  const float* bias1;     // [F] (forward) or null
  const float* bias2;     // [KD] or null
  const bf16_t* resid;    // [R, KD] or null
  bf16_t* out;            // [R, KD]
  bf16_t* side0;          // MODE 1: gelu output [R, F];  MODE 2: d(pre-activation) [R, F];  may be null in MODE 1
  bf16_t* side1;          // MODE 1: pre-activation out [R, F];  MODE 2: pre-activation in [R, F]
  float* colsum;          // MODE 2: [ceil(R/32)][F] or null
  int R, F;
  unsigned x_bytes, w_bytes, side_bytes;
  // LayerNorm fused at both ends (LN instantiations only; null = off): X holds RAW rows that are normalised with (ln_g, ln_b) on their way
  // into the S-waves' registers; the finished output rows are normalised once more with (nln_g, nln_b) -- the next block's norm1
  const float* ln_g; const float* ln_b;      // [KD]
  bf16_t* ln_out; float* ln_mean; float* ln_rstd;          // normalised X rows [R, KD] and their statistics [R] (what a backward keeps), or null
  const float* nln_g; const float* nln_b;    // [KD]
  bf16_t* nln_out; float* nln_mean; float* nln_rstd;       // LayerNorm of `out` [R, KD] (+ statistics, or null)
  float ln_eps;
  // MODE 2 with LN (dig_mlp_chain_bwd_ln): the LayerNorm BACKWARD of norm2 fused behind the data gradient.  X = dy is also the gradient of the
  // residual path; resid = the rows norm2 normalised (x_mid); ln_g = its gamma; out receives dx_mid = dy + LN'(dX) instead of dX
  const float* lnb_mean; const float* lnb_rstd;            // [R] statistics the forward kept
  float* lnb_ws;                                           // [ceil(R/128)][3][KD] fp32 partial column sums: d(gamma), d(beta), column sums of dy
  // forward with DROP: out = resid + drop_path(dropout(fc2(.) + b2)) (Mlp.drop behind fc2 and the block's drop_path, modeling_finetune.py:59,158)
  dig_dropout_t drop;
  // ... and the attention projection's data gradient behind that (optional): proj_out[R, KD] = out Wproj, projt = Wproj^T [KD in][KD out]
  const bf16_t* projt; bf16_t* proj_out;
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_write8(unsigned addr, uint2 v) {
  // an LDS store the compiler does not see: a visible one would be ordered behind the LDS-DMA in flight with s_waitcnt vmcnt(0)
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }

// NOTE on LDS reads: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of an LDS load that carries no alias information while an
// LDS-DMA is in flight; loads through ext_vector types (bf16x8, bf16x4, f32x4, dig_u32x4) carry TBAA tags and are left alone --
// every LDS read below is of that kind (checked in the ISA: no vmcnt(0) inside the tick loops).  LDS writes are inline asm.
// MODE 0: forward, no side outputs (momentum branch / evaluation);  1: forward + pre-activation and GELU output (online branch:
// what the backward reads);  2: backward (data gradient through both layers + d(pre-activation) + fc1 bias-gradient partials)
// The third parameter is DROP for the forward forms (dropout / drop-path in the epilogue) and, for MODE 2 -- where there is no dropout: the
// backward takes the masked gradient as its input -- PROJ: the opt-in projection phase behind norm2's backward (dig_mlp_chain_bwd_ln_proj).
// Its own instantiation: compiled into the default backward kernel, the phase's 22 spilled registers gave the whole kernel a scratch
// segment (private_segment_fixed_size 92) that the default path never touched but every wave had set up.
template <int MODE, bool LN, bool DROP_OR_PROJ>
__global__ __launch_bounds__(512) void mlp_chain_kernel(ChainParams p) {
  constexpr bool DROP = DROP_OR_PROJ && MODE != 2;
  constexpr bool PROJ = DROP_OR_PROJ && MODE == 2;
  static_assert(!PROJ || LN, "the projection phase follows norm2's backward");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool LNB = LN && MODE == 2;                             // backward with norm2's backward in the O-waves' epilogue
  // Lab switch (off): VMEM operations retire in issue order on gfx950 (one vmcnt for loads and stores), so a wave that waits (counted) for a ring
  // piece it issued BEHIND its side-output stores also waits for those stores' acknowledgements.  SDMA lets the S-waves (which never store in
  // the loop) bring all pieces of both rings; the O-waves then wait on no vmcnt in the loop.  Measured: no gain (see DIG_CHAIN_SDMA above).
  constexpr bool SDMA = (MODE == 1) && (DIG_CHAIN_SDMA != 0);
  constexpr int LNB_RS = KD * 2 + 16;                               // LNB: row pitch of the data-gradient tile [BM][KD] bf16 at LDS offset 0
  constexpr int LNB_COLRED = BM * LNB_RS;                           //      [8 waves][3][KD] fp32 column sums
  static_assert(!LNB || LNB_COLRED + 8 * 3 * KD * 4 <= X_OFF + 2 * SLOT, "LayerNorm-backward phase: LDS map");
  constexpr int B1S_OFF = X_OFF + (MODE == 1 ? SLOT : 0);          // forward: b1 [F], b2 [KD] (, LN: ln_g, ln_b, nln_g, nln_b [KD] each) fp32 in LDS
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3, role = wave >> 2;
  const int hi = lane >> 5, rr = lane & 31;
  const int m0 = blockIdx.x * BM;
  const int F = p.F;
  const int NC = F / FC;
  DIG_CHAIN_T_BEGIN()

  const auto rB1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.B1, 0, p.w_bytes, 0x00020000);
  const auto rB2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.B2, 0, p.w_bytes, 0x00020000);
  const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
  const auto rSide1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.side1, 0, p.side1 ? p.side_bytes : 0, 0x00020000);


  // ---- weight rings: per-thread source offsets (the swizzle lives on the source side: LDS-DMA destinations are lane-linear)
  // (SDMA: an S-wave also brings the pieces of wave + 4 -- sub = 1: thread tid + 256 of the same image)
  unsigned v1[4], v2[4], vp[2];                                     // (sized by a constant: `v1[SDMA ? 4 : 2]` made the HOST pass drop every kernel stub of this file, silently)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
#pragma unroll
    for (int sub = 0; sub < (SDMA ? 2 : 1); ++sub) {
      const int pc = it * 512 + tid + 256 * sub;
      const int r1 = pc >> 4, c1 = (pc & 15) ^ (r1 & 15);            // B1 slot: [64 f][128 k], 16 chunks of 16 B per row
      v1[it + 2 * sub] = (unsigned)((r1 * KD + 8 * c1) * 2);
      const int r2 = pc >> 3, c2 = (pc & 7) ^ ((r2 >> 1) & 7);       // B2 slot: [128 j][64 f], 8 chunks per row
      v2[it + 2 * sub] = (unsigned)((r2 * F + 8 * c2) * 2);
      if (sub == 0) vp[it] = (unsigned)(((size_t)(m0 + r2) * F + 8 * c2) * 2);     // MODE 2 pre tile: [128 tokens][64 f], same image as P
    }
  }
  auto issue_ring = [&](int slot, unsigned s1, unsigned s2) {
    if (SDMA && wave >= 4) return;
#pragma unroll
    for (int sub = 0; sub < (SDMA ? 2 : 1); ++sub) {
      unsigned char* d1 = smem + W1_RING + slot * SLOT + (wave + 4 * sub) * 1024;
      unsigned char* d2 = smem + W2_RING + slot * SLOT + (wave + 4 * sub) * 1024;
#pragma unroll
      for (int it = 0; it < 2; ++it) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB1, LDS_PTR(d1 + it * 8192), 16, v1[it + 2 * sub], s1, 0, 0);
#pragma unroll
      for (int it = 0; it < 2; ++it) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB2, LDS_PTR(d2 + it * 8192), 16, v2[it + 2 * sub], s2, 0, 0);
    }
  };
  // Barrier of tick (c, TAU): this wave's DMA pieces of the tick have landed (NW = VMEM operations it has issued behind them), its
  // own LDS traffic is complete.  The DMA for the tick after next (ring slot = position of that tick in its chunk) is then issued
  // piece by piece BETWEEN the tick's MFMA groups (dma_piece: an LDS-DMA costs its wave 40-70 issue cycles, tools/experiments/
  // chain_lab.hip), behind the tick's other VMEM operations (side-output stores, the backward's pre-activation tile).
  unsigned dma_s1 = 0, dma_s2 = 0;
  auto tick_sync = [&](auto tau_tag, auto nw_tag, int c) {
    constexpr int TAU = decltype(tau_tag)::value;
    DIG_CHAIN_T(3)
    if constexpr (SDMA) {
      if (wave < 4) wait_vm<8>();                                     // (an S-wave's eight pieces of the tick after this one stay in flight)
    } else {
      wait_vm<decltype(nw_tag)::value>();
    }
    DIG_CHAIN_T(0)
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DIG_CHAIN_T(1)
    constexpr int tn = (TAU + 2) % 3;
    const int cn = TAU == 0 ? c : c + 1;
    const int d1 = cn < NC ? cn : 0;                                  // past the end: re-read a valid piece into the free slot
    const int d2 = (cn >= 2 && cn < NC + 2) ? cn - 2 : 0;
    dma_s1 = (unsigned)((d1 * FC * KD + tn * 128) * 2);
    dma_s2 = (unsigned)((tn * 128 * F + d2 * FC) * 2);
    if (MODE == 2 && TAU == 0) {                                      // pre-activation tile of chunk c (used in period c + 1)
      const int dc = c < NC ? c : NC - 1;
      unsigned char* dp = smem + X_OFF + (c & 1) * SLOT + wave * 1024;
#pragma unroll
      for (int it = 0; it < 2; ++it) __builtin_amdgcn_raw_ptr_buffer_load_lds(rSide1, LDS_PTR(dp + it * 8192), 16, vp[it], (unsigned)(dc * FC * 2), 0, 0);
    }
    DIG_CHAIN_T(2)
  };
  auto dma_piece = [&](auto tau_tag, auto k_tag) {                    // piece k (0..3) of the DMA issued during tick TAU
    constexpr int TAU = decltype(tau_tag)::value, K = decltype(k_tag)::value;
    constexpr int tn = (TAU + 2) % 3;
    if (DIG_CHAIN_ABL & 2) return;
    if (SDMA) return;                                                 // (the S-waves' dma_piece8 below brings everything)
    if (K < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB1, LDS_PTR(smem + W1_RING + tn * SLOT + wave * 1024 + K * 8192), 16, v1[K], dma_s1, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB2, LDS_PTR(smem + W2_RING + tn * SLOT + wave * 1024 + (K - 2) * 8192), 16, v2[K - 2], dma_s2, 0, 0);
  };
  auto dma_piece8 = [&](auto tau_tag, auto s_tag) {                   // SDMA, S-waves: piece S (0..7) of the tick = (K = S >> 1, sub = S & 1), one per k-step
    constexpr int TAU = decltype(tau_tag)::value, S8 = decltype(s_tag)::value, K = S8 >> 1, SUB = S8 & 1;
    constexpr int tn = (TAU + 2) % 3;
    if (DIG_CHAIN_ABL & 2) return;
    if (K < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB1, LDS_PTR(smem + W1_RING + tn * SLOT + (wave + 4 * SUB) * 1024 + K * 8192), 16, v1[K + 2 * SUB], dma_s1, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB2, LDS_PTR(smem + W2_RING + tn * SLOT + (wave + 4 * SUB) * 1024 + (K - 2) * 8192), 16, v2[K - 2 + 2 * SUB], dma_s2, 0, 0);
  };
  constexpr int E_ALL = MODE == 2 ? 2 : 0;                            // VMEM operations every wave issues behind the ring DMA of a tick 0
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using TRUE_ = std::true_type;
  using FALSE_ = std::false_type;

  // Start-up order: the ring DMA of ticks (0,0) and (0,1) first, then the bias vectors b1 [F] and b2 [KD] -> LDS (stores in inline asm: a
  // visible LDS store would drain the DMA just issued; the barrier of tick (0,0) publishes them), then the S-waves' X rows.
  issue_ring(0, 0u, 0u);
  issue_ring(1, (unsigned)(128 * 2), (unsigned)((128 * F) * 2));
  // (the vector values are requested first, the S-waves' X rows behind them, and only then are the values written to LDS: the wait in front
  //  of the first LDS store then covers the ring DMA and these few loads, not the X rows.  Both steps live inside the role branches, so
  //  that the values are branch-local: kept across the role split they were spilled around the O-waves' loop.)
  float bv[4], bv2 = 0.f, lnv[4];
  auto load_vectors = [&]() {
    if (MODE == 2) return;
    const auto rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias1, 0, p.bias1 ? F * 4 : 0, 0x00020000);     // null: zeros
    const auto rb2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias2, 0, p.bias2 ? KD * 4 : 0, 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u) bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb1, (unsigned)((u * 512 + tid) * 4), 0, 0));
    bv2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb2, (unsigned)(tid * 4), 0, 0));
    if (LN) {
      const float* const src[4] = {p.ln_g, p.ln_b, p.nln_g, p.nln_b};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const auto rl = __builtin_amdgcn_make_buffer_rsrc((void*)src[u], 0, src[u] ? KD * 4 : 0, 0x00020000);
        lnv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (unsigned)(tid * 4), 0, 0));
      }
    }
  };
  // the vector values go to LDS behind the S-waves' X-row requests (stores in inline asm: a visible LDS store would drain the DMA just
  // issued); the barrier of tick (0, 0) publishes b1 / b2, the LayerNorm vectors need their own barrier (the S-waves read them first)
  auto stage_vectors = [&]() {
    if (MODE == 2) return;
    const auto rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias1, 0, p.bias1 ? F * 4 : 0, 0x00020000);
    const unsigned base = lds_addr(smem + B1S_OFF);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u * 512 + tid < F) asm volatile("ds_write_b32 %0, %1" ::"v"(base + (unsigned)((u * 512 + tid) * 4)), "v"(bv[u]) : "memory");
    for (int i0 = 2048; i0 < F; i0 += 2048) {                          // (hidden widths beyond 2048)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb1, (unsigned)((i0 + u * 512 + tid) * 4), 0, 0));
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * 512 + tid < F) asm volatile("ds_write_b32 %0, %1" ::"v"(base + (unsigned)((i0 + u * 512 + tid) * 4)), "v"(v[u]) : "memory");
    }
    if (tid < KD) asm volatile("ds_write_b32 %0, %1" ::"v"(base + (unsigned)((F + tid) * 4)), "v"(bv2) : "memory");
    if (LN) {
      if (tid < KD) {
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("ds_write_b32 %0, %1" ::"v"(base + (unsigned)((F + (u + 1) * KD + tid) * 4)), "v"(lnv[u]) : "memory");
      }
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };

  // ---- LNB: the rows of the LayerNorm-backward phase behind the role split.  Wave w owns tokens 16 w .. 16 w + 15 of the workgroup; a lane holds
  // the column pairs q = lane + 64 j (j < 3) of each: dword loads and stores of 256 contiguous bytes.  x_mid and dy rows are requested as soon
  // as a wave has left the pipeline (S-waves: in front of the barrier they wait at for the O-waves; O-waves: behind their tile stores).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  unsigned xr[LNB ? 16 : 1][3], yr[LNB ? 16 : 1][3];
  float ln_m = 0.f, ln_r = 0.f;                                      // statistics of token (lane & 15) of the wave's 16
  f32x2 gm[3];                                                        // gamma of the lane's columns
  auto ln_rows_load = [&]() {
    if constexpr (LNB) {
      int t = threadIdx.x;
      asm volatile("" : "+v"(t));
      const unsigned vo = (unsigned)((t & 63) * 4);
      const int rc = min(m0 + wave * 16 + (t & 15), p.R - 1);
      ln_m = p.lnb_mean[rc];
      ln_r = p.lnb_rstd[rc];
#pragma unroll
      for (int j = 0; j < 3; ++j) gm[j] = *reinterpret_cast<const f32x2*>(p.ln_g + 2 * ((t & 63) + 64 * j));
      const auto rXm = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, p.x_bytes, 0x00020000);      // rows beyond R: zeros
      const auto rDy = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((m0 + wave * 16 + i) * (KD * 2));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (DIG_CHAIN_LNB_ABL & 2) { xr[i][j] = yr[i][j] = 0x3f803f80u; continue; }
          xr[i][j] = __builtin_amdgcn_raw_buffer_load_b32(rXm, vo + (unsigned)(256 * j), so, 0);
          yr[i][j] = __builtin_amdgcn_raw_buffer_load_b32((DIG_CHAIN_LNB_ABL & 1) ? rXm : rDy, vo + (unsigned)(256 * j), so, 0);
        }
      }
    }
  };

  const int psw = (rr >> 1) & 7;
  if (DIG_CHAIN_PRIO == 1 && role == 0) __builtin_amdgcn_s_setprio(1);
  if (DIG_CHAIN_PRIO == 2 && role == 1) __builtin_amdgcn_s_setprio(1);
  if (role == 0) {
    // =========================================================== S-wave =====================================================
    load_vectors();
    dig_u32x4 xf[KD / 16];                                              // the 32 x D slice of X as MFMA B fragments (96 VGPRs), as dword vectors
    const unsigned xo = (unsigned)(((size_t)(m0 + pair * 32 + rr) * KD + hi * 8) * 2);         // rows beyond R read as zero
#pragma unroll
    for (int s = 0; s < KD / 16; ++s) xf[s] = __builtin_amdgcn_raw_buffer_load_b128(rX, xo + s * 32, 0, 0);
    stage_vectors();
    if (LN && MODE != 2 && p.ln_g) {
      // LayerNorm of the lane's token on the way in.  The row is split between lanes rr and rr + 32 (k = 16 s + 8 hi ..): statistics by
      // v_dot2 on the packed pairs (sum and sum of squares, fp32), one cross-half exchange, then y = (x - mean) rstd g + b in place.
      float s1 = 0.f, s2 = 0.f;
      const dig_bf16x2 ones = __builtin_bit_cast(dig_bf16x2, 0x3F803F80u);
#pragma unroll
      for (int s = 0; s < KD / 16; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const dig_bf16x2 pr = dig_as_bf16x2(xf[s][q]);
          s1 = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, s1, false);
          s2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2, false);
        }
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const float mean = s1 * (1.0f / KD);
      const float rs = rsqrtf(fmaxf(s2 * (1.0f / KD) - mean * mean, 0.f) + p.ln_eps);
      const float* gs = reinterpret_cast<const float*>(smem + B1S_OFF) + F + KD + 8 * hi;
#pragma unroll
      for (int s = 0; s < KD / 16; ++s) {
        dig_u32x4 w = xf[s];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(gs + 16 * s + 4 * h);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(gs + KD + 16 * s + 4 * h);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const unsigned u = w[2 * h + q];
            const float y0 = fmaf(__uint_as_float(u << 16) - mean, rs * g4[2 * q], b4[2 * q]);
            const float y1 = fmaf(__uint_as_float(u & 0xffff0000u) - mean, rs * g4[2 * q + 1], b4[2 * q + 1]);
            w[2 * h + q] = pack_bf2(y0, y1);
          }
        }
        xf[s] = w;
      }
      if (p.ln_out) {
        const auto rLn = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_out, 0, p.x_bytes, 0x00020000);   // rows beyond R: dropped
#pragma unroll
        for (int s = 0; s < KD / 16; ++s) __builtin_amdgcn_raw_buffer_store_b128(xf[s], rLn, xo + s * 32, 0, DIG_CHAIN_SIDE_AUX);   // (kept for the backward only)
      }
      if (p.ln_mean && hi == 0 && m0 + pair * 32 + rr < p.R) {
        p.ln_mean[m0 + pair * 32 + rr] = mean;
        p.ln_rstd[m0 + pair * 32 + rr] = rs;
      }
    }
    int a1off[KS_PER_TICK];
#pragma unroll
    for (int s = 0; s < KS_PER_TICK; ++s) a1off[s] = rr * 256 + (((2 * s + hi) ^ (rr & 15)) << 4);
    const unsigned pw_base = lds_addr(smem + P_OFF + pair * 4096 + rr * 128 + hi * 8);
    // selection matrices of the column-sum transposer (MODE 2): E_u[(hi', i)][n] = 1 iff n == 16 u + 8 (i >> 2) + 4 hi' + (i & 3)
    bf16x8 esel[2];
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) esel[u][i] = (rr == 16 * u + 8 * (i >> 2) + 4 * hi + (i & 3)) ? (short)0x3F80 : (short)0;
    }
    f32x16 Sa[2], Sb[2];

    // One chunk period of the S-wave.  The instruction order below IS the schedule: every MFMA is followed by one "gap" of
    // elementwise work for chunk c - 1 (one hidden element = ~14 VALU ops, which issue in the shadow of that MFMA and of the
    // O-wave's) and a sched_barrier, so the compiler neither clusters the VALU work nor serialises LDS read -> MFMA pairs; the
    // B1 fragments of k-step s + 1 are requested before the MFMAs of k-step s.
    auto s_period = [&](auto m1_tag, auto g_tag, f32x16 (&Sc)[2], f32x16 (&Sp)[2], int c) {
      constexpr bool M1 = decltype(m1_tag)::value, G = decltype(g_tag)::value;
      uint2 pk[8];                                                    // packed results of chunk c - 1: quad q = (block q >> 2, g = q & 3)
      uint2 pkpre[3], pkpre_t;                                        // MODE 1: pre-activation quads held over tick 0 / of the quad in progress
      bf16x4 prq[3];                                                  // MODE 2: pre-activation quads of the current tick
      float ga[4];
      auto quad_addr = [&](int q) { return pw_base + (unsigned)((q ^ psw) << 4); };          // 16-byte chunk 4 b + g = q of the token's row
      auto colsum_block = [&](int b) {                                // MODE 2: column sums over this wave's 32 tokens of block b of chunk c - 1
        if (!(G && MODE == 2)) return;
        if (!p.colsum) return;
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 a4 = make_uint4(pk[4 * b + 2 * u].x, pk[4 * b + 2 * u].y, pk[4 * b + 2 * u + 1].x, pk[4 * b + 2 * u + 1].y);
          z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), esel[u], z, 0, 0, 0);
        }
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sm += z[e];
        sm += __shfl_xor(sm, 32, 64);
        if (hi == 0) p.colsum[(size_t)(blockIdx.x * 4 + pair) * F + (c - 1) * FC + b * 32 + rr] = sm;
      };
      // Elementwise work of a chunk: 32 hidden elements per lane = 16 PAIRS (two independent dependency chains: a lone polynomial
      // chain issues at ~6.6 cycles per op, two interleaved ones at ~4), 6 / 5 / 5 per tick, placed behind the first MFMA of a
      // k-step.  Global pair P = elements (2 (P & 1), 2 (P & 1) + 1) of quad P >> 1.  The P tile is free from tick 1 on (the O-wave
      // takes chunk c - 2 out of it during tick 0): quads finished in tick 0 (0, 1 and half of 2) are held in registers.
      auto pairwork = [&](auto tau_tag, auto j_tag) {
        constexpr int TAU = decltype(tau_tag)::value, J = decltype(j_tag)::value;
        constexpr int NP = TAU == 0 ? 6 : 5, P0 = TAU == 0 ? 0 : (TAU == 1 ? 6 : 11);
        if (!(G && !(DIG_CHAIN_ABL & 1)) || J >= NP) return;
        constexpr int PG = P0 + J, q = PG >> 1, e0 = 2 * (PG & 1), b = q >> 2, g = q & 3;
        float v0 = Sp[b][4 * g + e0], v1 = Sp[b][4 * g + e0 + 1];
        if (MODE == 2) {
          dgelu2(bf2f((bf16_t)prq[q - (TAU == 0 ? 0 : (TAU == 1 ? 3 : 5))][e0]), bf2f((bf16_t)prq[q - (TAU == 0 ? 0 : (TAU == 1 ? 3 : 5))][e0 + 1]), v0, v1);
        } else {
          if (MODE == 1) {
            const unsigned w = pack_bf2(v0, v1);
            if (q < 3) { if (e0 == 0) pkpre[q].x = w; else pkpre[q].y = w; }
            else { if (e0 == 0) pkpre_t.x = w; else pkpre_t.y = w; }
          }
          gelu_fast2(v0, v1);
        }
        if (e0 == 0) pk[q].x = pack_bf2(v0, v1);
        else pk[q].y = pack_bf2(v0, v1);                                // (the quad goes to LDS at the top of the next k-step: quad_flush)
      };
      // LDS writes of finished quads.  They are inline asm (see the note on LDS reads above), so the compiler's lgkmcnt bookkeeping does
      // not know them: issued BEHIND the fragment prefetch of a k-step they made every `s_waitcnt lgkmcnt(N)` in front of an MFMA wait for
      // the prefetched fragments as well (the write is the youngest operation).  Issued at the TOP of a k-step, ahead of that k-step's
      // prefetch, they are older than everything the compiler counts and cost the MFMAs nothing.
      auto quad_flush = [&](auto tau_tag, auto s_tag) {
        constexpr int TAU = decltype(tau_tag)::value, S = decltype(s_tag)::value;
        if (!(G && !(DIG_CHAIN_ABL & 1))) return;
        if (TAU == 1 && S < 3) {                                        // the three quads tick 0 finished (the P tile was not free yet)
          lds_write8(quad_addr(S), pk[S]);
          if (MODE == 1) lds_write8(quad_addr(S) + (unsigned)(X_OFF - P_OFF), pkpre[S]);
        }
        constexpr int NP = TAU == 0 ? 6 : 5, P0 = TAU == 0 ? 0 : (TAU == 1 ? 6 : 11);
        if (TAU > 0 && S >= 1 && S <= NP) {
          constexpr int PG = P0 + S - 1, q = PG >> 1;                   // the pair k-step S - 1 worked on
          if ((PG & 1) && q >= 3) {
            lds_write8(quad_addr(q), pk[q]);
            if (MODE == 1) lds_write8(quad_addr(q) + (unsigned)(X_OFF - P_OFF), pkpre_t);
          }
        }
      };
      auto tick = [&](auto tau_tag) {
        constexpr int TAU = decltype(tau_tag)::value;
        if (G && MODE == 2) {                                         // pre-activation quads of this tick's pairs (DMA'd two periods ago)
          constexpr int Q0 = TAU == 0 ? 0 : (TAU == 1 ? 3 : 5);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            prq[k] = *reinterpret_cast<const bf16x4*>(smem + X_OFF + ((c - 1) & 1) * SLOT + pair * 4096 + rr * 128 + hi * 8 + (((Q0 + k) ^ psw) << 4));
        }
        const unsigned char* w = smem + W1_RING + TAU * SLOT;
        bf16x8 wa[2], wb[2];
        if (M1) { wa[0] = *reinterpret_cast<const bf16x8*>(w + a1off[0]); wa[1] = *reinterpret_cast<const bf16x8*>(w + 8192 + a1off[0]); }
        __builtin_amdgcn_sched_barrier(0);
        auto kstep = [&](auto s_tag, bf16x8 (&cur)[2], bf16x8 (&nxt)[2]) {
          constexpr int S = decltype(s_tag)::value;
          quad_flush(tau_tag, s_tag);
          if (M1 && S < KS_PER_TICK - 1) {
            nxt[0] = *reinterpret_cast<const bf16x8*>(w + a1off[S + 1]);
            nxt[1] = *reinterpret_cast<const bf16x8*>(w + 8192 + a1off[S + 1]);
          }
          if (M1 && !(DIG_CHAIN_ABL & 8)) Sc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], __builtin_bit_cast(bf16x8, xf[TAU * KS_PER_TICK + S]), Sc[0], 0, 0, 0);
          pairwork(tau_tag, s_tag);
          __builtin_amdgcn_sched_barrier(0);
          if (M1 && !(DIG_CHAIN_ABL & 8)) Sc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[1], __builtin_bit_cast(bf16x8, xf[TAU * KS_PER_TICK + S]), Sc[1], 0, 0, 0);
          if constexpr (SDMA) dma_piece8(tau_tag, s_tag);
          else if (S & 1) dma_piece(tau_tag, std::integral_constant<int, (S >> 1)>{});
          if (G && !(DIG_CHAIN_ABL & 1) && MODE == 2 && TAU == 1 && S == 6) colsum_block(0);
          if (G && !(DIG_CHAIN_ABL & 1) && MODE == 2 && TAU == 2 && S == 4) colsum_block(1);
          __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{}, wa, wb); kstep(std::integral_constant<int, 1>{}, wb, wa);
        kstep(std::integral_constant<int, 2>{}, wa, wb); kstep(std::integral_constant<int, 3>{}, wb, wa);
        kstep(std::integral_constant<int, 4>{}, wa, wb); kstep(std::integral_constant<int, 5>{}, wb, wa);
        kstep(std::integral_constant<int, 6>{}, wa, wb); kstep(std::integral_constant<int, 7>{}, wb, wa);
      };
      // ---- tick 0
      tick_sync(T0{}, std::integral_constant<int, 4>{}, c);
      if (M1) {
        if (MODE != 2) {
          const float* b1s = reinterpret_cast<const float*>(smem + B1S_OFF) + c * FC + 4 * hi;
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + b * 32 + g * 8);
              Sc[b][4 * g] = bv[0]; Sc[b][4 * g + 1] = bv[1]; Sc[b][4 * g + 2] = bv[2]; Sc[b][4 * g + 3] = bv[3];
            }
        } else {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) Sc[b][e] = 0.f;
        }
      }
      tick(T0{});
      // ---- tick 1: the O-wave took chunk c - 2 out of the P tile during tick 0; chunk c - 1 may go in now
      tick_sync(T1{}, std::integral_constant<int, 4 + E_ALL>{}, c);
      tick(T1{});
      // ---- tick 2
      tick_sync(T2{}, std::integral_constant<int, 4>{}, c);
      tick(T2{});
    };

    s_period(TRUE_{}, FALSE_{}, Sa, Sb, 0);
    int c = 1;
    for (; c + 1 < NC; c += 2) {
      s_period(TRUE_{}, TRUE_{}, Sb, Sa, c);
      s_period(TRUE_{}, TRUE_{}, Sa, Sb, c + 1);
    }
    // NC is even (host-checked): c == NC - 1 here, its accumulators are Sb
    s_period(TRUE_{}, TRUE_{}, Sb, Sa, c);
    s_period(FALSE_{}, TRUE_{}, Sa, Sb, c + 1);
    s_period(FALSE_{}, FALSE_{}, Sa, Sb, c + 2);
    if (DIG_CHAIN_ABL & 1) asm volatile("" ::"v"(Sa[0]), "v"(Sa[1]), "v"(Sb[0]), "v"(Sb[1]));
    DIG_CHAIN_T(3)
    DIG_CHAIN_T_END()
    if constexpr (LNB) {
      wait_vm<0>();                                                    // this wave's last (past-the-end) DMA has landed
      ln_rows_load();
      __builtin_amdgcn_s_barrier();                                    // every wave has left the rings and tiles: the O-waves store their rows there
      asm volatile("" ::: "memory");
    }
  } else {
    // =========================================================== O-wave =====================================================
    // Output accumulators start as bias + residual (the epilogue is then a convert-and-store).  The O-wave has nothing to multiply
    // during the first two chunk periods (the S-wave is two chunks ahead), so the residual rows are fetched and unpacked THERE, half
    // of the column blocks per tick (48 registers of loads in flight), instead of in front of the pipeline.
    load_vectors();
    stage_vectors();
    f32x16 D2[NJB];
    const auto rRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, (p.resid && !LNB && !DROP) ? p.x_bytes : 0, 0x00020000);   // null: zeros; rows beyond R: zeros
    // (LNB: resid holds the rows norm2 normalised -- an operand of the epilogue, not a term of the accumulators.  DROP: the residual rows
    //  are added behind the mask, in the epilogue: the accumulators start as the bias alone)
    dig_u32x4 rq[4];                                                   // residual chunks of one group (2 column blocks) in flight
    const unsigned ro16 = (unsigned)(((size_t)(m0 + pair * 32 + rr) * KD + 16 * hi) * 2);
    // group k (0..5) = column blocks 2k, 2k + 1: requested in idle tick k, unpacked in tick k + 1 (one tick of latency cover).  A lane
    // fetches 2 x 16 contiguous bytes per column block (columns 16 hi .. 16 hi + 15) instead of its own four 8-byte quads: half the
    // load instructions and row segments; v_permlane32_swap then trades the halves so that lane (token, hi) ends up with its quads
    // 8 g + 4 hi .. + 3 (the inverse of the epilogue's store shuffle).
    auto init_load = [&](auto k_tag) {
      constexpr int K = decltype(k_tag)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          rq[2 * j + c] = __builtin_amdgcn_raw_buffer_load_b128(rRes, ro16 + ((2 * K + j) * 32 + 8 * c) * 2, 0, 0);
    };
    auto init_finish = [&](auto k_tag) {
      constexpr int K = decltype(k_tag)::value;
      const float* b2s = reinterpret_cast<const float*>(smem + B1S_OFF) + F + 4 * hi;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          constexpr int jb0 = 2 * K;
          // chunk c of the lane's 32 bytes: dwords (x, y) = its low 4 columns, (z, w) = its high 4.  Lower lanes give their high half and
          // receive the upper lanes' low half: afterwards (x, y) = quad g = c, (z, w) = quad g = c + 2 on every lane.
          unsigned x = rq[2 * j + c][0], y = rq[2 * j + c][1], z = rq[2 * j + c][2], w = rq[2 * j + c][3];
          const auto s0 = __builtin_amdgcn_permlane32_swap(x, z, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(y, w, false, false);
          x = s0[0]; z = s0[1]; y = s1[0]; w = s1[1];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int g = c + 2 * h;
            const unsigned wx = h ? z : x, wy = h ? w : y;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (MODE != 2) bv = *reinterpret_cast<const f32x4*>(b2s + (jb0 + j) * 32 + g * 8);
            D2[jb0 + j][4 * g] = bv[0] + bf2f((bf16_t)(wx & 0xffff));
            D2[jb0 + j][4 * g + 1] = bv[1] + bf2f((bf16_t)(wx >> 16));
            D2[jb0 + j][4 * g + 2] = bv[2] + bf2f((bf16_t)(wy & 0xffff));
            D2[jb0 + j][4 * g + 3] = bv[3] + bf2f((bf16_t)(wy >> 16));
          }
        }
    };
    // idle tick KT (0..5): unpack the group requested a tick ago, request the next one; the last tick also waits for its own group
    auto init_step = [&](auto kt_tag) {
      constexpr int KT = decltype(kt_tag)::value;
      if (KT >= 1 && KT <= 5) init_finish(std::integral_constant<int, (KT >= 1 && KT <= 5) ? KT - 1 : 0>{});
      if (KT >= 0 && KT <= 5) init_load(std::integral_constant<int, (KT >= 0 && KT <= 5) ? KT : 0>{});
      if (KT == 5) init_finish(std::integral_constant<int, 5>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    int a2off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a2off[s] = rr * 128 + (((2 * s + hi) ^ psw) << 4);
    const auto rS0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.side0, 0, p.side0 ? p.side_bytes : 0, 0x00020000);
    constexpr int N_SIDE = MODE == 0 ? 0 : (MODE == 1 ? 8 : 4);        // side-output stores of an active period (issued in its tick 0)

    auto o_period = [&](auto on_tag, auto init_tag, int c) {
      constexpr bool ON = decltype(on_tag)::value;
      constexpr int KT0 = decltype(init_tag)::value;                  // first init tick of this period (0, 3) or -9: none
      constexpr int EX = E_ALL + (ON ? N_SIDE : 0);
      bf16x8 pf[4];
      auto jtick = [&](auto tau_tag) {
        constexpr int TAU = decltype(tau_tag)::value;
        const unsigned char* w = smem + W2_RING + TAU * SLOT;
        bf16x8 wfa[JB_PER_TICK], wfb[JB_PER_TICK];
        if (ON) {
#pragma unroll
          for (int jb = 0; jb < JB_PER_TICK; ++jb) wfa[jb] = *reinterpret_cast<const bf16x8*>(w + jb * 4096 + a2off[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // B2 fragments of k-step S + 1 are requested while the MFMAs of k-step S issue, in two halves so that the second half
        // can take over the registers the first two MFMAs have just released (24 fragment registers live instead of 32)
        auto kstep = [&](auto s_tag, bf16x8 (&cur)[JB_PER_TICK], bf16x8 (&nxt)[JB_PER_TICK]) {
          constexpr int S = decltype(s_tag)::value;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (ON && S < 3) {
              nxt[2 * h] = *reinterpret_cast<const bf16x8*>(w + (2 * h) * 4096 + a2off[S + 1]);
              nxt[2 * h + 1] = *reinterpret_cast<const bf16x8*>(w + (2 * h + 1) * 4096 + a2off[S + 1]);
            }
            if (ON) {
#pragma unroll
              for (int jb = 2 * h; jb < 2 * h + 2; ++jb)
                if (!(DIG_CHAIN_ABL & 4)) D2[TAU * JB_PER_TICK + jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[jb], pf[S], D2[TAU * JB_PER_TICK + jb], 0, 0, 0);
                else asm volatile("" ::"v"(cur[jb]));
            }
            if (h == 0) __builtin_amdgcn_sched_barrier(0);
          }
          dma_piece(tau_tag, s_tag);
          __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{}, wfa, wfb); kstep(std::integral_constant<int, 1>{}, wfb, wfa);
        kstep(std::integral_constant<int, 2>{}, wfa, wfb); kstep(std::integral_constant<int, 3>{}, wfb, wfa);
      };
      // ---- tick 0: chunk c - 2 leaves the P tile (operand fragments into registers, side outputs to HBM)
      tick_sync(T0{}, std::integral_constant<int, 4>{}, c);
      if (ON) {
        const unsigned char* pt = smem + P_OFF + pair * 4096;
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[s] = *reinterpret_cast<const bf16x8*>(pt + a2off[s]);
        if (MODE != 0) {
          const int r8 = lane >> 3, cp = lane & 7;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + r8;
            const int ch = cp ^ ((row >> 1) & 7);
            const unsigned go = (unsigned)(((size_t)(m0 + pair * 32 + row) * F + (c - 2) * FC + 8 * ch) * 2);   // rows beyond R: dropped
            const dig_u32x4 a = *reinterpret_cast<const dig_u32x4*>(pt + row * 128 + cp * 16);
            // (forward: the GELU output and the pre-activation are read again only by the backward -- non-temporal stores, aux = 2: the 400 MB of a
            //  launch do not go through write-allocated L2 lines)
            __builtin_amdgcn_raw_buffer_store_b128(a, rS0, go, 0, MODE == 1 ? DIG_CHAIN_SIDE_AUX : DIG_CHAIN_BWD_AUX);
            if (MODE == 1) {
              const dig_u32x4 b = *reinterpret_cast<const dig_u32x4*>(pt + (X_OFF - P_OFF) + row * 128 + cp * 16);
              __builtin_amdgcn_raw_buffer_store_b128(b, rSide1, go, 0, DIG_CHAIN_SIDE_AUX);
            }
          }
        }
      }
      init_step(std::integral_constant<int, KT0>{});
      __builtin_amdgcn_sched_barrier(0);                              // the side-output stores stay in front of the tick's ring DMA
      jtick(T0{});
      tick_sync(T1{}, std::integral_constant<int, 4 + EX>{}, c);
      init_step(std::integral_constant<int, KT0 + 1>{});
      jtick(T1{});
      tick_sync(T2{}, std::integral_constant<int, 4>{}, c);
      init_step(std::integral_constant<int, KT0 + 2>{});
      jtick(T2{});
    };
    o_period(FALSE_{}, std::integral_constant<int, 0>{}, 0);
    o_period(FALSE_{}, std::integral_constant<int, 3>{}, 1);
    for (int c = 2; c < NC + 2; ++c) o_period(TRUE_{}, std::integral_constant<int, -9>{}, c);
    DIG_CHAIN_T(3)
    DIG_CHAIN_T_END()

    // ---- epilogue: bias and residual are already in the accumulators.  v_permlane32_swap trades column groups between lane l and
    // l + 32 so that each lane owns 16 contiguous columns per 32-column block: 16-byte stores.
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));                                    // re-derive the lane's row here: kept live across the loop it would be spilled
    const int row = m0 + pair * 32 + (tid2 & 31), hi2 = (tid2 >> 5) & 1;
    auto store_block = [&](bf16_t* dst, unsigned (&Pk)[4][2]) {      // one 32-column block of the lane's row: 16 contiguous columns per lane
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(Pk[0][k], Pk[2][k], false, false);
        Pk[0][k] = r0[0]; Pk[2][k] = r0[1];
        const auto r1 = __builtin_amdgcn_permlane32_swap(Pk[1][k], Pk[3][k], false, false);
        Pk[1][k] = r1[0]; Pk[3][k] = r1[1];
      }
      *reinterpret_cast<uint4*>(dst) = make_uint4(Pk[0][0], Pk[0][1], Pk[2][0], Pk[2][1]);
      *reinterpret_cast<uint4*>(dst + 8) = make_uint4(Pk[1][0], Pk[1][1], Pk[3][0], Pk[3][1]);
    };
    if constexpr (LNB) {
      // ---- norm2's backward follows for all eight waves (below the role split): the data gradient goes to LDS as the bf16 rows a separate
      // LayerNorm launch would read back -- [128 tokens][KD] bf16, row pitch LNB_RS (16 bytes of padding: the 32 tokens of a wave's store
      // land in different banks), over the rings and tiles every wave has left behind the barrier.
      wait_vm<0>();                                                    // this wave's last (past-the-end) ring DMA has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned gbase = lds_addr(smem) + (unsigned)((pair * 32 + (tid2 & 31)) * LNB_RS + hi2 * 8);
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          lds_write8(gbase + (unsigned)((jb * 16 + 4 * g) * 4),
                     make_uint2(pack_bf2(D2[jb][4 * g], D2[jb][4 * g + 1]), pack_bf2(D2[jb][4 * g + 2], D2[jb][4 * g + 3])));
      ln_rows_load();
    } else
    if (row < p.R) {
      const bool lno = LN && p.nln_out != nullptr;
      float s1 = 0.f, s2 = 0.f;
      bf16_t* orow = p.out + (size_t)row * KD + hi2 * 16;
      if constexpr (DROP) {
        // ---- out = resid + drop_path(dropout(acc)), one column block at a time: keep / drop of element (row, col) from the keyed counter hash
        // of its index row * KD + col (common.h dig_drop_keep: the GEMM epilogue's rule, the same pattern bit for bit), drop-path per sample; the
        // residual rows are fetched here (the prologue's 32-byte pieces and lane exchange), the block is rounded, counted into the next
        // LayerNorm's sums and stored at once.  (As a pass of its own over the accumulators in front of the usual epilogue it cost 1-2 KB of
        // scratch: the compiler rebuilt every accumulator tuple.)
        // (the nine key words are read from the kernel-argument segment HERE: as ordinary members of p they are fetched at the kernel's entry
        //  and held across the main loop, whose scalar registers are full -- the spill costs the loop a vector register it does not have)
        typedef const __attribute__((address_space(4))) unsigned char* kptr_t;
        kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const __attribute__((address_space(4))) unsigned* kw =
            reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(ka + offsetof(ChainParams, drop));
        dig_dropout_t d;
        d.k0 = kw[0]; d.k1 = kw[1]; d.thr = kw[2]; d.scale = __uint_as_float(kw[3]);
        d.pk0 = kw[4]; d.pk1 = kw[5]; d.pthr = kw[6]; d.pscale = __uint_as_float(kw[7]); d.rows_per_sample = (int)kw[8];
        float sc = d.thr ? d.scale : 1.f;
        if (d.pthr) sc = dig_drop_keep(d.pk0, d.pk1, (unsigned)(row / d.rows_per_sample), 0u, d.pthr) ? sc * d.pscale : 0.f;
        const auto rRes2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.resid, 0, p.resid ? p.x_bytes : 0, 0x00020000);
        const unsigned ro = (unsigned)(((size_t)row * KD + 16 * hi2) * 2);
        const unsigned ebase = (unsigned)row * (unsigned)KD + 4u * (unsigned)hi2;
        const dig_bf16x2 ones = __builtin_bit_cast(dig_bf16x2, 0x3F803F80u);
        dig_u32x4 qa[2], qb[2];
        auto dload = [&](int jb, dig_u32x4 (&r)[2]) {
#pragma unroll
          for (int c = 0; c < 2; ++c) r[c] = __builtin_amdgcn_raw_buffer_load_b128(rRes2, ro + (unsigned)((jb * 32 + 8 * c) * 2), 0, 0);
        };
        auto dblock = [&](auto jb_tag, dig_u32x4 (&cur)[2], dig_u32x4 (&nxt)[2]) {
          constexpr int jb = decltype(jb_tag)::value;
          if (jb + 1 < NJB) dload(jb + 1, nxt);
          unsigned rw[4][2];                                            // the residual pairs in accumulator order: rw[g][k] = columns 8 g + 4 hi + 2 k, + 1
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(cur[c][0], cur[c][2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(cur[c][1], cur[c][3], false, false);
            rw[c][0] = s0[0]; rw[c + 2][0] = s0[1];
            rw[c][1] = s1[0]; rw[c + 2][1] = s1[1];
          }
          unsigned Pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const unsigned idx = ebase + (unsigned)(jb * 32 + 8 * g + 2 * k);
              // (thr = 0, drop-path alone: every hash passes -- no branch here)
              const float v0 = (dig_drop_keep(d.k0, d.k1, idx, 0u, d.thr) ? D2[jb][4 * g + 2 * k] * sc : 0.f) + __uint_as_float(rw[g][k] << 16);
              const float v1 = (dig_drop_keep(d.k0, d.k1, idx + 1u, 0u, d.thr) ? D2[jb][4 * g + 2 * k + 1] * sc : 0.f) +
                               __uint_as_float(rw[g][k] & 0xffff0000u);
              const unsigned u = pack_bf2(v0, v1);
              Pk[g][k] = u;
              if (lno) {
                const dig_bf16x2 pr = __builtin_bit_cast(dig_bf16x2, u);
                s1 = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, s1, false);
                s2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2, false);
                D2[jb][4 * g + 2 * k] = __uint_as_float(u << 16);
                D2[jb][4 * g + 2 * k + 1] = __uint_as_float(u & 0xffff0000u);
              }
            }
          store_block(orow + jb * 32, Pk);
          __builtin_amdgcn_sched_barrier(0);                           // (the hashes of later blocks hoisted in front cost registers)
        };
        dload(0, qa);
        dblock(std::integral_constant<int, 0>{}, qa, qb); dblock(std::integral_constant<int, 1>{}, qb, qa);
        dblock(std::integral_constant<int, 2>{}, qa, qb); dblock(std::integral_constant<int, 3>{}, qb, qa);
        dblock(std::integral_constant<int, 4>{}, qa, qb); dblock(std::integral_constant<int, 5>{}, qb, qa);
        dblock(std::integral_constant<int, 6>{}, qa, qb); dblock(std::integral_constant<int, 7>{}, qb, qa);
        dblock(std::integral_constant<int, 8>{}, qa, qb); dblock(std::integral_constant<int, 9>{}, qb, qa);
        dblock(std::integral_constant<int, 10>{}, qa, qb); dblock(std::integral_constant<int, 11>{}, qb, qa);
      } else {
        if (lno) {
          // LayerNorm of the finished rows (the next block's norm1), taken over the bf16 values that are stored -- what a separate LayerNorm
          // launch would read back: round the accumulators in place, sum and sum of squares by v_dot2 on the packed pairs
          const dig_bf16x2 ones = __builtin_bit_cast(dig_bf16x2, 0x3F803F80u);
#pragma unroll
          for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
              const unsigned u = pack_bf2(D2[jb][2 * g], D2[jb][2 * g + 1]);
              const dig_bf16x2 pr = __builtin_bit_cast(dig_bf16x2, u);
              s1 = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, s1, false);
              s2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2, false);
              D2[jb][2 * g] = __uint_as_float(u << 16);
              D2[jb][2 * g + 1] = __uint_as_float(u & 0xffff0000u);
            }
        }
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          unsigned Pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            Pk[g][0] = pack_bf2(D2[jb][g * 4], D2[jb][g * 4 + 1]);
            Pk[g][1] = pack_bf2(D2[jb][g * 4 + 2], D2[jb][g * 4 + 3]);
          }
          store_block(orow + jb * 32, Pk);
        }
      }
      if (lno) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float mean = s1 * (1.0f / KD);
        const float rs = rsqrtf(fmaxf(s2 * (1.0f / KD) - mean * mean, 0.f) + p.ln_eps);
        const float* ng = reinterpret_cast<const float*>(smem + B1S_OFF) + p.F + 3 * KD + 4 * hi2;
        bf16_t* nrow = p.nln_out + (size_t)row * KD + hi2 * 16;
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          unsigned Pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(ng + jb * 32 + g * 8);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(ng + KD + jb * 32 + g * 8);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf(D2[jb][g * 4 + e] - mean, rs * g4[e], b4[e]);
            Pk[g][0] = pack_bf2(y[0], y[1]);
            Pk[g][1] = pack_bf2(y[2], y[3]);
          }
          store_block(nrow + jb * 32, Pk);
        }
        if (p.nln_mean && hi2 == 0) { p.nln_mean[row] = mean; p.nln_rstd[row] = rs; }
      }
    }
  }
  if constexpr (LNB) {
    // =========================================== norm2's backward on the workgroup's 128 rows, all eight waves =====================
    // dx_mid = dy + rstd (gamma g - mean_c(gamma g) - xhat mean_c(gamma g xhat)), g = the data gradient in LDS.  A wave takes 16 tokens, a
    // lane 6 columns of each: the two row sums of the 16 tokens are reduced together (32 values: halves traded by v_permlane32_swap, rows by
    // v_permlane16_swap, the last 16 lanes by DPP -- 80 operations instead of 192 cross-lane round trips) and come back as wave-uniform
    // scalars; the parameter-gradient column sums (d gamma, d beta, column sums of dy = fc2's bias gradient) are plain per-lane sums over the
    // wave's tokens, summed over the eight waves through LDS in a fixed order: ONE partial row per workgroup.
    wait_lgkm0();                                                      // (O-waves: the tile stores)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int ln = t & 63;
    const int r0 = m0 + wave * 16;
    auto lo16 = [](unsigned u) { return __uint_as_float(u << 16); };
    auto up16 = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
    auto fbits = [](float f) { return __float_as_uint(f); };
    const float ln_n = -ln_m * ln_r;                                   // xhat = x rstd + nmr
    const unsigned char* grow = smem + (wave * 16) * LNB_RS + ln * 4;
    float rstd[16], nmr[16];                                           // (wave-uniform)
    float v[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      rstd[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ln_r), i));
      nmr[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ln_n), i));
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned gw = *reinterpret_cast<const unsigned*>(grow + i * LNB_RS + 256 * j);
        const float a0 = lo16(gw) * gm[j][0], a1 = up16(gw) * gm[j][1];
        s1 += a0 + a1;
        s2 = fmaf(a0, fmaf(lo16(xr[i][j]), rstd[i], nmr[i]), s2);
        s2 = fmaf(a1, fmaf(up16(xr[i][j]), rstd[i], nmr[i]), s2);
      }
      v[i] = s1; v[16 + i] = s2;
    }
    float w16[16], u8[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) {                                     // lower half of the wave: values j, upper half: values 16 + j
      const auto r = __builtin_amdgcn_permlane32_swap(fbits(v[j]), fbits(v[j + 16]), false, false);
      const unsigned a = r[0], b = r[1];
      w16[j] = __uint_as_float(a) + __uint_as_float(b);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                      // row q of 16 lanes: values 8 q + j
      const auto r = __builtin_amdgcn_permlane16_swap(fbits(w16[j]), fbits(w16[j + 8]), false, false);
      const unsigned a = r[0], b = r[1];
      float x = __uint_as_float(a) + __uint_as_float(b);
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));      // lanes ^ 1
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));      // lanes ^ 2
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));     // row_half_mirror
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));     // row_mirror
      u8[j] = x * (1.0f / KD);
    }
    const auto rOut = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.x_bytes, 0x00020000);          // rows beyond R: not written
    const unsigned vo = (unsigned)(ln * 4);
    f32x2 dg[3], db[3], dc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { dg[j] = f32x2{0.f, 0.f}; db[j] = f32x2{0.f, 0.f}; dc[j] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u8[i & 7]), 16 * (i >> 3)));
      const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u8[i & 7]), 32 + 16 * (i >> 3)));
      const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((r0 + i) * (KD * 2));   // (left to the compiler it sits in a VGPR: a waterfall loop per store)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned gw = *reinterpret_cast<const unsigned*>(grow + i * LNB_RS + 256 * j);
        const float g0 = lo16(gw), g1 = up16(gw);
        unsigned xw = xr[i][j];
        asm volatile("" : "+v"(xw));                                   // (or the first pass's 96 xhat values are kept for this one, in scratch)
        const float xh0 = fmaf(lo16(xw), rstd[i], nmr[i]), xh1 = fmaf(up16(xw), rstd[i], nmr[i]);
        const float y0 = lo16(yr[i][j]), y1 = up16(yr[i][j]);
        const float d0 = fmaf(rstd[i], fmaf(-xh0, c2, g0 * gm[j][0] - c1), y0);
        const float d1 = fmaf(rstd[i], fmaf(-xh1, c2, g1 * gm[j][1] - c1), y1);
        const unsigned ow = pack_bf2(d0, d1);
        if (!(DIG_CHAIN_LNB_ABL & 4)) __builtin_amdgcn_raw_buffer_store_b32(ow, rOut, vo + (unsigned)(256 * j), so, 0);
        else asm volatile("" ::"v"(ow));
        // the lane's own words of the tile, for the projection phase (unconditional: 48 branches here made the compiler sink the column sums
        // behind them and keep their operands in scratch)
        asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(grow) + (unsigned)(i * LNB_RS + 256 * j)), "v"(ow) : "memory");
        dg[j][0] = fmaf(g0, xh0, dg[j][0]); dg[j][1] = fmaf(g1, xh1, dg[j][1]);
        db[j][0] += g0; db[j][1] += g1;
        dc[j][0] += y0; dc[j][1] += y1;
      }
    }
    float* cr = reinterpret_cast<float*>(smem + LNB_COLRED) + wave * (3 * KD) + 2 * ln;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      *reinterpret_cast<f32x2*>(cr + 128 * j) = dg[j];
      *reinterpret_cast<f32x2*>(cr + KD + 128 * j) = db[j];
      *reinterpret_cast<f32x2*>(cr + 2 * KD + 128 * j) = dc[j];
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* cs = reinterpret_cast<const float*>(smem + LNB_COLRED);
    for (int c = t; c < 3 * KD; c += 512) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += cs[w * (3 * KD) + c];
      p.lnb_ws[(size_t)blockIdx.x * (3 * KD) + c] = a;
    }
    if constexpr (PROJ) {
      // ================================ the attention projection's data gradient on the rows just made =============================
      // dctx[128, KD] = dx_mid[128, KD] Wproj: the tile in LDS holds dx_mid (bf16, what the GEMM launch would read back).  Wave (tg, ch)
      // takes tokens 32 tg .. + 31 and output columns 192 ch .. + 191: its 32 x 384 slice of the tile goes to registers as MFMA
      // fragments (96), then the LDS is Wproj^T's -- [384 i][192 o] at a time (two halves of the reduction, 147 KB each, ring-slot swizzle,
      // fetched by LDS-DMA, every wave 18 pieces) -- and each half is 12 k-steps x 6 column blocks of MFMAs per wave.
      const int tg = wave & 3, ch = wave >> 2;
      int t3 = threadIdx.x;
      asm volatile("" : "+v"(t3));                                     // (addresses derived from here: hoisted into the LayerNorm phase they were spilled there)
      const int ln3 = t3 & 63, rr2 = ln3 & 31, hh = ln3 >> 5;
      bf16x8 xa[KD / 16];
      {
        const unsigned char* arow = smem + (tg * 32 + rr2) * LNB_RS + hh * 16;
#pragma unroll
        for (int s = 0; s < KD / 16; ++s) xa[s] = *reinterpret_cast<const bf16x8*>(arow + s * 32);
      }
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();                                    // every wave holds its rows (and has read the column sums): the LDS is free
      asm volatile("" ::: "memory");
      const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.projt, 0, KD * KD * 2, 0x00020000);
      f32x16 acc[6];
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      const int pswp = (rr2 >> 1) & 7;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int tt = 0; tt < 18; ++tt) {
          const unsigned U = (unsigned)(64 * (wave + 8 * tt) + ln3); // 16-byte unit of the half image: row n = U / 24, swizzled chunk U % 24
          const unsigned n = U / 24u, cq = U - 24u * n;
          const unsigned c = (cq & 24u) | ((cq & 7u) ^ ((n >> 1) & 7u));
          if (!(DIG_CHAIN_LNB_ABL & 16))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(smem + (wave + 8 * tt) * 1024), 16, n * (unsigned)(KD * 2) + 16u * c,
                                                     (unsigned)(half * KD), 0, 0);
        }
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the six weight fragments of k-step s + 1 are requested in front of the MFMAs of k-step s (left to the compiler, every MFMA waited
        // for its own LDS read: 19 us of the launch instead of 8)
        const unsigned char* wbase = smem + (ch * 6 * 32 + rr2) * KD;
        auto wload = [&](int s, bf16x8 (&w)[6]) {
          const int u = 2 * s + hh;                                   // the lane's 16-byte chunk of the row, ring-slot swizzle
          const unsigned char* q = wbase + (((u & 24) | ((u & 7) ^ pswp)) << 4);
#pragma unroll
          for (int j = 0; j < 6; ++j) w[j] = *reinterpret_cast<const bf16x8*>(q + j * 32 * KD);
        };
        bf16x8 wa[6], wb[6];
        if (!(DIG_CHAIN_LNB_ABL & 8)) {
          wload(0, wa);
          auto kstep = [&](auto s_tag, bf16x8 (&cur)[6], bf16x8 (&nxt)[6]) {
            constexpr int S = decltype(s_tag)::value;
            if (S + 1 < 12) wload(S + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);                         // (the requests stay in front: the scheduler moves them behind the MFMAs)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[j], xa[half * 12 + S], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          };
          kstep(std::integral_constant<int, 0>{}, wa, wb); kstep(std::integral_constant<int, 1>{}, wb, wa);
          kstep(std::integral_constant<int, 2>{}, wa, wb); kstep(std::integral_constant<int, 3>{}, wb, wa);
          kstep(std::integral_constant<int, 4>{}, wa, wb); kstep(std::integral_constant<int, 5>{}, wb, wa);
          kstep(std::integral_constant<int, 6>{}, wa, wb); kstep(std::integral_constant<int, 7>{}, wb, wa);
          kstep(std::integral_constant<int, 8>{}, wa, wb); kstep(std::integral_constant<int, 9>{}, wb, wa);
          kstep(std::integral_constant<int, 10>{}, wa, wb); kstep(std::integral_constant<int, 11>{}, wb, wa);
        }
        if (half == 0) {
          __builtin_amdgcn_s_barrier();                                // every wave is done with the first half before the second lands on it
          asm volatile("" ::: "memory");
        }
      }
      // rows out through LDS (the weight image is dead behind the barrier): lanes l and l + 32 trade column groups (store_block of the
      // O-waves) and put 16-byte pieces into the [128][KD] tile; then a wave writes 16 whole 768-byte rows -- as 32-byte pieces straight from
      // the accumulators the stores cost 16 us of the launch (every instruction touched 32 lines, a quarter of each).  Rows beyond R are dropped.
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const auto rPo = __builtin_amdgcn_make_buffer_rsrc((void*)p.proj_out, 0, p.x_bytes, 0x00020000);
      const unsigned po = lds_addr(smem) + (unsigned)((tg * 32 + rr2) * LNB_RS + hh * 32);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        unsigned Pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Pk[g][0] = pack_bf2(acc[j][4 * g], acc[j][4 * g + 1]);
          Pk[g][1] = pack_bf2(acc[j][4 * g + 2], acc[j][4 * g + 3]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto q0 = __builtin_amdgcn_permlane32_swap(Pk[0][k], Pk[2][k], false, false);
          Pk[0][k] = q0[0]; Pk[2][k] = q0[1];
          const auto q1 = __builtin_amdgcn_permlane32_swap(Pk[1][k], Pk[3][k], false, false);
          Pk[1][k] = q1[0]; Pk[3][k] = q1[1];
        }
        const unsigned co = po + (unsigned)(((ch * 6 + j) * 32) * 2);
        asm volatile("ds_write_b128 %0, %1" ::"v"(co), "v"(dig_u32x4{Pk[0][0], Pk[0][1], Pk[2][0], Pk[2][1]}) : "memory");
        asm volatile("ds_write_b128 %0, %1" ::"v"(co + 16u), "v"(dig_u32x4{Pk[1][0], Pk[1][1], Pk[3][0], Pk[3][1]}) : "memory");
      }
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(DIG_CHAIN_LNB_ABL & 32)) {
        int t4 = threadIdx.x;
        asm volatile("" : "+v"(t4));
        const unsigned lo4 = (unsigned)((t4 & 63) * 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((m0 + wave * 16 + i) * (KD * 2));
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const unsigned wv = *reinterpret_cast<const unsigned*>(smem + (wave * 16 + i) * LNB_RS + lo4 + 256 * j);
            __builtin_amdgcn_raw_buffer_store_b32(wv, rPo, lo4 + (unsigned)(256 * j), so, 0);
          }
        }
      }
    }
  }
}

template <int MODE, bool LN = false, bool DROP = false>
int launch_chain(const ChainParams& p, hipStream_t stream) {
  const int lds = X_OFF + (MODE == 0 ? (p.F + KD) * 4 : (MODE == 1 ? SLOT + (p.F + KD) * 4 : 2 * SLOT)) + (LN ? 4 * KD * 4 : 0);
#ifdef DIG_CHAIN_LDS_ALL
  const int lds_launch = 160 * 1024;                                   // (lab build: stamps live in the spare LDS)
#else
  const int lds_launch = lds;
#endif
  if (lds > 160 * 1024) return DIG_ERR_UNSUPPORTED;
  static int attr_lds[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (lds_launch > attr_lds[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain_kernel<MODE, LN, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_launch) != hipSuccess)
      return DIG_ERR_LAUNCH;
    attr_lds[dev] = lds_launch;
  }
  dig_launch(mlp_chain_kernel<MODE, LN, DROP>, dim3((p.R + BM - 1) / BM), dim3(512), lds_launch, stream, p);
  return dig_check_launch();
}

void no_layernorm(ChainParams& p) {
  p.ln_g = p.ln_b = p.nln_g = p.nln_b = nullptr;
  p.ln_out = p.nln_out = nullptr;
  p.ln_mean = p.ln_rstd = p.nln_mean = p.nln_rstd = nullptr;
  p.ln_eps = 0.f;
  p.lnb_mean = p.lnb_rstd = nullptr;
  p.lnb_ws = nullptr;
  p.projt = nullptr; p.proj_out = nullptr;
  p.drop = dig_dropout_t{};
}

int check_common(const void* x, const void* b1, const void* b2, const void* out, int R, int D, int F) {
  if (!x || !b1 || !b2 || !out || R <= 0) return DIG_ERR_ARG;
  if (D != KD || F < 2 * FC || (F % (2 * FC))) return DIG_ERR_UNSUPPORTED;       // an even number of 64-wide chunks
  if (!aligned16(x) || !aligned16(b1) || !aligned16(b2) || !aligned16(out)) return DIG_ERR_ALIGN;
  if ((size_t)R * F * 2 >= (1ull << 32) || (size_t)F * D * 2 >= (1ull << 31)) return DIG_ERR_UNSUPPORTED;
  return DIG_OK;
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_mlp_chain_supported(int D, int F) { return D == KD && F >= 2 * FC && F % (2 * FC) == 0 && F <= 6144; }

extern "C" int dig_mlp_chain_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                                 void* out, void* pre_out, void* act_out, int R, int D, int F, hipStream_t stream) {
  const int rc = check_common(x, w1, w2, out, R, D, F);
  if (rc != DIG_OK) return rc;
  if ((b1 && !aligned16(b1)) || (b2 && !aligned16(b2)) || (resid && !aligned16(resid)) || (pre_out && !aligned16(pre_out)) ||
      (act_out && !aligned16(act_out)))
    return DIG_ERR_ALIGN;
  if ((pre_out == nullptr) != (act_out == nullptr)) return DIG_ERR_ARG;             // both side outputs or none
  ChainParams p;
  p.X = (const bf16_t*)x; p.B1 = (const bf16_t*)w1; p.B2 = (const bf16_t*)w2; p.bias1 = b1; p.bias2 = b2;
  p.resid = (const bf16_t*)resid; p.out = (bf16_t*)out; p.side0 = (bf16_t*)act_out; p.side1 = (bf16_t*)pre_out; p.colsum = nullptr;
  p.R = R; p.F = F;
  p.x_bytes = (unsigned)((size_t)R * D * 2); p.w_bytes = (unsigned)((size_t)F * D * 2); p.side_bytes = (unsigned)((size_t)R * F * 2);
  no_layernorm(p);
  return pre_out ? launch_chain<1>(p, stream) : launch_chain<0>(p, stream);
}

// The same with the block's LayerNorms fused at both ends: x holds the RAW residual rows (resid = x is what is added back); they are
// normalised with (ln_g, ln_b) on the way in -- ln_out / ln_mean / ln_rstd receive what the backward keeps, or are null -- or, with ln_g
// null, x holds rows that are normalised already (resid = the raw rows).  When nln_g is given, the output rows are normalised again with
// (nln_g, nln_b) into nln_out (+ nln_mean / nln_rstd, or null).
// drop (optional): out = resid + drop_path(dropout(fc2(.) + b2)) -- Mlp.drop behind fc2 and the block's drop_path on the MLP branch
extern "C" int dig_mlp_chain_fwd_ln_dropout(const void* x, const void* resid, const float* ln_g, const float* ln_b, float eps, void* ln_out,
                                            float* ln_mean, float* ln_rstd, const void* w1, const float* b1, const void* w2, const float* b2,
                                            void* out, void* pre_out, void* act_out, const float* nln_g, const float* nln_b, void* nln_out,
                                            float* nln_mean, float* nln_rstd, int R, int D, int F, const dig_dropout_t* drop,
                                            hipStream_t stream) {
  const int rc = check_common(x, w1, w2, out, R, D, F);
  if (rc != DIG_OK) return rc;
  if ((ln_g == nullptr) != (ln_b == nullptr) || (nln_g == nullptr) != (nln_b == nullptr) || (nln_g == nullptr) != (nln_out == nullptr)) return DIG_ERR_ARG;
  if ((ln_mean == nullptr) != (ln_rstd == nullptr) || (nln_mean == nullptr) != (nln_rstd == nullptr)) return DIG_ERR_ARG;
  if (!ln_g && (ln_out || ln_mean)) return DIG_ERR_ARG;                              // x is normalised already: nothing to report about it
  if (resid && !aligned16(resid)) return DIG_ERR_ALIGN;
  if ((pre_out == nullptr) != (act_out == nullptr)) return DIG_ERR_ARG;
  if (F > 2048) return DIG_ERR_UNSUPPORTED;                                         // (LDS: the LayerNorm vectors sit behind b1 / b2)
  if ((b1 && !aligned16(b1)) || (b2 && !aligned16(b2)) || (pre_out && !aligned16(pre_out)) || (act_out && !aligned16(act_out)) ||
      (ln_out && !aligned16(ln_out)) || (nln_out && !aligned16(nln_out)))
    return DIG_ERR_ALIGN;
  ChainParams p;
  p.X = (const bf16_t*)x; p.B1 = (const bf16_t*)w1; p.B2 = (const bf16_t*)w2; p.bias1 = b1; p.bias2 = b2;
  p.resid = (const bf16_t*)resid; p.out = (bf16_t*)out; p.side0 = (bf16_t*)act_out; p.side1 = (bf16_t*)pre_out; p.colsum = nullptr;
  p.R = R; p.F = F;
  p.x_bytes = (unsigned)((size_t)R * D * 2); p.w_bytes = (unsigned)((size_t)F * D * 2); p.side_bytes = (unsigned)((size_t)R * F * 2);
  no_layernorm(p);
  p.ln_g = ln_g; p.ln_b = ln_b; p.ln_out = (bf16_t*)ln_out; p.ln_mean = ln_mean; p.ln_rstd = ln_rstd;
  p.nln_g = nln_g; p.nln_b = nln_b; p.nln_out = (bf16_t*)nln_out; p.nln_mean = nln_mean; p.nln_rstd = nln_rstd; p.ln_eps = eps;
  if (drop && (drop->thr || drop->pthr)) {
    if (drop->pthr && drop->rows_per_sample <= 0) return DIG_ERR_ARG;
    if ((size_t)R * D >= (1ull << 32)) return DIG_ERR_UNSUPPORTED;                  // (element index of the mask hash: 32 bits)
    p.drop = *drop;
    return pre_out ? launch_chain<1, true, true>(p, stream) : launch_chain<0, true, true>(p, stream);
  }
  return pre_out ? launch_chain<1, true>(p, stream) : launch_chain<0, true>(p, stream);
}

extern "C" int dig_mlp_chain_fwd_ln(const void* x, const void* resid, const float* ln_g, const float* ln_b, float eps, void* ln_out, float* ln_mean,
                                    float* ln_rstd, const void* w1, const float* b1, const void* w2, const float* b2, void* out,
                                    void* pre_out, void* act_out, const float* nln_g, const float* nln_b, void* nln_out, float* nln_mean,
                                    float* nln_rstd, int R, int D, int F, hipStream_t stream) {
  return dig_mlp_chain_fwd_ln_dropout(x, resid, ln_g, ln_b, eps, ln_out, ln_mean, ln_rstd, w1, b1, w2, b2, out, pre_out, act_out, nln_g, nln_b, nln_out,
                                      nln_mean, nln_rstd, R, D, F, nullptr, stream);
}

// dig_mlp_chain_bwd with norm2's backward behind it: dx_mid = dy + LN2'(dX) in place of dX, and the three parameter-gradient partial sums
// ([dig_mlp_chain_ln_parts(R)][3][D] fp32: d(gamma), d(beta), column sums of dy = fc2's bias gradient) for dig_layernorm_bwd_finalize_parts;
// with projt / dctx_out, the attention projection's data gradient dctx = dx_mid Wproj behind that (projt = Wproj^T, dig_transpose_bf16)
extern "C" int dig_mlp_chain_bwd_ln_proj(const void* dy, const void* w2t, const void* pre, const void* w1t, void* dpre_out, const void* x_mid,
                                         const float* ln_g, const float* ln_mean, const float* ln_rstd, void* dx_mid_out, float* colsum_partials,
                                         float* ln_partials, const void* projt, void* dctx_out, int R, int D, int F, hipStream_t stream) {
  const int rc = check_common(dy, w2t, w1t, dx_mid_out, R, D, F);
  if (rc != DIG_OK) return rc;
  if (!pre || !dpre_out || !x_mid || !ln_g || !ln_mean || !ln_rstd || !ln_partials || ((projt == nullptr) != (dctx_out == nullptr))) return DIG_ERR_ARG;
  if ((size_t)(R + BM) * D * 2 >= (1ull << 32)) return DIG_ERR_UNSUPPORTED;       // (row offsets of the LayerNorm phase are 32-bit)
  if (!aligned16(pre) || !aligned16(dpre_out) || !aligned16(x_mid) || (colsum_partials && !aligned16(colsum_partials)) ||
      (projt && (!aligned16(projt) || !aligned16(dctx_out))))
    return DIG_ERR_ALIGN;
  ChainParams p;
  p.X = (const bf16_t*)dy; p.B1 = (const bf16_t*)w2t; p.B2 = (const bf16_t*)w1t; p.bias1 = nullptr; p.bias2 = nullptr;
  p.resid = (const bf16_t*)x_mid; p.out = (bf16_t*)dx_mid_out; p.side0 = (bf16_t*)dpre_out; p.side1 = (bf16_t*)pre; p.colsum = colsum_partials;
  p.R = R; p.F = F;
  p.x_bytes = (unsigned)((size_t)R * D * 2); p.w_bytes = (unsigned)((size_t)F * D * 2); p.side_bytes = (unsigned)((size_t)R * F * 2);
  no_layernorm(p);
  p.ln_g = ln_g; p.lnb_mean = ln_mean; p.lnb_rstd = ln_rstd; p.lnb_ws = ln_partials;
  p.projt = (const bf16_t*)projt; p.proj_out = (bf16_t*)dctx_out;
  return projt ? launch_chain<2, true, true>(p, stream) : launch_chain<2, true>(p, stream);
}

extern "C" int dig_mlp_chain_bwd_ln(const void* dy, const void* w2t, const void* pre, const void* w1t, void* dpre_out, const void* x_mid,
                                    const float* ln_g, const float* ln_mean, const float* ln_rstd, void* dx_mid_out, float* colsum_partials,
                                    float* ln_partials, int R, int D, int F, hipStream_t stream) {
  return dig_mlp_chain_bwd_ln_proj(dy, w2t, pre, w1t, dpre_out, x_mid, ln_g, ln_mean, ln_rstd, dx_mid_out, colsum_partials, ln_partials, nullptr,
                                   nullptr, R, D, F, stream);
}

extern "C" int dig_mlp_chain_bwd(const void* dy, const void* w2t, const void* pre, const void* w1t, void* dpre_out, void* dx_out,
                                 float* colsum_partials, int R, int D, int F, hipStream_t stream) {
  const int rc = check_common(dy, w2t, w1t, dx_out, R, D, F);
  if (rc != DIG_OK) return rc;
  if (!pre || !dpre_out) return DIG_ERR_ARG;
  if (!aligned16(pre) || !aligned16(dpre_out) || (colsum_partials && !aligned16(colsum_partials))) return DIG_ERR_ALIGN;
  ChainParams p;
  p.X = (const bf16_t*)dy; p.B1 = (const bf16_t*)w2t; p.B2 = (const bf16_t*)w1t; p.bias1 = nullptr; p.bias2 = nullptr;
  p.resid = nullptr; p.out = (bf16_t*)dx_out; p.side0 = (bf16_t*)dpre_out; p.side1 = (bf16_t*)pre; p.colsum = colsum_partials;
  p.R = R; p.F = F;
  p.x_bytes = (unsigned)((size_t)R * D * 2); p.w_bytes = (unsigned)((size_t)F * D * 2); p.side_bytes = (unsigned)((size_t)R * F * 2);
  no_layernorm(p);
  return launch_chain<2>(p, stream);
}

// number of partial rows dig_mlp_chain_bwd_ln writes into ln_partials ([rows][3][D] fp32): one per workgroup of 128 tokens
extern "C" int dig_mlp_chain_ln_parts(int R) { return (R + BM - 1) / BM; }

// number of partial rows dig_mlp_chain_bwd writes into colsum_partials ([rows][F] fp32)
extern "C" int dig_mlp_chain_colsum_rows(int R) { return ((R + BM - 1) / BM) * 4; }

// dst[cols, rows] = src[rows, cols]^T (bf16): the K-contiguous copies of W2 / W1 the backward chain reads (1.2 MB each, once per step)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * cols + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows) dst[(size_t)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}
// the same for a list of equally shaped matrices in ONE launch (all MLP weights of an encoder: 24 launches of ~10 us -> 1)
#define DIG_TRANSPOSE_MAX 32
struct TransposeList { const bf16_t* src[DIG_TRANSPOSE_MAX]; bf16_t* dst[DIG_TRANSPOSE_MAX]; };
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(TransposeList l, int rows, int cols) {
  __shared__ bf16_t tile[64][66];
  const bf16_t* __restrict__ src = l.src[blockIdx.z];
  bf16_t* __restrict__ dst = l.dst[blockIdx.z];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(size_t)(r0 + r) * cols + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows) dst[(size_t)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}
extern "C" int dig_transpose_bf16_multi(const void* const* srcs, void* const* dsts, int count, int rows, int cols, hipStream_t stream) {
  if (!srcs || !dsts || count < 1 || count > DIG_TRANSPOSE_MAX || rows <= 0 || cols <= 0) return DIG_ERR_ARG;
  TransposeList l;
  for (int k = 0; k < count; ++k) {
    if (!srcs[k] || !dsts[k]) return DIG_ERR_ARG;
    l.src[k] = (const bf16_t*)srcs[k]; l.dst[k] = (bf16_t*)dsts[k];
  }
  hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, count), dim3(256), 0, stream, l, rows, cols);
  return dig_check_launch();
}
extern "C" int dig_transpose_bf16(const void* src, void* dst, int rows, int cols, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, stream, (const bf16_t*)src,
                     (bf16_t*)dst, rows, cols);
  return dig_check_launch();
}
