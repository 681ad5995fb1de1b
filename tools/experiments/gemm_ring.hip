// Forward Linear layer on an LDS-DMA ring for gfx950 (MI355X):  C[R, N] = A[R, K] W[N, K]^T  (+ bias, column scale, + residual), bf16 out.
//
// Reference math: nn.Linear inside a transformer block -- the qkv projection (modeling_finetune.py:103-109, q columns pre-scaled) and the
// attention output projection + residual (:119, :156).  Both operands are read along their contiguous (k) axis.
//
// Why beside gemm_pwide_kernel (csrc/gemm.hip): the short-K layers of the encoder (K = 384: six K-steps of 64) spend as long in the
// prologue / epilogue of a tile as in its loop, and the compiler drains every LDS-DMA it can see in front of the next LDS read
// (s_waitcnt vmcnt(0)), so a workgroup never overlaps its own operand stream with its MFMAs.  This kernel is csrc/wgrad.hip's mainloop
// turned to k-contiguous operands:
//   * tile 256 rows x 128 FN columns (FN = 3: 384, FN = 2: 256), 8 waves as 2 x 4, a wave owns 128 x 32 FN = 4 x FN accumulators of
//     v_mfma_f32_32x32x16_bf16 (192 registers at FN = 3): (4 + FN) 16-byte fragment reads per 4 FN MFMAs;
//   * operands HBM / L2 -> LDS by buffer_load_dwordx4 ... lds from INLINE ASM (unseen by the compiler: no drains), a run-time ring of
//     NSTG slots x 32 k (a slot: [256 rows x 64 B of A | 128 FN rows x 64 B of W]; 64-byte row segments = half cache lines: with 16-k
//     slots and 32-byte segments the same kernel was bound by the REQUEST rate of the L2 path, 511 vs 764 TFLOP/s for the tiles of
//     csrc/gemm.hip), NSTG - 1 slots in flight, counted s_waitcnt vmcnt, one raw s_barrier per slot (two 16-k MFMA steps); the four
//     16-byte chunks of a row sit swizzled by bits 2-3 of the row (ds_read_b128 of 16 consecutive rows hits 64 distinct banks), the
//     swizzle applied on the SOURCE side of the DMA (its LDS side is lane-linear);
//   * PERSISTENT: a workgroup walks a contiguous run of tiles (row block major: the three column tiles of a row block re-read its rows
//     from the XCD's L2), and the operand stream runs AHEAD across tile boundaries -- while a tile's accumulators are converted and stored
//     the first NSTG - 1 stages of the next tile are already on their way into LDS;
//   * epilogue straight from the accumulators: v_permlane32_swap makes each lane the owner of 16 contiguous columns of its row
//     (two 16-byte stores per 32 x 32 block), bias / scale / residual applied in fp32 in that layout.
#include "common.h"
#include <type_traits>

// phase time stamps for tools/experiments/fwd_ring_lab.hip (empty in the product build)
#ifndef DIG_FR_TS
#define DIG_FR_TS_DECL()
#define DIG_FR_TS(k)
#define DIG_FR_TS_END()
#endif

namespace {

constexpr int FR_BK = 32;                 // k per ring slot (two MFMA steps of 16)

struct FrParams {
  const bf16_t* A; const bf16_t* W; bf16_t* C; const float* bias; const bf16_t* resid;
  int R, N, K, lda, ldw, ldc, ldr;
  float alpha; int alpha_cols;
  int tiles_j, n_tiles;                   // column tiles per row block; tiles in all
};

// LDS-DMA the compiler does not see: M0 = LDS destination of the wave (lane l lands at +16 l), written in the same statement that reads it
__device__ __forceinline__ void fr_dma16(unsigned lds_dst, unsigned voff, dig_u32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// the same for the first 32 lanes only (a 512-byte piece): exec is narrowed inside the statement
__device__ __forceinline__ void fr_dma16_half(unsigned lds_dst, unsigned voff, dig_u32x4 rsrc, unsigned soff) {
  unsigned long long keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_hi, 0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void fr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ dig_u32x4 fr_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  dig_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

typedef __attribute__((address_space(3))) const bf16x8* fr_lds_frag_p;
__device__ __forceinline__ bf16x8 fr_frag(unsigned addr) { return *(fr_lds_frag_p)(uintptr_t)addr; }

template <int FN, int NSTG>
__global__ __launch_bounds__(512, 2) void fwd_ring_kernel(FrParams p) {
  constexpr int TN = 128 * FN;
  constexpr int A_BYTES = 256 * 64;                               // 16 KiB: 256 rows x 32 k
  constexpr int STAGE = A_BYTES + TN * 64;                        // 40 KiB (FN 3), 32 KiB (FN 2)
  constexpr int NPA = 2, NPB = FN;                                // 1-KiB LDS-DMA pieces (16 rows x 64 B) per wave per slot: A rows, W rows
  constexpr int NPW = NPA + NPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int wa = wave >> 2, wb = wave & 3;
  const int ns = p.K / FR_BK;                                     // slots per tile (host-checked: K % 32 == 0)

  // this workgroup's run of tiles [t_beg, t_end): contiguous, row block major
  const int G = gridDim.x;
  const int t_beg = (int)(((long long)blockIdx.x * p.n_tiles) / G), t_end = (int)(((long long)(blockIdx.x + 1) * p.n_tiles) / G);
  if (t_beg >= t_end) return;

  // ---- operand stream (runs ahead of the multiplications, across tile boundaries).  A wave brings rows [32 w, 32 w + 32) of A and rows
  // [16 FN w, 16 FN (w + 1)) of W of every slot in pieces of 16 rows; lane l of a piece = row l >> 2, 16-byte position l & 3, which
  // holds the row's k chunk (l & 3) ^ bits 2-3 of the row.
  const dig_u32x4 rA = fr_rsrc(p.A, (unsigned)((size_t)p.R * p.lda * 2));
  const dig_u32x4 rW = fr_rsrc(p.W, (unsigned)((size_t)p.N * p.ldw * 2));
  const int prow = lane >> 2, pch = (lane & 3) ^ ((lane >> 4) & 3);
  // (one address register per operand: the pieces of a wave differ by wave-uniform row offsets, which ride in the scalar offset -- five
  //  per-piece address registers on top of 192 accumulators + 40 fragment registers spilled, and a reload in front of an LDS-DMA is an
  //  s_waitcnt vmcnt(0) on the whole ring)
  const unsigned pvA = (unsigned)((prow * p.lda + pch * 8) * 2), pvW = (unsigned)((prow * p.ldw + pch * 8) * 2);
  unsigned pso[NPW], pdst[NPW];
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    pso[k] = (unsigned)__builtin_amdgcn_readfirstlane((32 * wave + 16 * k) * p.lda * 2);
    pdst[k] = (unsigned)((32 * wave + 16 * k) * 64);
  }
#pragma unroll
  for (int k = 0; k < NPB; ++k) {
    pso[NPA + k] = (unsigned)__builtin_amdgcn_readfirstlane((16 * FN * wave + 16 * k) * p.ldw * 2);
    pdst[NPA + k] = (unsigned)(A_BYTES + (16 * FN * wave + 16 * k) * 64);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  int q_tile = t_beg, q_k = 0;                                     // the stream's position: tile and slot within it
  unsigned soA, soW;                                               // scalar byte offsets of that slot's A rows / W rows
  auto stream_tile = [&]() {
    const int tl = q_tile < t_end ? q_tile : t_end - 1;            // past the run's end the stream repeats the last tile (never read)
    const int rb = tl / p.tiles_j, tj = tl - rb * p.tiles_j;
    soA = (unsigned)rb * (unsigned)(256 * p.lda * 2);
    soW = (unsigned)tj * (unsigned)(TN * p.ldw * 2);
  };
  stream_tile();
  auto issue_piece = [&](unsigned slot_off, int k) {
    if (k == 0 && q_k == ns) { q_k = 0; ++q_tile; stream_tile(); }
    fr_dma16(lds0 + slot_off + pdst[k], k < NPA ? pvA : pvW, k < NPA ? rA : rW, (k < NPA ? soA : soW) + pso[k]);
    if (k == NPW - 1) { soA += 64; soW += 64; ++q_k; }
  };
#pragma unroll
  for (int q = 0; q < NSTG - 1; ++q) {
#pragma unroll
    for (int k = 0; k < NPW; ++k) issue_piece((unsigned)(q * STAGE), k);
  }

  // ---- fragments: lane (row = lane & 31, k half = hi) of MFMA step h reads the 16-byte chunk 2 h + hi of its row, which sits at
  // position (2 h + hi) ^ bits 2-3 of the row
  const int fsw = (lane >> 2) & 3;
  const unsigned fa = lds0 + (unsigned)((wa * 128 + (lane & 31)) * 64);
  const unsigned fb = lds0 + (unsigned)(A_BYTES + (wb * 32 * FN + (lane & 31)) * 64);
  const unsigned fo[2] = {(unsigned)(((0 + hi) ^ fsw) * 16), (unsigned)(((2 + hi) ^ fsw) * 16)};
  bf16x8 af[4], bfr[2][FN];
  auto read_a = [&](unsigned so, int h, int u) { af[u] = fr_frag(fa + so + fo[h] + u * 2048); };
  auto read_b = [&](unsigned so, int h, auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int v = 0; v < FN; ++v) bfr[BUF][v] = fr_frag(fb + so + fo[h] + v * 2048);
  };
  f32x16 acc[4][FN];
  // slot offsets (scalars): `sc` = the slot being multiplied, `sp` = the slot before it (free: refilled during this slot), `sn` = the next
  unsigned sp = (unsigned)((NSTG - 1) * STAGE), sc = 0u, sn = STAGE;
  auto advance = [&]() {
    sp = sc; sc = sn;
    sn = (sn == (unsigned)((NSTG - 1) * STAGE)) ? 0u : sn + STAGE;
  };
  // One barrier per slot = two MFMA steps.  Step h multiplies fragments that are in registers and requests the fragments of the next step
  // from LDS (step 1 of this slot, or step 0 of the next slot); the slot's two steps together refill the PREVIOUS slot with slot + NSTG - 1
  // of the stream.  At a slot's barrier every wave has waited for its own pieces of the NEXT slot (counted vmcnt: the NSTG - 3 younger
  // slots may still be on their way) and for its own fragment reads of this slot's first step: behind the barrier the next slot is
  // complete in LDS and the previous slot is free.
  auto step = [&](auto h_tag, auto next_tag) {
    constexpr int H = decltype(h_tag)::value;
    constexpr bool NEXT = decltype(next_tag)::value;
    using CB = std::integral_constant<int, H>;
    using NB = std::integral_constant<int, H ^ 1>;
    if (H == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      fr_wait_vm<(NSTG - 3) * NPW>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const unsigned so = H == 0 ? sc : sn;                           // where the next step's fragments are
    if (NEXT) read_b(so, H ^ 1, NB{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int v = 0; v < FN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[CB::value][v], af[u], acc[u][v], 0, 0, 0);
      if (NEXT) read_a(so, H ^ 1, u);
      // the slot's NPW pieces go out between the MFMA groups of its two steps (an LDS-DMA costs its wave 60-100 issue cycles)
      if (H == 0 && u < 3) issue_piece(sp, u);
      if (H == 1 && 3 + u < NPW) issue_piece(sp, 3 + u);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (H == 1) advance();
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  for (int tile = t_beg; tile < t_end; ++tile) {
    // everything the stream has requested so far (NSTG - 1 slots from this tile on) and the previous tile's stores are done
    fr_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(sc, 0, H0{});
#pragma unroll
    for (int u = 0; u < 4; ++u) read_a(sc, 0, u);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < FN; ++v)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][v][e] = 0.f;
    for (int t = 0; t + 1 < ns; ++t) {
      step(H0{}, std::true_type{});
      step(H1{}, std::true_type{});
    }
    step(H0{}, std::true_type{});
    step(H1{}, std::false_type{});

    // ---- epilogue.  acc[u][v]: lane = row (lane & 31), registers = columns 8 g + 4 hi + (0..3); after the swaps lane (row, hi) owns
    // columns 16 hi .. 16 hi + 15 of the 32 x 32 block
    const int rbk = tile / p.tiles_j, tj = tile - rbk * p.tiles_j;
    const int irow0 = rbk * 256 + wa * 128 + (lane & 31);
    const int jcol0 = tj * TN + wb * 32 * FN + hi * 16;
#pragma unroll
    for (int v = 0; v < FN; ++v) {
      const int j = jcol0 + v * 32;
      const bool jok = j < p.N;                                    // (N is a multiple of 16)
      float bias16[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) bias16[e] = 0.f;
      if (p.bias && jok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + j + 4 * g);
          bias16[4 * g] = b4.x; bias16[4 * g + 1] = b4.y; bias16[4 * g + 2] = b4.z; bias16[4 * g + 3] = b4.w;
        }
      }
      const float al = (tj * TN + wb * 32 * FN + v * 32) < p.alpha_cols ? p.alpha : 1.0f;       // alpha_cols is a multiple of 32
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float x[16];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          // groups g = 0, 2 and g = 1, 3 trade halves: lanes of the low half end with column groups (0, 1 | 2, 3) of THEIR 16 columns
          const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[u][v][qd]), __float_as_uint(acc[u][v][8 + qd]), false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[u][v][4 + qd]), __float_as_uint(acc[u][v][12 + qd]), false, false);
          x[qd] = __uint_as_float(s0[0]); x[4 + qd] = __uint_as_float(s0[1]);
          x[8 + qd] = __uint_as_float(s1[0]); x[12 + qd] = __uint_as_float(s1[1]);
        }
        const int i = irow0 + u * 32;
        const bool live = jok && i < p.R;
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = (x[e] + bias16[e]) * al;
        if (p.resid && live) {
          const uint4 r0 = *reinterpret_cast<const uint4*>(p.resid + (size_t)i * p.ldr + j);
          const uint4 r1 = *reinterpret_cast<const uint4*>(p.resid + (size_t)i * p.ldr + j + 8);
          const unsigned w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) { x[2 * e] += bf2f((bf16_t)(w[e] & 0xffff)); x[2 * e + 1] += bf2f((bf16_t)(w[e] >> 16)); }
        }
        if (live) {
          bf16_t* o = p.C + (size_t)i * p.ldc + j;
          *reinterpret_cast<uint4*>(o) = make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
          *reinterpret_cast<uint4*>(o + 8) = make_uint4(pack_bf2(x[8], x[9]), pack_bf2(x[10], x[11]), pack_bf2(x[12], x[13]), pack_bf2(x[14], x[15]));
        }
      }
    }
  }
  fr_wait_vm<0>();                                                  // the run-ahead pieces still on their way into LDS
}

template <int FN, int NSTG>
int launch_fwd_ring(const FrParams& p, hipStream_t stream) {
  constexpr int LDS = NSTG * (256 * 64 + 128 * FN * 64);
  static bool attr_set[DIG_MAX_DEVICES] = {};
  static int n_cu[DIG_MAX_DEVICES] = {};
  const int dev = dig_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_ring_kernel<FN, NSTG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipDeviceProp_t prop;
    n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    attr_set[dev] = true;
  }
  const int grid = p.n_tiles < n_cu[dev] ? p.n_tiles : n_cu[dev];
  dig_launch(fwd_ring_kernel<FN, NSTG>, dim3(grid), dim3(512), LDS, stream, p);
  return dig_check_launch();
}

}  // namespace

// (lab only: not part of the library -- see tools/experiments/fwd_ring_lab.hip for the measurements that kept it out)
extern "C" int dig_gemm_ring_supported(int R, int N, int K) {
  return R > 0 && N > 0 && K >= 32 && (K % 32) == 0 && (N % 16) == 0 && ((N % 384) == 0 || (N % 256) == 0) ? 1 : 0;
}

extern "C" int dig_gemm_ring_fwd(const void* A, const void* W, void* C, int R, int N, int K, int lda, int ldw, int ldc, const float* bias,
                                 const void* resid, int ldr, float alpha, int alpha_cols, hipStream_t stream) {
  if (!A || !W || !C || R <= 0 || N <= 0 || K <= 0) return DIG_ERR_ARG;
  if (!dig_gemm_ring_supported(R, N, K) || (alpha_cols % 32) != 0) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(A) || !aligned16(W) || !aligned16(C) || (resid && !aligned16(resid)) || (bias && !aligned16(bias)) || (lda & 7) || (ldw & 7) || (ldc & 7) ||
      (resid && (ldr & 7)) || lda < K || ldw < K || ldc < N)
    return DIG_ERR_ALIGN;
  if ((size_t)R * lda * 2 >= (1ull << 32) || (size_t)N * ldw * 2 >= (1ull << 32)) return DIG_ERR_UNSUPPORTED;
  FrParams p{(const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, bias, (const bf16_t*)resid, R, N, K, lda, ldw, ldc, ldr, alpha, alpha_cols, 0, 0};
  const int fn = (N % 384) == 0 ? 3 : 2;
  p.tiles_j = N / (128 * fn);
  p.n_tiles = ((R + 255) / 256) * p.tiles_j;
  return fn == 3 ? launch_fwd_ring<3, 4>(p, stream) : launch_fwd_ring<2, 5>(p, stream);
}
