// Tile helpers shared by the attention kernels (csrc/attention.hip, csrc/attn_block.hip): LDS layout "U" of a [256 x 64] bf16 operand tile,
// its two fragment reads, accumulator packing and the row store of a 32 x 64 result.  See the header of attention.hip for the layout.
#pragma once
#include "common.h"

namespace {

constexpr int N_TOK = 256;
constexpr int DH = 64;
constexpr int TILE = N_TOK * DH * 2;  // 32 KiB

__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int u_addr(int row, int col) { return row * 128 + ((((col >> 3) ^ swz(row))) << 4) + (col & 7) * 2; }

// Stage a [256 x 64] bf16 tile (row stride ld elements, starting at element offset base) into LDS layout U.
// 2048 16-B pieces; NT threads.
// STEP = 2 (NT = 256: one iteration = one 32-row block): only the even 32-row blocks
// PIECES = 512: a [64 x 64] tile (the first 64 rows of the same layout)
template <int NT, int STEP = 1, int PIECES = 2048>
__device__ __forceinline__ void stage_tile(unsigned char* lds, __amdgpu_buffer_rsrc_t rs, unsigned base_bytes, int ld,
                                           int tid, int wave) {
#pragma unroll
  for (int it = 0; it < PIECES / NT; it += STEP) {
    const int piece = it * NT + tid;
    const int row = piece >> 3, pc = piece & 7;
    const int c = pc ^ swz(row);
    const unsigned off = base_bytes + (unsigned)((row * ld + c * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + (it * NT + wave * 64) * 16), 16, off, 0, 0, 0);
  }
}

// 8 contiguous bf16 (cols 16*s + 8*hi ..) of row (rowoff + lane&31)
__device__ __forceinline__ bf16x8 frag_direct(const unsigned char* tile, int rowoff, int s, int lane) {
  const int row = rowoff + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((chunk ^ swz(row)) << 4));
}

// Transposed fragment: for column (coloff + lane&31) return rows {r0 + 4*hi + 0..3, r0 + 8 + 4*hi + 0..3}
// (the row permutation of a 32x32 MFMA accumulator quad pair).
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* tile, int r0, int coloff, int lane) {
  const int hi = lane >> 5;
  const int i = lane & 15;
  const int col = coloff + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
  const int ra = r0 + 4 * hi + (i >> 2);
  const int rb = ra + 8;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + u_addr(ra, col)));
  bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + u_addr(rb, col)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
  return r;
}

// Loop forms of the two fragment reads.  Inside a loop over 32-row tiles the XOR swizzle of layout U depends only on the low
// row bits, i.e. on the lane: the byte offsets below are computed once per kernel and a tile is `base + tile * 4096` plus
// instruction immediates (the generic helpers recomputed ~40 VALU address instructions per tile).
struct FragOff {
  int d[4];        // frag_direct: 16-B chunk (2s + hi) of row (lane & 31), s = 0..3
  int t[2][2];     // frag_tr: [dt][row half a / b] for r0 = 0, coloff = 32 * dt
};
__device__ __forceinline__ FragOff frag_offsets(int lane) {
  FragOff o;
  const int row = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int s = 0; s < 4; ++s) o.d[s] = row * 128 + (((2 * s + hi) ^ swz(row)) << 4);
  const int i = lane & 15;
  const int ra = 4 * hi + (i >> 2);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int col = dt * 32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
    o.t[dt][0] = u_addr(ra, col);
    o.t[dt][1] = u_addr(ra + 8, col);
  }
  return o;
}
__device__ __forceinline__ bf16x8 frag_direct_o(const unsigned char* tile32, const FragOff& o, int s) {      // tile32 = tile + 32-row block * 4096
  return *reinterpret_cast<const bf16x8*>(tile32 + o.d[s]);
}
__device__ __forceinline__ bf16x8 frag_tr_o(const unsigned char* tile16, const FragOff& o, int dt) {         // tile16 = tile + 16-row block * 2048
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile16 + o.t[dt][0]));
  bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile16 + o.t[dt][1]));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = h4[0]; r[5] = h4[1]; r[6] = h4[2]; r[7] = h4[3];
  return r;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int u) {
  const uint4 w = make_uint4(pack_bf2(a[u * 8], a[u * 8 + 1]), pack_bf2(a[u * 8 + 2], a[u * 8 + 3]),
                             pack_bf2(a[u * 8 + 4], a[u * 8 + 5]), pack_bf2(a[u * 8 + 6], a[u * 8 + 7]));
  return __builtin_bit_cast(bf16x8, w);
}

// ------------------------------------------------------------------------------------------------
// Store one lane-row of a 32 x 64 result held as two 32x32 MFMA accumulators (lane = row, registers = 4-column groups
// interleaved between the two lane halves).  v_permlane32_swap trades column groups between lane l and l+32 so that each
// lane owns 16 contiguous columns per accumulator: 4 x 16-byte stores per row instead of 8 x 8-byte ones.
template <bool NT = false>
__device__ __forceinline__ void store_rows(bf16_t* row, const f32x16 (&acc)[2], int hi) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    unsigned P[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      P[g][0] = pack_bf2(acc[dt][g * 4], acc[dt][g * 4 + 1]);
      P[g][1] = pack_bf2(acc[dt][g * 4 + 2], acc[dt][g * 4 + 3]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(P[0][k], P[2][k], false, false);
      P[0][k] = r0[0]; P[2][k] = r0[1];
      const auto r1 = __builtin_amdgcn_permlane32_swap(P[1][k], P[3][k], false, false);
      P[1][k] = r1[0]; P[3][k] = r1[1];
    }
    bf16_t* o = row + dt * 32 + hi * 16;
    const dig_u32x4 w0 = {P[0][0], P[0][1], P[2][0], P[2][1]}, w1 = {P[1][0], P[1][1], P[3][0], P[3][1]};
    if (NT) {                                                            // results nobody re-reads through this L2 soon: no write-allocate
      __builtin_nontemporal_store(w0, reinterpret_cast<dig_u32x4*>(o));
      __builtin_nontemporal_store(w1, reinterpret_cast<dig_u32x4*>(o + 8));
    } else {
      *reinterpret_cast<dig_u32x4*>(o) = w0;
      *reinterpret_cast<dig_u32x4*>(o + 8) = w1;
    }
  }
}

// The same 32 x 64 result leaving in FULL 128-byte lines: store_rows writes 32 bytes of 32 different rows per instruction (a lane pair per row) --
// 32 partial-line write requests where 8 full lines would do -- and the attention backward's 151 MB of results cost it a third of its time that
// way (lab: 133 us with the row stores, 99 us with no stores at all, 45 us for all its loads).  The block goes through 2 KiB of wave-private LDS,
// 16 rows at a time (64-bit pieces at 16-byte position (4 dt + g) ^ (row & 7): conflict-free both ways), and is read back 8 adjacent lanes per
// row: one store instruction = 8 rows x 128 contiguous bytes.  `blk` = &T[the wave's first row][first column], ld in elements.
// SUM: the rows' column sums ride along -- lane (r8 = lane >> 3, c = lane & 7) adds the bf16 values it stores (columns 8 c .. 8 c + 7 of rows r8 and
// 8 + r8 of a pass) to cs[8]; after the caller's last block, lines_colsum_finish() folds the eight row groups of the wave.
template <bool NT = false, bool SUM = false>
__device__ __forceinline__ void store_rows_lines(bf16_t* blk, int ld, const f32x16 (&acc)[2], unsigned char* stg, int lane, float* cs = nullptr) {
  asm volatile("" : "+v"(lane));            // the addresses below are derived HERE from a copy the optimiser cannot see through: hoisted out of the
                                            // caller's loops they stay alive across its MFMA phases (a spilled register in the attention backward)
  const int rr = lane & 31, hi = lane >> 5;
  const int wr = (rr & 15) * 128 + hi * 8, ws = rr & 7;
  const int r8 = lane >> 3, c = lane & 7;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if ((rr >> 4) == pass) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(stg + wr + (((4 * dt + g) ^ ws) << 4)) =
              make_uint2(pack_bf2(acc[dt][g * 4], acc[dt][g * 4 + 1]), pack_bf2(acc[dt][g * 4 + 2], acc[dt][g * 4 + 3]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const dig_u32x4 v0 = *reinterpret_cast<const dig_u32x4*>(stg + r8 * 128 + ((c ^ (r8 & 7)) << 4));
    const dig_u32x4 v1 = *reinterpret_cast<const dig_u32x4*>(stg + (8 + r8) * 128 + ((c ^ (r8 & 7)) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    dig_u32x4* d0 = reinterpret_cast<dig_u32x4*>(blk + (size_t)(16 * pass + r8) * ld + c * 8);
    dig_u32x4* d1 = reinterpret_cast<dig_u32x4*>(blk + (size_t)(16 * pass + 8 + r8) * ld + c * 8);
    if (NT) { __builtin_nontemporal_store(v0, d0); __builtin_nontemporal_store(v1, d1); }
    else { *d0 = v0; *d1 = v1; }
    if (SUM) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cs[2 * i] += __uint_as_float(v0[i] << 16) + __uint_as_float(v1[i] << 16);
        cs[2 * i + 1] += __uint_as_float(v0[i] & 0xffff0000u) + __uint_as_float(v1[i] & 0xffff0000u);
      }
    }
  }
}
// cs[j] of lane (r8, c) -> the wave's sum over r8 = 0..7 in the lanes r8 = 0, which write out[8 c + j]  (out: 64 floats)
__device__ __forceinline__ void lines_colsum_finish(float (&cs)[8], float* out, int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = cs[j];
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8 (lane ^ 8 within a 16-lane row)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    cs[j] = v;
  }
  if ((lane >> 3) == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[8 * (lane & 7) + j] = cs[j];
  }
}

}  // namespace
