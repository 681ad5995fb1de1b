// Short-sequence attention for the recognition decoder in TRAINING: MultiHeadAttention.forward of
// models/transformer_layer.py:238-281 over whole sequences (teacher forcing, models/decoder.py:196-222) and its gradient.
//   self-attention : 25 queries x 25 keys per (sample, head), mask = causal & (key < length[sample])   (decoder.py:183-184)
//   cross-attention: 25 queries x 256 memory keys, no mask
// A few GFLOP per batch: one workgroup per (sample, head), K and V of that head in LDS (bf16, padded rows), scores / probabilities
// in LDS (fp32), plain FMA loops -- these kernels are launch- and latency-bound, not worth MFMA tiles.  Head dim 64.
// logits = (q . k) * scale; masked logits -> probability 0; lse saved for the backward, which recomputes the probabilities.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace {

constexpr int DK = 64;
constexpr int KROW = DK + 2;                   // bf16 elements per K / V row in LDS (132 B: consecutive rows hit different banks)
constexpr int MAXQ = 32;

struct SeqAttnParams {
  const bf16_t *q, *k, *v;
  int ldq, ldk, ldv;
  int Lq, Lk;
  float scale;
  int causal;
  const long long* lens;                       // per sample valid key count (self-attention), or null
  dig_dropout_t drop;                          // attention dropout on the probabilities (thr = 0: off); see csrc/common.h
};

__device__ __forceinline__ float drop_factor(const SeqAttnParams& p, int bh, int i, int j) {
  if (!p.drop.thr) return 1.f;
  return dig_drop_keep(p.drop.k0, p.drop.k1, ((unsigned)i << 16) | (unsigned)j, (unsigned)bh, p.drop.thr) ? p.drop.scale : 0.f;
}

__device__ __forceinline__ bool key_ok(const SeqAttnParams& p, int i, int j, long long len) {
  return (!p.causal || j <= i) && (!p.lens || j < len);
}

// L8 (few keys: the 25 x 25 self-attention, where a thread per key would leave 231 of 256 threads idle): 8 lanes share one key,
// 8 channels each (a wave load covers 8 whole 128-byte rows), 32 keys per pass, dot products finished by three lane shuffles.
// Work split otherwise.  Score-shaped products (S = Q K^T, dP = dO V^T, dK = dS^T Q, dV = P^T dO): a thread owns ONE key, its K / V rows
// (or dK / dV accumulators) live in registers and the query-side rows are read from LDS as broadcasts -- pure FMA streams.
// Output-shaped products (O = P V, dQ = dS K): a thread owns one channel d and every 4th query, K / V come from LDS (bf16,
// consecutive lanes = consecutive channels) and the probabilities are broadcast reads.

__device__ __forceinline__ void load_row64(const bf16_t* __restrict__ src, float (&r)[DK]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint4 v = reinterpret_cast<const uint4*>(src)[c];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[c * 8 + 2 * e] = bf2f((bf16_t)(w[e] & 0xffff)); r[c * 8 + 2 * e + 1] = bf2f((bf16_t)(w[e] >> 16)); }
  }
}

__device__ __forceinline__ void store_row64(bf16_t* __restrict__ dst, const float (&r)[DK], float scale) {
#pragma unroll
  for (int c = 0; c < 8; ++c)
    reinterpret_cast<uint4*>(dst)[c] = make_uint4(pack_bf2(r[c * 8] * scale, r[c * 8 + 1] * scale), pack_bf2(r[c * 8 + 2] * scale, r[c * 8 + 3] * scale),
                                                 pack_bf2(r[c * 8 + 4] * scale, r[c * 8 + 5] * scale), pack_bf2(r[c * 8 + 6] * scale, r[c * 8 + 7] * scale));
}

__device__ __forceinline__ float dot64(const float* __restrict__ q, const float (&k)[DK]) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                        // four independent chains (a single one is latency-bound)
#pragma unroll
  for (int d4 = 0; d4 < DK / 4; ++d4) {
    const float4 x = reinterpret_cast<const float4*>(q)[d4];
    s0 += x.x * k[4 * d4]; s1 += x.y * k[4 * d4 + 1]; s2 += x.z * k[4 * d4 + 2]; s3 += x.w * k[4 * d4 + 3];
  }
  return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ void load8(const bf16_t* __restrict__ src, float (&r)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(src);
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[2 * e] = bf2f((bf16_t)(w[e] & 0xffff)); r[2 * e + 1] = bf2f((bf16_t)(w[e] >> 16)); }
}

__device__ __forceinline__ float dot8_reduce(const float* __restrict__ q8, const float (&k)[8]) {   // 8 lanes x 8 channels -> full dot in all 8
  const float4 a = reinterpret_cast<const float4*>(q8)[0], b = reinterpret_cast<const float4*>(q8)[1];
  float s = (a.x * k[0] + a.y * k[1]) + (a.z * k[2] + a.w * k[3]) + (b.x * k[4] + b.y * k[5]) + (b.z * k[6] + b.w * k[7]);
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  return s;
}

// stage one [Lk][64] bf16 operand of this (sample, head) into LDS, 16 bytes per thread and step
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ src, int ld, int b, int h, int Lk, bf16_t* __restrict__ dst) {
  for (int e = threadIdx.x; e < Lk * 8; e += blockDim.x) {
    const int j = e >> 3, c = e & 7;
    reinterpret_cast<uint4*>(dst)[e] = reinterpret_cast<const uint4*>(src + ((size_t)b * Lk + j) * ld + h * DK)[c];
  }
}

// out[i][d] = scale_i * sum_j W[i][j] * X[j][d] for the queries i = (tid>>6), +4, ...  (W fp32 [Lq][Lk] in LDS, X bf16 [Lk][64] in LDS)
template <typename Store>
__device__ __forceinline__ void rows_times_x(const float* __restrict__ W, const bf16_t* __restrict__ X, int Lq, int Lk, Store store) {
  const int d = threadIdx.x & 63, g = threadIdx.x >> 6;
  float acc[MAXQ / 4];
#pragma unroll
  for (int u = 0; u < MAXQ / 4; ++u) acc[u] = 0.f;
  for (int j = 0; j < Lk; ++j) {
    const float x = bf2f(X[j * DK + d]);
#pragma unroll
    for (int u = 0; u < MAXQ / 4; ++u) {
      const int i = g + 4 * u;
      if (i < Lq) acc[u] += W[i * Lk + j] * x;
    }
  }
#pragma unroll
  for (int u = 0; u < MAXQ / 4; ++u) {
    const int i = g + 4 * u;
    if (i < Lq) store(i, d, acc[u]);
  }
}

// LDS: Qs [MAXQ][64] f32 | S [MAXQ][Lk] f32 | rowinv [MAXQ] f32 | Vs [Lk][64] bf16
template <bool L8>
__global__ __launch_bounds__(256) void seq_attn_fwd_kernel(SeqAttnParams p, bf16_t* __restrict__ out, int ldo, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Qs = reinterpret_cast<float*>(smem);
  float* S = Qs + MAXQ * DK;
  float* rowinv = S + MAXQ * p.Lk;
  bf16_t* Vs = reinterpret_cast<bf16_t*>(rowinv + MAXQ);
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long len = p.lens ? p.lens[b] : (long long)p.Lk;
  for (int e = tid; e < p.Lq * DK; e += blockDim.x) {
    const int i = e / DK, d = e - i * DK;
    Qs[e] = bf2f(p.q[((size_t)b * p.Lq + i) * p.ldq + h * DK + d]) * p.scale;
  }
  stage64(p.v, p.ldv, b, h, p.Lk, Vs);
  __syncthreads();
  if (L8) {
    const int ch = tid & 7;
    for (int j0 = 0; j0 < p.Lk; j0 += 32) {
      const int j = j0 + (tid >> 3);
      float k8[8];
      if (j < p.Lk) load8(p.k + ((size_t)b * p.Lk + j) * p.ldk + h * DK + ch * 8, k8);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) k8[e] = 0.f;
      }
      for (int i = 0; i < p.Lq; ++i) {
        const float sc = dot8_reduce(Qs + i * DK + ch * 8, k8);
        if (ch == 0 && j < p.Lk) S[i * p.Lk + j] = key_ok(p, i, j, len) ? sc : -INFINITY;
      }
    }
  } else {
    for (int j = tid; j < p.Lk; j += 256) {                              // logits of my key against every query
      float kr[DK];
      load_row64(p.k + ((size_t)b * p.Lk + j) * p.ldk + h * DK, kr);
      for (int i = 0; i < p.Lq; ++i) S[i * p.Lk + j] = key_ok(p, i, j, len) ? dot64(Qs + i * DK, kr) : -INFINITY;
    }
  }
  __syncthreads();
  for (int i = wave; i < p.Lq; i += 4) {                                 // one wave per query row: softmax statistics
    float m = -INFINITY;
    for (int j = lane; j < p.Lk; j += 64) m = fmaxf(m, S[i * p.Lk + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      const float e = (S[i * p.Lk + j] == -INFINITY) ? 0.f : __expf(S[i * p.Lk + j] - m);
      S[i * p.Lk + j] = e * drop_factor(p, b * gridDim.y + h, i, j);     // dropout after the normalisation by the full row sum
      s += e;
    }
    s = wave_sum(s);
    if (lane == 0) {
      rowinv[i] = 1.f / s;                                              // (drop_factor already carries 1/(1-p))
      lse[((size_t)b * gridDim.y + h) * p.Lq + i] = m + __logf(s);
    }
  }
  __syncthreads();
  rows_times_x(S, Vs, p.Lq, p.Lk, [&](int i, int d, float a) { out[((size_t)b * p.Lq + i) * ldo + h * DK + d] = f2bf(a * rowinv[i]); });
}

// LDS: Qs, Gs [MAXQ][64] f32 | P, dS [MAXQ][Lk] f32 | Ks [Lk][64] bf16
template <bool L8>
__global__ __launch_bounds__(256) void seq_attn_bwd_kernel(SeqAttnParams p, const bf16_t* __restrict__ dout, int ldo, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dk, int lddk,
                                                           bf16_t* __restrict__ dv, int lddv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Qs = reinterpret_cast<float*>(smem);
  float* Gs = Qs + MAXQ * DK;
  float* P = Gs + MAXQ * DK;
  float* dS = P + MAXQ * p.Lk;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(dS + MAXQ * p.Lk);
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long len = p.lens ? p.lens[b] : (long long)p.Lk;
  const float* lse_row = lse + ((size_t)b * gridDim.y + h) * p.Lq;
  for (int e = tid; e < p.Lq * DK; e += blockDim.x) {
    const int i = e / DK, d = e - i * DK;
    const size_t r = (size_t)b * p.Lq + i;
    Qs[e] = bf2f(p.q[r * p.ldq + h * DK + d]);
    Gs[e] = bf2f(dout[r * ldo + h * DK + d]);
  }
  stage64(p.k, p.ldk, b, h, p.Lk, Ks);
  __syncthreads();
  if (L8) {
    const int ch = tid & 7;
    for (int j0 = 0; j0 < p.Lk; j0 += 32) {                              // P[:, j] and dP[:, j]
      const int j = j0 + (tid >> 3);
      float k8[8], v8[8];
      if (j < p.Lk) {
        load8(p.k + ((size_t)b * p.Lk + j) * p.ldk + h * DK + ch * 8, k8);
        load8(p.v + ((size_t)b * p.Lk + j) * p.ldv + h * DK + ch * 8, v8);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { k8[e] = 0.f; v8[e] = 0.f; }
      }
      for (int i = 0; i < p.Lq; ++i) {
        const float sc = dot8_reduce(Qs + i * DK + ch * 8, k8) * p.scale;
        const float dp = dot8_reduce(Gs + i * DK + ch * 8, v8);
        if (ch == 0 && j < p.Lk) {
          const bool ok = key_ok(p, i, j, len);
          P[i * p.Lk + j] = ok ? __expf(sc - lse_row[i]) : 0.f;
          dS[i * p.Lk + j] = ok ? dp * drop_factor(p, b * gridDim.y + h, i, j) : 0.f;
        }
      }
    }
  } else {
    for (int j = tid; j < p.Lk; j += 256) {                                // my key: P[:, j] and dP[:, j]
      float kr[DK], vr[DK];
      load_row64(p.k + ((size_t)b * p.Lk + j) * p.ldk + h * DK, kr);
      load_row64(p.v + ((size_t)b * p.Lk + j) * p.ldv + h * DK, vr);
      for (int i = 0; i < p.Lq; ++i) {
        float pr = 0.f, dp = 0.f;
        if (key_ok(p, i, j, len)) {
          pr = __expf(dot64(Qs + i * DK, kr) * p.scale - lse_row[i]);
          dp = dot64(Gs + i * DK, vr) * drop_factor(p, b * gridDim.y + h, i, j);   // dP = mask * (dO V^T) / (1 - p)
        }
        P[i * p.Lk + j] = pr;
        dS[i * p.Lk + j] = dp;
      }
    }
  }
  __syncthreads();
  for (int i = wave; i < p.Lq; i += 4) {                                 // dS = P * (dP - sum_j P dP)
    float del = 0.f;
    for (int j = lane; j < p.Lk; j += 64) del += P[i * p.Lk + j] * dS[i * p.Lk + j];
    del = wave_sum(del);
    for (int j = lane; j < p.Lk; j += 64) dS[i * p.Lk + j] = P[i * p.Lk + j] * (dS[i * p.Lk + j] - del);
  }
  __syncthreads();
  if (L8) {
    const int ch = tid & 7;
    for (int j0 = 0; j0 < p.Lk; j0 += 32) {                              // dK[j, 8 ch] = scale * sum_i dS[i,j] Q[i, ch], dV[j, 8 ch] = sum_i P~[i,j] dO[i, ch]
      const int j = j0 + (tid >> 3);
      if (j >= p.Lk) continue;
      float ak[8], av[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { ak[e] = 0.f; av[e] = 0.f; }
      for (int i = 0; i < p.Lq; ++i) {
        const float ds = dS[i * p.Lk + j], pr = P[i * p.Lk + j] * drop_factor(p, b * gridDim.y + h, i, j);
        const float4 q0 = reinterpret_cast<const float4*>(Qs + i * DK + ch * 8)[0], q1 = reinterpret_cast<const float4*>(Qs + i * DK + ch * 8)[1];
        const float4 g0 = reinterpret_cast<const float4*>(Gs + i * DK + ch * 8)[0], g1 = reinterpret_cast<const float4*>(Gs + i * DK + ch * 8)[1];
        ak[0] += ds * q0.x; ak[1] += ds * q0.y; ak[2] += ds * q0.z; ak[3] += ds * q0.w; ak[4] += ds * q1.x; ak[5] += ds * q1.y; ak[6] += ds * q1.z; ak[7] += ds * q1.w;
        av[0] += pr * g0.x; av[1] += pr * g0.y; av[2] += pr * g0.z; av[3] += pr * g0.w; av[4] += pr * g1.x; av[5] += pr * g1.y; av[6] += pr * g1.z; av[7] += pr * g1.w;
      }
      const size_t r = (size_t)b * p.Lk + j;
      const float sc = p.scale;
      *reinterpret_cast<uint4*>(dk + r * lddk + h * DK + ch * 8) = make_uint4(pack_bf2(ak[0] * sc, ak[1] * sc), pack_bf2(ak[2] * sc, ak[3] * sc),
                                                                              pack_bf2(ak[4] * sc, ak[5] * sc), pack_bf2(ak[6] * sc, ak[7] * sc));
      *reinterpret_cast<uint4*>(dv + r * lddv + h * DK + ch * 8) = make_uint4(pack_bf2(av[0], av[1]), pack_bf2(av[2], av[3]), pack_bf2(av[4], av[5]), pack_bf2(av[6], av[7]));
    }
  } else {
    for (int j = tid; j < p.Lk; j += 256) {                                // my key: dK[j, :] = scale * sum_i dS[i,j] Q[i,:], dV[j, :] = sum_i P[i,j] dO[i,:]
      float ak[DK], av[DK];
  #pragma unroll
      for (int d = 0; d < DK; ++d) { ak[d] = 0.f; av[d] = 0.f; }
      for (int i = 0; i < p.Lq; ++i) {
        const float ds = dS[i * p.Lk + j], pr = P[i * p.Lk + j] * drop_factor(p, b * gridDim.y + h, i, j);   // dV sees the dropped P
  #pragma unroll
        for (int d4 = 0; d4 < DK / 4; ++d4) {
          const float4 qv = reinterpret_cast<const float4*>(Qs + i * DK)[d4];
          const float4 gv = reinterpret_cast<const float4*>(Gs + i * DK)[d4];
          ak[4 * d4] += ds * qv.x; ak[4 * d4 + 1] += ds * qv.y; ak[4 * d4 + 2] += ds * qv.z; ak[4 * d4 + 3] += ds * qv.w;
          av[4 * d4] += pr * gv.x; av[4 * d4 + 1] += pr * gv.y; av[4 * d4 + 2] += pr * gv.z; av[4 * d4 + 3] += pr * gv.w;
        }
      }
      const size_t r = (size_t)b * p.Lk + j;
      store_row64(dk + r * lddk + h * DK, ak, p.scale);
      store_row64(dv + r * lddv + h * DK, av, 1.f);
    }
  }
  rows_times_x(dS, Ks, p.Lq, p.Lk, [&](int i, int d, float a) { dq[((size_t)b * p.Lq + i) * lddq + h * DK + d] = f2bf(a * p.scale); });
}

// x[b*T + t, :] = emb[token[b, t], :] + pos[t, :]   (decoder.py:173-181: trg_word_emb + PositionalEncoding; dropout p = 0)
__global__ __launch_bounds__(256) void seq_embed_fwd_kernel(const long long* __restrict__ tok, const float* __restrict__ emb,
                                                            const float* __restrict__ pos, bf16_t* __restrict__ x, int n_tok, int T, int d,
                                                            int vocab) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n_tok * d) return;
  const int r = (int)(i / d), c = (int)(i - (size_t)r * d);
  long long t = tok[r];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  x[i] = f2bf(emb[(size_t)t * d + c] + pos[(size_t)(r % T) * d + c]);
}

// demb[v, :] += sum over the tokens equal to v of dx[token row, :].  One block per (vocabulary row, 256-channel chunk): the matching
// token rows are first collected IN ORDER (ballot + prefix per 256-token chunk), then every thread sums its channel over that
// list -- a fixed summation order (deterministic).  With `lens`, positions t >= lens[b] are skipped: under teacher forcing their
// dx rows are exact zeros (no loss term, causal + length masks), and the padding token would otherwise make one block sum half
// of all rows.
__global__ __launch_bounds__(256) void seq_embed_bwd_kernel(const long long* __restrict__ tok, const bf16_t* __restrict__ dx,
                                                            float* __restrict__ demb, int n_tok, int d, int T,
                                                            const long long* __restrict__ lens) {
  extern __shared__ int list[];                                         // [n_tok] matching rows, ascending
  __shared__ int wcount[4];
  __shared__ int total;
  const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n_tok; r0 += 256) {
    const int r = r0 + tid;
    bool hit = r < n_tok && tok[r] == v;
    if (hit && lens) hit = (r % T) < lens[r / T];
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int base = total;
    for (int w = 0; w < wave; ++w) base += wcount[w];
    if (hit) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
    __syncthreads();
    if (tid == 0) total += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
  const int n = total;
  if (n == 0) return;
  const int c = blockIdx.y * 256 + tid;
  if (c >= d) return;
  float a0 = 0.f, a1 = 0.f;                                             // two chains, fixed pairing: still one order
  int k = 0;
  for (; k + 1 < n; k += 2) {
    a0 += bf2f(dx[(size_t)list[k] * d + c]);
    a1 += bf2f(dx[(size_t)list[k + 1] * d + c]);
  }
  if (k < n) a0 += bf2f(dx[(size_t)list[k] * d + c]);
  demb[(size_t)v * d + c] += a0 + a1;
}

// dlogits[b,t,:] = g * (softmax(logits[b,t,:]) - onehot(target)) / B for t < length[b], else 0  (gradient of SeqCrossEntropyLoss)
__global__ __launch_bounds__(64) void seq_ce_bwd_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target,
                                                        const long long* __restrict__ length, const float* __restrict__ g, int T, int C,
                                                        float inv_B, bf16_t* __restrict__ dlogits, int ldd) {
  const int row = blockIdx.x, b = row / T, t = row - b * T, lane = threadIdx.x;
  bf16_t* out = dlogits + (size_t)row * ldd;
  if (t >= length[b]) {
    for (int c = lane; c < ldd; c += 64) out[c] = 0;
    return;
  }
  const float* x = logits + (size_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = wave_sum(s);
  const float sc = (g ? g[0] : 1.f) * inv_B, inv = 1.f / s;
  long long y = target[row];
  y = y < 0 ? 0 : (y >= C ? C - 1 : y);
  for (int c = lane; c < ldd; c += 64) out[c] = c < C ? f2bf(sc * (__expf(x[c] - m) * inv - (c == y ? 1.f : 0.f))) : (bf16_t)0;
}

size_t lds_fwd(int Lk) { return (size_t)MAXQ * DK * 4 + (size_t)MAXQ * Lk * 4 + MAXQ * 4 + (size_t)Lk * DK * 2; }
size_t lds_bwd(int Lk) { return (size_t)2 * MAXQ * DK * 4 + (size_t)2 * MAXQ * Lk * 4 + (size_t)Lk * DK * 2; }

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_seq_attn_fwd_dropout(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, float* lse,
                                        int B, int heads, int Lq, int Lk, float scale, int causal, const long long* lens,
                                        const dig_dropout_t* drop, hipStream_t stream) {
  if (!q || !k || !v || !out || !lse || B <= 0 || heads <= 0 || Lq <= 0 || Lq > MAXQ || Lk <= 0 || Lk > 512) return DIG_ERR_ARG;
  if ((ldk & 7) || (ldv & 7) || !aligned16(k) || !aligned16(v)) return DIG_ERR_ALIGN;   // 16-byte row reads
  SeqAttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, Lq, Lk, scale, causal, lens, drop ? *drop : dig_dropout_t{}};
  const size_t lds = lds_fwd(Lk);
  static size_t attr = 0;
  static size_t attr8 = 0;
  if (Lk <= 64) {                                                            // few keys: 8 lanes per key
    if (lds > attr8) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr8 = lds; }
    hipLaunchKernelGGL(seq_attn_fwd_kernel<true>, dim3(B, heads), dim3(256), lds, stream, p, (bf16_t*)out, ldo, lse);
    return dig_check_launch();
  }
  if (lds > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
  hipLaunchKernelGGL(seq_attn_fwd_kernel<false>, dim3(B, heads), dim3(256), lds, stream, p, (bf16_t*)out, ldo, lse);
  return dig_check_launch();
}

extern "C" int dig_seq_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, float* lse, int B,
                                int heads, int Lq, int Lk, float scale, int causal, const long long* lens, hipStream_t stream) {
  return dig_seq_attn_fwd_dropout(q, ldq, k, ldk, v, ldv, out, ldo, lse, B, heads, Lq, Lk, scale, causal, lens, nullptr, stream);
}

extern "C" int dig_seq_attn_bwd_dropout(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo,
                                        const float* lse, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int heads, int Lq,
                                        int Lk, float scale, int causal, const long long* lens, const dig_dropout_t* drop,
                                        hipStream_t stream) {
  if (!q || !k || !v || !dout || !lse || !dq || !dk || !dv || B <= 0 || heads <= 0 || Lq <= 0 || Lq > MAXQ || Lk <= 0 || Lk > 512) return DIG_ERR_ARG;
  if ((ldk & 7) || (ldv & 7) || !aligned16(k) || !aligned16(v)) return DIG_ERR_ALIGN;   // 16-byte row reads
  SeqAttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, Lq, Lk, scale, causal, lens, drop ? *drop : dig_dropout_t{}};
  const size_t lds = lds_bwd(Lk);
  static size_t attr = 0;
  static size_t attr8 = 0;
  if (Lk <= 64 && !(lddk & 7) && !(lddv & 7) && aligned16(dk) && aligned16(dv)) {   // few keys: 8 lanes per key (16-byte dK / dV stores)
    if (lds > attr8) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr8 = lds; }
    hipLaunchKernelGGL(seq_attn_bwd_kernel<true>, dim3(B, heads), dim3(256), lds, stream, p, (const bf16_t*)dout, ldo, lse, (bf16_t*)dq, lddq,
                       (bf16_t*)dk, lddk, (bf16_t*)dv, lddv);
    return dig_check_launch();
  }
  if (lds > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
  hipLaunchKernelGGL(seq_attn_bwd_kernel<false>, dim3(B, heads), dim3(256), lds, stream, p, (const bf16_t*)dout, ldo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk,
                     lddk, (bf16_t*)dv, lddv);
  return dig_check_launch();
}

extern "C" int dig_seq_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo,
                                const float* lse, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int heads, int Lq, int Lk,
                                float scale, int causal, const long long* lens, hipStream_t stream) {
  return dig_seq_attn_bwd_dropout(q, ldq, k, ldk, v, ldv, dout, ldo, lse, dq, lddq, dk, lddk, dv, lddv, B, heads, Lq, Lk, scale, causal, lens,
                                  nullptr, stream);
}

extern "C" int dig_seq_embed_fwd(const long long* tokens, const float* emb, const float* pos_table, void* x, int B, int T, int d, int vocab,
                                 hipStream_t stream) {
  if (!tokens || !emb || !pos_table || !x || B <= 0 || T <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  const size_t n = (size_t)B * T * d;
  hipLaunchKernelGGL(seq_embed_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tokens, emb, pos_table, (bf16_t*)x, B * T, T, d, vocab);
  return dig_check_launch();
}

extern "C" int dig_seq_embed_bwd_lens(const long long* tokens, const void* dx, float* demb, int n_tok, int d, int vocab, int T,
                                      const long long* lens, hipStream_t stream) {
  if (!tokens || !dx || !demb || n_tok <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  if (lens && (T <= 0 || n_tok % T)) return DIG_ERR_ARG;
  if ((size_t)n_tok * sizeof(int) > 60 * 1024) return DIG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(seq_embed_bwd_kernel, dim3(vocab, (d + 255) / 256), dim3(256), (size_t)n_tok * sizeof(int), stream, tokens,
                     (const bf16_t*)dx, demb, n_tok, d, T > 0 ? T : 1, lens);
  return dig_check_launch();
}

extern "C" int dig_seq_embed_bwd(const long long* tokens, const void* dx, float* demb, int n_tok, int d, int vocab, hipStream_t stream) {
  return dig_seq_embed_bwd_lens(tokens, dx, demb, n_tok, d, vocab, 0, nullptr, stream);
}

extern "C" int dig_seq_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar,
                                         int B, int T, int C, void* dlogits, int ldd, hipStream_t stream) {
  if (!logits || !target || !length || !dlogits || B <= 0 || T <= 0 || C <= 0 || ld < C || ldd < C) return DIG_ERR_ARG;
  hipLaunchKernelGGL(seq_ce_bwd_kernel, dim3(B * T), dim3(64), 0, stream, logits, ld, target, length, gscalar, T, C, 1.0f / (float)B,
                     (bf16_t*)dlogits, ldd);
  return dig_check_launch();
}
