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
};

__device__ __forceinline__ bool key_ok(const SeqAttnParams& p, int i, int j, long long len) {
  return (!p.causal || j <= i) && (!p.lens || j < len);
}

__device__ __forceinline__ void load_kv(const SeqAttnParams& p, int b, int h, bf16_t* Ks, bf16_t* Vs) {
  for (int e = threadIdx.x; e < p.Lk * (DK / 2); e += blockDim.x) {
    const int j = e / (DK / 2), c = e - j * (DK / 2);
    const size_t row = (size_t)b * p.Lk + j;
    reinterpret_cast<unsigned*>(Ks + j * KROW)[c] = reinterpret_cast<const unsigned*>(p.k + row * p.ldk + h * DK)[c];
    reinterpret_cast<unsigned*>(Vs + j * KROW)[c] = reinterpret_cast<const unsigned*>(p.v + row * p.ldv + h * DK)[c];
  }
}

// S[i][j] = masked logits -> P (unnormalised exp kept with the row max / sum), all in LDS
__device__ __forceinline__ void scores(const SeqAttnParams& p, long long len, const float* Qs, const bf16_t* Ks, float* S) {
  for (int e = threadIdx.x; e < p.Lq * p.Lk; e += blockDim.x) {
    const int i = e / p.Lk, j = e - i * p.Lk;
    float s = -INFINITY;
    if (key_ok(p, i, j, len)) {
      s = 0.f;
      const float* qr = Qs + i * DK;
      const bf16_t* kr = Ks + j * KROW;
#pragma unroll 16
      for (int d = 0; d < DK; ++d) s += qr[d] * bf2f(kr[d]);
      s *= p.scale;
    }
    S[e] = s;
  }
}

__global__ __launch_bounds__(256) void seq_attn_fwd_kernel(SeqAttnParams p, bf16_t* __restrict__ out, int ldo, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + p.Lk * KROW;
  float* Qs = reinterpret_cast<float*>(Vs + p.Lk * KROW);
  float* S = Qs + MAXQ * DK;
  float* rowinv = S + MAXQ * p.Lk;
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const long long len = p.lens ? p.lens[b] : (long long)p.Lk;
  load_kv(p, b, h, Ks, Vs);
  for (int e = tid; e < p.Lq * DK; e += blockDim.x) {
    const int i = e / DK, d = e - i * DK;
    Qs[e] = bf2f(p.q[((size_t)b * p.Lq + i) * p.ldq + h * DK + d]);
  }
  __syncthreads();
  scores(p, len, Qs, Ks, S);
  __syncthreads();
  for (int i = tid >> 6; i < p.Lq; i += 4) {                            // one wave per query row: softmax statistics
    const int lane = tid & 63;
    float m = -INFINITY;
    for (int j = lane; j < p.Lk; j += 64) m = fmaxf(m, S[i * p.Lk + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < p.Lk; j += 64) {
      const float e = (S[i * p.Lk + j] == -INFINITY) ? 0.f : __expf(S[i * p.Lk + j] - m);
      S[i * p.Lk + j] = e;
      s += e;
    }
    s = wave_sum(s);
    if (lane == 0) {
      rowinv[i] = 1.f / s;
      lse[((size_t)b * gridDim.y + h) * p.Lq + i] = m + __logf(s);
    }
  }
  __syncthreads();
  for (int e = tid; e < p.Lq * DK; e += blockDim.x) {
    const int i = e / DK, d = e - i * DK;
    float a = 0.f;
    for (int j = 0; j < p.Lk; ++j) a += S[i * p.Lk + j] * bf2f(Vs[j * KROW + d]);
    out[((size_t)b * p.Lq + i) * ldo + h * DK + d] = f2bf(a * rowinv[i]);
  }
}

__global__ __launch_bounds__(256) void seq_attn_bwd_kernel(SeqAttnParams p, const bf16_t* __restrict__ dout, int ldo, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dk, int lddk,
                                                           bf16_t* __restrict__ dv, int lddv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + p.Lk * KROW;
  float* Qs = reinterpret_cast<float*>(Vs + p.Lk * KROW);
  float* Gs = Qs + MAXQ * DK;                                           // dO
  float* P = Gs + MAXQ * DK;                                            // probabilities
  float* dS = P + MAXQ * p.Lk;                                          // dP, then dS
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const long long len = p.lens ? p.lens[b] : (long long)p.Lk;
  load_kv(p, b, h, Ks, Vs);
  for (int e = tid; e < p.Lq * DK; e += blockDim.x) {
    const int i = e / DK, d = e - i * DK;
    const size_t r = (size_t)b * p.Lq + i;
    Qs[e] = bf2f(p.q[r * p.ldq + h * DK + d]);
    Gs[e] = bf2f(dout[r * ldo + h * DK + d]);
  }
  __syncthreads();
  scores(p, len, Qs, Ks, P);
  __syncthreads();
  for (int e = tid; e < p.Lq * p.Lk; e += blockDim.x) {                 // P = exp(S - lse);  dP = dO . V
    const int i = e / p.Lk, j = e - i * p.Lk;
    const float s = P[e];
    float pr = 0.f, dp = 0.f;
    if (s != -INFINITY) {
      pr = __expf(s - lse[((size_t)b * gridDim.y + h) * p.Lq + i]);
      const float* gr = Gs + i * DK;
      const bf16_t* vr = Vs + j * KROW;
#pragma unroll 16
      for (int d = 0; d < DK; ++d) dp += gr[d] * bf2f(vr[d]);
    }
    P[e] = pr;
    dS[e] = dp;
  }
  __syncthreads();
  for (int i = tid >> 6; i < p.Lq; i += 4) {                            // dS = P * (dP - sum_j P dP)
    const int lane = tid & 63;
    float del = 0.f;
    for (int j = lane; j < p.Lk; j += 64) del += P[i * p.Lk + j] * dS[i * p.Lk + j];
    del = wave_sum(del);
    for (int j = lane; j < p.Lk; j += 64) dS[i * p.Lk + j] = P[i * p.Lk + j] * (dS[i * p.Lk + j] - del);
  }
  __syncthreads();
  for (int e = tid; e < p.Lq * DK; e += blockDim.x) {                   // dQ = scale * dS K
    const int i = e / DK, d = e - i * DK;
    float a = 0.f;
    for (int j = 0; j < p.Lk; ++j) a += dS[i * p.Lk + j] * bf2f(Ks[j * KROW + d]);
    dq[((size_t)b * p.Lq + i) * lddq + h * DK + d] = f2bf(a * p.scale);
  }
  for (int e = tid; e < p.Lk * DK; e += blockDim.x) {                   // dK = scale * dS^T Q ;  dV = P^T dO
    const int j = e / DK, d = e - j * DK;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < p.Lq; ++i) {
      ak += dS[i * p.Lk + j] * Qs[i * DK + d];
      av += P[i * p.Lk + j] * Gs[i * DK + d];
    }
    const size_t r = (size_t)b * p.Lk + j;
    dk[r * lddk + h * DK + d] = f2bf(ak * p.scale);
    dv[r * lddv + h * DK + d] = f2bf(av);
  }
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

// demb[v, :] += sum over the tokens equal to v of dx[token row, :]; one block per vocabulary row, fixed scan order (deterministic)
__global__ __launch_bounds__(256) void seq_embed_bwd_kernel(const long long* __restrict__ tok, const bf16_t* __restrict__ dx,
                                                            float* __restrict__ demb, int n_tok, int d) {
  const int v = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < n_tok; ++r)
      if (tok[r] == v) a += bf2f(dx[(size_t)r * d + c]);
    if (a != 0.f) demb[(size_t)v * d + c] += a;
  }
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

size_t lds_fwd(int Lk) { return (size_t)2 * Lk * KROW * 2 + (size_t)MAXQ * DK * 4 + (size_t)MAXQ * Lk * 4 + MAXQ * 4; }
size_t lds_bwd(int Lk) { return (size_t)2 * Lk * KROW * 2 + (size_t)2 * MAXQ * DK * 4 + (size_t)2 * MAXQ * Lk * 4; }

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_seq_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, float* lse, int B,
                                int heads, int Lq, int Lk, float scale, int causal, const long long* lens, hipStream_t stream) {
  if (!q || !k || !v || !out || !lse || B <= 0 || heads <= 0 || Lq <= 0 || Lq > MAXQ || Lk <= 0 || Lk > 512) return DIG_ERR_ARG;
  if ((ldk & 1) || (ldv & 1)) return DIG_ERR_ALIGN;
  SeqAttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, Lq, Lk, scale, causal, lens};
  const size_t lds = lds_fwd(Lk);
  static size_t attr = 0;
  if (lds > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
  hipLaunchKernelGGL(seq_attn_fwd_kernel, dim3(B, heads), dim3(256), lds, stream, p, (bf16_t*)out, ldo, lse);
  return dig_check_launch();
}

extern "C" int dig_seq_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int ldo,
                                const float* lse, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int heads, int Lq, int Lk,
                                float scale, int causal, const long long* lens, hipStream_t stream) {
  if (!q || !k || !v || !dout || !lse || !dq || !dk || !dv || B <= 0 || heads <= 0 || Lq <= 0 || Lq > MAXQ || Lk <= 0 || Lk > 512) return DIG_ERR_ARG;
  if ((ldk & 1) || (ldv & 1)) return DIG_ERR_ALIGN;
  SeqAttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, Lq, Lk, scale, causal, lens};
  const size_t lds = lds_bwd(Lk);
  static size_t attr = 0;
  if (lds > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
  hipLaunchKernelGGL(seq_attn_bwd_kernel, dim3(B, heads), dim3(256), lds, stream, p, (const bf16_t*)dout, ldo, lse, (bf16_t*)dq, lddq, (bf16_t*)dk,
                     lddk, (bf16_t*)dv, lddv);
  return dig_check_launch();
}

extern "C" int dig_seq_embed_fwd(const long long* tokens, const float* emb, const float* pos_table, void* x, int B, int T, int d, int vocab,
                                 hipStream_t stream) {
  if (!tokens || !emb || !pos_table || !x || B <= 0 || T <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  const size_t n = (size_t)B * T * d;
  hipLaunchKernelGGL(seq_embed_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tokens, emb, pos_table, (bf16_t*)x, B * T, T, d, vocab);
  return dig_check_launch();
}

extern "C" int dig_seq_embed_bwd(const long long* tokens, const void* dx, float* demb, int n_tok, int d, int vocab, hipStream_t stream) {
  if (!tokens || !dx || !demb || n_tok <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(seq_embed_bwd_kernel, dim3(vocab), dim3(256), 0, stream, tokens, (const bf16_t*)dx, demb, n_tok, d);
  return dig_check_launch();
}

extern "C" int dig_seq_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar,
                                         int B, int T, int C, void* dlogits, int ldd, hipStream_t stream) {
  if (!logits || !target || !length || !dlogits || B <= 0 || T <= 0 || C <= 0 || ld < C || ldd < C) return DIG_ERR_ARG;
  hipLaunchKernelGGL(seq_ce_bwd_kernel, dim3(B * T), dim3(64), 0, stream, logits, ld, target, length, gscalar, T, C, 1.0f / (float)B,
                     (bf16_t*)dlogits, ldd);
  return dig_check_launch();
}
