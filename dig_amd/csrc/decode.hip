// Greedy decode with a K/V cache: the per-step pieces of TFDecoder.forward_test
// (models/decoder.py:224-252) that are not GEMMs / LayerNorms.  The reference re-runs the whole decoder over all 26 positions
// at every one of the 25 steps; position t only depends on tokens <= t, so the device path feeds one token per step, keeps
// the self-attention keys/values of each layer in HBM ([B, T, 3*H*64] bf16: the fused q|k|v projection writes row t in place)
// and projects the cross-attention keys/values of the encoder memory once per layer.
//
//   dig_decode_embed       x[b,:] = trg_word_emb[token[b],:] + position_table[t,:]              (decoder.py:173-181)
//   dig_decode_self_attn   one query (row t) against cached rows 0..t, per (sample, head)        (transformer_layer.py:238-281)
//   dig_decode_cross_attn  one query against the Nm memory keys/values, per (sample, head); optional per-head weights
//   dig_softmax_argmax     probabilities + greedy token of a logit row                           (decoder.py:238-246)
// Head dimension 64 (d_k = d_v = 64: `tf_decoder` and `small_tf_decoder`).  All HBM-bound and tiny; one wave (self) or four
// waves (cross) per (sample, head), fp32 softmax.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace {

constexpr int DK = 64;

__global__ __launch_bounds__(256) void embed_kernel(const long long* __restrict__ tok, const float* __restrict__ emb,
                                                    const float* __restrict__ pe_row, bf16_t* __restrict__ x, int B, int d, int vocab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * d) return;
  const int b = i / d, c = i - b * d;
  long long t = tok[b];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  x[i] = f2bf(emb[(size_t)t * d + c] + pe_row[c]);
}

// qkv: [B, T, 3*hk] bf16 (q | k | v), row t holds this step's projections; out: [B, hk] bf16
__global__ __launch_bounds__(64) void self_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T, int hk, int t,
                                                       float scale) {
  const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const size_t row = (size_t)3 * hk;
  const bf16_t* base = qkv + (size_t)b * T * row + h * DK + lane;
  const float q = bf2f(base[(size_t)t * row]) * scale;
  float m = -INFINITY, l = 0.f, acc = 0.f;                             // online softmax over the t+1 cached positions
  for (int j = 0; j <= t; ++j) {
    const float s = wave_sum(q * bf2f(base[(size_t)j * row + hk]));
    const float mn = fmaxf(m, s);
    const float c = __expf(m - mn), p = __expf(s - mn);
    l = l * c + p;
    acc = acc * c + p * bf2f(base[(size_t)j * row + 2 * hk]);
    m = mn;
  }
  out[(size_t)b * hk + h * DK + lane] = f2bf(acc / l);
}

// q: [B, hk] bf16; kv: [B, Nm, 2*hk] bf16 (k | v); out: [B, hk] bf16; weights (optional): [B, H, Nm] fp32
// 256 threads = 32 key slots x 8 channel chunks: a lane loads 16 bytes (8 channels) of one key row, so 8 consecutive lanes
// cover a whole 128-byte row and every load instruction of a wave moves 8 full rows (coalesced; the first version gave each
// thread its own row and ran at 1.8 TB/s).  Scores: partial dot over the lane's 8 channels + 3 shuffles.  Values: each lane
// accumulates its 8 channels over its key slot's keys, then shuffles / LDS combine the 32 slots.
__global__ __launch_bounds__(256) void cross_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv, bf16_t* __restrict__ out,
                                                         float* __restrict__ weights, int Nm, int hk, float scale, int slots_per_mem) {
  extern __shared__ float sm[];                                        // [Nm] scores / probabilities, then [4][64] partial outputs
  __shared__ float red[8];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, H = gridDim.y;
  const int chunk = tid & 7, slot = tid >> 3;                           // 8 channels [chunk*8, +8); keys slot, slot+32, ...
  float qv[8];
  {
    const uint4 v = *reinterpret_cast<const uint4*>(q + (size_t)b * hk + h * DK + chunk * 8);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { qv[2 * e] = bf2f((bf16_t)(w[e] & 0xffff)) * scale; qv[2 * e + 1] = bf2f((bf16_t)(w[e] >> 16)) * scale; }
  }
  const bf16_t* kb = kv + (size_t)(b / slots_per_mem) * Nm * 2 * hk + h * DK + chunk * 8;   // beam search: `slots_per_mem` queries share a memory
  float lmax = -INFINITY;
  for (int j = slot; j < Nm; j += 32) {
    const uint4 v = *reinterpret_cast<const uint4*>(kb + (size_t)j * 2 * hk);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += qv[2 * e] * bf2f((bf16_t)(w[e] & 0xffff)) + qv[2 * e + 1] * bf2f((bf16_t)(w[e] >> 16));
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (chunk == 0) sm[j] = s;
    lmax = fmaxf(lmax, s);
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.f;
  for (int j = tid; j < Nm; j += 256) {
    const float p = __expf(sm[j] - gmax);
    sm[j] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = lsum;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  if (weights)
    for (int j = tid; j < Nm; j += 256) weights[((size_t)b * H + h) * Nm + j] = sm[j] * inv;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int j = slot; j < Nm; j += 32) {
    const uint4 v = *reinterpret_cast<const uint4*>(kb + (size_t)j * 2 * hk + hk);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    const float p = sm[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += p * bf2f((bf16_t)(w[e] & 0xffff)); acc[2 * e + 1] += p * bf2f((bf16_t)(w[e] >> 16)); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {                                         // combine the 8 key slots of this wave (lanes differ in bits 3..5)
    acc[e] += __shfl_xor(acc[e], 8, 64); acc[e] += __shfl_xor(acc[e], 16, 64); acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  __syncthreads();                                                      // probabilities are no longer needed
  float* part = sm;
  if ((tid & 63) < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[(tid >> 6) * DK + chunk * 8 + e] = acc[e];
  }
  __syncthreads();
  if (tid < DK) out[(size_t)b * hk + h * DK + tid] = f2bf((part[tid] + part[DK + tid] + part[2 * DK + tid] + part[3 * DK + tid]) * inv);
}

// one wave per row: probs[b, :C] = softmax(logits[b, :C]); token[b] = first index of the maximum (torch.max semantics)
__global__ __launch_bounds__(64) void softmax_argmax_kernel(const float* __restrict__ logits, int ld, float* __restrict__ probs,
                                                            long long* __restrict__ token, int C) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (size_t)b * ld;
  float m = -INFINITY;
  int am = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = row[c];
    if (v > m) { m = v; am = c; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(row[c] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < C; c += 64) probs[(size_t)b * C + c] = __expf(row[c] - m) * inv;
  if (lane == 0) token[b] = am;
}

// One step of TFDecoder.beam_search (decoder.py:283-307) for one sample: log-softmax of its `bw` slots' logits, + the slots' running
// scores, top-`bw` of the bw*C candidates (ties: the lower candidate index, i.e. torch.topk's result wherever it is defined), then the
// bookkeeping of that step: symbol = candidate % C, predecessor = candidate / C + b*bw, stored score, and the running score that
// the next step starts from (-inf once a slot has emitted EOS, :299-301).  One 256-thread workgroup per sample; bw <= 16.
__global__ __launch_bounds__(256) void beam_step_kernel(const float* __restrict__ logits, int ld, float* __restrict__ seq_scores, int bw, int C,
                                                        int eos, long long* __restrict__ symbols, long long* __restrict__ preds,
                                                        float* __restrict__ stored) {
  extern __shared__ float cand[];                                       // [bw * C]
  __shared__ float lse[16];
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = wave; k < bw; k += 4) {                                   // a wave per slot: log-sum-exp of the row
    const float* row = logits + (size_t)(b * bw + k) * ld;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(row[c] - m);
    s = wave_sum(s);
    if (lane == 0) lse[k] = m + __logf(s);
  }
  __syncthreads();
  const int n = bw * C;
  for (int i = tid; i < n; i += 256) {
    const int k = i / C, c = i - k * C;
    cand[i] = seq_scores[b * bw + k] + (logits[(size_t)(b * bw + k) * ld + c] - lse[k]);
  }
  __syncthreads();
  for (int r = 0; r < bw; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
      const float v = cand[i];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
      if (bi == 0x7fffffff) bi = r;                                      // every candidate is NaN / -inf and already taken: keep indices valid
      const int sym = bi % C;
      symbols[b * bw + r] = sym;
      preds[b * bw + r] = bi / C + b * bw;
      stored[b * bw + r] = bv;
      seq_scores[b * bw + r] = sym == eos ? -INFINITY : bv;
      cand[bi] = -INFINITY;                                              // taken (a -inf candidate can be taken again: the reference's
    }                                                                    //  order among -inf ties is unspecified as well)
    __syncthreads();
  }
}

// Accuracy (evaluation_metric/metrics.py:19-81): both label rows are cut at EOS, UNKNOWN and every class that is not a digit
// or letter are dropped, letters compare case-insensitively.  canon[c] = canonical code of class c (0 = dropped, EOS and
// UNKNOWN included).  One thread per sample; match[b] = 1 when the normalised strings are equal.
__global__ void string_match_kernel(const long long* __restrict__ pred, const long long* __restrict__ target,
                                    const unsigned char* __restrict__ canon, int n_classes, int eos, int B, int T,
                                    unsigned char* __restrict__ match) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const long long* p = pred + (size_t)b * T;
  const long long* t = target + (size_t)b * T;
  int i = 0, j = 0;
  bool ok = true;
  while (ok) {
    unsigned char a = 0, c = 0;
    while (i < T && p[i] != eos) {                                      // next kept character of the prediction
      const long long v = p[i++];
      a = (v >= 0 && v < n_classes) ? canon[v] : 0;
      if (a) break;
    }
    if (i < T && p[i] == eos && !a) i = T;
    while (j < T && t[j] != eos) {
      const long long v = t[j++];
      c = (v >= 0 && v < n_classes) ? canon[v] : 0;
      if (c) break;
    }
    if (j < T && t[j] == eos && !c) j = T;
    if (a != c) ok = false;
    if (!a && !c) break;                                                // both strings ended
  }
  match[b] = ok ? 1 : 0;
}

// SeqCrossEntropyLoss.forward (loss/seqCrossEntropyLoss.py:47-63): row (b,t) contributes -log_softmax(input[b,t])[target[b,t]]
// when t < length[b].  One wave per row writes its term; a single block then adds the B*T terms in a fixed order.
__global__ __launch_bounds__(64) void seq_ce_rows_kernel(const float* __restrict__ input, const long long* __restrict__ target,
                                                         const long long* __restrict__ length, int T, int C, float* __restrict__ rowloss) {
  const int row = blockIdx.x, b = row / T, t = row - b * T, lane = threadIdx.x;
  if (t >= length[b]) {
    if (lane == 0) rowloss[row] = 0.f;
    return;
  }
  const float* x = input + (size_t)row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = wave_sum(s);
  if (lane == 0) {
    long long y = target[row];
    y = y < 0 ? 0 : (y >= C ? C - 1 : y);
    rowloss[row] = -(x[y] - m - __logf(s));
  }
}

__global__ __launch_bounds__(256) void fixed_order_sum_kernel(const float* __restrict__ v, int n, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += v[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// SeqLabelSmoothingCrossEntropyLoss.forward (loss/seqLabelSmoothingCrossEntropyLoss.py:48-70), as the reference computes it: the
// `nll_loss` column [BT,1] and the `smooth_loss` product of a [BT] vector with the [BT,1] mask broadcast to a [BT,BT] matrix, so
//   loss = ( confidence * BT * sum_i mask_i * nll_i  +  smoothing * (sum_i mask_i) * sum_j s_j ) / B,   s_j = -mean_c log_softmax(x_j)_c
// with s_j taken over ALL rows j (also the padded ones).  One wave per row writes (mask_i * nll_i, s_i); a single block then adds
// the terms in a fixed order.
__global__ __launch_bounds__(64) void seq_ls_ce_rows_kernel(const float* __restrict__ input, const long long* __restrict__ target,
                                                            const long long* __restrict__ length, int T, int C, float* __restrict__ rows2) {
  const int row = blockIdx.x, b = row / T, t = row - b * T, lane = threadIdx.x, n = gridDim.x;
  const float* x = input + (size_t)row * C;
  float m = -INFINITY, sx = 0.f;
  for (int c = lane; c < C; c += 64) { m = fmaxf(m, x[c]); sx += x[c]; }
  m = wave_max(m);
  sx = wave_sum(sx);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = wave_sum(s);
  if (lane == 0) {
    const float lse = m + __logf(s);
    long long y = target[row];
    y = y < 0 ? 0 : (y >= C ? C - 1 : y);
    rows2[row] = t < length[b] ? lse - x[y] : 0.f;
    rows2[n + row] = lse - sx / (float)C;
  }
}

__global__ __launch_bounds__(256) void seq_ls_ce_sum_kernel(const float* __restrict__ rows2, const long long* __restrict__ length, int B, int T,
                                                            float confidence, float smoothing, float* __restrict__ out) {
  __shared__ float red[3][256];
  const int n = B * T;
  float a = 0.f, b = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { a += rows2[i]; b += rows2[n + i]; }
  for (int i = threadIdx.x; i < B; i += 256) { const long long l = length[i]; cnt += (float)(l < 0 ? 0 : (l > T ? T : l)); }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = cnt;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (confidence * (float)n * red[0][0] + smoothing * red[2][0] * red[1][0]) / (float)B;
}

// gradient: dL/dx[i,c] = g / B * ( confidence * BT * mask_i * (p_ic - [c == y_i]) + smoothing * M * (p_ic - 1/C) ),  M = sum_i mask_i
__global__ __launch_bounds__(64) void seq_ls_ce_bwd_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target,
                                                           const long long* __restrict__ length, const float* __restrict__ g, int B, int T, int C,
                                                           float confidence, float smoothing, bf16_t* __restrict__ dlogits, int ldd) {
  const int row = blockIdx.x, b = row / T, t = row - b * T, lane = threadIdx.x;
  float cnt = 0.f;
  for (int i = lane; i < B; i += 64) { const long long l = length[i]; cnt += (float)(l < 0 ? 0 : (l > T ? T : l)); }
  cnt = wave_sum(cnt);
  const float* x = logits + (size_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = wave_sum(s);
  const float sc = (g ? g[0] : 1.f) / (float)B, inv = 1.f / s;
  const float wn = t < length[b] ? confidence * (float)(B * T) : 0.f, ws = smoothing * cnt;
  long long y = target[row];
  y = y < 0 ? 0 : (y >= C ? C - 1 : y);
  bf16_t* out = dlogits + (size_t)row * ldd;
  for (int c = lane; c < ldd; c += 64) {
    float v = 0.f;
    if (c < C) {
      const float p = __expf(x[c] - m) * inv;
      v = sc * (wn * (p - (c == y ? 1.f : 0.f)) + ws * (p - 1.f / (float)C));
    }
    out[c] = f2bf(v);
  }
}

// recognition_f_measure (evaluation_metric/metrics.py:83-100): per sample, the SETS of kept characters of prediction and target
// (same cut / drop / case-fold rules as the accuracy; canon codes 1..63 -> one bit each), p = n/(|P|+1e-5), r = n/(|T|+1e-5),
// f = 2pr/(p+r+1e-5), all in double as the reference's Python floats.
__global__ void char_fmeasure_kernel(const long long* __restrict__ pred, const long long* __restrict__ target,
                                     const unsigned char* __restrict__ canon, int n_classes, int eos, int B, int T,
                                     double* __restrict__ f_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  unsigned long long ps = 0, ts = 0;
  for (int i = 0; i < T; ++i) {
    const long long v = pred[(size_t)b * T + i];
    if (v == eos) break;
    const unsigned char c = (v >= 0 && v < n_classes) ? canon[v] : 0;
    if (c) ps |= 1ull << (c & 63);
  }
  for (int i = 0; i < T; ++i) {
    const long long v = target[(size_t)b * T + i];
    if (v == eos) break;
    const unsigned char c = (v >= 0 && v < n_classes) ? canon[v] : 0;
    if (c) ts |= 1ull << (c & 63);
  }
  const double n = (double)__popcll(ps & ts);
  const double p = n / ((double)__popcll(ps) + 1e-5), r = n / ((double)__popcll(ts) + 1e-5);
  f_out[b] = 2 * p * r / (p + r + 1e-5);
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_decode_embed(const long long* tokens, const float* emb, const float* pe_row, void* x, int B, int d, int vocab,
                                hipStream_t stream) {
  if (!tokens || !emb || !pe_row || !x || B <= 0 || d <= 0 || vocab <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(embed_kernel, dim3((B * d + 255) / 256), dim3(256), 0, stream, tokens, emb, pe_row, (bf16_t*)x, B, d, vocab);
  return dig_check_launch();
}

extern "C" int dig_decode_self_attn(const void* qkv_cache, void* out, int B, int T, int heads, int head_dim, int t, float scale,
                                    hipStream_t stream) {
  if (!qkv_cache || !out || B <= 0 || heads <= 0 || t < 0 || t >= T) return DIG_ERR_ARG;
  if (head_dim != DK) return DIG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(self_attn_kernel, dim3(B, heads), dim3(64), 0, stream, (const bf16_t*)qkv_cache, (bf16_t*)out, T, heads * DK, t, scale);
  return dig_check_launch();
}

extern "C" int dig_decode_cross_attn(const void* q, const void* kv_mem, void* out, float* weights, int B, int n_mem, int heads,
                                     int head_dim, float scale, int slots_per_mem, hipStream_t stream) {
  if (!q || !kv_mem || !out || B <= 0 || n_mem <= 0 || heads <= 0 || slots_per_mem < 1 || B % slots_per_mem) return DIG_ERR_ARG;
  if (head_dim != DK || n_mem > 8192) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(kv_mem)) return DIG_ERR_ALIGN;
  const size_t lds = (size_t)std::max(n_mem, 4 * DK) * sizeof(float);
  hipLaunchKernelGGL(cross_attn_kernel, dim3(B, heads), dim3(256), lds, stream, (const bf16_t*)q, (const bf16_t*)kv_mem, (bf16_t*)out, weights,
                     n_mem, heads * DK, scale, slots_per_mem);
  return dig_check_launch();
}

extern "C" int dig_softmax_argmax(const float* logits, int ld, float* probs, long long* tokens, int B, int C, hipStream_t stream) {
  if (!logits || !probs || !tokens || B <= 0 || C <= 0 || ld < C) return DIG_ERR_ARG;
  hipLaunchKernelGGL(softmax_argmax_kernel, dim3(B), dim3(64), 0, stream, logits, ld, probs, tokens, C);
  return dig_check_launch();
}

extern "C" int dig_beam_step(const float* logits, int ld, float* seq_scores, int B, int beam_width, int C, int eos, long long* symbols,
                             long long* predecessors, float* stored_scores, hipStream_t stream) {
  if (!logits || !seq_scores || !symbols || !predecessors || !stored_scores || B <= 0 || C <= 0 || ld < C) return DIG_ERR_ARG;
  if (beam_width < 1 || beam_width > 16 || (size_t)beam_width * C * sizeof(float) > 60000) return DIG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), (size_t)beam_width * C * sizeof(float), stream, logits, ld, seq_scores, beam_width, C,
                     eos, symbols, predecessors, stored_scores);
  return dig_check_launch();
}

extern "C" int dig_string_match(const long long* pred, const long long* target, const unsigned char* canon, int n_classes, int eos,
                                int B, int T, unsigned char* match, hipStream_t stream) {
  if (!pred || !target || !canon || !match || n_classes <= 0 || B <= 0 || T <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(string_match_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, pred, target, canon, n_classes, eos, B, T, match);
  return dig_check_launch();
}

extern "C" int dig_seq_cross_entropy(const float* input, const long long* target, const long long* length, int B, int T, int C,
                                     float* row_workspace, float* loss, hipStream_t stream) {
  if (!input || !target || !length || !row_workspace || !loss || B <= 0 || T <= 0 || C <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(seq_ce_rows_kernel, dim3(B * T), dim3(64), 0, stream, input, target, length, T, C, row_workspace);
  hipLaunchKernelGGL(fixed_order_sum_kernel, dim3(1), dim3(256), 0, stream, row_workspace, B * T, 1.0f / (float)B, loss);
  return dig_check_launch();
}

extern "C" int dig_char_fmeasure(const long long* pred, const long long* target, const unsigned char* canon, int n_classes, int eos,
                                 int B, int T, double* f_per_sample, hipStream_t stream) {
  if (!pred || !target || !canon || !f_per_sample || n_classes <= 0 || B <= 0 || T <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(char_fmeasure_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, pred, target, canon, n_classes, eos, B, T, f_per_sample);
  return dig_check_launch();
}

extern "C" int dig_seq_ls_cross_entropy(const float* input, const long long* target, const long long* length, int B, int T, int C,
                                        float smoothing, float* row_workspace, float* loss, hipStream_t stream) {
  if (!input || !target || !length || !row_workspace || !loss || B <= 0 || T <= 0 || C <= 0 || smoothing < 0.f || smoothing > 1.f) return DIG_ERR_ARG;
  hipLaunchKernelGGL(seq_ls_ce_rows_kernel, dim3(B * T), dim3(64), 0, stream, input, target, length, T, C, row_workspace);
  hipLaunchKernelGGL(seq_ls_ce_sum_kernel, dim3(1), dim3(256), 0, stream, row_workspace, length, B, T, 1.f - smoothing, smoothing, loss);
  return dig_check_launch();
}

extern "C" int dig_seq_ls_cross_entropy_bwd(const float* logits, int ld, const long long* target, const long long* length, const float* gscalar,
                                            int B, int T, int C, float smoothing, void* dlogits, int ldd, hipStream_t stream) {
  if (!logits || !target || !length || !dlogits || B <= 0 || T <= 0 || C <= 0 || ld < C || ldd < C || smoothing < 0.f || smoothing > 1.f) return DIG_ERR_ARG;
  hipLaunchKernelGGL(seq_ls_ce_bwd_kernel, dim3(B * T), dim3(64), 0, stream, logits, ld, target, length, gscalar, B, T, C, 1.f - smoothing,
                     smoothing, (bf16_t*)dlogits, ldd);
  return dig_check_launch();
}
