// GRU attention recognition head (second decoder): `DecoderUnit.forward` of models/attn_decoder.py:258-272 --
// additive attention over the encoder tokens (`AttentionUnit.forward`, :215-233), one nn.GRU cell -- and its gradient, one launch per
// time step (the recurrence makes the steps dependent; every step is a few small kernels, launch-latency-bound by construction).
//   alpha[b, :] = softmax_n( w . tanh(sProj[b, :] + xProj[b, n, :]) )        context[b, :] = sum_n alpha[b, n] x[b, n, :]
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), s' = (1 - z) * n + z * s      (torch.nn.GRU, gates r|z|n)
// Activations that are GEMM operands are bf16, the recurrent state and every reduction fp32.  The (B, N, A) gradient of xProj and the
// (B, N, X) gradient of x are NOT accumulated step by step (that would re-write 230 MB per step): the per-step softmax gradients are
// kept ([B, N] floats per step) and one final kernel per tensor sums over the steps in registers.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(2.f * fminf(fmaxf(x, -15.f), 15.f));
  return (e - 1.f) * __frcp_rn(e + 1.f);
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }

__device__ __forceinline__ void load8f(const bf16_t* __restrict__ src, float (&r)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(src);
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[2 * e] = bf2f((bf16_t)(w[e] & 0xffff)); r[2 * e + 1] = bf2f((bf16_t)(w[e] >> 16)); }
}

constexpr int MAXA = 1024;     // attention width (attDim): <= 1024, multiple of 8
constexpr int MAXN = 512;      // encoder tokens per sample

// scores of one sample: a wave per token (lanes over the A channels, 8 per lane and pass), wave reduction.
// sp: sProj row staged in LDS (fp32), w in LDS.  v[n] = sum_a w[a] * tanh(sp[a] + xproj[n, a])
__device__ __forceinline__ float token_score(const bf16_t* __restrict__ xp_row, const float* __restrict__ sp, const float* __restrict__ w, int A, int lane) {
  float acc = 0.f;
  for (int a0 = lane * 8; a0 < A; a0 += 512) {
    float xv[8];
    load8f(xp_row + a0, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += w[a0 + e] * tanh_fast(sp[a0 + e] + xv[e]);
  }
  return wave_sum(acc);
}

// ---- forward: alpha (fp32, saved) and context (bf16, written at ctx + b * ldc: the caller points it into the GRU input [yProj | context])
constexpr int NT = 512, NW = NT / 64;     // one workgroup per sample and only B of them: 8 waves keep enough loads in flight per CU

__device__ __forceinline__ float block_max(float v, float* red, int lane, int wave) {
  v = wave_max(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  return m;
}
__device__ __forceinline__ float block_sum(float v, float* red, int lane, int wave) {
  v = wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) m += red[i];
  __syncthreads();
  return m;
}

__global__ __launch_bounds__(NT) void addattn_fwd_kernel(const bf16_t* __restrict__ xproj, const bf16_t* __restrict__ sproj,
                                                         const float* __restrict__ w, const bf16_t* __restrict__ x, float* __restrict__ alpha,
                                                         bf16_t* __restrict__ ctx, int ldc, int N, int A, int X) {
  __shared__ float sp[MAXA], ws[MAXA], v[MAXN], part[2 * MAXA];
  __shared__ float red[NW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int a = tid; a < A; a += NT) { sp[a] = bf2f(sproj[(size_t)b * A + a]); ws[a] = w[a]; }
  __syncthreads();
  for (int n = wave; n < N; n += NW) {
    const float s = token_score(xproj + ((size_t)b * N + n) * A, sp, ws, A, lane);
    if (lane == 0) v[n] = s;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int n = tid; n < N; n += NT) m = fmaxf(m, v[n]);
  m = block_max(m, red, lane, wave);
  float s = 0.f;
  for (int n = tid; n < N; n += NT) { const float e = __expf(v[n] - m); v[n] = e; s += e; }
  const float inv = 1.f / block_sum(s, red, lane, wave);
  for (int n = tid; n < N; n += NT) { const float a = v[n] * inv; v[n] = a; alpha[(size_t)b * N + n] = a; }
  __syncthreads();
  // context: a thread owns two channels (rows read coalesced); when the pairs fit in half the block, the two halves of the block
  // take one half of the tokens each and meet in LDS
  const int CP = X / 2;
  if (CP <= NT / 2) {
    const int half = tid >= NT / 2 ? 1 : 0, cp = tid - half * (NT / 2);
    float a0 = 0.f, a1 = 0.f;
    if (cp < CP) {
      const bf16_t* xb = x + (size_t)b * N * X + 2 * cp;
      const int n_hi = half ? N : N / 2;
      for (int n = half ? N / 2 : 0; n < n_hi; ++n) {
        const unsigned p = *reinterpret_cast<const unsigned*>(xb + (size_t)n * X);
        a0 += v[n] * bf2f((bf16_t)(p & 0xffff));
        a1 += v[n] * bf2f((bf16_t)(p >> 16));
      }
      if (half) { part[2 * cp] = a0; part[2 * cp + 1] = a1; }
    }
    __syncthreads();
    if (!half && cp < CP) *reinterpret_cast<unsigned*>(ctx + (size_t)b * ldc + 2 * cp) = pack_bf2(a0 + part[2 * cp], a1 + part[2 * cp + 1]);
  } else {
    for (int c = tid * 2; c < X; c += 2 * NT) {
      float a0 = 0.f, a1 = 0.f;
      const bf16_t* xb = x + (size_t)b * N * X + c;
      for (int n = 0; n < N; ++n) {
        const unsigned p = *reinterpret_cast<const unsigned*>(xb + (size_t)n * X);
        a0 += v[n] * bf2f((bf16_t)(p & 0xffff));
        a1 += v[n] * bf2f((bf16_t)(p >> 16));
      }
      *reinterpret_cast<unsigned*>(ctx + (size_t)b * ldc + c) = pack_bf2(a0, a1);
    }
  }
}

// ---- backward of one step: dctx [B, X] (bf16, row stride ldd) -> dv [B, N] (fp32, kept for the final kernels), dsproj [B, A] (bf16),
// dw_acc [B, A] (fp32, += : per-sample partials of the wEmbed gradient, summed over samples at the end)
__global__ __launch_bounds__(NT) void addattn_bwd_kernel(const bf16_t* __restrict__ xproj, const bf16_t* __restrict__ sproj,
                                                         const float* __restrict__ w, const bf16_t* __restrict__ x, const float* __restrict__ alpha,
                                                         const bf16_t* __restrict__ dctx, int ldd, float* __restrict__ dv_out,
                                                         bf16_t* __restrict__ dsproj, float* __restrict__ dw_acc, int N, int A, int X) {
  __shared__ float sp[MAXA], ws[MAXA], dv[MAXN], dc[MAXA], part[4 * (NT / 2)];
  __shared__ float red[NW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int a = tid; a < A; a += NT) { sp[a] = bf2f(sproj[(size_t)b * A + a]); ws[a] = w[a]; }
  for (int c = tid; c < X; c += NT) dc[c] = bf2f(dctx[(size_t)b * ldd + c]);           // (X <= MAXA)
  __syncthreads();
  // dalpha[n] = dctx . x[n, :]: a wave per token
  for (int n = wave; n < N; n += NW) {
    float acc = 0.f;
    const bf16_t* xr = x + ((size_t)b * N + n) * X;
    for (int c0 = lane * 8; c0 < X; c0 += 512) {
      float xv[8];
      load8f(xr + c0, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += dc[c0 + e] * xv[e];
    }
    acc = wave_sum(acc);
    if (lane == 0) dv[n] = acc;
  }
  __syncthreads();
  float dot = 0.f;
  for (int n = tid; n < N; n += NT) dot += alpha[(size_t)b * N + n] * dv[n];
  dot = block_sum(dot, red, lane, wave);
  for (int n = tid; n < N; n += NT) {
    const float g = alpha[(size_t)b * N + n] * (dv[n] - dot);            // softmax backward
    dv[n] = g;
    dv_out[(size_t)b * N + n] = g;
  }
  __syncthreads();
  // dsproj[a] = sum_n dv[n] * w[a] * (1 - t^2),  dw[a] += sum_n dv[n] * t,  t = tanh(sp[a] + xproj[n, a]): a thread owns two channels;
  // when the pairs fit in half the block the two halves take one half of the tokens each
  const int AP = A / 2;
  if (AP <= NT / 2) {
    const int half = tid >= NT / 2 ? 1 : 0, ap = tid - half * (NT / 2), a = 2 * ap;
    float g0 = 0.f, g1 = 0.f, w0 = 0.f, w1 = 0.f;
    if (ap < AP) {
      const bf16_t* xb = xproj + (size_t)b * N * A + a;
      const int n_hi = half ? N : N / 2;
      for (int n = half ? N / 2 : 0; n < n_hi; ++n) {
        const unsigned p = *reinterpret_cast<const unsigned*>(xb + (size_t)n * A);
        const float t0 = tanh_fast(sp[a] + bf2f((bf16_t)(p & 0xffff))), t1 = tanh_fast(sp[a + 1] + bf2f((bf16_t)(p >> 16)));
        const float d = dv[n];
        g0 += d * (1.f - t0 * t0); g1 += d * (1.f - t1 * t1);
        w0 += d * t0; w1 += d * t1;
      }
      if (half) { part[4 * ap] = g0; part[4 * ap + 1] = g1; part[4 * ap + 2] = w0; part[4 * ap + 3] = w1; }
    }
    __syncthreads();
    if (!half && ap < AP) {
      g0 += part[4 * ap]; g1 += part[4 * ap + 1]; w0 += part[4 * ap + 2]; w1 += part[4 * ap + 3];
      *reinterpret_cast<unsigned*>(dsproj + (size_t)b * A + a) = pack_bf2(g0 * ws[a], g1 * ws[a + 1]);
      dw_acc[(size_t)b * A + a] += w0;
      dw_acc[(size_t)b * A + a + 1] += w1;
    }
  } else {
    for (int a = tid * 2; a < A; a += 2 * NT) {
      float g0 = 0.f, g1 = 0.f, w0 = 0.f, w1 = 0.f;
      const bf16_t* xb = xproj + (size_t)b * N * A + a;
      for (int n = 0; n < N; ++n) {
        const unsigned p = *reinterpret_cast<const unsigned*>(xb + (size_t)n * A);
        const float t0 = tanh_fast(sp[a] + bf2f((bf16_t)(p & 0xffff))), t1 = tanh_fast(sp[a + 1] + bf2f((bf16_t)(p >> 16)));
        const float d = dv[n];
        g0 += d * (1.f - t0 * t0); g1 += d * (1.f - t1 * t1);
        w0 += d * t0; w1 += d * t1;
      }
      *reinterpret_cast<unsigned*>(dsproj + (size_t)b * A + a) = pack_bf2(g0 * ws[a], g1 * ws[a + 1]);
      dw_acc[(size_t)b * A + a] += w0;
      dw_acc[(size_t)b * A + a + 1] += w1;
    }
  }
}

// ---- after the last step: dxproj[b, n, a] = sum_t dv_t[b, n] * w[a] * (1 - tanh^2(sproj_t[b, a] + xproj[b, n, a]))   (bf16 out)
// dv_all: [T][B][N] fp32, sproj_all: [T][B][A] bf16.  Block = (sample, 64 tokens); a thread owns two channels and walks its tokens.
__global__ __launch_bounds__(256) void addattn_dxproj_kernel(const bf16_t* __restrict__ xproj, const bf16_t* __restrict__ sproj_all,
                                                             const float* __restrict__ w, const float* __restrict__ dv_all,
                                                             bf16_t* __restrict__ dxproj, int T, int B, int N, int A) {
  extern __shared__ float sm[];                                            // sp [T][A] | dvs [T][64]
  float* sp = sm;
  float* dvs = sm + (size_t)T * A;
  const int b = blockIdx.x, n0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int i = tid; i < T * A; i += 256) { const int t = i / A, a = i - t * A; sp[i] = bf2f(sproj_all[((size_t)t * B + b) * A + a]); }
  for (int i = tid; i < T * 64; i += 256) { const int t = i >> 6, n = n0 + (i & 63); dvs[i] = n < N ? dv_all[((size_t)t * B + b) * N + n] : 0.f; }
  __syncthreads();
  for (int a = tid * 2; a < A; a += 512) {
    const float w0 = w[a], w1 = w[a + 1];
    for (int j = 0; j < 64 && n0 + j < N; ++j) {
      const size_t off = ((size_t)b * N + n0 + j) * A + a;
      const unsigned p = *reinterpret_cast<const unsigned*>(xproj + off);
      const float x0 = bf2f((bf16_t)(p & 0xffff)), x1 = bf2f((bf16_t)(p >> 16));
      float g0 = 0.f, g1 = 0.f;
      for (int t = 0; t < T; ++t) {
        const float t0 = tanh_fast(sp[t * A + a] + x0), t1 = tanh_fast(sp[t * A + a + 1] + x1), d = dvs[t * 64 + j];
        g0 += d * (1.f - t0 * t0); g1 += d * (1.f - t1 * t1);
      }
      *reinterpret_cast<unsigned*>(dxproj + off) = pack_bf2(g0 * w0, g1 * w1);
    }
  }
}

// dx[b, n, c] = sum_t alpha_t[b, n] * dctx_t[b, c]   (bf16 out; alpha_all [T][B][N] fp32, dctx_all rows (t, b) of stride ldd)
__global__ __launch_bounds__(256) void addattn_dx_kernel(const float* __restrict__ alpha_all, const bf16_t* __restrict__ dctx_all, int ldd,
                                                         bf16_t* __restrict__ dx, int T, int B, int N, int X) {
  extern __shared__ float sm[];                                            // dc [T][X] | al [T][64]
  float* dc = sm;
  float* al = sm + (size_t)T * X;
  const int b = blockIdx.x, n0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int i = tid; i < T * X; i += 256) { const int t = i / X, c = i - t * X; dc[i] = bf2f(dctx_all[((size_t)t * B + b) * ldd + c]); }
  for (int i = tid; i < T * 64; i += 256) { const int t = i >> 6, n = n0 + (i & 63); al[i] = n < N ? alpha_all[((size_t)t * B + b) * N + n] : 0.f; }
  __syncthreads();
  for (int c = tid * 2; c < X; c += 512)
    for (int j = 0; j < 64 && n0 + j < N; ++j) {
      float g0 = 0.f, g1 = 0.f;
      for (int t = 0; t < T; ++t) { const float a = al[t * 64 + j]; g0 += a * dc[t * X + c]; g1 += a * dc[t * X + c + 1]; }
      *reinterpret_cast<unsigned*>(dx + ((size_t)b * N + n0 + j) * X + c) = pack_bf2(g0, g1);
    }
}

// ---- GRU cell.  gi, gh: [B, 3S] bf16 (biases already added by the GEMMs); s_prev fp32 -> s fp32 + bf16 copy; r, z, n, gh_n kept (fp32)
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const bf16_t* __restrict__ gi, const bf16_t* __restrict__ gh, const float* __restrict__ s_prev,
                                                           float* __restrict__ s, bf16_t* __restrict__ s_bf, float* __restrict__ gates, int B, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  const int b = i / S, k = i - b * S;
  const size_t g0 = (size_t)b * 3 * S + k;
  const float hn = bf2f(gh[g0 + 2 * S]);
  const float r = sigmoid_fast(bf2f(gi[g0]) + bf2f(gh[g0]));
  const float z = sigmoid_fast(bf2f(gi[g0 + S]) + bf2f(gh[g0 + S]));
  const float n = tanh_fast(bf2f(gi[g0 + 2 * S]) + r * hn);
  const float sp = s_prev ? s_prev[i] : 0.f;
  const float v = (1.f - z) * n + z * sp;
  s[i] = v;
  s_bf[i] = f2bf(v);
  float* gt = gates + (size_t)b * 4 * S + k;
  gt[0] = r; gt[S] = z; gt[2 * S] = n; gt[3 * S] = hn;
}

// ds = ds_a + ds_b + ds_c + ds_d (fp32 [B, S], any of them may be null): gradient w.r.t. this step's new state.
// -> dgi, dgh [B, 3S] bf16, ds_prev [B, S] fp32 (the direct z * ds path only; the GEMM paths are added by the caller's next call)
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ ds_a, const float* __restrict__ ds_b, const float* __restrict__ ds_c,
                                                           const float* __restrict__ ds_d, const float* __restrict__ gates,
                                                           const float* __restrict__ s_prev, bf16_t* __restrict__ dgi, bf16_t* __restrict__ dgh,
                                                           float* __restrict__ ds_prev, int B, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  const int b = i / S, k = i - b * S;
  float ds = ds_a[i];
  if (ds_b) ds += ds_b[i];
  if (ds_c) ds += ds_c[i];
  if (ds_d) ds += ds_d[i];
  const float* gt = gates + (size_t)b * 4 * S + k;
  const float r = gt[0], z = gt[S], n = gt[2 * S], hn = gt[3 * S];
  const float sp = s_prev ? s_prev[i] : 0.f;
  const float dn = ds * (1.f - z), dz = ds * (sp - n);
  const float dpre = dn * (1.f - n * n);
  const float dr = dpre * hn;
  const float gr = dr * r * (1.f - r), gz = dz * z * (1.f - z);
  const size_t g0 = (size_t)b * 3 * S + k;
  dgi[g0] = f2bf(gr); dgi[g0 + S] = f2bf(gz); dgi[g0 + 2 * S] = f2bf(dpre);
  dgh[g0] = f2bf(gr); dgh[g0 + S] = f2bf(gz); dgh[g0 + 2 * S] = f2bf(dpre * r);
  ds_prev[i] = ds * z;
}

// out[r, :cols] (bf16, row stride ld) = table[token[r], :cols] (fp32); tokens clamped to [0, vocab)
__global__ __launch_bounds__(256) void embed_rows_kernel(const long long* __restrict__ tok, const float* __restrict__ table, bf16_t* __restrict__ out,
                                                         int ld, int rows, int cols, int vocab) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
  long long t = tok[r];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  out[(size_t)r * ld + c] = f2bf(table[(size_t)t * cols + c]);
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_addattn_fwd(const void* xproj, const void* sproj, const float* w, const void* x, float* alpha, void* ctx, int ldc, int B, int N,
                               int A, int X, hipStream_t stream) {
  if (!xproj || !sproj || !w || !x || !alpha || !ctx || B <= 0 || N <= 0 || N > MAXN || A <= 0 || A > MAXA || (A & 7) || X <= 0 || (X & 1) || (ldc & 1))
    return DIG_ERR_ARG;
  if (!aligned16(xproj)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(addattn_fwd_kernel, dim3(B), dim3(NT), 0, stream, (const bf16_t*)xproj, (const bf16_t*)sproj, w, (const bf16_t*)x, alpha,
                     (bf16_t*)ctx, ldc, N, A, X);
  return dig_check_launch();
}

extern "C" int dig_addattn_bwd(const void* xproj, const void* sproj, const float* w, const void* x, const float* alpha, const void* dctx, int ldd,
                               float* dv, void* dsproj, float* dw_acc, int B, int N, int A, int X, hipStream_t stream) {
  if (!xproj || !sproj || !w || !x || !alpha || !dctx || !dv || !dsproj || !dw_acc || B <= 0 || N <= 0 || N > MAXN || A <= 0 || A > MAXA || (A & 7) ||
      X <= 0 || X > MAXA || (X & 7))
    return DIG_ERR_ARG;
  if (!aligned16(xproj) || !aligned16(x)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(addattn_bwd_kernel, dim3(B), dim3(NT), 0, stream, (const bf16_t*)xproj, (const bf16_t*)sproj, w, (const bf16_t*)x, alpha,
                     (const bf16_t*)dctx, ldd, dv, (bf16_t*)dsproj, dw_acc, N, A, X);
  return dig_check_launch();
}

extern "C" int dig_addattn_bwd_tokens(const void* xproj, const void* sproj_all, const float* w, const float* dv_all, const float* alpha_all,
                                      const void* dctx_all, int ldd, void* dxproj, void* dx, int T, int B, int N, int A, int X,
                                      hipStream_t stream) {
  if (!xproj || !sproj_all || !w || !dv_all || !alpha_all || !dctx_all || !dxproj || !dx || T <= 0 || B <= 0 || N <= 0 || A <= 0 || (A & 1) || X <= 0 ||
      (X & 1))
    return DIG_ERR_ARG;
  const size_t l1 = ((size_t)T * A + (size_t)T * 64) * 4, l2 = ((size_t)T * X + (size_t)T * 64) * 4;
  if (l1 > 150 * 1024 || l2 > 150 * 1024) return DIG_ERR_UNSUPPORTED;
  static size_t a1 = 0, a2 = 0;
  if (l1 > a1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(addattn_dxproj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1); a1 = l1; }
  if (l2 > a2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(addattn_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2); a2 = l2; }
  const dim3 grid(B, (N + 63) / 64);
  hipLaunchKernelGGL(addattn_dxproj_kernel, grid, dim3(256), l1, stream, (const bf16_t*)xproj, (const bf16_t*)sproj_all, w, dv_all, (bf16_t*)dxproj, T, B, N,
                     A);
  hipLaunchKernelGGL(addattn_dx_kernel, grid, dim3(256), l2, stream, alpha_all, (const bf16_t*)dctx_all, ldd, (bf16_t*)dx, T, B, N, X);
  return dig_check_launch();
}

extern "C" int dig_gru_cell_fwd(const void* gi, const void* gh, const float* s_prev, float* s, void* s_bf16, float* gates, int B, int S,
                                hipStream_t stream) {
  if (!gi || !gh || !s || !s_bf16 || !gates || B <= 0 || S <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((B * S + 255) / 256), dim3(256), 0, stream, (const bf16_t*)gi, (const bf16_t*)gh, s_prev, s, (bf16_t*)s_bf16,
                     gates, B, S);
  return dig_check_launch();
}

extern "C" int dig_gru_cell_bwd(const float* ds_a, const float* ds_b, const float* ds_c, const float* ds_d, const float* gates, const float* s_prev,
                                void* dgi, void* dgh, float* ds_prev, int B, int S, hipStream_t stream) {
  if (!ds_a || !gates || !dgi || !dgh || !ds_prev || B <= 0 || S <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((B * S + 255) / 256), dim3(256), 0, stream, ds_a, ds_b, ds_c, ds_d, gates, s_prev, (bf16_t*)dgi, (bf16_t*)dgh,
                     ds_prev, B, S);
  return dig_check_launch();
}

extern "C" int dig_embed_rows(const long long* tokens, const float* table, void* out, int ld, int rows, int cols, int vocab, hipStream_t stream) {
  if (!tokens || !table || !out || rows <= 0 || cols <= 0 || ld < cols || vocab <= 0) return DIG_ERR_ARG;
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tokens, table, (bf16_t*)out, ld, rows, cols, vocab);
  return dig_check_launch();
}
