// Token-level data movement kernels of the DiG step (all HBM-bound, coalesced on the channel axis).
//   patch_embed_{fwd,bwd}: PatchEmbed conv k4 s4 as a 48-wide dot per token, fused with the mask-token mix and the
//       sinusoid position add (modeling_finetune.py:173-196, modeling_pretrain_vit.py:89-99).
//   window_pool_{fwd,bwd}: PatchNet 'no_patchtrans' = adaptive_avg_pool2d of the 8x32 token grid to (1, 4)
//       (modeling_pretrain_moco_mim_ori.py:189-193).
//   mask_to_index: boolean mask -> ascending token indices per sample (the order boolean indexing yields,
//       engine_for_pretraining_moco.py:107, modeling_pretrain_moco_mim_ori.py:567) -- bit-exact integer work.
//   gather_rows / scatter_rows_add: masked-token row gather for the SimMIM decoder and its backward.
//   mim_target: un-normalise + 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' patchify + masked gather
//       (engine_for_pretraining_moco.py:85-111), one pass.
//   mse_fwd_bwd: F.mse_loss(reduction='mean') and its gradient (engine_for_pretraining_moco.py:141).
#include "common.h"

namespace {

constexpr int PE_TOK = 64;   // tokens per workgroup (forward)

// grid: ceil(n_tok / PE_TOK); block: D threads (one output channel each)
__global__ void patch_embed_fwd_kernel(const float* __restrict__ img, const float* __restrict__ W, const float* __restrict__ bias,
                                       const unsigned char* __restrict__ mask, const float* __restrict__ mask_token,
                                       const float* __restrict__ pos, bf16_t* __restrict__ out, int n_tok, int D, int gh, int gw,
                                       int Himg, int Wimg) {
  __shared__ float patch[PE_TOK][48];
  __shared__ unsigned char mk[PE_TOK];
  const int t0 = blockIdx.x * PE_TOK;
  const int ntok_img = gh * gw;
  for (int e = threadIdx.x; e < PE_TOK * 48; e += blockDim.x) {
    const int tl = e / 48, k = e - tl * 48;
    const int t = t0 + tl;
    float v = 0.f;
    if (t < n_tok) {
      const int b = t / ntok_img, n = t - b * ntok_img;
      const int ph = n / gw, pw = n - ph * gw;
      const int c = k >> 4, p1 = (k >> 2) & 3, p2 = k & 3;       // conv weight order (c, p1, p2)
      v = img[(((size_t)b * 3 + c) * Himg + ph * 4 + p1) * Wimg + pw * 4 + p2];
    }
    patch[tl][k] = v;
  }
  for (int e = threadIdx.x; e < PE_TOK; e += blockDim.x) mk[e] = (t0 + e < n_tok && mask) ? mask[t0 + e] : 0;
  const int d = threadIdx.x;
  float w[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) w[k] = W[d * 48 + k];
  const float bd = bias[d], mt = mask_token[d];
  __syncthreads();
  for (int tl = 0; tl < PE_TOK; ++tl) {
    const int t = t0 + tl;
    if (t >= n_tok) break;
    float acc = bd;
#pragma unroll
    for (int k = 0; k < 48; ++k) acc += patch[tl][k] * w[k];
    const int n = t % ntok_img;
    const float v = (mk[tl] ? mt : acc) + pos[(size_t)n * D + d];
    out[(size_t)t * D + d] = f2bf(v);
  }
}

// The same on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation = an fmaf chain): the thread-per-channel
// form above spends one LDS operand read per FMA and runs at a fifth of the fp32 rate (37 us per 128-image view; 4 launches per step at the
// head of both encoders).  A wave owns 32 tokens; its 48 patch values per token sit in registers (lane = token, 24 values per lane half: the
// k pairs are chosen so that a lane reads 12 aligned float2), the weights are staged once per workgroup into LDS as Wt[k'][D] and feed the
// A operand; D / 32 column blocks x 24 MFMAs per token tile.  Swapped roles (A = weights, B = patches) leave a lane with ITS token's
// channels in registers: bias / mask token / position are added in that layout and stored as 8-byte bf16 groups.
// k' = 2 j + h with j = c * 8 + p1 * 2 + q  <->  conv weight index (c, p1, p2 = 2 h + q).
__global__ __launch_bounds__(512) void patch_embed_fwd_mfma_kernel(const float* __restrict__ img, const float* __restrict__ W,
                                                                   const float* __restrict__ bias, const unsigned char* __restrict__ mask,
                                                                   const float* __restrict__ mask_token, const float* __restrict__ pos,
                                                                   bf16_t* __restrict__ out, int n_tok, int D, int gh, int gw, int Himg, int Wimg) {
  extern __shared__ __attribute__((aligned(16))) float Wt[];           // [48][D + 1] (odd pitch: the transposing fill is conflict-free)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, tl = lane & 31;
  const int WP = D + 1;
  if (tid < 480) {                                                     // 10 weight rows per pass, a thread keeps its k (and k')
    const int k = tid % 48, d0 = tid / 48;
    const int c = k >> 4, p1 = (k >> 2) & 3, p2 = k & 3;
    const int kk = 2 * (c * 8 + p1 * 2 + (p2 & 1)) + (p2 >> 1);
    for (int d = d0; d < D; d += 10) Wt[kk * WP + d] = W[d * 48 + k];
  }
  __syncthreads();
  const int ntok_img = gh * gw;
  const int n_tiles = (n_tok + 31) / 32;
  // 8 waves: a pair of waves shares a token tile and splits the column blocks (two waves per SIMD: one's epilogue loads / stores under
  // the other's MFMA chain)
  const int half = wave & 1, ncb = D / 32, cb0 = half * ((ncb + 1) / 2), cb1 = half ? ncb : (ncb + 1) / 2;
  for (int tile = blockIdx.x * 4 + (wave >> 1); tile < n_tiles; tile += gridDim.x * 4) {
    const int t = tile * 32 + tl;
    const bool tok_ok = t < n_tok;
    const int tc = tok_ok ? t : n_tok - 1;
    const int b = tc / ntok_img, n = tc - b * ntok_img;
    const int ph = n / gw, pw = n - ph * gw;
    float pv[24];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int p1 = 0; p1 < 4; ++p1) {
        const float2 v = *reinterpret_cast<const float2*>(img + (((size_t)b * 3 + c) * Himg + ph * 4 + p1) * Wimg + pw * 4 + 2 * hi);
        pv[c * 8 + p1 * 2] = v.x; pv[c * 8 + p1 * 2 + 1] = v.y;
      }
    const bool masked = mask && mask[tc];
    const float* posr = pos + (size_t)n * D;
    bf16_t* orow = out + (size_t)tc * D;
    for (int cb = cb0; cb < cb1; ++cb) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const int ch = cb * 32 + 16 * hi;
      float4 b4[4], m4[4], p4[4];                                          // the epilogue's operands: requested in front of the MFMA chain
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        b4[g] = *reinterpret_cast<const float4*>(bias + ch + 4 * g); m4[g] = *reinterpret_cast<const float4*>(mask_token + ch + 4 * g);
        p4[g] = *reinterpret_cast<const float4*>(posr + ch + 4 * g);
      }
      const float* wl = Wt + hi * WP + cb * 32 + tl;
#pragma unroll
      for (int j = 0; j < 24; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[2 * j * WP], pv[j], acc, 0, 0, 0);
      // acc: lane = token, registers = channels 8 g + 4 hi + (0..3); after the swaps lane (token, hi) owns channels 16 hi .. 16 hi + 15
      float x[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[qd]), __float_as_uint(acc[8 + qd]), false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 + qd]), __float_as_uint(acc[12 + qd]), false, false);
        x[qd] = __uint_as_float(s0[0]); x[4 + qd] = __uint_as_float(s0[1]);
        x[8 + qd] = __uint_as_float(s1[0]); x[12 + qd] = __uint_as_float(s1[1]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        x[4 * g] = (masked ? m4[g].x : x[4 * g] + b4[g].x) + p4[g].x; x[4 * g + 1] = (masked ? m4[g].y : x[4 * g + 1] + b4[g].y) + p4[g].y;
        x[4 * g + 2] = (masked ? m4[g].z : x[4 * g + 2] + b4[g].z) + p4[g].z; x[4 * g + 3] = (masked ? m4[g].w : x[4 * g + 3] + b4[g].w) + p4[g].w;
      }
      if (tok_ok) {
        *reinterpret_cast<uint4*>(orow + ch) = make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
        *reinterpret_cast<uint4*>(orow + ch + 8) = make_uint4(pack_bf2(x[8], x[9]), pack_bf2(x[10], x[11]), pack_bf2(x[12], x[13]), pack_bf2(x[14], x[15]));
      }
    }
  }
}

// grid: chunks of tokens; block: D threads.  dW[d][k] += sum_t (1-m_t) dy[t,d] patch[t,k]; dbias[d]; dmask_token[d]
__global__ void patch_embed_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ img,
                                       const unsigned char* __restrict__ mask, float* __restrict__ dW, float* __restrict__ dbias,
                                       float* __restrict__ dmask_token, int n_tok, int D, int gh, int gw, int Himg, int Wimg,
                                       int tok_per_block) {
  __shared__ float patch[PE_TOK][48];
  __shared__ unsigned char mk[PE_TOK];
  const int d = threadIdx.x;
  const int ntok_img = gh * gw;
  float acc[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) acc[k] = 0.f;
  float ab = 0.f, am = 0.f;
  const int tb = blockIdx.x * tok_per_block;
  const int te = min(n_tok, tb + tok_per_block);
  for (int t0 = tb; t0 < te; t0 += PE_TOK) {
    __syncthreads();
    for (int e = threadIdx.x; e < PE_TOK * 48; e += blockDim.x) {
      const int tl = e / 48, k = e - tl * 48;
      const int t = t0 + tl;
      float v = 0.f;
      if (t < te) {
        const int b = t / ntok_img, n = t - b * ntok_img;
        const int ph = n / gw, pw = n - ph * gw;
        const int c = k >> 4, p1 = (k >> 2) & 3, p2 = k & 3;
        v = img[(((size_t)b * 3 + c) * Himg + ph * 4 + p1) * Wimg + pw * 4 + p2];
      }
      patch[tl][k] = v;
    }
    for (int e = threadIdx.x; e < PE_TOK; e += blockDim.x) mk[e] = (t0 + e < te && mask) ? mask[t0 + e] : 0;
    __syncthreads();
    for (int tl = 0; tl < PE_TOK && t0 + tl < te; ++tl) {
      const float g = bf2f(dy[(size_t)(t0 + tl) * D + d]);
      if (mk[tl]) {
        am += g;
      } else {
        ab += g;
#pragma unroll
        for (int k = 0; k < 48; ++k) acc[k] += g * patch[tl][k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 48; ++k) atomicAdd(dW + d * 48 + k, acc[k]);
  atomicAdd(dbias + d, ab);
  atomicAdd(dmask_token + d, am);
}

// Column range of pooling window `win`: adaptive_avg_pool2d's bins [floor(win gw / nwin), ceil((win + 1) gw / nwin)) -- equal windows when nwin
// divides gw (README: --num_windows 4 on 32 columns), bins of 7 / 7 / 8 / 7 / 7 columns that overlap by one for the argparse default of 5
// (run_mae_pretraining_moco.py:143; PatchNet.forward, modeling_pretrain_moco_mim_ori.py:189-193)
__device__ __forceinline__ void pool_window(int win, int gw, int nwin, int& c_lo, int& wlen) {
  c_lo = (win * gw) / nwin;
  wlen = ((win + 1) * gw + nwin - 1) / nwin - c_lo;
}
// Last window whose bin holds column `col` (the first is floor(col nwin / gw)): at most the next one when nwin <= gw; with MORE windows than
// columns (ConvPatchNet pools its 1 x 4 map to (1, num_windows), :254) a column feeds ceil(nwin / gw) + 1 of them.
__device__ __forceinline__ int pool_last_window(int col, int gw, int nwin) { return ((col + 1) * nwin + gw - 1) / gw - 1; }

// x [n_img, gh*gw, D] bf16 -> out [n_img*nwin, D] (fp32 or bf16): mean over all gh rows and the window's columns
template <typename OutT>
__global__ void window_pool_fwd_kernel(const bf16_t* __restrict__ x, OutT* __restrict__ out, int n_img, int gh, int gw,
                                       int nwin, int D) {
  const int idx = blockIdx.x;                 // img * nwin + win
  const int img = idx / nwin, win = idx - img * nwin;
  int c_lo, wlen;
  pool_window(win, gw, nwin, c_lo, wlen);
  const float inv = 1.0f / (gh * wlen);
  for (int d2 = threadIdx.x; d2 < D / 2; d2 += blockDim.x) {
    float a0 = 0.f, a1 = 0.f;
    for (int r = 0; r < gh; ++r)
      for (int c = 0; c < wlen; ++c) {
        const int n = r * gw + c_lo + c;
        const unsigned u = *reinterpret_cast<const unsigned*>(x + ((size_t)img * gh * gw + n) * D + d2 * 2);
        a0 += bf2f((bf16_t)(u & 0xffff));
        a1 += bf2f((bf16_t)(u >> 16));
      }
    if constexpr (sizeof(OutT) == 4) {
      out[(size_t)idx * D + d2 * 2] = a0 * inv;
      out[(size_t)idx * D + d2 * 2 + 1] = a1 * inv;
    } else {
      *reinterpret_cast<unsigned*>(out + (size_t)idx * D + d2 * 2) = pack_bf2(a0 * inv, a1 * inv);
    }
  }
}

__device__ __forceinline__ void ld8_bf16(const bf16_t* p, float* v) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = bf2f((bf16_t)(w[k] & 0xffff)); v[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16)); }
}
__device__ __forceinline__ void st8_bf16(bf16_t* p, const float* v) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

// The same with 16-byte loads (D % 8 == 0, D / 8 <= 256): a workgroup = one (image, window); thread (g, ch) adds tokens g, g + G, ... of its
// 8 channels (G = 256 / (D / 8) token rows in flight instead of one 4-byte load per thread at a time: 20 -> 6 us for 128 x 4 windows), the
// G partial sums are folded through LDS in g order (fixed order: bit-reproducible)
template <typename OutT>
__global__ __launch_bounds__(256) void window_pool_fwd16_kernel(const bf16_t* __restrict__ x, OutT* __restrict__ out, int n_img, int gh, int gw,
                                                                int nwin, int D) {
  __shared__ float red[256][9];
  const int idx = blockIdx.x;
  const int img = idx / nwin, win = idx - img * nwin;
  int c_lo, wlen;
  pool_window(win, gw, nwin, c_lo, wlen);
  const int ntw = gh * wlen;
  const int c8 = D >> 3, G = 256 / c8;
  const int tid = threadIdx.x, g = tid / c8, ch = tid - g * c8;
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.f;
  if (g < G) {
    for (int t = g; t < ntw; t += G) {
      const int r = t / wlen, c = t - r * wlen;
      const int n = r * gw + c_lo + c;
      float v[8];
      ld8_bf16(x + ((size_t)img * gh * gw + n) * D + ch * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[tid][k] = a[k];
  __syncthreads();
  if (g == 0) {
    const float inv = 1.0f / ntw;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += red[q * c8 + ch][k];
      s[k] = t * inv;
    }
    if constexpr (sizeof(OutT) == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) out[(size_t)idx * D + ch * 8 + k] = s[k];
    } else {
      st8_bf16(out + (size_t)idx * D + ch * 8, s);
    }
  }
}

// dx[img, n, :] (+)= sum over the windows that hold n's column of dpool[img*nwin + win, :] / (gh*wlen(win)) with 16-byte accesses: one item =
// 8 channels of one token.  (One window per column when nwin divides gw; neighbouring uneven windows share a column: both contribute, in
// window order.)
__global__ __launch_bounds__(256) void window_pool_bwd16_kernel(const bf16_t* __restrict__ dpool, bf16_t* __restrict__ dx, int n_img, int gh, int gw,
                                                                int nwin, int D, int accumulate) {
  const int c8 = D >> 3, ntok = gh * gw;
  const size_t total = (size_t)n_img * ntok * c8;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(e % c8);
    const size_t t = e / c8;
    const int img = (int)(t / ntok), n = (int)(t - (size_t)img * ntok);
    const int col = n % gw, w0 = (col * nwin) / gw;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    for (int win = max(0, w0 - 1); win <= min(nwin - 1, pool_last_window(col, gw, nwin)); ++win) {
      int c_lo, wlen;
      pool_window(win, gw, nwin, c_lo, wlen);
      if (col < c_lo || col >= c_lo + wlen) continue;
      const float inv = 1.0f / (gh * wlen);
      float u[8];
      ld8_bf16(dpool + ((size_t)img * nwin + win) * D + ch * 8, u);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += u[k] * inv;
    }
    bf16_t* o = dx + t * D + ch * 8;
    if (accumulate) {
      float w[8];
      ld8_bf16(o, w);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += w[k];
    }
    st8_bf16(o, v);
  }
}

// dx[img, n, :] (+)= dpool[img*nwin + win(n), :] / (gh*wlen)
__global__ void window_pool_bwd_kernel(const bf16_t* __restrict__ dpool, bf16_t* __restrict__ dx, int n_img, int gh, int gw,
                                       int nwin, int D, int accumulate) {
  const int t = blockIdx.x;                    // token index over n_img*gh*gw
  const int ntok = gh * gw;
  const int img = t / ntok, n = t - img * ntok;
  const int col = n % gw, w0 = (col * nwin) / gw;
  for (int d2 = threadIdx.x; d2 < D / 2; d2 += blockDim.x) {
    float a0 = 0.f, a1 = 0.f;
    for (int win = max(0, w0 - 1); win <= min(nwin - 1, pool_last_window(col, gw, nwin)); ++win) {
      int c_lo, wlen;
      pool_window(win, gw, nwin, c_lo, wlen);
      if (col < c_lo || col >= c_lo + wlen) continue;
      const float inv = 1.0f / (gh * wlen);
      const unsigned u = *reinterpret_cast<const unsigned*>(dpool + ((size_t)img * nwin + win) * D + d2 * 2);
      a0 += bf2f((bf16_t)(u & 0xffff)) * inv;
      a1 += bf2f((bf16_t)(u >> 16)) * inv;
    }
    unsigned* o = reinterpret_cast<unsigned*>(dx + (size_t)t * D + d2 * 2);
    if (accumulate) {
      const unsigned v = *o;
      a0 += bf2f((bf16_t)(v & 0xffff));
      a1 += bf2f((bf16_t)(v >> 16));
    }
    *o = pack_bf2(a0, a1);
  }
}

// one wave per sample: idx[b, j] = b*N + (j-th set position of mask[b, :]) ; count[b] = popcount
__global__ void mask_to_index_kernel(const unsigned char* __restrict__ mask, int* __restrict__ idx, int* __restrict__ count,
                                     int B, int N, int max_per_sample) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int base = 0;
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int n = n0 + lane;
    const bool set = n < N && mask[(size_t)b * N + n] != 0;
    const unsigned long long bal = __ballot(set);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (set && base + before < max_per_sample) idx[(size_t)b * max_per_sample + base + before] = b * N + n;
    base += __popcll(bal);
  }
  if (lane == 0) count[b] = base;
}

// dst[m, :] = src[idx[m], :]  (rows m >= M are zero-filled up to M_pad)
__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst, int M,
                                   int D8) {
  const int m = blockIdx.x;
  const uint4* s = m < M ? reinterpret_cast<const uint4*>(src) + (size_t)idx[m] * D8 : nullptr;
  uint4* d = reinterpret_cast<uint4*>(dst) + (size_t)m * D8;
  for (int c = threadIdx.x; c < D8; c += blockDim.x) d[c] = s ? s[c] : make_uint4(0, 0, 0, 0);
}

// dst[idx[m], :] += src[m, :]   (idx unique)
__global__ void scatter_rows_add_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst,
                                        int M, int D8) {
  const int m = blockIdx.x;
  const uint4* s = reinterpret_cast<const uint4*>(src) + (size_t)m * D8;
  uint4* d = reinterpret_cast<uint4*>(dst) + (size_t)idx[m] * D8;
  for (int c = threadIdx.x; c < D8; c += blockDim.x) {
    const uint4 a = s[c], b = d[c];
    const unsigned aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = pack_bf2(bf2f((bf16_t)(aa[k] & 0xffff)) + bf2f((bf16_t)(bb[k] & 0xffff)), bf2f((bf16_t)(aa[k] >> 16)) + bf2f((bf16_t)(bb[k] >> 16)));
    d[c] = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

// target[m, (p1*4+p2)*3 + c] = img[b, c, ph*4+p1, pw*4+p2] * 0.5 + 0.5  for token idx[m] = b*N + n
// normalize != 0 (`normlize_target`, engine_for_pretraining_moco.py:88-93): per patch and channel, (x - mean) / (sqrt(unbiased var) + 1e-6)
// over the patch's 16 pixels of the un-normalised image
__global__ void mim_target_kernel(const float* __restrict__ img, const int* __restrict__ idx, float* __restrict__ target, int M,
                                  int gh, int gw, int Himg, int Wimg, int normalize) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * 48) return;
  const int m = e / 48, k = e - m * 48;
  const int t = idx[m];
  const int ntok = gh * gw;
  const int b = t / ntok, n = t - b * ntok;
  const int ph = n / gw, pw = n - ph * gw;
  const int c = k % 3, p = k / 3, p1 = p >> 2, p2 = p & 3;
  const float* base = img + (((size_t)b * 3 + c) * Himg + ph * 4) * Wimg + pw * 4;
  const float x = base[p1 * Wimg + p2] * 0.5f + 0.5f;
  if (!normalize) { target[e] = x; return; }
  float v[16], mean = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { v[q] = base[(q >> 2) * Wimg + (q & 3)] * 0.5f + 0.5f; mean += v[q]; }
  mean *= (1.0f / 16.0f);
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) var += (v[q] - mean) * (v[q] - mean);
  target[e] = (x - mean) / (sqrtf(var * (1.0f / 15.0f)) + 1e-6f);
}

// loss += sum (pred - target)^2 * inv_count ;  dpred = gscale * 2 * (pred - target) * inv_count (bf16, ld_d, pad cols zeroed)
__global__ void mse_fwd_bwd_kernel(const float* __restrict__ pred, int ld_p, const float* __restrict__ target, int M, int C,
                                   float inv_count, float gscale, float* __restrict__ loss, bf16_t* __restrict__ dpred, int ld_d,
                                   float* __restrict__ ws) {
  float acc = 0.f;
  const int total = M * ld_d;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int m = e / ld_d, c = e - m * ld_d;
    float g = 0.f;
    if (c < C) {
      const float df = pred[(size_t)m * ld_p + c] - target[(size_t)m * C + c];
      acc += df * df;
      g = 2.f * df * inv_count * gscale;
    }
    if (dpred) dpred[e] = f2bf(g);
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (!loss) return;
  const float bs = (part[0] + part[1] + part[2] + part[3]) * inv_count;
  if (!ws) {
    if (threadIdx.x == 0) atomicAdd(loss, bs);                                                         // one atomic per block
    return;
  }
  // with a workspace: the block sums are added in block order by the last workgroup to finish (bit-reproducible loss value)
  __shared__ float red[256];
  const float mine[1] = {bs};
  float tot[1];
  if (dig_grid_sum_last<1>(ws, mine, tot, red)) loss[0] += tot[0];
}

__global__ void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = reinterpret_cast<const uint4*>(a)[i], y = reinterpret_cast<const uint4*>(b)[i];
    const unsigned xx[4] = {x.x, x.y, x.z, x.w}, yy[4] = {y.x, y.y, y.z, y.w};
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = pack_bf2(bf2f((bf16_t)(xx[k] & 0xffff)) + bf2f((bf16_t)(yy[k] & 0xffff)), bf2f((bf16_t)(xx[k] >> 16)) + bf2f((bf16_t)(yy[k] >> 16)));
    reinterpret_cast<uint4*>(o)[i] = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

// d(pre-activation) = d(act) * gelu'(pre)   (FFN backward between the fc2 dgrad and the fc1 dgrad/wgrad)
__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ pre, bf16_t* __restrict__ dpre, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = reinterpret_cast<const uint4*>(dact)[i], y = reinterpret_cast<const uint4*>(pre)[i];
    const unsigned xx[4] = {x.x, x.y, x.z, x.w}, yy[4] = {y.x, y.y, y.z, y.w};
    unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = pack_bf2(bf2f((bf16_t)(xx[k] & 0xffff)) * dgelu_acc_f(bf2f((bf16_t)(yy[k] & 0xffff))),
                      bf2f((bf16_t)(xx[k] >> 16)) * dgelu_acc_f(bf2f((bf16_t)(yy[k] >> 16))));
    reinterpret_cast<uint4*>(dpre)[i] = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

// partial[blockIdx.y][c] = sum over this block's rows of x[r, c]   (bias gradients, stage 1; no atomics: same-address
// device-scope atomics from ~2000 workgroups serialise at the memory side).  16-B loads: a wave covers 512 columns.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int rows, int C, int ld,
                                                     int rows_per_block) {
  __shared__ float red[4][64][8 + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int r = r0 + wv;
    for (; r + 12 < r1; r += 16) {                     // 4 independent 16-B loads in flight per lane
      uint4 u[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const uint4*>(x + (size_t)(r + 4 * q) * ld + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w[4] = {u[q].x, u[q].y, u[q].z, u[q].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[2 * k] += bf2f((bf16_t)(w[k] & 0xffff)); a[2 * k + 1] += bf2f((bf16_t)(w[k] >> 16)); }
      }
    }
    for (; r < r1; r += 4) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (size_t)r * ld + c);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[2 * k] += bf2f((bf16_t)(w[k] & 0xffff)); a[2 * k + 1] += bf2f((bf16_t)(w[k] >> 16)); }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wv][lane][k] = a[k];
  __syncthreads();
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int l = e >> 3, k = e & 7;
    const int cc = blockIdx.x * 512 + e;
    if (cc < C) partial[(size_t)blockIdx.y * C + cc] = red[0][l][k] + red[1][l][k] + red[2][l][k] + red[3][l][k];
  }
}

// out[c] += sum_b partial[b][c]; a block owns 8 columns: 2 float4 lanes x 128 row-groups (C % 8 == 0)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nb, int C, float* __restrict__ out) {
  __shared__ float4 red[128][2];
  const int cx = threadIdx.x & 1, ry = threadIdx.x >> 1;
  const int c = blockIdx.x * 8 + cx * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = ry; b < nb; b += 128) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)b * C + c);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  red[ry][cx] = a;
  __syncthreads();
  for (int st = 64; st > 0; st >>= 1) {
    if (ry < st) {
      const float4 o = red[ry + st][cx];
      float4 m = red[ry][cx];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      red[ry][cx] = m;
    }
    __syncthreads();
  }
  if (ry == 0) {
    const float4 m = red[0][cx];
    out[c] += m.x; out[c + 1] += m.y; out[c + 2] += m.z; out[c + 3] += m.w;
  }
}

// the same for up to DIG_COLSUM_MAX_SEGS partial sets in ONE launch (an encoder block's LayerNorm and bias gradients): partial rows of
// segment k are `stride` floats apart (a LayerNorm workspace interleaves its three vectors), blocks [first[k], first[k+1]) own 8 columns each
struct ColsumSegs {
  const float* part[DIG_COLSUM_MAX_SEGS];
  float* out[DIG_COLSUM_MAX_SEGS];
  long long stride[DIG_COLSUM_MAX_SEGS];
  int n_parts[DIG_COLSUM_MAX_SEGS];
  int first[DIG_COLSUM_MAX_SEGS + 1];
  int n_segs;
};
__global__ __launch_bounds__(256) void colsum_finalize_multi_kernel(ColsumSegs a) {
  __shared__ float4 red[128][2];
  int k = 0;
  while (k + 1 < a.n_segs && (int)blockIdx.x >= a.first[k + 1]) ++k;
  const float* __restrict__ partial = a.part[k];
  const long long stride = a.stride[k];
  const int nb = a.n_parts[k];
  const int cx = threadIdx.x & 1, ry = threadIdx.x >> 1;
  const int c = ((int)blockIdx.x - a.first[k]) * 8 + cx * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = ry; b < nb; b += 128) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)b * stride + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[ry][cx] = acc;
  __syncthreads();
  for (int st = 64; st > 0; st >>= 1) {
    if (ry < st) {
      const float4 o = red[ry + st][cx];
      float4 m = red[ry][cx];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      red[ry][cx] = m;
    }
    __syncthreads();
  }
  if (ry == 0) {
    float* out = a.out[k];
    const float4 m = red[0][cx];
    out[c] += m.x; out[c + 1] += m.y; out[c + 2] += m.z; out[c + 3] += m.w;
  }
}

// The same with 32 columns per block (every C a multiple of 32): a partial row is read in full 128-byte lines, 8 lanes per row, 32 row groups --
// the 8-column form above reads 32 bytes per row and line, and a launch over all twelve encoder blocks' partial rows (216 MB, the deferred
// reductions of the single-process backward) took 141 us on the caller's stream with it.
__global__ __launch_bounds__(256) void colsum_finalize_multi32_kernel(ColsumSegs a) {
  __shared__ float4 red[32][8];
  int k = 0;
  while (k + 1 < a.n_segs && (int)blockIdx.x >= a.first[k + 1]) ++k;
  const float* __restrict__ partial = a.part[k];
  const long long stride = a.stride[k];
  const int nb = a.n_parts[k];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = ((int)blockIdx.x - a.first[k]) * 32 + cx * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = ry; b < nb; b += 32) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)b * stride + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[ry][cx] = acc;
  __syncthreads();
  for (int st = 16; st > 0; st >>= 1) {
    if (ry < st) {
      const float4 o = red[ry + st][cx];
      float4 m = red[ry][cx];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      red[ry][cx] = m;
    }
    __syncthreads();
  }
  if (ry == 0) {
    float* out = a.out[k];
    const float4 m = red[0][cx];
    out[c] += m.x; out[c + 1] += m.y; out[c + 2] += m.z; out[c + 3] += m.w;
  }
}

// P[t, k] (bf16, ld 64) = masked ? 0 : img patch element k (conv order c,p1,p2), k < 48; pad columns 48..63 = 0.
// Lets the patch-embed weight gradient run as an MFMA wgrad GEMM (dW[D,48] = dy^T P) instead of a VALU loop.
__global__ __launch_bounds__(256) void patchify_bf16_kernel(const float* __restrict__ img, const unsigned char* __restrict__ mask,
                                                            bf16_t* __restrict__ out, int n_tok, int gh, int gw, int Himg, int Wimg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (token, 4-element group): 16 groups per token
  if (e >= n_tok * 16) return;
  const int t = e >> 4, grp = e & 15;
  const int ntok = gh * gw;
  const int b = t / ntok, n = t - b * ntok;
  const int ph = n / gw, pw = n - ph * gw;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (grp < 12 && !(mask && mask[t])) {
    const int c = grp >> 2, p1 = grp & 3;                  // k = c*16 + p1*4 + p2
    const float4 px = *reinterpret_cast<const float4*>(img + (((size_t)b * 3 + c) * Himg + ph * 4 + p1) * Wimg + pw * 4);
    v[0] = px.x; v[1] = px.y; v[2] = px.z; v[3] = px.w;
  }
  *reinterpret_cast<uint2*>(out + (size_t)t * 64 + grp * 4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}

// out0[c] += sum over rows with mask==0 of x[r,c]; out1[c] += sum over rows with mask!=0   (patch-embed bias / mask_token grads)
__global__ __launch_bounds__(256) void colsum_masked_kernel(const bf16_t* __restrict__ x, const unsigned char* __restrict__ mask,
                                                            float* __restrict__ part0, float* __restrict__ part1, int rows, int C,
                                                            int rows_per_block) {
  __shared__ float red[2][4][64][8 + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a[2][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[0][k] = 0.f; a[1][k] = 0.f; }
  if (c < C) {
    for (int r = r0 + wv; r < r1; r += 4) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (size_t)r * C + c);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
      const bool m = mask[r] != 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = bf2f((bf16_t)(w[k] & 0xffff)), hi = bf2f((bf16_t)(w[k] >> 16));
        a[0][2 * k] += m ? 0.f : lo; a[0][2 * k + 1] += m ? 0.f : hi;
        a[1][2 * k] += m ? lo : 0.f; a[1][2 * k + 1] += m ? hi : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][wv][lane][k] = a[0][k]; red[1][wv][lane][k] = a[1][k]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int which = e >> 9, l = (e & 511) >> 3, k = e & 7;
    const int cc = blockIdx.x * 512 + (e & 511);
    if (cc < C) (which ? part1 : part0)[(size_t)blockIdx.y * C + cc] = red[which][0][l][k] + red[which][1][l][k] + red[which][2][l][k] + red[which][3][l][k];
  }
}

// out = in * (element mask / (1-p)) * (drop-path factor of the row's sample): nn.Dropout / timm drop_path applied to a [rows, cols]
// bf16 tensor, forward (pos_drop, decoder embedding dropout) or backward (the gradient of every dropped branch); 8 columns per thread
__global__ __launch_bounds__(256) void dropout_apply_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long long n8, int cols,
                                                            dig_dropout_t d) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n8) return;
  const int c8 = cols >> 3;
  const int i = (int)(t / c8), j = (int)(t - (long long)i * c8) * 8;
  const uint4 x = reinterpret_cast<const uint4*>(in)[t];
  const unsigned w[4] = {x.x, x.y, x.z, x.w};
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = bf2f((bf16_t)(w[e] & 0xffff)); v[2 * e + 1] = bf2f((bf16_t)(w[e] >> 16)); }
  dig_drop_apply8(v, d, i, j, cols);
  reinterpret_cast<uint4*>(out)[t] = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

}  // namespace

extern "C" int dig_patch_embed_fwd(const float* img, const float* W, const float* bias, const unsigned char* mask,
                                   const float* mask_token, const float* pos, void* out, int n_img, int gh, int gw, int D,
                                   hipStream_t stream) {
  if (!img || !W || !bias || !mask_token || !pos || !out || n_img <= 0 || D <= 0 || D > 1024 || (D & 63)) return DIG_ERR_ARG;
  const int n_tok = n_img * gh * gw;
  if ((D & 31) == 0 && 48 * D * 4 <= 96 * 1024 && aligned16(img) && aligned16(bias) && aligned16(mask_token) && aligned16(pos) && aligned16(out)) {
    // fp32 MFMA form (exact f32 arithmetic, another summation order than the thread-per-channel kernel: both add k in a fixed order)
    static bool attr[DIG_MAX_DEVICES] = {};
    const int dev = dig_device();
    if (!attr[dev]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(patch_embed_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr[dev] = true;
    }
    const int tiles = (n_tok + 31) / 32;
    const int grid = std::min(256, (tiles + 3) / 4);
    hipLaunchKernelGGL(patch_embed_fwd_mfma_kernel, dim3(grid), dim3(512), 48 * (D + 1) * 4, stream, img, W, bias, mask, mask_token, pos, (bf16_t*)out,
                       n_tok, D, gh, gw, gh * 4, gw * 4);
    return dig_check_launch();
  }
  hipLaunchKernelGGL(patch_embed_fwd_kernel, dim3((n_tok + PE_TOK - 1) / PE_TOK), dim3(D), 0, stream, img, W, bias, mask,
                     mask_token, pos, (bf16_t*)out, n_tok, D, gh, gw, gh * 4, gw * 4);
  return dig_check_launch();
}

extern "C" int dig_patch_embed_bwd(const void* dy, const float* img, const unsigned char* mask, float* dW, float* dbias,
                                   float* dmask_token, int n_img, int gh, int gw, int D, hipStream_t stream) {
  if (!dy || !img || !dW || !dbias || !dmask_token || n_img <= 0 || D <= 0 || D > 1024 || (D & 63)) return DIG_ERR_ARG;
  const int n_tok = n_img * gh * gw;
  int tpb = 512;
  while ((n_tok + tpb - 1) / tpb > 256) tpb *= 2;
  hipLaunchKernelGGL(patch_embed_bwd_kernel, dim3((n_tok + tpb - 1) / tpb), dim3(D), 0, stream, (const bf16_t*)dy, img, mask,
                     dW, dbias, dmask_token, n_tok, D, gh, gw, gh * 4, gw * 4, tpb);
  return dig_check_launch();
}

extern "C" int dig_window_pool_fwd(const void* x, void* out, int out_is_f32, int n_img, int gh, int gw, int nwin, int D,
                                   hipStream_t stream) {
  if (!x || !out || n_img <= 0 || nwin <= 0 || gh <= 0 || gw <= 0 || (D & 1)) return DIG_ERR_ARG;
  if ((D & 7) == 0 && D / 8 <= 256 && aligned16(x) && aligned16(out)) {
    if (out_is_f32)
      hipLaunchKernelGGL(window_pool_fwd16_kernel<float>, dim3(n_img * nwin), dim3(256), 0, stream, (const bf16_t*)x, (float*)out, n_img, gh, gw, nwin, D);
    else
      hipLaunchKernelGGL(window_pool_fwd16_kernel<bf16_t>, dim3(n_img * nwin), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, n_img, gh, gw, nwin, D);
    return dig_check_launch();
  }
  if (out_is_f32)
    hipLaunchKernelGGL(window_pool_fwd_kernel<float>, dim3(n_img * nwin), dim3(std::min(256, D / 2)), 0, stream, (const bf16_t*)x,
                       (float*)out, n_img, gh, gw, nwin, D);
  else
    hipLaunchKernelGGL(window_pool_fwd_kernel<bf16_t>, dim3(n_img * nwin), dim3(std::min(256, D / 2)), 0, stream, (const bf16_t*)x,
                       (bf16_t*)out, n_img, gh, gw, nwin, D);
  return dig_check_launch();
}

extern "C" int dig_window_pool_bwd(const void* dpool, void* dx, int n_img, int gh, int gw, int nwin, int D, int accumulate,
                                   hipStream_t stream) {
  if (!dpool || !dx || n_img <= 0 || nwin <= 0 || gh <= 0 || gw <= 0 || (D & 1)) return DIG_ERR_ARG;
  if ((D & 7) == 0 && aligned16(dpool) && aligned16(dx)) {
    const size_t total = (size_t)n_img * gh * gw * (D / 8);
    hipLaunchKernelGGL(window_pool_bwd16_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)dpool, (bf16_t*)dx, n_img, gh, gw, nwin, D, accumulate);
    return dig_check_launch();
  }
  hipLaunchKernelGGL(window_pool_bwd_kernel, dim3(n_img * gh * gw), dim3(std::min(256, D / 2)), 0, stream, (const bf16_t*)dpool,
                     (bf16_t*)dx, n_img, gh, gw, nwin, D, accumulate);
  return dig_check_launch();
}

// out[v * B + b][n] = (mask[b][v][n] != 0) && v < keep_views: the loader's [B, V, N] mask (any of five element types) as the view-major
// uint8 rows the encoder reads, with the views that are not masked (only_mim_on_ori_img: view 1) zeroed -- the bool cast, the fill, the
// permute copy and the uint8 cast of engine_for_pretraining_moco.py:99-104 / modeling_pretrain_moco_mim_ori.py:497 in one launch
template <typename T>
__global__ void mask_views_u8_kernel(const T* __restrict__ mask, unsigned char* __restrict__ out, int B, int V, int N, int keep_views) {
  const size_t total = (size_t)B * V * N;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(e % N);
    const size_t r = e / N;
    const int b = (int)(r % B), v = (int)(r / B);
    out[e] = (v < keep_views && mask[((size_t)b * V + v) * N + n] != (T)0) ? 1 : 0;
  }
}

extern "C" int dig_mask_views_u8(const void* mask, int elem_kind, int B, int V, int N, int keep_views, unsigned char* out, hipStream_t stream) {
  if (!mask || !out || B <= 0 || V <= 0 || N <= 0 || keep_views < 0) return DIG_ERR_ARG;
  const size_t total = (size_t)B * V * N;
  const int grid = (int)std::min<size_t>(1024, (total + 255) / 256);
  switch (elem_kind) {                                                 // 0: 1-byte (bool / uint8), 1: fp32, 2: fp64, 3: int32, 4: int64
    case 0: hipLaunchKernelGGL(mask_views_u8_kernel<unsigned char>, dim3(grid), dim3(256), 0, stream, (const unsigned char*)mask, out, B, V, N, keep_views); break;
    case 1: hipLaunchKernelGGL(mask_views_u8_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)mask, out, B, V, N, keep_views); break;
    case 2: hipLaunchKernelGGL(mask_views_u8_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)mask, out, B, V, N, keep_views); break;
    case 3: hipLaunchKernelGGL(mask_views_u8_kernel<int>, dim3(grid), dim3(256), 0, stream, (const int*)mask, out, B, V, N, keep_views); break;
    case 4: hipLaunchKernelGGL(mask_views_u8_kernel<long long>, dim3(grid), dim3(256), 0, stream, (const long long*)mask, out, B, V, N, keep_views); break;
    default: return DIG_ERR_UNSUPPORTED;
  }
  return dig_check_launch();
}

extern "C" int dig_mask_to_index(const unsigned char* mask, int* idx, int* count, int B, int N, int max_per_sample,
                                 hipStream_t stream) {
  if (!mask || !idx || !count || B <= 0 || N <= 0 || max_per_sample <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(mask_to_index_kernel, dim3(B), dim3(64), 0, stream, mask, idx, count, B, N, max_per_sample);
  return dig_check_launch();
}

extern "C" int dig_gather_rows(const void* src, const int* idx, void* dst, int M, int M_pad, int D, hipStream_t stream) {
  if (!src || !idx || !dst || M <= 0 || M_pad < M || (D & 7)) return DIG_ERR_ARG;
  if (!aligned16(src) || !aligned16(dst)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(M_pad), dim3(64), 0, stream, (const bf16_t*)src, idx, (bf16_t*)dst, M, D / 8);
  return dig_check_launch();
}

extern "C" int dig_scatter_rows_add(const void* src, const int* idx, void* dst, int M, int D, hipStream_t stream) {
  if (!src || !idx || !dst || M <= 0 || (D & 7)) return DIG_ERR_ARG;
  if (!aligned16(src) || !aligned16(dst)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(M), dim3(64), 0, stream, (const bf16_t*)src, idx, (bf16_t*)dst, M, D / 8);
  return dig_check_launch();
}

extern "C" int dig_mim_target(const float* img, const int* idx, float* target, int M, int gh, int gw, int normalize, hipStream_t stream) {
  if (!img || !idx || !target || M <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(mim_target_kernel, dim3((M * 48 + 255) / 256), dim3(256), 0, stream, img, idx, target, M, gh, gw, gh * 4,
                     gw * 4, normalize);
  return dig_check_launch();
}

extern "C" int dig_mse_fwd_bwd_ws(const float* pred, int ld_pred, const float* target, int M, int C, float gscale, float* loss,
                                  void* dpred, int ld_dpred, float* workspace, hipStream_t stream) {
  if (!pred || !target || M <= 0 || C <= 0 || ld_pred < C || (dpred && ld_dpred < C)) return DIG_ERR_ARG;
  const int ldd = dpred ? ld_dpred : C;
  const int total = M * ldd;
  hipLaunchKernelGGL(mse_fwd_bwd_kernel, dim3(std::min(256, (total + 255) / 256)), dim3(256), 0, stream, pred, ld_pred, target, M, C,
                     1.0f / ((float)M * C), gscale, loss, (bf16_t*)dpred, ldd, workspace);
  return dig_check_launch();
}
extern "C" int dig_mse_fwd_bwd(const float* pred, int ld_pred, const float* target, int M, int C, float gscale, float* loss,
                               void* dpred, int ld_dpred, hipStream_t stream) {
  return dig_mse_fwd_bwd_ws(pred, ld_pred, target, M, C, gscale, loss, dpred, ld_dpred, nullptr, stream);
}

extern "C" int dig_add_bf16(const void* a, const void* b, void* out, long long n, hipStream_t stream) {
  if (!a || !b || !out || n <= 0 || (n & 7)) return DIG_ERR_ARG;
  if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return DIG_ERR_ALIGN;
  const size_t n8 = (size_t)n / 8;
  hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)std::min<size_t>(4096, (n8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)a,
                     (const bf16_t*)b, (bf16_t*)out, n8);
  return dig_check_launch();
}

extern "C" int dig_gelu_bwd(const void* dact, const void* pre, void* dpre, long long n, hipStream_t stream) {
  if (!dact || !pre || !dpre || n <= 0 || (n & 7)) return DIG_ERR_ARG;
  if (!aligned16(dact) || !aligned16(pre) || !aligned16(dpre)) return DIG_ERR_ALIGN;
  const size_t n8 = (size_t)n / 8;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)std::min<size_t>(8192, (n8 + 255) / 256)), dim3(256), 0, stream,
                     (const bf16_t*)dact, (const bf16_t*)pre, (bf16_t*)dpre, n8);
  return dig_check_launch();
}

static inline int colsum_rows_per_block(int rows, int C) {
  const int cb = (C + 511) / 512;
  int rpb = 64;
  while ((long)cb * ((rows + rpb - 1) / rpb) > 512) rpb *= 2;
  return rpb;
}

extern "C" long long dig_colsum_workspace_bytes(int rows, int C) {
  const int rpb = colsum_rows_per_block(rows, C);
  return (long long)((rows + rpb - 1) / rpb) * C * sizeof(float);
}

extern "C" int dig_colsum(const void* x, float* out, float* workspace, int rows, int C, int ld, hipStream_t stream) {
  if (!x || !out || !workspace || rows <= 0 || C <= 0 || (C & 7) || (ld & 7)) return DIG_ERR_ARG;
  if (!aligned16(x)) return DIG_ERR_ALIGN;
  const int cb = (C + 511) / 512;
  const int rpb = colsum_rows_per_block(rows, C);
  const int nb = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_kernel, dim3(cb, nb), dim3(256), 0, stream, (const bf16_t*)x, workspace, rows, C, ld, rpb);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C / 8), dim3(256), 0, stream, workspace, nb, C, out);
  return dig_check_launch();
}

// out[c] += sum_b partials[b][c]  (the per-64-row column sums dig_gemm_bf16 writes with colsum_partials)
extern "C" int dig_colsum_partials(const float* partials, int n_parts, int C, float* out, hipStream_t stream) {
  if (!partials || !out || n_parts <= 0 || C <= 0 || (C & 7)) return DIG_ERR_ARG;
  if (!aligned16(partials)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C / 8), dim3(256), 0, stream, partials, n_parts, C, out);
  return dig_check_launch();
}

extern "C" int dig_colsum_partials_multi(const dig_colsum_seg_t* segs, int n_segs, hipStream_t stream) {
  if (!segs || n_segs < 1 || n_segs > DIG_COLSUM_MAX_SEGS) return DIG_ERR_ARG;
  ColsumSegs a;
  a.n_segs = n_segs;
  // more than one block's worth of segments with every width a multiple of 32: the full-line form (32 columns per workgroup)
  bool wide = n_segs > 12;
  for (int k = 0; k < n_segs && wide; ++k) wide = (segs[k].C & 31) == 0;
  const int cols = wide ? 32 : 8;
  int blocks = 0;
  for (int k = 0; k < n_segs; ++k) {
    const dig_colsum_seg_t& g = segs[k];
    if (!g.partials || !g.out || g.n_parts <= 0 || g.C <= 0 || (g.C & 7) || g.stride < g.C || (g.stride & 3)) return DIG_ERR_ARG;
    if (!aligned16(g.partials)) return DIG_ERR_ALIGN;
    a.part[k] = g.partials; a.out[k] = g.out; a.stride[k] = g.stride; a.n_parts[k] = g.n_parts;
    a.first[k] = blocks;
    blocks += g.C / cols;
  }
  a.first[n_segs] = blocks;
  if (wide) hipLaunchKernelGGL(colsum_finalize_multi32_kernel, dim3(blocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(colsum_finalize_multi_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return dig_check_launch();
}

extern "C" int dig_patchify_bf16(const float* img, const unsigned char* mask, void* out, int n_img, int gh, int gw, hipStream_t stream) {
  if (!img || !out || n_img <= 0) return DIG_ERR_ARG;
  if (!aligned16(img) || !aligned16(out)) return DIG_ERR_ALIGN;
  const int n_tok = n_img * gh * gw;
  hipLaunchKernelGGL(patchify_bf16_kernel, dim3((n_tok * 16 + 255) / 256), dim3(256), 0, stream, img, mask, (bf16_t*)out, n_tok, gh, gw,
                     gh * 4, gw * 4);
  return dig_check_launch();
}

extern "C" int dig_colsum_masked(const void* x, const unsigned char* mask, float* out_unmasked, float* out_masked,
                                 float* workspace /* 2 x dig_colsum_workspace_bytes(rows, C) */, int rows, int C, hipStream_t stream) {
  if (!x || !mask || !out_unmasked || !out_masked || !workspace || rows <= 0 || C <= 0 || (C & 7)) return DIG_ERR_ARG;
  if (!aligned16(x)) return DIG_ERR_ALIGN;
  const int cb = (C + 511) / 512;
  const int rpb = colsum_rows_per_block(rows, C);
  const int nb = (rows + rpb - 1) / rpb;
  float* p0 = workspace;
  float* p1 = workspace + (size_t)nb * C;
  hipLaunchKernelGGL(colsum_masked_kernel, dim3(cb, nb), dim3(256), 0, stream, (const bf16_t*)x, mask, p0, p1, rows, C, rpb);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C / 8), dim3(256), 0, stream, p0, nb, C, out_unmasked);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C / 8), dim3(256), 0, stream, p1, nb, C, out_masked);
  return dig_check_launch();
}

extern "C" int dig_dropout_apply(const void* in, void* out, long long rows, int cols, const dig_dropout_t* drop, hipStream_t stream) {
  if (!in || !out || !drop || rows <= 0 || cols <= 0 || (cols & 7) || rows * cols >= (1ll << 32)) return DIG_ERR_ARG;
  if (drop->pthr && drop->rows_per_sample <= 0) return DIG_ERR_ARG;
  if (!aligned16(in) || !aligned16(out)) return DIG_ERR_ALIGN;
  const long long n8 = rows * cols / 8;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, n8, cols, *drop);
  return dig_check_launch();
}
