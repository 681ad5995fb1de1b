// LayerNorm (row statistics) and BatchNorm1d (column statistics, cross-rank capable) for gfx950.
// Reference: nn.LayerNorm(eps=1e-6) in Block (modeling_finetune.py:134,140) and pix_decoder
// (modeling_pretrain_moco_mim_ori.py:424); nn.BatchNorm1d inside _build_mlp (:463-482) converted to
// SyncBatchNorm by run_mae_pretraining_moco.py:390.
//
// All of these are HBM-bound: bf16 in / bf16 out, fp32 statistics, one pass over the data per kernel.
//   LayerNorm: one 64-lane wave per row, D/64 contiguous elements per lane held in registers (row is read
//   once), wave-shuffle reductions, grid-stride over rows; the backward also folds the skip-path gradient
//   add and reduces dgamma/dbeta per wave -> per workgroup (LDS) -> one fp32 atomic per column per workgroup.
//   BatchNorm: statistics are column sums over rows.  A workgroup owns 128 columns x a strip of rows (each
//   lane two adjacent columns -> 256-B coalesced row segments), 4 waves on different rows, LDS combine,
//   fp32 atomics into [2,C].  The [2,C] vector is what goes through RCCL all-reduce between the
//   `stats` and `apply` kernels when world_size > 1 (SyncBN semantics: statistics over all ranks' rows).
#include "common.h"

namespace {

template <int EPL>
__device__ __forceinline__ void load_row(const bf16_t* p, float* v) {
  if constexpr (EPL == 8) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = bf2f((bf16_t)(w[k] & 0xffff)); v[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16)); }
  } else if constexpr (EPL % 2 == 0) {
    const unsigned* q = reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int k = 0; k < EPL / 2; ++k) { const unsigned w = q[k]; v[2 * k] = bf2f((bf16_t)(w & 0xffff)); v[2 * k + 1] = bf2f((bf16_t)(w >> 16)); }
  } else {
#pragma unroll
    for (int k = 0; k < EPL; ++k) v[k] = bf2f(p[k]);
  }
}
template <int EPL>
__device__ __forceinline__ void store_row(bf16_t* p, const float* v) {
  if constexpr (EPL == 8) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  } else if constexpr (EPL % 2 == 0) {
    unsigned* q = reinterpret_cast<unsigned*>(p);
#pragma unroll
    for (int k = 0; k < EPL / 2; ++k) q[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < EPL; ++k) p[k] = f2bf(v[k]);
  }
}

template <int EPL, bool GELU>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps) {
  constexpr int D = EPL * 64;
  // rows per wave per iteration: all U row loads are issued before the first reduction (a wave with one 1-KiB row in flight and two
  // dependent 6-step wave reductions behind it keeps 32 KiB per CU in flight -- half of what the HBM latency x bandwidth product asks for);
  // each row's arithmetic and summation order are those of the one-row form (bit-identical results)
  constexpr int U = GELU ? 2 : 4;
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  float g[EPL], b[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) { g[k] = gamma[lane * EPL + k]; b[k] = beta[lane * EPL + k]; }
  for (int r0 = w * U; r0 < rows; r0 += nw * U) {
    float v[U][EPL];
#pragma unroll
    for (int u = 0; u < U; ++u) load_row<EPL>(x + (size_t)min(r0 + u, rows - 1) * D + lane * EPL, v[u]);
    float mu[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) s += v[u][k];
      mu[u] = wave_sum(s) * (1.0f / D);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) { v[u][k] -= mu[u]; q += v[u][k] * v[u][k]; }
      rs[u] = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + u >= rows) break;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        v[u][k] = v[u][k] * rs[u] * g[k] + b[k];
        if (GELU) v[u][k] = gelu_f(v[u][k]);
      }
      store_row<EPL>(y + (size_t)(r0 + u) * D + lane * EPL, v[u]);
      if (lane == 0) { mean[r0 + u] = mu[u]; rstd[r0 + u] = rs[u]; }
    }
  }
}

// dx = [dres +] rstd * (dyg - mean(dyg) - xhat * mean(dyg*xhat)),  dyg = dy' * gamma, dy' = dy (* gelu'(ln) if GELU)
template <int EPL, bool GELU>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                     float* __restrict__ partial /*[grid][3][D]*/, int rows) {
  constexpr int D = EPL * 64;
  __shared__ float red[3][4][D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  float g[EPL], bt[EPL], dg[EPL], db[EPL], dc[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) { g[k] = gamma[lane * EPL + k]; bt[k] = GELU ? beta[lane * EPL + k] : 0.f; dg[k] = 0.f; db[k] = 0.f; dc[k] = 0.f; }
  // two rows per iteration: all six row loads are issued before the first reduction, so each wave keeps twice the bytes
  // in flight (the kernel is latency-bound per row: 3 loads -> 2 wave reductions -> 1 store)
  for (int r0 = w * 2; r0 < rows; r0 += nw * 2) {
    float v[2][EPL], d[2][EPL], e[2][EPL];
    float mu[2], rs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = min(r0 + u, rows - 1);
      load_row<EPL>(x + (size_t)r * D + lane * EPL, v[u]);
      load_row<EPL>(dy + (size_t)r * D + lane * EPL, d[u]);
      if (dres) load_row<EPL>(dres + (size_t)r * D + lane * EPL, e[u]);
      mu[u] = mean[r]; rs[u] = rstd[r];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (r0 + u >= rows) break;
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        v[u][k] = (v[u][k] - mu[u]) * rs[u];                     // xhat
        if (GELU) d[u][k] *= dgelu_acc_f(v[u][k] * g[k] + bt[k]);
        dg[k] += d[u][k] * v[u][k];
        db[k] += d[u][k];
        d[u][k] *= g[k];
        c1 += d[u][k];
        c2 += d[u][k] * v[u][k];
      }
      c1 = wave_sum(c1) * (1.0f / D);
      c2 = wave_sum(c2) * (1.0f / D);
      float o[EPL];
#pragma unroll
      for (int k = 0; k < EPL; ++k) o[k] = rs[u] * (d[u][k] - c1 - v[u][k] * c2);
      if (dres) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) { o[k] += e[u][k]; dc[k] += e[u][k]; }   // dc: column sums of the skip-path gradient
      }
      store_row<EPL>(dx + (size_t)(r0 + u) * D + lane * EPL, o);
    }
  }
#pragma unroll
  for (int k = 0; k < EPL; ++k) { red[0][wv][lane * EPL + k] = dg[k]; red[1][wv][lane * EPL + k] = db[k]; red[2][wv][lane * EPL + k] = dc[k]; }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 3 * D;
  for (int c = threadIdx.x; c < 3 * D; c += blockDim.x) {
    const int w = c / D, cc = c - w * D;
    out[c] = red[w][0][cc] + red[w][1][cc] + red[w][2][cc] + red[w][3][cc];
  }
}

// dgamma += sum_b partial[b][0], dbeta += sum_b partial[b][1], dcol (optional) += sum_b partial[b][2]   (deterministic).
// A block owns 16 columns: 4 float4 lanes x 64 row-groups, so the nb-long sum runs 64-way parallel with 16-B loads,
// then one LDS tree over the row-groups (this kernel sits on the backward critical path: latency matters, not bytes).
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ dcol) {
  __shared__ float4 red[64][4];
  const int cx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + cx * 4;                       // 3*D % 16 == 0 for every supported D
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = ry; b < nblocks; b += 64) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)b * 3 * D + c);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  red[ry][cx] = a;
  __syncthreads();
  for (int st = 32; st > 0; st >>= 1) {
    if (ry < st) {
      const float4 o = red[ry + st][cx];
      float4 m = red[ry][cx];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      red[ry][cx] = m;
    }
    __syncthreads();
  }
  if (ry == 0) {
    const int w = c / D, cc = c - w * D;                        // a float4 never straddles two of the three vectors (D % 4 == 0)
    float* dst = w == 0 ? dgamma : (w == 1 ? dbeta : dcol);
    if (dst) {
      const float4 m = red[0][cx];
      dst[cc] += m.x; dst[cc + 1] += m.y; dst[cc + 2] += m.z; dst[cc + 3] += m.w;
    }
  }
}

// ---------------- BatchNorm ----------------
__device__ __forceinline__ void load8f(const float* p, float* v) {      // 8 consecutive fp32 (32-byte aligned: column groups of 8)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// MODE 0: out[0][c] += sum_r x, out[1][c] += sum_r x^2
// MODE 1: g = dy * (relu ? (gamma*xhat+beta > 0) : 1); out[0][c] += sum g, out[1][c] += sum g*xhat
template <int MODE>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int relu, float* __restrict__ out, int rows, int C, int rows_per_block) {
  // A thread owns one group of 8 columns (16-byte loads; round 4: the 4-byte form ran the backward statistics of a [32 768, 512] layer at
  // 1.5 TB/s) and walks the rows of its strip; with fewer than 256 column groups several rows are in flight per workgroup (lanes
  // g = tid / c8 take rows r0 + g, r0 + g + G, ...), and their partial sums are folded through LDS in a fixed order.
  __shared__ float red[256][17];
  const int c8 = C >> 3;
  const int tpr = min(c8, 256);                                      // threads per row pass
  const int G = 256 / tpr;                                           // rows in flight
  const int tid = threadIdx.x;
  const int g = tid / tpr, ct = tid - g * tpr;
  const int cg = blockIdx.x * 256 + ct;                              // column group (grid.x = ceil(c8 / 256))
  const bool live = g < G && cg < c8;
  const int c = cg * 8;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = b[k] = 0.f;
  if (live) {
    float mu[8], rs[8], gm[8], bt[8];
    if (MODE == 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 m4 = *reinterpret_cast<const float4*>(mean + c + 4 * h), r4 = *reinterpret_cast<const float4*>(rstd + c + 4 * h);
        mu[4 * h] = m4.x; mu[4 * h + 1] = m4.y; mu[4 * h + 2] = m4.z; mu[4 * h + 3] = m4.w;
        rs[4 * h] = r4.x; rs[4 * h + 1] = r4.y; rs[4 * h + 2] = r4.z; rs[4 * h + 3] = r4.w;
        float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gamma) { g4 = *reinterpret_cast<const float4*>(gamma + c + 4 * h); b4 = *reinterpret_cast<const float4*>(beta + c + 4 * h); }
        gm[4 * h] = g4.x; gm[4 * h + 1] = g4.y; gm[4 * h + 2] = g4.z; gm[4 * h + 3] = g4.w;
        bt[4 * h] = b4.x; bt[4 * h + 1] = b4.y; bt[4 * h + 2] = b4.z; bt[4 * h + 3] = b4.w;
      }
    }
    for (int r = r0 + g; r < r1; r += G) {
      float xv[8];
      load_row<8>(x + (size_t)r * C + c, xv);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] += xv[k]; b[k] += xv[k] * xv[k]; }
      } else {
        float dv[8];
        load_row<8>(dy + (size_t)r * C + c, dv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = (xv[k] - mu[k]) * rs[k];
          float d = dv[k];
          if (relu && !(gm[k] * xh + bt[k] > 0.f)) d = 0.f;
          a[k] += d; b[k] += d * xh;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[tid][k] = a[k]; red[tid][8 + k] = b[k]; }
  __syncthreads();
  if (g == 0 && cg < c8) {                      // per-row-strip partials (no atomics: statistics are bit-reproducible)
    float* o = out + (size_t)blockIdx.y * 2 * C;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float sa = 0.f, sb = 0.f;
      for (int q = 0; q < G; ++q) { sa += red[q * tpr + ct][k]; sb += red[q * tpr + ct][8 + k]; }
      o[c + k] = sa;
      o[C + c + k] = sb;
    }
  }
}

// sums[e] = sum_b partial[b][e], e < 2*C  (fixed order)
// acc0 / acc1 (optional, [n / 2] each): += the first / second half of sums -- the BatchNorm affine gradients (dbeta, dgamma) taken from the LOCAL
// sums before any cross-rank reduction touches them (two axpy launches per layer folded into this one)
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ partial, int nb, int n, float* __restrict__ sums,
                                                                float* __restrict__ acc0, float* __restrict__ acc1) {
  // 8 columns x 32 strip groups per workgroup (round 4: 32 x 8 left a [512 strips, 1 024] problem to 32 workgroups of 64 dependent loads: 20 us)
  __shared__ float red[32][9];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cx;
  float a = 0.f;
  if (c < n)
    for (int b = ry; b < nb; b += 32) a += partial[(size_t)b * n + c];
  red[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) t += red[q][cx];
    sums[c] = t;
    if (acc0) {
      const int h = n >> 1;
      if (c < h) acc0[c] += t; else acc1[c - h] += t;
    }
  }
}

// y = [relu]( gamma * (x - mean) * rstd + beta );  mean/rstd derived from global sums (sum, sumsq) and 1/n.
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ sums,
                                                           float inv_n, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int relu, bf16_t* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           size_t total8, int C, float n_total, float momentum,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var) {
  // the launch makes (gridDim.x * 256) a multiple of C/8, so a thread keeps one 8-column group for its whole row walk and the
  // per-column scale / shift are computed once (they used to be re-derived, with a 64-bit modulo, for every element)
  const int c8 = C >> 3;
  const unsigned t0 = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(t0 % (unsigned)c8) * 8;
  float mu[8], rs[8], gm[8], bt[8];
  {
    float s1[8], s2[8];
    load8f(sums + c, s1); load8f(sums + C + c, s2);
    if (gamma) { load8f(gamma + c, gm); load8f(beta + c, bt); }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mu[k] = s1[k] * inv_n;
      const float var = fmaxf(s2[k] * inv_n - mu[k] * mu[k], 0.f);
      rs[k] = rsqrtf(var + eps);
      if (!gamma) { gm[k] = 1.f; bt[k] = 0.f; }
      if (t0 < (unsigned)c8) {
        mean_out[c + k] = mu[k]; rstd_out[c + k] = rs[k];
        if (run_mean) {                                                  // the running statistics, with bn_running_kernel's own expressions
          const float mr = s1[k] / n_total;
          const float vr = fmaxf(s2[k] / n_total - mr * mr, 0.f);
          run_mean[c + k] = run_mean[c + k] * (1.f - momentum) + mr * momentum;
          run_var[c + k] = run_var[c + k] * (1.f - momentum) + vr * (n_total / (n_total - 1.f)) * momentum;
        }
      }
    }
  }
  for (size_t i = t0; i < total8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    load_row<8>(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float o = (v[k] - mu[k]) * rs[k];                                 // same expression as the backward's xhat (ReLU mask)
      if (gamma) o = o * gm[k] + bt[k];
      if (relu) o = fmaxf(o, 0.f);
      v[k] = o;
    }
    store_row<8>(y + i * 8, v);
  }
}

// dx = gamma * rstd * (g - S1/n - xhat * S2/n), S = all-rank sums of (g, g*xhat)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int relu, const float* __restrict__ sums, float inv_n,
                                                           bf16_t* __restrict__ dx, size_t total8, int C) {
  const int c8 = C >> 3;
  const unsigned t0 = blockIdx.x * blockDim.x + threadIdx.x;          // (gridDim.x * 256) % (C/8) == 0: fixed column group
  const int c = (int)(t0 % (unsigned)c8) * 8;
  float mu[8], rs[8], gm[8], bt[8], s1[8], s2[8];
  load8f(mean + c, mu); load8f(rstd + c, rs); load8f(sums + c, s1); load8f(sums + C + c, s2);
  if (gamma) { load8f(gamma + c, gm); load8f(beta + c, bt); }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (!gamma) { gm[k] = 1.f; bt[k] = 0.f; }
    s1[k] *= inv_n; s2[k] *= inv_n;
  }
  for (size_t i = t0; i < total8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8], d[8];
    load_row<8>(x + i * 8, v);
    load_row<8>(dy + i * 8, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (v[k] - mu[k]) * rs[k];
      float g = d[k];
      if (relu && !(gm[k] * xh + bt[k] > 0.f)) g = 0.f;
      v[k] = gm[k] * rs[k] * (g - s1[k] - xh * s2[k]);
    }
    store_row<8>(dx + i * 8, v);
  }
}

// running_mean/var <- (1-mom)*old + mom*(mean, unbiased var) from the (all-rank) sums
__global__ void bn_running_kernel(const float* __restrict__ sums, float n, float momentum, float* __restrict__ rm, float* __restrict__ rv, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = sums[c] / n;
  const float var = fmaxf(sums[C + c] / n - mu * mu, 0.f);
  rm[c] = rm[c] * (1.f - momentum) + mu * momentum;
  rv[c] = rv[c] * (1.f - momentum) + var * (n / (n - 1.f)) * momentum;
}


// ---------------- BatchNorm of a FEW-ROW layer in one launch (single rank: no cross-rank reduction sits between statistics and apply) ---------
// The BN-MLP heads' 4096 / 256-wide layers see 8 B = 1 024 pooled rows: three launches per layer (column statistics in row strips, their
// fold, apply) of 10-25 us each for 16 MB of traffic.  Here a workgroup owns a slab of 32 columns for ALL rows: thread (rl, cq) = 64 row
// lanes x 4 column groups of 8 walks rows rl, rl + 64, ... twice -- sums (folded over the row lanes through LDS in lane order: fixed order,
// bit-reproducible), then the apply on the second, L2-hot read of its 64 KiB slab.  Same expressions as the three-launch path; the
// summation ORDER over rows differs (row lanes instead of strips), so the two paths agree to fp32 round-off, not bit for bit.
constexpr int BNF_COLS = 32, BNF_LANES = 64;

// MODE 0 forward: y = [relu](gamma xhat + beta), mean / rstd out, running statistics;  MODE 1 backward: dx, += dbeta / dgamma
template <int MODE>
__global__ __launch_bounds__(256) void bn_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ mean_in,
                                                       const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu, float eps, bf16_t* __restrict__ out,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out, float momentum,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ dbeta_acc,
                                                       float* __restrict__ dgamma_acc, int rows, int C) {
  __shared__ float red[BNF_LANES][4][17];
  __shared__ float tot[4][16];
  const int tid = threadIdx.x, cq = tid & 3, rl = tid >> 2;
  const int c = blockIdx.x * BNF_COLS + cq * 8;
  float mu[8], rs[8], gm[8], bt[8], a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = b[k] = 0.f; gm[k] = 1.f; bt[k] = 0.f; mu[k] = 0.f; rs[k] = 1.f; }
  if (gamma) { load8f(gamma + c, gm); load8f(beta + c, bt); }
  if (MODE == 1) { load8f(mean_in + c, mu); load8f(rstd_in + c, rs); }
  for (int r = rl; r < rows; r += BNF_LANES) {
    float xv[8];
    load_row<8>(x + (size_t)r * C + c, xv);
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] += xv[k]; b[k] += xv[k] * xv[k]; }
    } else {
      float dv[8];
      load_row<8>(dy + (size_t)r * C + c, dv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (xv[k] - mu[k]) * rs[k];
        float d = dv[k];
        if (relu && !(gm[k] * xh + bt[k] > 0.f)) d = 0.f;
        a[k] += d; b[k] += d * xh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[rl][cq][k] = a[k]; red[rl][cq][8 + k] = b[k]; }
  __syncthreads();
  if (tid < 64) {                                                     // 4 column groups x 16 sums: one thread each, row lanes in order
    const int q = tid >> 4, e = tid & 15;
    float t = 0.f;
    for (int l = 0; l < BNF_LANES; ++l) t += red[l][q][e];
    tot[q][e] = t;
  }
  __syncthreads();
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = tot[cq][k]; s2[k] = tot[cq][8 + k]; }
  if (MODE == 0) {
    const float n_total = (float)rows, inv_n = 1.0f / n_total;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mu[k] = s1[k] * inv_n;
      const float var = fmaxf(s2[k] * inv_n - mu[k] * mu[k], 0.f);
      rs[k] = rsqrtf(var + eps);
      if (rl == 0) {
        mean_out[c + k] = mu[k]; rstd_out[c + k] = rs[k];
        if (run_mean) {                                                // bn_running_kernel's own expressions
          const float mr = s1[k] / n_total;
          const float vr = fmaxf(s2[k] / n_total - mr * mr, 0.f);
          run_mean[c + k] = run_mean[c + k] * (1.f - momentum) + mr * momentum;
          run_var[c + k] = run_var[c + k] * (1.f - momentum) + vr * (n_total / (n_total - 1.f)) * momentum;
        }
      }
    }
    for (int r = rl; r < rows; r += BNF_LANES) {
      float v[8];
      load_row<8>(x + (size_t)r * C + c, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float o = (v[k] - mu[k]) * rs[k];
        if (gamma) o = o * gm[k] + bt[k];
        if (relu) o = fmaxf(o, 0.f);
        v[k] = o;
      }
      store_row<8>(out + (size_t)r * C + c, v);
    }
  } else {
    if (rl == 0 && dbeta_acc) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { dbeta_acc[c + k] += s1[k]; dgamma_acc[c + k] += s2[k]; }
    }
    const float inv_n = 1.0f / (float)rows;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] *= inv_n; s2[k] *= inv_n; }
    for (int r = rl; r < rows; r += BNF_LANES) {
      float v[8], d[8];
      load_row<8>(x + (size_t)r * C + c, v);
      load_row<8>(dy + (size_t)r * C + c, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (v[k] - mu[k]) * rs[k];
        float g = d[k];
        if (relu && !(gm[k] * xh + bt[k] > 0.f)) g = 0.f;
        v[k] = gm[k] * rs[k] * (g - s1[k] - xh * s2[k]);
      }
      store_row<8>(out + (size_t)r * C + c, v);
    }
  }
}

inline int ln_grid(int rows) { return std::max(1, std::min(2048, (rows + 3) / 4)); }
inline int ln_fwd_grid(int rows, bool gelu) { const int per = 4 * (gelu ? 2 : 4); return std::max(1, std::min(2048, (rows + per - 1) / per)); }

}  // namespace

#define LN_DISPATCH(D, GELU, KERNEL, ...)                                                          \
  switch (D) {                                                                                     \
    case 64: hipLaunchKernelGGL((KERNEL<1, GELU>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;  \
    case 128: hipLaunchKernelGGL((KERNEL<2, GELU>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break; \
    case 192: hipLaunchKernelGGL((KERNEL<3, GELU>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break; \
    case 256: hipLaunchKernelGGL((KERNEL<4, GELU>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break; \
    case 384: hipLaunchKernelGGL((KERNEL<6, GELU>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break; \
    case 512: hipLaunchKernelGGL((KERNEL<8, GELU>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break; \
    default: return DIG_ERR_UNSUPPORTED;                                                           \
  }

extern "C" int dig_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int rows, int D, float eps, int fuse_gelu, hipStream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(y)) return DIG_ERR_ALIGN;
  const int grid = ln_fwd_grid(rows, fuse_gelu != 0);
  if (fuse_gelu) { LN_DISPATCH(D, true, ln_fwd_kernel, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, eps) }
  else { LN_DISPATCH(D, false, ln_fwd_kernel, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, eps) }
  return dig_check_launch();
}

extern "C" long long dig_layernorm_bwd_workspace_bytes(int rows, int D) {
  return (long long)std::max(1, std::min(1024, (rows + 15) / 16)) * 3 * D * sizeof(float);
}

// dcolsum (nullable): receives += column sums of dres (the bias gradient of the layer that produced the skip-path sum)
// number of partial rows ([parts][3][D] fp32) the LayerNorm backward leaves in its workspace
extern "C" int dig_layernorm_bwd_parts(int rows) { return std::max(1, std::min(1024, (rows + 15) / 16)); }

// dx only; dgamma / dbeta / dres column-sum partials stay in `workspace` for dig_layernorm_bwd_finalize (which a caller may
// run on another stream: nothing on the data-gradient chain depends on it)
extern "C" int dig_layernorm_bwd_partials(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                          const float* rstd, const void* dres, void* dx, float* workspace, int rows, int D,
                                          int fuse_gelu, hipStream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !workspace || rows <= 0) return DIG_ERR_ARG;
  if (fuse_gelu && !beta) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || (dres && !aligned16(dres))) return DIG_ERR_ALIGN;
  const int grid = dig_layernorm_bwd_parts(rows);
  if (fuse_gelu) { LN_DISPATCH(D, true, ln_bwd_kernel, (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, workspace, rows) }
  else { LN_DISPATCH(D, false, ln_bwd_kernel, (const bf16_t*)dy, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, workspace, rows) }
  return dig_check_launch();
}

extern "C" int dig_layernorm_bwd_finalize(const float* workspace, int rows, int D, float* dgamma, float* dbeta, float* dcolsum,
                                          hipStream_t stream) {
  if (!workspace || !dgamma || !dbeta || rows <= 0 || D <= 0 || (D & 15)) return DIG_ERR_ARG;
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * D) / 16), dim3(256), 0, stream, workspace, dig_layernorm_bwd_parts(rows), D, dgamma,
                     dbeta, dcolsum);
  return dig_check_launch();
}

// the same over a workspace of `parts` partial rows written by another producer (dig_mlp_chain_bwd_ln: one per 128 rows = one per workgroup, dig_mlp_chain_ln_parts)
extern "C" int dig_layernorm_bwd_finalize_parts(const float* workspace, int parts, int D, float* dgamma, float* dbeta, float* dcolsum,
                                                hipStream_t stream) {
  if (!workspace || !dgamma || !dbeta || parts <= 0 || D <= 0 || (D & 15)) return DIG_ERR_ARG;
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * D) / 16), dim3(256), 0, stream, workspace, parts, D, dgamma, dbeta, dcolsum);
  return dig_check_launch();
}

extern "C" int dig_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                 const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* dcolsum,
                                 float* workspace, int rows, int D, int fuse_gelu, hipStream_t stream) {
  if (!dgamma || !dbeta || (dcolsum && !dres)) return DIG_ERR_ARG;
  const int rc = dig_layernorm_bwd_partials(dy, x, gamma, beta, mean, rstd, dres, dx, workspace, rows, D, fuse_gelu, stream);
  if (rc != DIG_OK) return rc;
  return dig_layernorm_bwd_finalize(workspace, rows, D, dgamma, dbeta, dcolsum, stream);
}

static inline int bn_col_blocks(int C) { return ((C >> 3) + 255) / 256; }
static inline int bn_rows_per_block(int rows, int C) {
  const int cb = bn_col_blocks(C);
  int rpb = 32;
  while ((long)cb * ((rows + rpb - 1) / rpb) > 1024) rpb *= 2;
  return rpb;
}

extern "C" long long dig_bn_stats_workspace_bytes(int rows, int C) {
  const int rpb = bn_rows_per_block(rows, C);
  return (long long)((rows + rpb - 1) / rpb) * 2 * C * sizeof(float);
}

// sums[2,C] = (sum_r x, sum_r x^2), overwritten; two-stage and deterministic
extern "C" int dig_bn_stats(const void* x, float* sums, float* workspace, int rows, int C, hipStream_t stream) {
  if (!x || !sums || !workspace || rows <= 0 || C <= 0 || (C & 7)) return DIG_ERR_ARG;
  if (!aligned16(x)) return DIG_ERR_ALIGN;
  const int cb = bn_col_blocks(C);
  const int rpb = bn_rows_per_block(rows, C);
  const int nb = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(bn_colstats_kernel<0>, dim3(cb, nb), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0, workspace, rows, C, rpb);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, stream, workspace, nb, 2 * C, sums, (float*)nullptr, (float*)nullptr);
  return dig_check_launch();
}

// grid for the BN apply kernels: enough blocks to fill the chip, with (grid * 256) a multiple of C/8 so that every thread
// keeps one 8-column group (see bn_fwd_apply_kernel)
static inline int bn_apply_grid(size_t total8, int C) {
  const int c8 = C >> 3;
  int a = c8, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }                     // a = gcd(c8, 256)
  const int unit = c8 / a;                                             // blocks per whole column period
  const size_t want = std::max<size_t>(std::min<size_t>(256, (total8 + 255) / 256), std::min<size_t>(2048, (total8 + 2047) / 2048));   // ~8 groups per thread
  return (int)std::max<size_t>(1, (want + unit - 1) / unit) * unit;
}

extern "C" int dig_bn_fwd_apply_running(const void* x, const float* sums, float n_total, float eps, const float* gamma,
                                        const float* beta, int relu, void* y, float* mean_out, float* rstd_out, float momentum,
                                        float* running_mean, float* running_var, int rows, int C, hipStream_t stream) {
  if (!x || !sums || !y || !mean_out || !rstd_out || rows <= 0 || (C & 7) || n_total <= 0.f) return DIG_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr) || (running_mean && n_total <= 1.f)) return DIG_ERR_ARG;
  if ((gamma == nullptr) != (beta == nullptr)) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(y) || !aligned16(sums) || (gamma && (!aligned16(gamma) || !aligned16(beta)))) return DIG_ERR_ALIGN;
  const size_t total8 = (size_t)rows * C / 8;
  const int grid = bn_apply_grid(total8, C);
  hipLaunchKernelGGL(bn_fwd_apply_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, sums, 1.0f / n_total, eps, gamma,
                     beta, relu, (bf16_t*)y, mean_out, rstd_out, total8, C, n_total, momentum, running_mean, running_var);
  return dig_check_launch();
}

extern "C" int dig_bn_fwd_apply(const void* x, const float* sums, float n_total, float eps, const float* gamma,
                                const float* beta, int relu, void* y, float* mean_out, float* rstd_out, int rows, int C,
                                hipStream_t stream) {
  return dig_bn_fwd_apply_running(x, sums, n_total, eps, gamma, beta, relu, y, mean_out, rstd_out, 0.f, nullptr, nullptr, rows, C, stream);
}

extern "C" int dig_bn_bwd_stats_acc(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, int relu, float* sums, float* dbeta_acc, float* dgamma_acc, float* workspace, int rows, int C,
                                    hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !sums || !workspace || rows <= 0 || (C & 7)) return DIG_ERR_ARG;
  if ((dbeta_acc == nullptr) != (dgamma_acc == nullptr)) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(mean) || !aligned16(rstd) || (gamma && (!aligned16(gamma) || !aligned16(beta)))) return DIG_ERR_ALIGN;
  const int cb = bn_col_blocks(C);
  const int rpb = bn_rows_per_block(rows, C);
  const int nb = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(bn_colstats_kernel<1>, dim3(cb, nb), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, gamma,
                     beta, relu, workspace, rows, C, rpb);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, stream, workspace, nb, 2 * C, sums, dbeta_acc, dgamma_acc);
  return dig_check_launch();
}

extern "C" int dig_bn_bwd_stats(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, int relu, float* sums, float* workspace, int rows, int C, hipStream_t stream) {
  return dig_bn_bwd_stats_acc(dy, x, mean, rstd, gamma, beta, relu, sums, nullptr, nullptr, workspace, rows, C, stream);
}

extern "C" int dig_bn_bwd_apply(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, int relu, const float* sums, float n_total, void* dx, int rows, int C,
                                hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !sums || !dx || rows <= 0 || (C & 7) || n_total <= 0.f) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(mean) || !aligned16(rstd) || !aligned16(sums) ||
      (gamma && (!aligned16(gamma) || !aligned16(beta)))) return DIG_ERR_ALIGN;
  const size_t total8 = (size_t)rows * C / 8;
  const int grid = bn_apply_grid(total8, C);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd,
                     gamma, beta, relu, sums, 1.0f / n_total, (bf16_t*)dx, total8, C);
  return dig_check_launch();
}


// Few-row BatchNorm layers in ONE launch each (single rank only: with a process group the cross-rank reduction of the statistics sits between
// dig_bn_stats and dig_bn_fwd_apply).  rows <= 4096, C a multiple of 32, at least 32 column slabs.
extern "C" int dig_bn_fused_supported(int rows, int C) { return rows >= 2 && rows <= 4096 && C >= 256 && (C % BNF_COLS) == 0; }

extern "C" int dig_bn_fwd_fused(const void* x, float eps, const float* gamma, const float* beta, int relu, void* y, float* mean_out,
                                float* rstd_out, float momentum, float* running_mean, float* running_var, int rows, int C, hipStream_t stream) {
  if (!x || !y || !mean_out || !rstd_out || (gamma == nullptr) != (beta == nullptr) || (running_mean == nullptr) != (running_var == nullptr))
    return DIG_ERR_ARG;
  if (!dig_bn_fused_supported(rows, C)) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(y)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(bn_fused_kernel<0>, dim3(C / BNF_COLS), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, gamma, beta, relu, eps, (bf16_t*)y, mean_out, rstd_out, momentum, running_mean, running_var,
                     (float*)nullptr, (float*)nullptr, rows, C);
  return dig_check_launch();
}

extern "C" int dig_bn_bwd_fused(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                int relu, float* dbeta_acc, float* dgamma_acc, void* dx, int rows, int C, hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !dx || (gamma == nullptr) != (beta == nullptr) || (dbeta_acc == nullptr) != (dgamma_acc == nullptr))
    return DIG_ERR_ARG;
  if (!dig_bn_fused_supported(rows, C)) return DIG_ERR_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(bn_fused_kernel<1>, dim3(C / BNF_COLS), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, mean, rstd, gamma, beta,
                     relu, 0.f, (bf16_t*)dx, (float*)nullptr, (float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr, dbeta_acc, dgamma_acc, rows, C);
  return dig_check_launch();
}

extern "C" int dig_bn_update_running(const float* sums, float n_total, float momentum, float* running_mean, float* running_var,
                                     int C, hipStream_t stream) {
  if (!sums || !running_mean || !running_var || C <= 0 || n_total <= 1.f) return DIG_ERR_ARG;
  hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, n_total, momentum, running_mean, running_var, C);
  return dig_check_launch();
}
