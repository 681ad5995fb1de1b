// MoCo-v3 InfoNCE head in fp32 (reference: contrastive_loss / accuracy / label_smooth_loss,
// modeling_pretrain_moco_mim_ori.py:444-461, 593-625).  The logits are tiny ([4B, 4B*W] x 256) but feed a
// temperature-0.2 softmax, so everything here stays fp32: L2 normalise, a plain LDS-tiled fp32 GEMM for
// q k^T / T and for dq = dlogits k, and a fused row kernel for log-softmax / CE / top-1 / top-5 / dlogits.
#include "common.h"

namespace {

// y = x / max(||x||, eps) row-wise; one wave per row
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm, int n, int C,
                                  float eps) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[(size_t)r * C + c]; s += v * v; }
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), eps);
  for (int c = lane; c < C; c += 64) y[(size_t)r * C + c] = x[(size_t)r * C + c] * inv;
  if (lane == 0) inv_norm[r] = inv;
}

// dx = (dy - y * <y, dy>) * inv_norm
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv_norm,
                                  float* __restrict__ dx, int n, int C) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += y[(size_t)r * C + c] * dy[(size_t)r * C + c];
  s = wave_sum(s);
  const float inv = inv_norm[r];
  for (int c = lane; c < C; c += 64) dx[(size_t)r * C + c] = (dy[(size_t)r * C + c] - y[(size_t)r * C + c] * s) * inv;
}

// fp32 GEMM for the InfoNCE logits and their gradient (a few hundred rows: launch- and latency-bound, so small 32x32 tiles
// for parallelism and a 64-deep K step for few barrier round trips).  256 threads, 2x2 outputs per thread.
// TB=false: C = alpha * A[I,R] * B[J,R]^T ; TB=true: C = alpha * A[I,R] * B[R,J]
template <bool TB>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    int I, int J, int R, int lda, int ldb, int ldc, float alpha, int r_per_split) {
  constexpr int TS = 32, KS = 64;
  __shared__ float As[KS][TS + 1];
  __shared__ float Bs[KS][TS + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  float acc[2][2] = {};
  const int rbeg = blockIdx.z * r_per_split;                           // R-split z writes slab z of C ([splits][I][ldc])
  R = min(R, rbeg + r_per_split);
  C += (size_t)blockIdx.z * I * ldc;
  for (int r0 = rbeg; r0 < R; r0 += KS) {
    for (int e = threadIdx.x; e < TS * KS; e += 256) {
      const int rr = e & (KS - 1), ii = e >> 6;                       // consecutive threads walk K (contiguous in A, and in B^T)
      As[rr][ii] = (i0 + ii < I && r0 + rr < R) ? A[(size_t)(i0 + ii) * lda + r0 + rr] : 0.f;
      if (!TB) Bs[rr][ii] = (j0 + ii < J && r0 + rr < R) ? B[(size_t)(j0 + ii) * ldb + r0 + rr] : 0.f;
    }
    if (TB) {
      for (int e = threadIdx.x; e < TS * KS; e += 256) {
        const int jj = e & (TS - 1), rr = e >> 5;                     // consecutive threads walk J (contiguous in B)
        Bs[rr][jj] = (j0 + jj < J && r0 + rr < R) ? B[(size_t)(r0 + rr) * ldb + j0 + jj] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 16
    for (int rr = 0; rr < KS; ++rr) {
      const float a0 = As[rr][ty * 2], a1 = As[rr][ty * 2 + 1], b0 = Bs[rr][tx * 2], b1 = Bs[rr][tx * 2 + 1];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = i0 + ty * 2 + u, j = j0 + tx * 2 + v;
      if (i < I && j < J) C[(size_t)i * ldc + j] = acc[u][v] * alpha;
    }
}

// Per row i of logits [n, m]: lse, loss_i = lse - logit[label], rank of the label among the row, and (in place)
// dlogits = gscale * (softmax - onehot).  out[0] += sum_i loss_i, out[1] += #top1, out[2] += #top5.
__global__ __launch_bounds__(256) void ce_rows_kernel(float* __restrict__ logits, int n, int m, int label_offset, float gscale,
                                                      float* __restrict__ out, float* __restrict__ ws) {
  __shared__ float red[8];
  const int i = blockIdx.x;
  float* row = logits + (size_t)i * m;
  const int label = i + label_offset;
  float mx = -3.0e38f;
  for (int j = threadIdx.x; j < m; j += 256) mx = fmaxf(mx, row[j]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float zl = row[label];
  float s = 0.f, gt = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
    const float z = row[j];
    s += __expf(z - mx);
    gt += (z > zl) ? 1.f : 0.f;
  }
  s = wave_sum(s);
  gt = wave_sum(gt);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = gt; }
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  gt = red[4] + red[5] + red[6] + red[7];
  const float lse = mx + __logf(s);
  const float inv = 1.0f / s;
  for (int j = threadIdx.x; j < m; j += 256) {
    const float p = __expf(row[j] - mx) * inv;
    row[j] = gscale * (p - (j == label ? 1.f : 0.f));
  }
  if (!ws) {
    if (threadIdx.x == 0) {
      atomicAdd(out, lse - zl);
      if (gt < 0.5f) atomicAdd(out + 1, 1.f);
      if (gt < 4.5f) atomicAdd(out + 2, 1.f);
    }
    return;
  }
  // with a workspace: the rows' values are added in row order by the last workgroup to finish (bit-reproducible loss value)
  __shared__ float red2[256];
  const float mine[3] = {lse - zl, gt < 0.5f ? 1.f : 0.f, gt < 4.5f ? 1.f : 0.f};
  float tot[3];
  if (dig_grid_sum_last<3>(ws, mine, tot, red2)) { out[0] += tot[0]; out[1] += tot[1]; out[2] += tot[2]; }
}

// The scalar tail of the InfoNCE pair (modeling_pretrain_moco_mim_ori.py:444-461, 2 T * mean CE of both directions) and of a step's log line
// (engine_for_pretraining_moco.py:146-183): a handful of fp32 scalars combined in ONE launch each instead of five to seven framework ones.
__global__ void infonce_finish_kernel(const float* __restrict__ stats6, float loss_scale, float acc_scale, float* __restrict__ contra,
                                      float* __restrict__ accs4) {
  if (threadIdx.x == 0) contra[0] = (stats6[0] + stats6[3]) * loss_scale;
  if (threadIdx.x < 4) accs4[threadIdx.x] = stats6[(threadIdx.x >> 1) * 3 + 1 + (threadIdx.x & 1)] * acc_scale;
}

__global__ __launch_bounds__(64) void step_meters_kernel(const float* __restrict__ loss, const float* __restrict__ contra, const float* __restrict__ pixel,
                                                        const float* __restrict__ accs4, const int* __restrict__ counts, int n_counts,
                                                        const float* __restrict__ grad_norm, float* __restrict__ out10) {
  int lo = 0x7fffffff, hi = -0x7fffffff - 1;
  for (int i = threadIdx.x; i < n_counts; i += 64) { const int c = counts[i]; lo = min(lo, c); hi = max(hi, c); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
  if (threadIdx.x == 0) {
    out10[0] = loss[0]; out10[1] = contra[0]; out10[2] = pixel[0];
    out10[7] = (float)lo; out10[8] = (float)hi;
    out10[9] = grad_norm ? grad_norm[0] : __builtin_nanf("");
  }
  if (threadIdx.x < 4) out10[3 + threadIdx.x] = accs4[threadIdx.x];
}

}  // namespace

extern "C" int dig_infonce_finish(const float* stats6, float loss_scale, float acc_scale, float* contra, float* accs4, hipStream_t stream) {
  if (!stats6 || !contra || !accs4) return DIG_ERR_ARG;
  hipLaunchKernelGGL(infonce_finish_kernel, dim3(1), dim3(64), 0, stream, stats6, loss_scale, acc_scale, contra, accs4);
  return dig_check_launch();
}

extern "C" int dig_step_meters(const float* loss, const float* contra, const float* pixel, const float* accs4, const int* counts, int n_counts,
                               const float* grad_norm, float* out10, hipStream_t stream) {
  if (!loss || !contra || !pixel || !accs4 || !counts || n_counts <= 0 || !out10) return DIG_ERR_ARG;
  hipLaunchKernelGGL(step_meters_kernel, dim3(1), dim3(64), 0, stream, loss, contra, pixel, accs4, counts, n_counts, grad_norm, out10);
  return dig_check_launch();
}

extern "C" int dig_l2norm_fwd(const float* x, float* y, float* inv_norm, int n, int C, float eps, hipStream_t stream) {
  if (!x || !y || !inv_norm || n <= 0 || C <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, x, y, inv_norm, n, C, eps);
  return dig_check_launch();
}

extern "C" int dig_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int n, int C, hipStream_t stream) {
  if (!dy || !y || !inv_norm || !dx || n <= 0 || C <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, dy, y, inv_norm, dx, n, C);
  return dig_check_launch();
}

extern "C" int dig_sgemm(const float* A, const float* B, float* C, int I, int J, int R, int lda, int ldb, int ldc, int trans_b,
                         float alpha, int r_splits, hipStream_t stream) {
  if (!A || !B || !C || I <= 0 || J <= 0 || R <= 0 || r_splits < 1) return DIG_ERR_ARG;
  const int per = ((R + r_splits - 1) / r_splits + 63) / 64 * 64;      // whole 64-deep K steps per split
  if ((R + per - 1) / per != r_splits) return DIG_ERR_ARG;             // (callers pick r_splits with R % (64 * r_splits) == 0)
  dim3 grid((J + 31) / 32, (I + 31) / 32, r_splits);
  if (trans_b)
    hipLaunchKernelGGL(sgemm_kernel<true>, grid, dim3(256), 0, stream, A, B, C, I, J, R, lda, ldb, ldc, alpha, per);
  else
    hipLaunchKernelGGL(sgemm_kernel<false>, grid, dim3(256), 0, stream, A, B, C, I, J, R, lda, ldb, ldc, alpha, per);
  return dig_check_launch();
}

extern "C" int dig_ce_rows_ws(float* logits, int n, int m, int label_offset, float gscale, float* out3, float* workspace, hipStream_t stream) {
  if (!logits || !out3 || n <= 0 || m <= 0 || label_offset < 0 || label_offset + n > m) return DIG_ERR_ARG;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(n), dim3(256), 0, stream, logits, n, m, label_offset, gscale, out3, workspace);
  return dig_check_launch();
}
extern "C" int dig_ce_rows(float* logits, int n, int m, int label_offset, float gscale, float* out3, hipStream_t stream) {
  return dig_ce_rows_ws(logits, n, m, label_offset, gscale, out3, nullptr, stream);
}
