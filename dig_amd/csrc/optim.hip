// Flat-arena optimizer kernels (HBM-bound, 16 B per lane per stream):
//   adamw_step: decoupled-weight-decay Adam exactly as custom_optim/_functional.py:115-140 computes it, over a
//       contiguous fp32 range of the parameter arena; a uint8 table with one entry per 256-element granule (parameters are
//       padded to 256) selects the decay / no-decay group of optim_factory.py:57-100, so one launch covers both groups; it
//       also refreshes the bf16 shadow the GEMMs read.
//       Traffic: read p,g,m,v + write p,m,v (28 B/param) + 2 B/param bf16 shadow.
//   ema_update: p_m = p_m*m + p*(1-m)  (modeling_pretrain_moco_mim_ori.py:428-442) + bf16 shadow: 12 + 2 B/param.
//   sumsq_partial / sumsq_final: deterministic two-stage sum of squares for the global gradient norm
//       (utils/utils.py:507-519); sqrt is taken on the host side of the ABI.
//   cast / scale / fill helpers.
#include "common.h"

namespace {

// One element's AdamW update (custom_optim/_functional.py:115-140), shared by the flat and the tiled loop: identical arithmetic, in this order.
__device__ __forceinline__ void adamw_elem(float& P, float G, float& M, float& V, float decay, float step_size, float beta1, float beta2, float eps,
                                           float inv_sqrt_bc2, float grad_scale) {
  const float gk = G * grad_scale;
  P *= decay;
  M = M * beta1 + gk * (1.0f - beta1);
  V = V * beta2 + gk * gk * (1.0f - beta2);
  const float denom = sqrtf(V) * inv_sqrt_bc2 + eps;
  P -= step_size * (M / denom);
}

// A 2-D weight whose updated values also leave the launch TRANSPOSED in bf16 (W^T [cols, rows]: the K-contiguous operand of the fused MLP
// backward): element offset in the arena, element offset of W^T in `tr_out`, shape (multiples of 64), index of its first 64 x 64 tile.
struct AdamTrMat { long long off, dst_off; int rows, cols, tile0, pad; };
static_assert(sizeof(AdamTrMat) == 32, "dig_adamw_step_tr's table records are 32 bytes");

// group[i >> 8] (parameters are padded to 256 elements) selects the param group of element i: 0 = decay, 1 = no_decay, 2 = a parameter that
// never receives a gradient and is left untouched, as the reference's AdamW skips `p.grad is None` (custom_optim/adamw.py:78-79);
// bit 7 = the granule belongs to a weight of the `mats` table: the flat loop leaves it to the tile workgroups (blocks >= flat_blocks).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long long n4,
                                                    const unsigned char* __restrict__ group, float lr0, float wd0, float lr1,
                                                    float wd1, float beta1, float beta2, float eps, float inv_bc1,
                                                    float inv_sqrt_bc2, float grad_scale, const float* __restrict__ finite_gate,
                                                    const float* __restrict__ dev_scalars, const AdamTrMat* __restrict__ mats, int n_mats,
                                                    bf16_t* __restrict__ tr_out, int flat_blocks) {
  if (finite_gate && !isfinite(finite_gate[0])) return;                 // non-finite gradients: the whole update is a no-op (GradScaler's inf-skip)
  if (dev_scalars) {                                                    // per-step scalars from memory: a captured graph replays with new values
    lr0 = dev_scalars[0]; wd0 = dev_scalars[1]; lr1 = dev_scalars[2]; wd1 = dev_scalars[3];
    inv_bc1 = dev_scalars[4]; inv_sqrt_bc2 = dev_scalars[5];
  }
  if (mats && (int)blockIdx.x >= flat_blocks) {
    // ---- one 64 x 64 tile of a listed weight: the same update, + the bf16 shadow, + the tile transposed through LDS into W^T
    __shared__ bf16_t tile[64][68];
    const int t = (int)blockIdx.x - flat_blocks;
    int k = 0;
    while (k + 1 < n_mats && mats[k + 1].tile0 <= t) ++k;
    const AdamTrMat mt = mats[k];
    const int tcn = mt.cols >> 6, local = t - mt.tile0;
    const int r0 = (local / tcn) << 6, c0 = (local % tcn) << 6;
    const int grp = group[mt.off >> 8] & 1;
    const float lr = grp ? lr1 : lr0, wd = grp ? wd1 : wd0;
    const float decay = 1.0f - lr * wd, step_size = lr * inv_bc1;
    const int c4 = threadIdx.x & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (threadIdx.x >> 4) + 16 * q;
      const long long i = (mt.off + (long long)(r0 + r) * mt.cols + c0 + c4 * 4) >> 2;
      float4 pp = reinterpret_cast<float4*>(p)[i];
      float4 gg = reinterpret_cast<const float4*>(g)[i];
      float4 mm = reinterpret_cast<float4*>(m)[i];
      float4 vv = reinterpret_cast<float4*>(v)[i];
      float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w}, M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) adamw_elem(P[e], G[e], M[e], V[e], decay, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
      reinterpret_cast<float4*>(p)[i] = make_float4(P[0], P[1], P[2], P[3]);
      reinterpret_cast<float4*>(m)[i] = make_float4(M[0], M[1], M[2], M[3]);
      reinterpret_cast<float4*>(v)[i] = make_float4(V[0], V[1], V[2], V[3]);
      const uint2 sh = make_uint2(pack_bf2(P[0], P[1]), pack_bf2(P[2], P[3]));
      if (shadow) reinterpret_cast<uint2*>(shadow)[i] = sh;
      *reinterpret_cast<uint2*>(&tile[r][c4 * 4]) = sh;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = (threadIdx.x >> 4) + 16 * q, r4 = (threadIdx.x & 15) * 4;
      const unsigned lo = (unsigned)tile[r4][c] | ((unsigned)tile[r4 + 1][c] << 16), hi = (unsigned)tile[r4 + 2][c] | ((unsigned)tile[r4 + 3][c] << 16);
      *reinterpret_cast<uint2*>(tr_out + mt.dst_off + (long long)(c0 + c) * mt.rows + r0 + r4) = make_uint2(lo, hi);
    }
    return;
  }
  const long long stride = (long long)(flat_blocks > 0 ? flat_blocks : (int)gridDim.x) * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int gflag = group[i >> 6];
    if (gflag & 0x80) continue;                                         // a tiled weight's granule
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float P[4] = {pp.x, pp.y, pp.z, pp.w};
    if (gflag != 2) {
      const float lr = gflag ? lr1 : lr0, wd = gflag ? wd1 : wd0;
      float4 gg = reinterpret_cast<const float4*>(g)[i];
      float4 mm = reinterpret_cast<float4*>(m)[i];
      float4 vv = reinterpret_cast<float4*>(v)[i];
      float G[4] = {gg.x, gg.y, gg.z, gg.w}, M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
      const float decay = 1.0f - lr * wd, step_size = lr * inv_bc1;
#pragma unroll
      for (int k = 0; k < 4; ++k) adamw_elem(P[k], G[k], M[k], V[k], decay, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
      reinterpret_cast<float4*>(p)[i] = make_float4(P[0], P[1], P[2], P[3]);
      reinterpret_cast<float4*>(m)[i] = make_float4(M[0], M[1], M[2], M[3]);
      reinterpret_cast<float4*>(v)[i] = make_float4(V[0], V[1], V[2], V[3]);
    }
    if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack_bf2(P[0], P[1]), pack_bf2(P[2], P[3]));
  }
}

// The same update with any number of parameter groups (layer-wise lr decay of the fine-tune recipe, optim_factory.py:33-100 /
// run_class_finetuning.py:471-520): group_idx[i >> 8] selects (lr, weight_decay) from two small device tables; index 255 marks a
// granule without a gradient, which is left untouched exactly as the reference's AdamW skips `p.grad is None`.
__global__ __launch_bounds__(256) void adamw_groups_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, bf16_t* __restrict__ shadow, long long n4,
                                                           const unsigned char* __restrict__ group_idx, const float* __restrict__ lr_tab,
                                                           const float* __restrict__ wd_tab, float beta1, float beta2, float eps, float inv_bc1,
                                                           float inv_sqrt_bc2, float grad_scale, const float* __restrict__ finite_gate) {
  if (finite_gate && !isfinite(finite_gate[0])) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int grp = group_idx[i >> 6];
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float P[4] = {pp.x, pp.y, pp.z, pp.w};
    if (grp != 255) {
      const float lr = lr_tab[grp], wd = wd_tab[grp];
      float4 gg = reinterpret_cast<const float4*>(g)[i];
      float4 mm = reinterpret_cast<float4*>(m)[i];
      float4 vv = reinterpret_cast<float4*>(v)[i];
      float G[4] = {gg.x, gg.y, gg.z, gg.w}, M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
      const float decay = 1.0f - lr * wd, step_size = lr * inv_bc1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = G[k] * grad_scale;
        P[k] *= decay;
        M[k] = M[k] * beta1 + gk * (1.0f - beta1);
        V[k] = V[k] * beta2 + gk * gk * (1.0f - beta2);
        const float denom = sqrtf(V[k]) * inv_sqrt_bc2 + eps;
        P[k] -= step_size * (M[k] / denom);
      }
      reinterpret_cast<float4*>(p)[i] = make_float4(P[0], P[1], P[2], P[3]);
      reinterpret_cast<float4*>(m)[i] = make_float4(M[0], M[1], M[2], M[3]);
      reinterpret_cast<float4*>(v)[i] = make_float4(V[0], V[1], V[2], V[3]);
    }
    if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack_bf2(P[0], P[1]), pack_bf2(P[2], P[3]));
  }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pm, const float* __restrict__ p, bf16_t* __restrict__ shadow,
                                                  long long n4, float m, float one_minus_m, const float* __restrict__ dev_m) {
  if (dev_m) { m = dev_m[0]; one_minus_m = dev_m[1]; }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(pm)[i];
    const float4 b = reinterpret_cast<const float4*>(p)[i];
    a.x = a.x * m + b.x * one_minus_m;
    a.y = a.y * m + b.y * one_minus_m;
    a.z = a.z * m + b.z * one_minus_m;
    a.w = a.w * m + b.w * one_minus_m;
    reinterpret_cast<float4*>(pm)[i] = a;
    if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
  }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n4, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    acc += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
  }
}

__global__ __launch_bounds__(256) void cast_back_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const uint2 a = reinterpret_cast<const uint2*>(x)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(bf2f((bf16_t)(a.x & 0xffff)), bf2f((bf16_t)(a.x >> 16)), bf2f((bf16_t)(a.y & 0xffff)), bf2f((bf16_t)(a.y >> 16)));
  }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n4, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(x)[i];
    a.x *= s; a.y *= s; a.z *= s; a.w *= s;
    reinterpret_cast<float4*>(x)[i] = a;
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, long long n4, float val) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(x)[i] = make_float4(val, val, val, val);
}

// x *= *scalar (device scalar; keeps the upstream autograd gradient on the device)
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, long long n, const float* __restrict__ scalar, float extra) {
  const float s = scalar[0] * extra;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= s;
}

// dst bf16 [M_pad, ld] <- src f32 [M, C] (zero padding in rows >= M and cols >= C)
__global__ __launch_bounds__(256) void pad_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int M, int C, int M_pad, int ld) {
  const long long total = (long long)M_pad * ld;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(e / ld), c = (int)(e - (long long)r * ld);
    dst[e] = (r < M && c < C) ? f2bf(src[(size_t)r * C + c]) : (bf16_t)0;
  }
}

inline int flat_grid(long long n4) { return (int)std::min<long long>(4096, (n4 + 255) / 256); }

}  // namespace

extern "C" int dig_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n,
                              const unsigned char* group_flags, float lr0, float wd0, float lr1, float wd1, float beta1,
                              float beta2, float eps, int step, float grad_scale, const float* finite_gate, hipStream_t stream) {
  if (!p || !g || !m || !v || !group_flags || n <= 0 || (n & 255) || step < 1) return DIG_ERR_ARG;
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (bf16_shadow && (((uintptr_t)bf16_shadow) & 7))) return DIG_ERR_ALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)bf16_shadow, n / 4,
                     group_flags, lr0, wd0, lr1, wd1, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, finite_gate,
                     (const float*)nullptr, (const AdamTrMat*)nullptr, 0, (bf16_t*)nullptr, 0);
  return dig_check_launch();
}

// dig_adamw_step + the transposed bf16 copies of the weights listed in `mats` (device table of n_mats 32-byte records {int64 off, int64 dst_off, int32 rows, cols,
// tile0, 0} -- see include/dig_hip.h), whose granules carry bit 7 in group_flags; n_tiles = the table's total number of 64 x 64 tiles.
extern "C" int dig_adamw_step_tr(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n,
                                 const unsigned char* group_flags, float lr0, float wd0, float lr1, float wd1, float beta1,
                                 float beta2, float eps, int step, float grad_scale, const float* finite_gate, const void* mats, int n_mats,
                                 int n_tiles, void* tr_out, hipStream_t stream) {
  if (!p || !g || !m || !v || !group_flags || n <= 0 || (n & 255) || step < 1 || !mats || n_mats < 1 || n_tiles < 1 || !tr_out) return DIG_ERR_ARG;
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (bf16_shadow && (((uintptr_t)bf16_shadow) & 7)) || (((uintptr_t)tr_out) & 7))
    return DIG_ERR_ALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int fb = flat_grid(n / 4);
  hipLaunchKernelGGL(adamw_kernel, dim3(fb + n_tiles), dim3(256), 0, stream, p, g, m, v, (bf16_t*)bf16_shadow, n / 4,
                     group_flags, lr0, wd0, lr1, wd1, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, finite_gate,
                     (const float*)nullptr, (const AdamTrMat*)mats, n_mats, (bf16_t*)tr_out, fb);
  return dig_check_launch();
}

extern "C" int dig_adamw_bias_corrections(float beta1, float beta2, int step, float* out2) {
  if (!out2 || step < 1) return DIG_ERR_ARG;
  out2[0] = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
  out2[1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  return DIG_OK;
}

extern "C" int dig_adamw_step_dev(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n,
                                  const unsigned char* group_flags, const float* scalars6, float beta1, float beta2, float eps,
                                  float grad_scale, const float* finite_gate, hipStream_t stream) {
  if (!p || !g || !m || !v || !group_flags || !scalars6 || n <= 0 || (n & 255)) return DIG_ERR_ARG;
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (bf16_shadow && (((uintptr_t)bf16_shadow) & 7))) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(adamw_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)bf16_shadow, n / 4,
                     group_flags, 0.f, 0.f, 0.f, 0.f, beta1, beta2, eps, 1.f, 1.f, grad_scale, finite_gate, scalars6, (const AdamTrMat*)nullptr, 0,
                     (bf16_t*)nullptr, 0);
  return dig_check_launch();
}

extern "C" int dig_adamw_step_groups(float* p, const float* g, float* m, float* v, void* bf16_shadow, long long n,
                                     const unsigned char* group_idx, const float* lr_tab, const float* wd_tab, float beta1, float beta2,
                                     float eps, int step, float grad_scale, const float* finite_gate, hipStream_t stream) {
  if (!p || !g || !m || !v || !group_idx || !lr_tab || !wd_tab || n <= 0 || (n & 255) || step < 1) return DIG_ERR_ARG;
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || (bf16_shadow && (((uintptr_t)bf16_shadow) & 7))) return DIG_ERR_ALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_groups_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)bf16_shadow, n / 4, group_idx,
                     lr_tab, wd_tab, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, finite_gate);
  return dig_check_launch();
}

extern "C" int dig_ema_update(float* pm, const float* p, void* bf16_shadow, long long n, float m, hipStream_t stream) {
  if (!pm || !p || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(pm) || !aligned16(p) || (bf16_shadow && (((uintptr_t)bf16_shadow) & 7))) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, pm, p, (bf16_t*)bf16_shadow, n / 4, m,
                     (float)(1.0 - (double)m), (const float*)nullptr);
  return dig_check_launch();
}

extern "C" int dig_ema_update_dev(float* pm, const float* p, void* bf16_shadow, long long n, const float* m_and_one_minus_m,
                                  hipStream_t stream) {
  if (!pm || !p || !m_and_one_minus_m || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(pm) || !aligned16(p) || (bf16_shadow && (((uintptr_t)bf16_shadow) & 7))) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, pm, p, (bf16_t*)bf16_shadow, n / 4, 0.f, 0.f,
                     m_and_one_minus_m);
  return dig_check_launch();
}

extern "C" long long dig_sumsq_workspace_bytes(long long n) { return 1024 * sizeof(float); }

extern "C" int dig_sumsq(const float* x, long long n, float* workspace, float* out, hipStream_t stream) {
  if (!x || !workspace || !out || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(x)) return DIG_ERR_ALIGN;
  const int grid = (int)std::min<long long>(1024, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, stream, x, n / 4, workspace);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, workspace, grid, out);
  return dig_check_launch();
}

extern "C" int dig_cast_f32_to_bf16(const float* x, void* y, long long n, hipStream_t stream) {
  if (!x || !y || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(x) || (((uintptr_t)y) & 7)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(cast_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, x, (bf16_t*)y, n / 4);
  return dig_check_launch();
}

extern "C" int dig_cast_bf16_to_f32(const void* x, float* y, long long n, hipStream_t stream) {
  if (!x || !y || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(y) || (((uintptr_t)x) & 7)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(cast_back_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, (const bf16_t*)x, y, n / 4);
  return dig_check_launch();
}

extern "C" int dig_scale_f32(float* x, long long n, float s, hipStream_t stream) {
  if (!x || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(x)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(scale_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, x, n / 4, s);
  return dig_check_launch();
}

extern "C" int dig_fill_f32(float* x, long long n, float value, hipStream_t stream) {
  if (!x || n <= 0 || (n & 3)) return DIG_ERR_ARG;
  if (!aligned16(x)) return DIG_ERR_ALIGN;
  hipLaunchKernelGGL(fill_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, x, n / 4, value);
  return dig_check_launch();
}

extern "C" int dig_scale_by_device_scalar(float* x, long long n, const float* scalar, float extra, hipStream_t stream) {
  if (!x || !scalar || n <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(scale_dev_kernel, dim3(flat_grid((n + 3) / 4)), dim3(256), 0, stream, x, n, scalar, extra);
  return dig_check_launch();
}

extern "C" int dig_pad_cast_rows(const float* src, void* dst, int M, int C, int M_pad, int ld, hipStream_t stream) {
  if (!src || !dst || M <= 0 || C <= 0 || M_pad < M || ld < C) return DIG_ERR_ARG;
  hipLaunchKernelGGL(pad_cast_kernel, dim3(flat_grid(((long long)M_pad * ld + 3) / 4)), dim3(256), 0, stream, src, (bf16_t*)dst, M, C, M_pad, ld);
  return dig_check_launch();
}

namespace {
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long long n, float a) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
}  // namespace

extern "C" int dig_axpy_f32(float* y, const float* x, long long n, float a, hipStream_t stream) {
  if (!y || !x || n <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(axpy_kernel, dim3(flat_grid((n + 3) / 4)), dim3(256), 0, stream, y, x, n, a);
  return dig_check_launch();
}
