// Input transform on the device: the step either side of the hot path.
//
//   dig_resize_bicubic_normalize_u8 : uint8 RGB crops of any size -> fp32 [n, 3, out_h, out_w], bit-exact with
//       transforms.Resize((h, w), interpolation=3) + ToTensor + Normalize(mean, std) of the reference
//       (dataset/datasets.py:27-42 on the PIL crops of dataset/dataset_image.py:128-160).  Resize on a PIL image is
//       Pillow's ImagingResample: separable horizontal-then-vertical passes on 8-bit pixels, double-precision bicubic
//       (a = -0.5) coefficients over a support of 2*max(scale,1), normalised, rounded to 22-bit fixed point, int32
//       accumulation from 1<<21, clip to [0,255] after >>22 (restated in oracle/input_oracle.py, pinned against Pillow).
//   dig_random_masks : RandomMaskingGenerator (masking_generator.py:12-49) as a device generator: each row is a uniformly
//       random subset of exactly num_mask patches, chosen as the num_mask smallest Philox4x32-10 keys.
//
// HBM-bound integer work: one workgroup per crop; the fixed-point coefficient tables live in LDS; each thread produces
// output pixels (all three channels) by evaluating the horizontal pass on the fly for the rows its vertical window needs
// (an 8-bit intermediate exactly as Pillow's), so no intermediate image goes to memory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double bicubic_filter(double x) {
#pragma clang fp contract(off)
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output index xx (same operation order, no FMA contraction)
__device__ void coeffs_for(int xx, int in_size, int out_size, int ksize, int* __restrict__ kk, int* __restrict__ bounds) {
#pragma clang fp contract(off)
  double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const double center = ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic_filter(((double)(x + xmin) - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double w = bicubic_filter(((double)(x + xmin) - center + 0.5) * ss);
    if (ww != 0.0) w /= ww;
    kk[xx * ksize + x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resize_normalize_kernel(const unsigned char* __restrict__ packed,
                                                               const long long* __restrict__ offsets,
                                                               const int* __restrict__ heights, const int* __restrict__ widths,
                                                               float* __restrict__ out, int out_h, int out_w, int ksh, int ksv,
                                                               float mean, float stdv) {
  extern __shared__ int lds[];
  int* kh = lds;                               // [out_w][ksh]
  int* bh = kh + out_w * ksh;                  // [out_w][2]
  int* kv = bh + 2 * out_w;                    // [out_h][ksv]
  int* bv = kv + out_h * ksv;                  // [out_h][2]
  const int img = blockIdx.x;
  const int h = heights[img], w = widths[img];
  const unsigned char* src = packed + offsets[img];
  for (int t = threadIdx.x; t < out_w + out_h; t += blockDim.x) {
    if (t < out_w) coeffs_for(t, w, out_w, ksh, kh, bh);
    else coeffs_for(t - out_w, h, out_h, ksv, kv, bv);
  }
  __syncthreads();
  const bool pass_h = (w != out_w), pass_v = (h != out_h);        // ImagingResample skips a pass that keeps the size
  const size_t plane = (size_t)out_h * out_w;
  float* o = out + (size_t)img * 3 * plane;
  for (int p = threadIdx.x; p < out_h * out_w; p += blockDim.x) {
    const int yy = p / out_w, xx = p - yy * out_w;
    const int x0 = pass_h ? bh[2 * xx] : xx, nx = pass_h ? bh[2 * xx + 1] : 1;
    const int y0 = pass_v ? bv[2 * yy] : yy, ny = pass_v ? bv[2 * yy + 1] : 1;
    const int* kx = kh + xx * ksh;
    const int* ky = kv + yy * ksv;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    int r0 = 0, r1 = 0, r2 = 0;
    for (int y = 0; y < ny; ++y) {
      const unsigned char* row = src + ((size_t)(y0 + y) * w + x0) * 3;
      int h0, h1, h2;
      if (pass_h) {
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < nx; ++x) {
          const int k = kx[x];
          s0 += (int)row[3 * x] * k;
          s1 += (int)row[3 * x + 1] * k;
          s2 += (int)row[3 * x + 2] * k;
        }
        h0 = clip8(s0); h1 = clip8(s1); h2 = clip8(s2);
      } else {
        h0 = row[0]; h1 = row[1]; h2 = row[2];
      }
      if (pass_v) {
        const int k = ky[y];
        a0 += h0 * k; a1 += h1 * k; a2 += h2 * k;
      } else {
        r0 = h0; r1 = h1; r2 = h2;
      }
    }
    if (pass_v) { r0 = clip8(a0); r1 = clip8(a1); r2 = clip8(a2); }
    // ToTensor: uint8 / 255 in fp32; Normalize: (x - mean) / std  (IEEE division: this file is built without fast-math)
    o[p] = ((float)r0 / 255.0f - mean) / stdv;
    o[plane + p] = ((float)r1 / 255.0f - mean) / stdv;
    o[2 * plane + p] = ((float)r2 / 255.0f - mean) / stdv;
  }
}

// ---- Philox4x32-10
__device__ __forceinline__ unsigned philox_first(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

// one wave per mask row: the num_mask patches with the smallest (key, index) are set
__global__ __launch_bounds__(64) void random_masks_kernel(unsigned char* __restrict__ mask, int n_patches, int num_mask, unsigned k0,
                                                          unsigned k1, unsigned step) {
  extern __shared__ unsigned keys[];
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int p = lane; p < n_patches; p += 64) keys[p] = philox_first((unsigned)r, (unsigned)p, step, 0u, k0, k1);
  __syncthreads();
  for (int p = lane; p < n_patches; p += 64) {
    const unsigned mine = keys[p];
    int rank = 0;
    for (int q = 0; q < n_patches; ++q) {
      const unsigned kq = keys[q];
      rank += (kq < mine) || (kq == mine && q < p);
    }
    mask[(size_t)r * n_patches + p] = rank < num_mask ? 1 : 0;
  }
}

int ksize_for(int in_size, int out_size) {
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)std::ceil(2.0 * fs) * 2 + 1;
}

}  // namespace

// C-ABI: see include/dig_hip.h
extern "C" int dig_resize_bicubic_normalize_u8(const unsigned char* packed, const long long* offsets, const int* heights,
                                               const int* widths, int n_img, float* out, int out_h, int out_w, float mean,
                                               float std_, int max_h, int max_w, hipStream_t stream) {
  if (!packed || !offsets || !heights || !widths || !out || n_img <= 0 || out_h <= 0 || out_w <= 0 || max_h <= 0 || max_w <= 0 || std_ == 0.f)
    return DIG_ERR_ARG;
  const int ksh = ksize_for(max_w, out_w), ksv = ksize_for(max_h, out_h);
  const size_t lds = ((size_t)out_w * (ksh + 2) + (size_t)out_h * (ksv + 2)) * sizeof(int);
  if (lds > 160 * 1024) return DIG_ERR_UNSUPPORTED;                    // crops beyond ~ (64 x out) per axis: shrink on the host first
  static size_t attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resize_normalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  hipLaunchKernelGGL(resize_normalize_kernel, dim3(n_img), dim3(256), lds, stream, packed, offsets, heights, widths, out, out_h, out_w,
                     ksh, ksv, mean, std_);
  return dig_check_launch();
}

extern "C" int dig_random_masks(unsigned char* mask, int n_rows, int n_patches, int num_mask, unsigned long long seed,
                                unsigned step, hipStream_t stream) {
  if (!mask || n_rows <= 0 || n_patches <= 0 || n_patches > 16384 || num_mask < 0 || num_mask > n_patches) return DIG_ERR_ARG;
  hipLaunchKernelGGL(random_masks_kernel, dim3(n_rows), dim3(64), n_patches * sizeof(unsigned), stream, mask, n_patches, num_mask,
                     (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), step);
  return dig_check_launch();
}
