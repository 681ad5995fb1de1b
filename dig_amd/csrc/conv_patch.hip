// Data movement of ConvPatchNet (`--patchnet_name conv`, modeling_pretrain_moco_mim_ori.py:207-260): its 3x3 convolutions run as GEMMs of this
// library (csrc/gemm.hip) over an im2col matrix, in the reference's own weight layout; everything here is HBM-bound bf16 byte work on NHWC maps
// (the token matrix [n_img, 8 * 32, C] of the encoder IS the NHWC map of `seq_x.reshape(B, 8, 32, C).permute(0, 3, 1, 2)`, :251-253).
//   im2col3x3:   col[(b, y, x), c * 9 + ky * 3 + kx] = map[b, y + ky - 1, x + kx - 1, c] (zero outside), the column order of nn.Conv2d's
//                weight.view(C_out, C_in * 9): conv = col @ W^T, dW = dy^T @ col with the arena's own views.  Row pitch ldc >= 9 C, a multiple
//                of 64 (the GEMM's reduction granule): columns [9 C, ldc) are zero-filled.
//   conv3x3_weight_flip: Wt[c_in, c_out * 9 + t] = W[c_out, c_in * 9 + 8 - t]: the data gradient of the convolution is the convolution of dy with
//                the flipped, transposed taps -- dx = im2col(dy) @ Wt^T, a direct-form GEMM again (no col2im scatter).
//   maxpool2x2:  nn.MaxPool2d(2, 2) and its gradient: the FIRST maximum in (0,0), (0,1), (1,0), (1,1) order wins a tie (aten's `val > maxval`
//                scan from the window's first element) -- after a ReLU whole windows of zeros are common.  Index map: one byte per output.
#include "common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 u, unsigned short* v) {
  v[0] = u.x & 0xffff; v[1] = u.x >> 16; v[2] = u.y & 0xffff; v[3] = u.y >> 16;
  v[4] = u.z & 0xffff; v[5] = u.z >> 16; v[6] = u.w & 0xffff; v[7] = u.w >> 16;
}

// item = (map pixel r, group of 8 channels): nine 16-byte loads (a tap each), 72 consecutive output columns = nine 16-byte stores
// (8 g * 9 * 2 B = 144 g: 16-byte aligned).  The thread of group 0 also zero-fills the row's pad columns.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int n_img, int H, int W, int C,
                                                        int ldc) {
  const int c8 = C >> 3;
  const size_t total = (size_t)n_img * H * W * c8;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(e % c8);
    const size_t r = e / c8;
    const int xx = (int)(r % W), yy = (int)((r / W) % H);
    unsigned short in[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
      uint4 u = make_uint4(0, 0, 0, 0);
      if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W)
        u = *reinterpret_cast<const uint4*>(x + ((ptrdiff_t)r + (ptrdiff_t)(t / 3 - 1) * W + (t % 3 - 1)) * C + g * 8);
      unpack8(u, in[t]);
    }
    uint4* o = reinterpret_cast<uint4*>(col + r * (size_t)ldc + (size_t)g * 72);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      unsigned w[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k0 = q * 8 + 2 * p, k1 = k0 + 1;                  // output columns (relative): k = i * 9 + t
        w[p] = (unsigned)in[k0 % 9][k0 / 9] | ((unsigned)in[k1 % 9][k1 / 9] << 16);
      }
      o[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (g == 0) {
      uint4* z = reinterpret_cast<uint4*>(col + r * (size_t)ldc + (size_t)C * 9);
      for (int q = 0; q < (ldc - C * 9) >> 3; ++q) z[q] = make_uint4(0, 0, 0, 0);
    }
  }
}

// thread = (c_in, c_out): nine taps W[c_out, c_in * 9 + (8 - t)] -> Wt[c_in, c_out * 9 + t]; threads are adjacent in c_out (18-byte runs of
// the output next to each other).  <= 5.3 M elements per layer at ViT-S: microseconds.
__global__ __launch_bounds__(256) void conv3x3_weight_flip_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wt, int c_out, int c_in) {
  const size_t total = (size_t)c_out * c_in;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(e % c_out), ci = (int)(e / c_out);
    const bf16_t* s = w + ((size_t)co * c_in + ci) * 9;
    bf16_t* d = wt + ((size_t)ci * c_out + co) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) d[t] = s[8 - t];
  }
}

// item = (output pixel, group of 8 channels)
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, unsigned char* __restrict__ idx,
                                                             int n_img, int H, int W, int C) {
  const int c8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)n_img * Ho * Wo * c8;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(e % c8);
    const size_t ro = e / c8;
    const int xo = (int)(ro % Wo), yo = (int)((ro / Wo) % Ho);
    const size_t b = ro / ((size_t)Wo * Ho);
    const bf16_t* p = x + ((b * H + 2 * yo) * W + 2 * xo) * (size_t)C + g * 8;
    unsigned short m[8], v[8];
    unsigned char am[8];
    unpack8(*reinterpret_cast<const uint4*>(p), m);
#pragma unroll
    for (int k = 0; k < 8; ++k) am[k] = 0;
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      unpack8(*reinterpret_cast<const uint4*>(p + ((size_t)(t >> 1) * W + (t & 1)) * C), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float fv = bf2f(v[k]), fm = bf2f(m[k]);
        if (fv > fm || fv != fv) { m[k] = v[k]; am[k] = (unsigned char)t; }         // (aten: `(val > maxval) || isnan(val)`)
      }
    }
    *reinterpret_cast<uint4*>(y + ro * C + g * 8) = make_uint4(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16),
                                                                m[4] | ((unsigned)m[5] << 16), m[6] | ((unsigned)m[7] << 16));
    *reinterpret_cast<uint2*>(idx + ro * C + g * 8) = make_uint2(am[0] | (am[1] << 8) | (am[2] << 16) | ((unsigned)am[3] << 24),
                                                                 am[4] | (am[5] << 8) | (am[6] << 16) | ((unsigned)am[7] << 24));
  }
}

// every input pixel belongs to exactly one window (kernel = stride = 2): each dx element is written once, no accumulation
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                             bf16_t* __restrict__ dx, int n_img, int H, int W, int C) {
  const int c8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)n_img * Ho * Wo * c8;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(e % c8);
    const size_t ro = e / c8;
    const int xo = (int)(ro % Wo), yo = (int)((ro / Wo) % Ho);
    const size_t b = ro / ((size_t)Wo * Ho);
    unsigned short d[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + ro * C + g * 8), d);
    const uint2 a = *reinterpret_cast<const uint2*>(idx + ro * C + g * 8);
    unsigned char am[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { am[k] = (a.x >> (8 * k)) & 0xff; am[4 + k] = (a.y >> (8 * k)) & 0xff; }
    bf16_t* p = dx + ((b * H + 2 * yo) * W + 2 * xo) * (size_t)C + g * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned short o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = am[k] == t ? d[k] : (unsigned short)0;
      *reinterpret_cast<uint4*>(p + ((size_t)(t >> 1) * W + (t & 1)) * C) = make_uint4(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16),
                                                                                      o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16));
    }
  }
}

inline unsigned grid_for(size_t total) { return (unsigned)std::min<size_t>(8192, (total + 255) / 256); }

}  // namespace

extern "C" int dig_im2col3x3(const void* x, void* col, int n_img, int H, int W, int C, int ldc, hipStream_t stream) {
  if (!x || !col || n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || ldc < 9 * C || (ldc & 7)) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(col)) return DIG_ERR_ALIGN;
  const size_t total = (size_t)n_img * H * W * (C / 8);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)col, n_img, H, W, C, ldc);
  return dig_check_launch();
}

extern "C" int dig_conv3x3_weight_flip(const void* w, void* wt, int c_out, int c_in, hipStream_t stream) {
  if (!w || !wt || c_out <= 0 || c_in <= 0) return DIG_ERR_ARG;
  hipLaunchKernelGGL(conv3x3_weight_flip_kernel, dim3(grid_for((size_t)c_out * c_in)), dim3(256), 0, stream, (const bf16_t*)w, (bf16_t*)wt, c_out,
                     c_in);
  return dig_check_launch();
}

extern "C" int dig_maxpool2x2_fwd(const void* x, void* y, unsigned char* idx, int n_img, int H, int W, int C, hipStream_t stream) {
  if (!x || !y || !idx || n_img <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) return DIG_ERR_ARG;
  if (!aligned16(x) || !aligned16(y) || (reinterpret_cast<uintptr_t>(idx) & 7)) return DIG_ERR_ALIGN;
  const size_t total = (size_t)n_img * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, idx, n_img, H, W, C);
  return dig_check_launch();
}

extern "C" int dig_maxpool2x2_bwd(const void* dy, const unsigned char* idx, void* dx, int n_img, int H, int W, int C, hipStream_t stream) {
  if (!dy || !dx || !idx || n_img <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) return DIG_ERR_ARG;
  if (!aligned16(dy) || !aligned16(dx) || (reinterpret_cast<uintptr_t>(idx) & 7)) return DIG_ERR_ALIGN;
  const size_t total = (size_t)n_img * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)dy, idx, (bf16_t*)dx, n_img, H, W, C);
  return dig_check_launch();
}
