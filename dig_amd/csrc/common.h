// Shared device helpers for the gfx950 kernels of dig_amd.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned dig_u32x4 __attribute__((ext_vector_type(4)));

#define DIG_OK 0
#define DIG_ERR_ARG (-1)
#define DIG_ERR_ALIGN (-2)
#define DIG_ERR_LAUNCH (-3)
#define DIG_ERR_UNSUPPORTED (-4)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16, round-to-nearest-even, via the gfx950 packed converter (v_cvt_pk_bf16_f32: one instruction per PAIR;
// a hand-rolled integer rounding costs ~10 VALU ops per value and was the largest VALU item of every epilogue)
typedef __bf16 dig_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dig_f32x2 __attribute__((ext_vector_type(2)));
// A dword as a bf16 pair for v_dot2.  hipcc 7.2 folds `bit_cast<bf16x2>(v[q])` of a 4-dword vector element to element 0 for every q (the
// code then loads and multiplies dword 0 four times); the empty asm keeps the extracted dword a scalar value the fold cannot see through.
__device__ __forceinline__ dig_bf16x2 dig_as_bf16x2(unsigned u) {
  asm volatile("" : "+v"(u));
  return __builtin_bit_cast(dig_bf16x2, u);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const dig_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dig_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf(x) by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off class): 1 rcp + 1 exp + 6 fma instead
// of the ~40-instruction libm erff -- the exact-erf GELU of nn.GELU() sits in a GEMM epilogue where VALU time matters.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// erf(z) for GELU epilogues whose result is rounded to bf16: odd degree-15 polynomial on |z| <= 3 (clamped; minimax fit,
// |abs err| <= 8.8e-5 including the clamp, i.e. <= 4.4e-5 relative on gelu -- 1/90 of a bf16 ulp) -- 8 FMAs and no
// transcendental, about half the issue cost of erf_fast inside a GEMM epilogue.
__device__ __forceinline__ float erf_poly(float z) {
  const float zc = fminf(fmaxf(z, -3.0f), 3.0f);
  const float z2 = zc * zc;
  float p = -4.055360137e-07f;
  p = fmaf(p, z2, 1.715983126e-05f);
  p = fmaf(p, z2, -3.145953815e-04f);
  p = fmaf(p, z2, 3.318710718e-03f);
  p = fmaf(p, z2, -2.268579789e-02f);
  p = fmaf(p, z2, 1.077178270e-01f);
  p = fmaf(p, z2, -3.732314110e-01f);
  p = fmaf(p, z2, 1.127895713e+00f);
  return p * zc;
}
// erf-GELU (nn.GELU()) and its derivative, fp32 in / out, for bf16-rounded results
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_poly(x * 0.70710678118654752f)); }
// The same GELU with the constants folded into the polynomial (x erf_poly(x / sqrt 2) / 2 = x xc Q(xc^2), xc = x clamped to
// +-3 sqrt 2): 11 VALU ops instead of 13.5, same error bound -- for code whose VALU stream is placed between MFMAs by hand.
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.2426405f, 4.2426405f);      // (fminf(fmaxf()) on an MFMA result costs a canonicalising v_max)
  const float t = xc * xc;
  float p = -1.120145568e-09f;
  p = fmaf(p, t, 9.479557069e-08f);
  p = fmaf(p, t, -3.475820744e-06f);
  p = fmaf(p, t, 7.333383917e-05f);
  p = fmaf(p, t, -1.002580095e-03f);
  p = fmaf(p, t, 9.521000741e-03f);
  p = fmaf(p, t, -6.597861542e-02f);
  p = fmaf(p, t, 3.987713536e-01f);
  return x * fmaf(xc, p, 0.5f);
}
// two values at once, the two dependency chains written interleaved (a lone chain issues at ~6.6 cycles per op, two at ~4)
__device__ __forceinline__ void gelu_fast2(float& a, float& b) {
  const float ac = __builtin_amdgcn_fmed3f(a, -4.2426405f, 4.2426405f), bc = __builtin_amdgcn_fmed3f(b, -4.2426405f, 4.2426405f);
  const float ta = ac * ac, tb = bc * bc;
  float pa = -1.120145568e-09f, pb = -1.120145568e-09f;
#define DIG_G2(c) pa = fmaf(pa, ta, c); pb = fmaf(pb, tb, c);
  DIG_G2(9.479557069e-08f) DIG_G2(-3.475820744e-06f) DIG_G2(7.333383917e-05f) DIG_G2(-1.002580095e-03f)
  DIG_G2(9.521000741e-03f) DIG_G2(-6.597861542e-02f) DIG_G2(3.987713536e-01f)
#undef DIG_G2
  a *= fmaf(ac, pa, 0.5f);
  b *= fmaf(bc, pb, 0.5f);
}
// gelu'(x) = Phi(x) + x phi(x).  gelu' - 1/2 is odd: one odd degree-15 polynomial in z = x/4 on |z| <= 1 (clamped: beyond |x| = 4 the
// derivative is within 5e-4 of its limits 0 / 1), minimax fit, |abs err| <= 2.8e-4 (a seventh of a bf16 ulp of the O(1) factor it
// is) -- 11 VALU ops and no transcendental, against 19 for erf_poly + exp: the fc2-dgrad epilogue is VALU-bound (round-2 PMC).
__device__ __forceinline__ float dgelu_f(float x) {
  const float z = __builtin_amdgcn_fmed3f(x * 0.25f, -1.0f, 1.0f);
  const float t = z * z;
  float p = -1.763060760e+01f;
  p = fmaf(p, t, 8.145713806e+01f);
  p = fmaf(p, t, -1.613111115e+02f);
  p = fmaf(p, t, 1.802621155e+02f);
  p = fmaf(p, t, -1.259510422e+02f);
  p = fmaf(p, t, 5.725678253e+01f);
  p = fmaf(p, t, -1.676991081e+01f);
  p = fmaf(p, t, 3.186886549e+00f);
  return fmaf(p, z, 0.5f);
}
// the accurate form (erf polynomial + exp, |abs err| ~1e-5) for kernels that are not VALU-bound and whose fp32 parameter-gradient
// sums are checked to 1e-4 (LayerNorm + GELU of the pixel decoder, the standalone GELU backward)
__device__ __forceinline__ float dgelu_acc_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_poly(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ void dgelu2(float xa, float xb, float& ga, float& gb) {      // ga *= gelu'(xa), gb *= gelu'(xb), chains interleaved
  const float za = __builtin_amdgcn_fmed3f(xa * 0.25f, -1.0f, 1.0f), zb = __builtin_amdgcn_fmed3f(xb * 0.25f, -1.0f, 1.0f);
  const float ta = za * za, tb = zb * zb;
  float pa = -1.763060760e+01f, pb = -1.763060760e+01f;
#define DIG_G2(c) pa = fmaf(pa, ta, c); pb = fmaf(pb, tb, c);
  DIG_G2(8.145713806e+01f) DIG_G2(-1.613111115e+02f) DIG_G2(1.802621155e+02f) DIG_G2(-1.259510422e+02f)
  DIG_G2(5.725678253e+01f) DIG_G2(-1.676991081e+01f) DIG_G2(3.186886549e+00f)
#undef DIG_G2
  ga *= fmaf(pa, za, 0.5f);
  gb *= fmaf(pb, zb, 0.5f);
}

// ---- dropout / stochastic depth --------------------------------------------------------------------------------------
// Counter-based keep/drop decision: a keyed two-round multiply-xorshift hash of the element coordinates (a, b) under the
// site key (k0, k1); the element is dropped when the 32-bit hash is below thr = floor(p * 2^32).  Stateless, so forward and
// backward (and the CPU oracle, oracle/finetune_oracle.py `keep_mask`) regenerate the same mask from (key, coordinates)
// and no mask tensor is ever stored.  ~9 integer ops per element; coordinates: elementwise tensors a = row * cols + col,
// b = 0; attention probabilities a = (query << 16) | key, b = sample * heads + head; drop-path a = sample, b = 0.
// multi-segment reduction descriptors (include/dig_hip.h: dig_reduce_seg_t, dig_colsum_seg_t)
#define DIG_REDUCE_MAX_SEGS 8
#define DIG_COLSUM_MAX_SEGS 112
struct dig_reduce_seg_t { const float* partials; float* out; long long n; int splits; int reserved; };
struct dig_colsum_seg_t { const float* partials; float* out; long long stride; int n_parts; int C; };

struct dig_dropout_t {
  unsigned k0, k1;        // site key
  unsigned thr;           // 0: no element dropout
  float scale;            // 1 / (1 - p)
  unsigned pk0, pk1;      // drop-path key
  unsigned pthr;          // 0: no drop-path
  float pscale;           // 1 / (1 - drop_path)
  int rows_per_sample;    // drop-path: sample = row / rows_per_sample
};
__device__ __forceinline__ unsigned dig_drop_hash(unsigned k0, unsigned k1, unsigned a, unsigned b) {
  unsigned x = a ^ k0;
  x ^= x >> 16; x *= 0x7feb352du;
  x += k1 + b * 0x9e3779b9u;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool dig_drop_keep(unsigned k0, unsigned k1, unsigned a, unsigned b, unsigned thr) {
  return dig_drop_hash(k0, k1, a, b) >= thr;
}
// 8 consecutive elements of row i starting at column j of a [rows, cols] tensor: v *= element mask * scale * drop-path factor
__device__ __forceinline__ void dig_drop_apply8(float (&v)[8], const dig_dropout_t& d, int i, int j, int cols) {
  float s = 1.f;
  if (d.pthr) s = dig_drop_keep(d.pk0, d.pk1, (unsigned)(i / d.rows_per_sample), 0u, d.pthr) ? d.pscale : 0.f;
  if (d.thr) {
    const unsigned base = (unsigned)i * (unsigned)cols + (unsigned)j;
    s *= d.scale;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = dig_drop_keep(d.k0, d.k1, base + e, 0u, d.thr) ? v[e] * s : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= s;
  }
}

// hipFuncSetAttribute and the CU count belong to a DEVICE: the launchers keep their "done once" state per device ordinal (a process that
// uses a second GPU would otherwise launch with an unset dynamic-LDS limit, or size a persistent grid for the wrong chip).  The writes are
// idempotent, so two host threads racing on an entry are harmless.
#define DIG_MAX_DEVICES 64
static inline int dig_device() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < DIG_MAX_DEVICES) ? d : 0;
}

// ---- launch probe (csrc/probe.hip): the matrix-core launchers go through dig_launch(); with the probe on, a launch carries start / stop
// events (hipExtLaunchKernel) and its device-side duration can be read back by dig_probe_stop
bool dig_probe_on();
void dig_probe_events(hipEvent_t* e0, hipEvent_t* e1);
template <typename K, typename... Args>
static inline void dig_launch(K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t stream, Args... args) {
  if (dig_probe_on()) {
    hipEvent_t e0, e1;
    dig_probe_events(&e0, &e1);
    hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0, e1, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
  }
}

// Deterministic grid-wide sum without a second launch: every workgroup leaves its NV partial values in ws[1 + NV * block + v], then takes a
// ticket (ws[0], an integer: the only atomic); the workgroup that draws the last ticket sums all partials IN BLOCK ORDER (a fixed tree over
// its 256 threads) and returns true on its thread 0 with the totals in tot[]; it also resets the ticket for the next launch.  The fences
// make the partials of the other XCDs' L2s visible (release before the ticket, acquire after it).  ws: 1 + NV * gridDim.x floats, zero once.
template <int NV>
__device__ __forceinline__ bool dig_grid_sum_last(float* __restrict__ ws, const float (&mine)[NV], float (&tot)[NV], float* red /* [256] LDS */) {
  __shared__ unsigned last_flag;
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z, blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) ws[1 + NV * blk + v] = mine[v];
    __threadfence();
    last_flag = (atomicAdd(reinterpret_cast<unsigned*>(ws), 1u) == nblk - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last_flag) return false;
  __threadfence();
  // thread t adds blocks t, t + T, ... in order; a wave folds its 64 sums with a fixed butterfly; thread 0 adds the waves in order
  float a[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) a[v] = 0.f;
  for (unsigned b = threadIdx.x; b < nblk; b += blockDim.x) {
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] += __hip_atomic_load(ws + 1 + NV * b + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    a[v] = wave_sum(a[v]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * NV + v] = a[v];
  }
  __syncthreads();
  if (threadIdx.x != 0) return false;
  const unsigned nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float t = 0.f;
    for (unsigned w = 0; w < nw; ++w) t += red[w * NV + v];
    tot[v] = t;
  }
  reinterpret_cast<unsigned*>(ws)[0] = 0u;
  return true;
}

static inline int dig_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DIG_OK : DIG_ERR_LAUNCH;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
