// Phase timeline + ablations of the attention-backward kernel on one MI355X.  s_memtime stamps at the phase boundaries of every wave
// (hooks: DIG_ATTN_TS in dig_amd/csrc/attention.hip), averaged over all workgroups, plus the launch's wall time; `lab_kernel` is a
// copy of attn_bwd2_kernel with compile-time ablations (ABL bits: 1 no softmax arithmetic, 2 no LDS fragment re-reads inside the
// loops, 4 no result stores, 8 no delta loads, 16 skip phase A, 32 skip phase B).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I include -I dig_amd/csrc tools/experiments/attn_bwd_lab.hip -o build/lab/attn_bwd_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ long long* g_ts;
__device__ long long* g_rt;
#define NTS 8
#define DIG_ATTN_TS(i) if ((threadIdx.x & 63) == 0) { g_ts[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * NTS + (i)] = (long long)__builtin_amdgcn_s_memtime(); if ((i) == 0 || (i) == 6) g_rt[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + ((i) ? 1 : 0)] = (long long)__builtin_amdgcn_s_memrealtime(); }
#include "../../dig_amd/csrc/attention.hip"

namespace {
template <bool DROP, int ABL>
__global__ __launch_bounds__(256, 2) void lab_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ ctx,
                                                           const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, int D, int H, float scale,
                                                           unsigned qkv_bytes, unsigned ctx_bytes, float* __restrict__ qsum, float* __restrict__ vsum,
                                                           dig_dropout_t drop, int nqb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* T0 = smem;                                              // K, then Q
  unsigned char* T1 = smem + TILE;                                       // V, then dO
  float* lse_s = reinterpret_cast<float*>(smem + 2 * TILE);              // [256]
  float* del_s = lse_s + N_TOK;                                          // [256]
  float* csum_s = del_s + N_TOK;                                         // [8 blocks][2][64]: column sums of dQ and dV (qsum only)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / H, h = blockIdx.x - img * H;
  const int ld = 3 * D;
  const size_t tok0 = (size_t)img * N_TOK;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, qkv_bytes, 0x00020000);
  const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)dctx, 0, ctx_bytes, 0x00020000);
  const unsigned base = (unsigned)((tok0 * ld + h * DH) * 2);
  DIG_ATTN_TS(0)
  stage_tile<256>(T0, rs, base + (unsigned)(D * 2), ld, tid, wave);      // K
  stage_tile<256>(T1, rs, base + (unsigned)(2 * D * 2), ld, tid, wave);  // V
  const int hi = lane >> 5;
  // delta[q] = sum_d dO[q,d] * O[q,d]; one thread per query
  {
    const int q = tid;
    const bf16_t* o = ctx + (tok0 + q) * D + h * DH;
    const bf16_t* g = dctx + (tok0 + q) * D + h * DH;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < ((ABL & 8) ? 0 : 8); ++c) {
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o + c * 8);
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += bf2f((bf16_t)ov[e]) * bf2f((bf16_t)gv[e]);
    }
    del_s[q] = acc;
    lse_s[q] = lse[(size_t)blockIdx.x * N_TOK + q];
  }
  // Q / dO fragments of this wave's two query blocks, straight from global
  bf16x8 qf[4], gf[4];
  auto load_qg = [&](int qb) {
    const int q = qb * 32 + (lane & 31);
    const bf16_t* qp = qkv + (tok0 + q) * ld + h * DH + hi * 8;
    const bf16_t* gp = dctx + (tok0 + q) * D + h * DH + hi * 8;
    if (qb < nqb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(qp + s * 16);
        gf[s] = *reinterpret_cast<const bf16x8*>(gp + s * 16);
      }
    }
  };
  load_qg(wave * 2);
  DIG_ATTN_TS(1)
  __syncthreads();
  DIG_ATTN_TS(2)

  // ---------------- phase A: dQ for query blocks 2*wave, 2*wave+1 ----------------
  // Software pipeline over the 8 key tiles, two tiles per trip with named accumulators: the S / dP MFMAs of tile kt+1 are
  // issued BEFORE the softmax arithmetic of tile kt, so the matrix pipe works through them while this wave's VALU runs
  // (within a wave the chain MFMA -> exp -> MFMA is serial; with two waves per SIMD that chain left both pipes idle most
  // of the time: SQ_WAIT_INST_ANY was 2x SQ_ACTIVE_INST_ANY).
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int qb = wave * 2 + ps;
    if (qb >= nqb || (ABL & 16)) continue;
    const int q0 = qb * 32;
    const int q = q0 + (lane & 31);
    const float my_lse = lse_s[q], my_del = del_s[q];
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
    bf16x8 kfr[4], vfr[4], ktr[2][2];
    auto load_kv_tile = [&](int kt) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kfr[s] = frag_direct(T0, kt * 32, s, lane);
        vfr[s] = frag_direct(T1, kt * 32, s, lane);
      }
    };
    auto load_ktr = [&](int kt) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ktr[u][dt] = frag_tr(T0, kt * 32 + u * 16, dt * 32, lane);
    };
    auto mma_sdp = [&](f32x16& st, f32x16& dp) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], gf[s], dp, 0, 0, 0);
      }
    };
    auto soft_dq = [&](f32x16& st, f32x16& dp, int kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float g = dp[e];
        if (DROP) {
          const unsigned key = kt * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
          g = dig_drop_keep(drop.k0, drop.k1, ((unsigned)q << 16) | key, blockIdx.x, drop.thr) ? g * drop.scale : 0.f;
        }
        if (!(ABL & 1)) st[e] = __expf(st[e] - my_lse) * (g - my_del);                     // dS^T
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 ds = pack8(st, u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[u][dt], ds, dq[dt], 0, 0, 0);
      }
    };
    f32x16 stA, dpA, stB, dpB;
    load_kv_tile(0);
    mma_sdp(stA, dpA);
    __builtin_amdgcn_sched_barrier(0);
    load_kv_tile(1);
    for (int kt = 0; kt < 8; kt += 2) {
      if (!(ABL & 2) || kt == 0) load_ktr(kt);
      __builtin_amdgcn_sched_barrier(0);
      mma_sdp(stB, dpB);                                                   // tile kt+1 in the matrix pipe ...
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 2)) load_kv_tile((kt + 2) & 7);
      __builtin_amdgcn_sched_barrier(0);
      soft_dq(stA, dpA, kt);                                               // ... while tile kt's dS is formed
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 2)) load_ktr(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 2 < 8) mma_sdp(stA, dpA);                                   // tile kt+2
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 2)) load_kv_tile((kt + 3) & 7);
      __builtin_amdgcn_sched_barrier(0);
      soft_dq(stB, dpB, kt + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ps == 0) load_qg(qb + 1);                                         // next block's rows fly while this block's result is stored
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] *= scale;
    if (!(ABL & 4)) store_rows(dqkv + (tok0 + q) * ld + h * DH, dq, hi); else asm volatile("" :: "v"(dq[0][0]), "v"(dq[1][5]));
    if (qsum) wave_colsum(dq, csum_s + qb * 128, lane);
  }

  DIG_ATTN_TS(3)
  // ---------------- restage: Q -> T0, dO -> T1 (every wave is done with K, V) ----------------
  // K / V fragments of this wave's two key blocks come from global memory (L2-warm), requested before the barrier
  bf16x8 kf[4], vf[4];
  auto load_kv = [&](int kb) {
    const int key = kb * 32 + (lane & 31);
    const bf16_t* kp = qkv + (tok0 + key) * ld + D + h * DH + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + s * 16);
      vf[s] = *reinterpret_cast<const bf16x8*>(kp + D + s * 16);
    }
  };
  load_kv(wave * 2);
  __syncthreads();
  DIG_ATTN_TS(4)
  stage_tile<256>(T0, rs, base, ld, tid, wave);                                         // Q
  stage_tile<256>(T1, rg, (unsigned)((tok0 * D + h * DH) * 2), D, tid, wave);           // dO
  __syncthreads();
  DIG_ATTN_TS(5)

  // ---------------- phase B: dK, dV for key blocks 2*wave, 2*wave+1 ----------------
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int kb = wave * 2 + ps;
    if (ABL & 32) continue;
    const int k0 = kb * 32;
    const int key = k0 + (lane & 31);
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
    bf16x8 qfr[4], gfr[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qfr[s] = frag_direct(T0, 0, s, lane);
      gfr[s] = frag_direct(T1, 0, s, lane);
    }
#pragma unroll 2
    for (int qt = 0; qt < nqb; ++qt) {
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[s], vf[s], dp, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 gtr[2][2], qtr[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          gtr[u][dt] = frag_tr(T1, ((ABL & 2) ? 0 : qt) * 32 + u * 16, dt * 32, lane);
          qtr[u][dt] = frag_tr(T0, ((ABL & 2) ? 0 : qt) * 32 + u * 16, dt * 32, lane);
        }
      float ls[4][4], dl[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qr = qt * 32 + 8 * g + 4 * hi;
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qr);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + qr);
        ls[g][0] = l4.x; ls[g][1] = l4.y; ls[g][2] = l4.z; ls[g][3] = l4.w;
        dl[g][0] = d4.x; dl[g][1] = d4.y; dl[g][2] = d4.z; dl[g][3] = d4.w;
      }
      const int qtn = (qt + 1) & 7;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qfr[s] = frag_direct(T0, qtn * 32, s, lane);
        gfr[s] = frag_direct(T1, qtn * 32, s, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = (ABL & 1) ? st[g * 4 + e] : __expf(st[g * 4 + e] - ls[g][e]);
          float m = 1.f;
          if (DROP) {
            const unsigned qi = qt * 32 + 8 * g + 4 * hi + e;
            m = dig_drop_keep(drop.k0, drop.k1, (qi << 16) | (unsigned)key, blockIdx.x, drop.thr) ? drop.scale : 0.f;
          }
          st[g * 4 + e] = p * m;
          if (!(ABL & 1)) dp[g * 4 + e] = p * (dp[g * 4 + e] * m - dl[g][e]);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 pf = pack8(st, u);
        const bf16x8 ds = pack8(dp, u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtr[u][dt], pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[u][dt], ds, dk[dt], 0, 0, 0);
        }
      }
    }
    if (ps == 0) load_kv(kb + 1);
    bf16_t* okp = dqkv + (tok0 + key) * ld + D + h * DH;
    if (!(ABL & 4)) { store_rows(okp, dk, hi); store_rows(okp + D, dv, hi); } else asm volatile("" :: "v"(dk[0][0]), "v"(dv[1][5]), "v"(dk[1][3]), "v"(dv[0][2]));
    if (vsum) wave_colsum(dv, csum_s + kb * 128 + 64, lane);
  }
  DIG_ATTN_TS(6)
  if (qsum) {
    __syncthreads();
    if (tid < 128) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += csum_s[w * 128 + tid];
      float* dst = tid < 64 ? qsum : vsum;
      dst[(size_t)img * D + h * DH + (tid & 63)] = a;
    }
  }
}


}  // namespace

static const int Bn = 256, H = 6, D = H * 64;
static unsigned short *qkv, *ctx, *dctx, *dqkv; static float* lse; static long long* ts; static long long* rt;

template <int ABL> void launch_lab() {
  const size_t qb = (size_t)Bn * N_TOK * 3 * D * 2;
  const int lds = 2 * TILE + 2 * N_TOK * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(lab_kernel<false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((lab_kernel<false, ABL>), dim3(Bn * H), dim3(256), lds, 0, qkv, ctx, dctx, lse, dqkv, D, H, 0.125f, (unsigned)qb, (unsigned)(qb / 3),
                     (float*)nullptr, (float*)nullptr, dig_dropout_t{}, 8);
}

template <typename F> void timeline(const char* name, F fn) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(ts, 0, (size_t)Bn * H * 8 * NTS * 8);
  for (int it = 0; it < 400; ++it) fn();                     // ~70 ms of warm-up: the clocks ramp up over milliseconds
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int n = 50;
  for (int it = 0; it < n; ++it) fn();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t((size_t)Bn * H * 8 * NTS);
  hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost);
  double acc[NTS] = {0}; long cnt = 0; long long tmin = -1, tmax = 0;
  for (int b = 0; b < Bn * H; ++b)
    for (int w = 0; w < 8; ++w) {
      const long long* p = &t[((size_t)b * 8 + w) * NTS];
      if (!p[0] || !p[6]) continue;
      for (int i = 1; i <= 6; ++i) acc[i] += (double)(p[i] - p[i - 1]);
      if (tmin < 0 || p[0] < tmin) tmin = p[0];
      if (p[6] > tmax) tmax = p[6];
      ++cnt;
    }
  printf("%-34s %7.1f us |", name, ms / n * 1e3);
  double tot = 0;
  for (int i = 1; i <= 6; ++i) { printf(" %7.0f", cnt ? acc[i] / cnt : 0.0); tot += cnt ? acc[i] / cnt : 0.0; }
  { double a7 = 0; long c7 = 0; for (int b = 0; b < Bn * H; ++b) for (int w = 0; w < 4; ++w) { const long long* p = &t[((size_t)b * 8 + w) * NTS]; if (p[7] && p[3]) { a7 += (double)(p[7] - p[3]); ++c7; } }
    printf(" | wave %7.0f ticks; B block 0 loop %7.0f", tot, c7 ? a7 / c7 : 0.0); }
  { std::vector<long long> r((size_t)Bn * H * 8 * 2); hipMemcpy(r.data(), rt, r.size() * 8, hipMemcpyDeviceToHost);
    double life = 0; long c = 0; long long mn = -1, mx = 0;
    for (int b = 0; b < Bn * H; ++b) for (int w = 0; w < 4; ++w) { const long long* q = &r[((size_t)b * 8 + w) * 2]; if (q[0] && q[1]) { life += (double)(q[1] - q[0]); ++c; if (mn < 0 || q[0] < mn) mn = q[0]; if (q[1] > mx) mx = q[1]; } }
    printf("; realtime: wave life %.2f us, launch span %.1f us\n", c ? life / c / 100.0 : 0.0, (mx - mn) / 100.0); }
}

int main() {
  const size_t nq = (size_t)Bn * 256 * 3 * D, nc = (size_t)Bn * 256 * D;
  std::vector<unsigned short> h(nq);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 2.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&ctx, nc * 2); hipMalloc(&dctx, nc * 2); hipMalloc(&lse, (size_t)Bn * H * 256 * 4);
  hipMalloc(&ts, (size_t)Bn * H * 8 * NTS * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts));
  hipMalloc(&rt, (size_t)Bn * H * 8 * 2 * 8); hipMemset(rt, 0, (size_t)Bn * H * 8 * 2 * 8); hipMemcpyToSymbol(HIP_SYMBOL(g_rt), &rt, sizeof(rt));
  hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice);
  hipMemcpy(dctx, h.data(), nc * 2, hipMemcpyHostToDevice);
  dig_attn_fwd(qkv, ctx, lse, Bn, H, D, 0);
  printf("columns: issue-stage+delta | wait staging | phase A | kv frags+barrier | restage | phase B  (s_memtime ticks per wave)\n");
  timeline("product dig_attn_bwd", [] { dig_attn_bwd(qkv, ctx, dctx, lse, dqkv, Bn, H, D, 0.125f, nullptr, nullptr, 0); });
  timeline("lab full", [] { launch_lab<0>(); });
  timeline("lab -softmax VALU", [] { launch_lab<1>(); });
  timeline("lab -LDS re-reads", [] { launch_lab<2>(); });
  timeline("lab -stores", [] { launch_lab<4>(); });
  timeline("lab -delta loads", [] { launch_lab<8>(); });
  timeline("lab -phase A", [] { launch_lab<16>(); });
  timeline("lab -phase B", [] { launch_lab<32>(); });
  timeline("lab -VALU -LDS", [] { launch_lab<3>(); });
  timeline("lab -VALU -LDS -stores -delta", [] { launch_lab<15>(); });
  return 0;
}
