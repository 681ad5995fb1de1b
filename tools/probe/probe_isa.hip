// Probe: empirical semantics of gfx950 primitives this repo's kernels rely on.
//  (1) ds_read_b64_tr_b16 lane/element mapping
//  (2) v_mfma_f32_32x32x16_bf16 A/B/C fragment layout
//  (3) __builtin_amdgcn_global_load_lds (16 B) destination order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned short f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fff + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}

__global__ void k_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // each lane supplies its own 8-byte address: group g=l>>4 owns a 128-B block; lane i -> +i*8
  typedef __attribute__((ext_vector_type(4))) short v4s;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + (l >> 4) * 64 + (l & 15) * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

__global__ void k_mfma(const unsigned short* A, const unsigned short* B, float* C) {
  // A: [32][16] row-major (m,k), B: [16][32] row-major (k,n). lane l: a = A[l&31][(l>>5)*8 + j], b = B[(l>>5)*8+j][l&31]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)A[(l & 31) * 16 + (l >> 5) * 8 + j];
    b[j] = (short)B[((l >> 5) * 8 + j) * 32 + (l & 31)];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int col = l & 31;
    C[row * 32 + col] = c[r];
  }
}

__global__ void k_glds(const unsigned short* src, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 8 * 2];
  int l = threadIdx.x;  // 128 threads = 2 waves
  int w = l >> 6;
  // each lane copies 16 B from src + perm(l)*8 elements; perm = reverse within wave
  const unsigned short* g = src + (w * 64 + (63 - (l & 63))) * 8;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + w * 512), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 1024; i += 128) out[i] = lds[i];
}

int main() {
  // (1)
  unsigned short* d_out; hipMalloc(&d_out, 64 * 4 * 2);
  k_tr<<<1, 64>>>(d_out);
  std::vector<unsigned short> h(256);
  hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost);
  printf("TR: lane -> 4 elems (element index in LDS, block = 64 elems per 16-lane group)\n");
  int ok_tr = 1;
  for (int l = 0; l < 64; ++l) {
    printf("  l%02d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    for (int j = 0; j < 4; ++j) if (h[l*4+j] != (l>>4)*64 + j*16 + (l&15)) ok_tr = 0;
  }
  printf("TR_SEMANTIC_EXPECTED=%d\n", ok_tr);
  // (2)
  std::vector<unsigned short> A(32*16), B(16*32); std::vector<float> Af(32*16), Bf(16*32);
  auto tobf = [](float f){ unsigned u; std::memcpy(&u,&f,4); return (unsigned short)(u>>16); };
  for (int i = 0; i < 32*16; ++i) { float v = (float)((i*7)%13 - 6); Af[i]=v; A[i]=tobf(v); }
  for (int i = 0; i < 16*32; ++i) { float v = (float)((i*5)%11 - 5); Bf[i]=v; B[i]=tobf(v); }
  unsigned short *dA,*dB; float* dC; hipMalloc(&dA, 1024); hipMalloc(&dB,1024); hipMalloc(&dC, 32*32*4);
  hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice); hipMemcpy(dB,B.data(),1024,hipMemcpyHostToDevice);
  k_mfma<<<1,64>>>(dA,dB,dC);
  std::vector<float> C(1024); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m=0;m<32;++m) for (int n=0;n<32;++n){ float r=0; for(int k=0;k<16;++k) r+=Af[m*16+k]*Bf[k*32+n]; if (fabsf(r-C[m*32+n])>1e-3) ++bad; }
  printf("MFMA_LAYOUT_BAD=%d\n", bad);
  // (3)
  std::vector<unsigned short> S(1024); for (int i=0;i<1024;++i) S[i]=i;
  unsigned short *dS,*dO; hipMalloc(&dS,2048); hipMalloc(&dO,2048);
  hipMemcpy(dS,S.data(),2048,hipMemcpyHostToDevice);
  k_glds<<<1,128>>>(dS,dO);
  std::vector<unsigned short> O(1024); hipMemcpy(O.data(), dO, 2048, hipMemcpyDeviceToHost);
  int badg=0; for (int w=0;w<2;++w) for(int l=0;l<64;++l) for(int e=0;e<8;++e){ int exp=(w*64+(63-l))*8+e; if (O[w*512+l*8+e]!=exp) ++badg; }
  printf("GLDS_BAD=%d\n", badg);
  hipError_t e = hipDeviceSynchronize(); printf("status=%s\n", hipGetErrorString(e));
  return 0;
}
