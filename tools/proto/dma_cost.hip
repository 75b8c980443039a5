// micro-benchmark: cost of one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave) issued between dependent MFMAs,
// one wave per SIMD (256 threads / CU).  Variants: which waves issue, how the issues are spaced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define MFMA(a,b,c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8,(a)), __builtin_bit_cast(bf16x8,(b)), (c), 0,0,0)
template <int IMM>
__device__ __forceinline__ void blds(const char* base, uint32_t lane16, uint32_t soff, uint32_t lds_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, (int)lane16, (int)soff, IMM, 0);
}
// MODE 0: no DMA; 1: every wave issues 8 pieces in the first 8 gaps of each 48-MFMA tile; 2: every wave, one piece every 6 gaps;
// 3: like 2 but wave w shifted by w gaps; 4: only wave (tile & 3) issues, 32 pieces spread (2 per 3 gaps)
template <int MODE, bool SHARED>
__global__ __launch_bounds__(256) void k(const u32x4* in, const char* w, float* out, int tiles) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4 a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const char* src = w + (size_t)(SHARED ? 0 : blockIdx.x) * 65536;
  for (int t = 0; t < tiles; ++t) {
    const uint32_t dst = (t & 1) * 32768;
    const char* s = src + (t & 1) * 32768 + wave * 8192;
#pragma unroll
    for (int u = 0; u < 48; ++u) {
      acc = MFMA(a, b, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1 && u < 8) { if (u < 4) blds<0>(s, lane * 16, u * 1024, dst + wave * 8192 + u * 1024); else blds<0>(s, lane * 16, u * 1024, dst + wave * 8192 + u * 1024); }
      if (MODE == 2 && u % 6 == 0) blds<0>(s, lane * 16, (u / 6) * 1024, dst + wave * 8192 + (u / 6) * 1024);
      if (MODE == 3 && u >= 4 && (u - 4) % 5 == 0 && (u - 4) / 5 < 8) {
        // shifted per wave: wave w issues at u = 4 + 5 j - w  -> emulate by comparing
      }
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (u == 3 + 5 * j) { if (wave == 0) blds<0>(s, lane * 16, j * 1024, dst + wave * 8192 + j * 1024); }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (u == 4 + 5 * j) { if (wave == 1) blds<0>(s, lane * 16, j * 1024, dst + wave * 8192 + j * 1024); }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (u == 5 + 5 * j) { if (wave == 2) blds<0>(s, lane * 16, j * 1024, dst + wave * 8192 + j * 1024); }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (u == 6 + 5 * j) { if (wave == 3) blds<0>(s, lane * 16, j * 1024, dst + wave * 8192 + j * 1024); }
      }
      if (MODE == 4 && wave == (t & 3) && u < 47 && (u % 3) != 2) { const int p = (u / 3) * 2 + (u % 3); if (p < 32) blds<0>(src + (t & 1) * 32768, lane * 16, p * 1024, dst + p * 1024); }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float sum = 0.f;
  for (int i = 0; i < 16; ++i) sum += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum + smem[threadIdx.x];
}
template <int MODE, bool SHARED>
void run(const u32x4* in, const char* w, float* out) {
  const int tiles = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)(k<MODE, SHARED>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<MODE, SHARED>), dim3(256), dim3(256), 65536, 0, in, w, out, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, SHARED>), dim3(256), dim3(256), 65536, 0, in, w, out, tiles);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("shared %d mode %d: %.3f ms  ns/tile %.1f  (48 MFMAs ideal 640 ns @2.4GHz)\n", (int)SHARED, MODE, ms, ms * 1e6 / tiles);
}
int main() {
  u32x4* in; float* out; char* w;
  (void)hipMalloc(&in, 512 * 16); (void)hipMemset(in, 0x3c, 512 * 16); (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&w, (size_t)256 * 65536); (void)hipMemset(w, 1, (size_t)256 * 65536);
  for (int rep = 0; rep < 2; ++rep) { run<0,false>(in, w, out); run<1,false>(in, w, out); run<2,false>(in, w, out); run<1,true>(in, w, out); run<2,true>(in, w, out); }
  return 0;
}
