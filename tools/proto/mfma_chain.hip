// micro-benchmark: back-to-back DEPENDENT v_mfma_f32_32x32x16_bf16 (one accumulator) vs two / four interleaved chains, one wave
// per SIMD (256 threads, 1 block per CU), and the same with 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a,b,c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8,(a)), __builtin_bit_cast(bf16x8,(b)), (c), 0,0,0)
template <int NC>
__global__ __launch_bounds__(512) void k(const u32x4* in, float* out, int iters) {
  u32x4 a = in[threadIdx.x], b = in[threadIdx.x + 512];
  f32x16 acc[NC];
  for (int c = 0; c < NC; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u) {
      acc[u % NC] = MFMA(a, b, acc[u % NC]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < NC; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NC>
void run(int threads, const u32x4* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NC>, dim3(256), dim3(threads), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NC>, dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 48 * (threads / 64) * 256;
  const double tf = mf * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  printf("chains %d waves/SIMD %d: %.3f ms  %.0f TFLOP/s  (%.1f %% of 2516)  cycles/MFMA/SIMD @2.4GHz %.1f\n", NC, threads / 256, ms, tf, tf / 25.16,
         ms * 1e-3 * 2.4e9 / ((double)iters * 48 * (threads / 256)));
}
int main() {
  u32x4* in; float* out;
  hipMalloc(&in, 1024 * 16); hipMemset(in, 0, 1024 * 16); hipMalloc(&out, 256 * 512 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(256, in, out); run<2>(256, in, out); run<4>(256, in, out);
    run<1>(512, in, out); run<2>(512, in, out);
  }
  return 0;
}
