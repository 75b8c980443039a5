// micro-test (round 5): does v_mfma_f32_32x32x16_f16 honour fp16 DENORMAL inputs, and what does v_cvt_pk_f16_f32 (the
// kernel's RNE pack) produce for fp32 values below 2^-14?  Decides whether the fp16 (hi, lo) split of the bf16x3 NoF needs
// its lo term scaled into the normal range.  Also times the f16 instruction against the bf16 one (same shape).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/f16_denorm.hip -o build/proto/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* av, const float* bv, float* out, float* cvt) {
  // A[row][k] = av[0] for all, B[k][col] = bv[0]: every output = 16 * a * b (K = 16)
  f32x2 pa; pa[0] = av[0]; pa[1] = av[0];
  f32x2 pb; pb[0] = bv[0]; pb[1] = bv[0];
  const f16x2 ha = __builtin_convertvector(pa, f16x2), hb = __builtin_convertvector(pb, f16x2);
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ha[i & 1]; b[i] = hb[i & 1]; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; cvt[0] = (float)ha[0]; cvt[1] = (float)hb[0]; }
}

template <bool F16>
__global__ __launch_bounds__(256) void rate(const unsigned* in, float* out, int iters) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 a = ((const u32x4*)in)[threadIdx.x & 63], b = ((const u32x4*)in)[64 + (threadIdx.x & 63)];
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (F16) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc1, 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *av, *bv, *out, *cvt;
  hipMalloc(&av, 4); hipMalloc(&bv, 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cvt, 8);
  const float cases[][2] = {{1.f, 1.f}, {ldexpf(1.f, -16), 1.f}, {ldexpf(1.f, -20), 1.f}, {ldexpf(1.f, -24), 1.f}, {ldexpf(1.5f, -24), 1.f},
                            {ldexpf(1.f, -25), 1.f}, {1.f, ldexpf(1.f, -20)}, {ldexpf(1.f, -20), ldexpf(1.f, 10)}, {ldexpf(1.f, -12), ldexpf(1.f, -12)}};
  for (auto& c : cases) {
    hipMemcpy(av, &c[0], 4, hipMemcpyHostToDevice); hipMemcpy(bv, &c[1], 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, av, bv, out, cvt);
    float o, cv[2]; hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost); hipMemcpy(cv, cvt, 8, hipMemcpyDeviceToHost);
    printf("a %.6e (fp16 %.6e)  b %.6e (fp16 %.6e)  mfma 16ab = %.6e  expected %.6e  %s\n", c[0], cv[0], c[1], cv[1], o, 16.0 * cv[0] * cv[1],
           fabs(o - 16.0 * cv[0] * cv[1]) <= 1e-6 * fabs(16.0 * cv[0] * cv[1]) ? "ok" : "DIFFERS");
  }
  unsigned* in; hipMalloc(&in, 128 * 16);
  unsigned h[512];
  for (int i = 0; i < 512; ++i) h[i] = 0x3c003c00u ^ ((i * 2654435761u) & 0x03ff03ffu);     // fp16 values in [1, 2): random mantissas
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int f16 = 0; f16 < 2; ++f16) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int w = 0; w < 3; ++w) { if (f16) hipLaunchKernelGGL(rate<true>, dim3(256), dim3(256), 0, 0, in, out, iters); else hipLaunchKernelGGL(rate<false>, dim3(256), dim3(256), 0, 0, in, out, iters); }
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) { if (f16) hipLaunchKernelGGL(rate<true>, dim3(256), dim3(256), 0, 0, in, out, iters); else hipLaunchKernelGGL(rate<false>, dim3(256), dim3(256), 0, 0, in, out, iters); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double tf = (double)iters * 16 * 4 * 256 * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%s 32x32x16: %.3f ms  %.0f TFLOP/s (same bit patterns as operands)\n", f16 ? "f16 " : "bf16", ms, tf);
  }
  return 0;
}
