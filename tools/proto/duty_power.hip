// micro-benchmark (round 5): what bounds a bf16 MFMA-dense wave program on THIS chip -- issue slots or the power / current
// limiter?  v_mfma_f32_32x32x16_bf16 chains with
//   (1) zero vs random operands (round 3's mfma_chain.hip ran zeros: 95-98 % of 2 516 TFLOP/s says nothing about power),
//   (2) an injected idle phase per 16 MFMAs (s_sleep k: duty = 512 / (512 + 64 k) at one wave per SIMD): if the limiter holds
//       busy x clock constant, throughput stays flat while duty falls; if it does not, throughput falls with duty,
//   (3) one fresh 1 KiB A fragment from LDS per MFMA / per two MFMAs (the render kernels' fragment stream),
//   (4) the same with two waves per SIMD.
// Every variant runs ~0.3 s so that the power management settles; reported: wall TFLOP/s, % of 2 516, and the shader-clock
// ticks (s_memtime) per wall second as an indication of the effective clock.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/duty_power.hip -o build/proto/duty_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

extern __shared__ char smem[];

// SLEEP: s_sleep argument per 16 MFMAs (0: none).  LDSMODE: 0 operands in registers, 1 one ds_read_b128 per MFMA, 2 one per
// two MFMAs.  Two accumulator chains (as good as one: mfma_chain.hip).
template <int SLEEP, int LDSMODE>
__global__ __launch_bounds__(512) void k(const u32x4* in, float* out, int iters, unsigned long long* ticks) {
  const int lane = threadIdx.x & 63;
  u32x4 b[4], a[4];
  for (int i = 0; i < 4; ++i) { b[i] = in[(threadIdx.x & 63) + 64 * i]; a[i] = in[(threadIdx.x & 63) + 64 * (4 + i)]; }
  if (LDSMODE) {      // 16 fragment groups of 1 KiB in LDS (lane-linear, conflict-free ds_read_b128)
    for (int g = threadIdx.x >> 6; g < 16; g += blockDim.x >> 6) *(u32x4*)(smem + g * 1024 + lane * 16) = in[lane + 64 * (8 + g)];
    __syncthreads();
  }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (LDSMODE == 0) {
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        acc0 = MFMA(a[u & 3], b[(u >> 1) & 3], acc0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = MFMA(a[(u + 1) & 3], b[(u >> 1) & 3], acc1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      u32x4 r[4];
#pragma unroll
      for (int i = 0; i < 3; ++i) r[i] = *(const u32x4*)(smem + i * 1024 + lane * 16);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int f = LDSMODE == 1 ? u : u >> 1;                // fragment index of this MFMA
        acc0 = MFMA(r[f & 3], b[u & 3], acc0);
        __builtin_amdgcn_sched_barrier(0);
        const bool fetch = LDSMODE == 1 || (u & 1);
        if (fetch) r[(f + 3) & 3] = *(const u32x4*)(smem + ((f + 3) & 15) * 1024 + lane * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
}

template <int SLEEP, int LDSMODE>
void run(const char* what, int threads, const u32x4* in, float* out, unsigned long long* ticks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;         // x 16 MFMAs x 32 cycles = 20.5 M matrix cycles per wave: >= 8.5 ms per launch at one wave per SIMD
  const int launches = 24;
  hipLaunchKernelGGL((k<SLEEP, LDSMODE>), dim3(256), dim3(threads), 16384, 0, in, out, 100, ticks);
  for (int w = 0; w < 8; ++w) hipLaunchKernelGGL((k<SLEEP, LDSMODE>), dim3(256), dim3(threads), 16384, 0, in, out, iters, ticks);   // settle
  hipEventRecord(e0);
  for (int w = 0; w < launches; ++w) hipLaunchKernelGGL((k<SLEEP, LDSMODE>), dim3(256), dim3(threads), 16384, 0, in, out, iters, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= launches;
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 16 * (threads / 64) * 256;
  const double tf = mf * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  const double wps = threads / 256;
  // matrix-pipe duty at 2.4 GHz-independent terms: busy cycles per SIMD = iters * 16 * 32 * waves per SIMD; ticks = s_memtime span
  printf("%-44s waves/SIMD %d: %8.3f ms/launch %7.0f TFLOP/s (%5.1f %% of 2516)  s_memtime %.3f Gticks/s  MFMA busy per tick %.3f\n", what, (int)wps,
         ms, tf, tf / 25.16, t / (ms * 1e-3) / 1e9, (double)iters * 16 * 32 * wps / (double)t);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int nfrag = 8 + 16;
  std::vector<unsigned> h(nfrag * 64 * 4);
  u32x4* in; float* out; unsigned long long* ticks;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 8);
  for (int data = 0; data < 2; ++data) {
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    for (auto& w : h) {
      if (!data) { w = 0; continue; }
      // two random bf16 in [-1, 1): sign, exponent 118..126, 7 random mantissa bits
      unsigned v = 0;
      for (int half = 0; half < 2; ++half) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned r = (unsigned)(s >> 33);
        const unsigned bf = ((r & 1) << 15) | ((118 + ((r >> 1) % 9)) << 7) | ((r >> 8) & 0x7f);
        v |= bf << (16 * half);
      }
      w = v;
    }
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("=== operands: %s\n", data ? "random bf16 in [-1, 1)" : "zeros");
    run<0, 0>("registers, no idle", 256, in, out, ticks);
    run<0, 0>("registers, no idle", 512, in, out, ticks);
    if (!data) continue;
    run<1, 0>("registers, s_sleep 1 per 16 MFMAs", 256, in, out, ticks);
    run<2, 0>("registers, s_sleep 2 per 16 MFMAs", 256, in, out, ticks);
    run<4, 0>("registers, s_sleep 4 per 16 MFMAs", 256, in, out, ticks);
    run<8, 0>("registers, s_sleep 8 per 16 MFMAs", 256, in, out, ticks);
    run<4, 0>("registers, s_sleep 4 per 16 MFMAs", 512, in, out, ticks);
    run<8, 0>("registers, s_sleep 8 per 16 MFMAs", 512, in, out, ticks);
    run<0, 1>("LDS fragment per MFMA", 256, in, out, ticks);
    run<0, 1>("LDS fragment per MFMA", 512, in, out, ticks);
    run<0, 2>("LDS fragment per two MFMAs", 256, in, out, ticks);
    run<0, 2>("LDS fragment per two MFMAs", 512, in, out, ticks);
    run<4, 1>("LDS fragment per MFMA, s_sleep 4", 512, in, out, ticks);
  }
  return 0;
}
