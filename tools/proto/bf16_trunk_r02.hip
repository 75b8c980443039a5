// timing-only prototype: bf16 32x32x16 trunk, weights through a 3-slot LDS-DMA ring, NG sample groups per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define D __device__ __forceinline__
D void glds16(const char* g, uint32_t off) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + off), 16, 0, 0);
}
D u32x4 lds_u4(uint32_t off) { return *(const u32x4*)(smem + off); }
D unsigned pack2(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
D unsigned relu2(unsigned x) { i16x2 v = __builtin_bit_cast(i16x2, x); i16x2 z = {0, 0}; return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, z)); }
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
constexpr int KH = 16, PANEL = 16 * 1024, PD = 3;
template <int NG, int WAVES, int FL>
__global__ __launch_bounds__(WAVES * 64) void proto(const char* w, long long wbytes, float* out, int layers, int reps) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4 act[NG][KH], nxt[NG][KH];
  for (int g = 0; g < NG; ++g) for (int k = 0; k < KH; ++k) act[g][k] = u32x4{(unsigned)lane * 7u + k, 0x3f803f80u, (unsigned)g, 0x3f003f00u};
  const char* gnext = w + (blockIdx.x % 4) * PANEL;
  uint32_t off0 = 0, off1 = PANEL, off2 = 2 * PANEL;
  for (int grp = wave; grp < 16; grp += WAVES) { glds16(gnext + grp * 1024 + lane * 16, off0 + grp * 1024); glds16(gnext + PANEL + grp * 1024 + lane * 16, off1 + grp * 1024); }
  gnext += 2 * PANEL;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool late = (WAVES == 8) && wave < 4;
  u32x4 stg[16 / WAVES];
  uint32_t stg_dst = 0;
  bool stg_live = false;
  for (int r = 0; r < reps; ++r)
    for (int l = 0; l < layers; ++l) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t p = off0 + lane * 16, pn = off1 + lane * 16;
        auto hook = [&]() {
          if ((FL & 32) && stg_live) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < ((FL & 64) ? 1 : 16 / WAVES); ++q) *(u32x4*)(smem + stg_dst + q * WAVES * 1024) = stg[q];
            stg_live = false;
          }
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          if (!(FL & 1)) __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (FL & 64) {
            stg[0] = *(const u32x4*)(gnext + wave * 1024 + lane * 16);
            stg[1] = stg[0];
            glds16(gnext + (wave + WAVES) * 1024 + lane * 16, off2 + (wave + WAVES) * 1024);
            stg_dst = off2 + wave * 1024 + lane * 16;
            stg_live = true;
          } else
          if (FL & 16) {
#pragma unroll
            for (int q = 0; q < 16 / WAVES; ++q) stg[q] = *(const u32x4*)(gnext + (wave + q * WAVES) * 1024 + lane * 16);
            stg_dst = off2 + wave * 1024 + lane * 16;
            stg_live = true;
          } else
          if (!(FL & 2)) for (int grp = wave; grp < 16; grp += WAVES) glds16(gnext + grp * 1024 + lane * 16, off2 + grp * 1024);
          gnext += PANEL;
          if (gnext + PANEL > w + wbytes) gnext = w;
        };
        f32x16 acc[NG];
        for (int g = 0; g < NG; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
        u32x4 fr[PD];
#pragma unroll
        for (int k = 0; k < PD; ++k) fr[k] = lds_u4(p + k * 1024);
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          if (k == (late ? 8 : 0)) hook();
          if ((FL & 16) && !(FL & 32) && k == (late ? 2 : 10) && stg_live) {      // the loads had ~10 MFMAs to land: park them in LDS
#pragma unroll
            for (int q = 0; q < ((FL & 64) ? 1 : 16 / WAVES); ++q) *(u32x4*)(smem + stg_dst + q * WAVES * 1024) = stg[q];
            stg_live = false;
          }
          const u32x4 a = fr[k % PD];
          if (!(FL & 4) && k + PD < KH) fr[k % PD] = lds_u4(p + (k + PD) * 1024);
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[g] = MFMA(a, act[g][k], acc[g]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (FL & 8) { for (int q = 0; q < 4; ++q) { nxt[g][2 * t][q] = __builtin_bit_cast(unsigned, acc[g][q]); nxt[g][2 * t + 1][q] = __builtin_bit_cast(unsigned, acc[g][8 + q]); } } else
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            nxt[g][2 * t][q] = relu2(pack2(acc[g][2 * q], acc[g][2 * q + 1]));
            nxt[g][2 * t + 1][q] = relu2(pack2(acc[g][8 + 2 * q], acc[g][8 + 2 * q + 1]));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(nxt[g][2 * t][q]), "+v"(nxt[g][2 * t + 1][q]));
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t tmp = off0; off0 = off1; off1 = off2; off2 = tmp;
      }
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < KH; ++k) act[g][k] = nxt[g][k];
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned s = 0;
  for (int g = 0; g < NG; ++g) for (int k = 0; k < KH; ++k) s += act[g][k][0] ^ act[g][k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
template <int NG, int WAVES, int FL>
static void run(const char* name, int gridmul, const char* w, long long wbytes, float* out, int layers, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(proto<NG, WAVES, FL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PANEL);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * gridmul;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL((proto<NG, WAVES, FL>), dim3(grid), dim3(WAVES * 64), 3 * PANEL, 0, w, wbytes, out, layers, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double samples = (double)grid * (WAVES * 32 * NG) * reps, flops = samples * layers * 256.0 * 256.0 * 2.0;
    if (it == 2) printf("%-26s: %.3f ms  %.1f TFLOP/s (%.1f %% of 2516)  err=%d\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.16, (int)hipGetLastError());
  }
}
int main() {
  const long long wbytes = 64ll << 20;
  char* w; float* out;
  hipMalloc(&w, wbytes); hipMemset(w, 0x3c, wbytes); hipMalloc(&out, 256 * 512 * 4);
  run<1, 8, 0>("8w x 32 base", 1, w, wbytes, out, 8, 16);
  run<1, 8, 1>("8w x 32 no barrier", 1, w, wbytes, out, 8, 16);
  run<1, 8, 2>("8w x 32 no DMA", 1, w, wbytes, out, 8, 16);
  run<1, 8, 4>("8w x 32 no frag reads", 1, w, wbytes, out, 8, 16);
  run<1, 8, 8>("8w x 32 no epilogue", 1, w, wbytes, out, 8, 16);
  run<1, 8, 15>("8w x 32 MFMA only", 1, w, wbytes, out, 8, 16);
  run<1, 8, 16>("8w x 32 load+ds_write", 1, w, wbytes, out, 8, 16);
  run<1, 8, 17>("8w x 32 ld+dsw no bar", 1, w, wbytes, out, 8, 16);
  run<1, 8, 48>("8w x 32 ld+dsw at hook", 1, w, wbytes, out, 8, 16);
  run<1, 8, 80>("8w x 32 half VGPR half DMA", 1, w, wbytes, out, 8, 16);
  run<1, 8, 112>("8w x 32 half/half at hook", 1, w, wbytes, out, 8, 16);
  run<1, 4, 0>("2 WG/CU of 4w x 32", 2, w, wbytes, out, 8, 16);
  run<2, 4, 0>("4w x 64", 1, w, wbytes, out, 8, 16);
  run<2, 4, 15>("4w x 64 MFMA only", 1, w, wbytes, out, 8, 16);
  return 0;
}
