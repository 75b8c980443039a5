// micro-benchmark (round 5): what an LDS-DMA piece costs INSIDE the fast bf16 tile loop, and who should issue it.
// The hidden-layer panel loop of csrc/mf_bf16.hpp in isolation: 8 waves (two per SIMD), per panel and wave 16 x
// v_mfma_f32_32x32x16_bf16 on one accumulator chain, one fresh 1 KiB A fragment per MFMA through a register ring (prefetch
// distance 3) out of a 3-slot LDS ring of 16 KiB panels that LDS-DMA fills two panels ahead, one workgroup barrier per
// panel, random bf16 operands (the clock limiter sees real data).  MODE = who issues the 16 pieces of a panel, and where:
//   0  every wave 2 pieces, in the gaps behind MFMA 1 and 2 (the shipped schedule)
//   1  no DMA at all (the ring keeps its first contents)                   -> the price of the pieces
//   2  waves 0-3 only, 4 pieces each, gaps 1-4
//   3  ONE wave per panel (panel & 7), all 16 pieces, one per gap          -> its SIMD partner covers the issuing wave
//   4  every wave 2 pieces as a burst straight behind the barrier (gap 0)
//   5  every wave 2 pieces late in the panel (gaps 12, 13)
//   6  every wave 2 pieces, waves 4-7 six gaps later than waves 0-3
//   8  no DMA, no barrier
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/dma_mix.hip -o build/proto/dma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ void blds(const char* base, uint32_t lane16, uint32_t soff, uint32_t lds_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, (int)lane16, (int)soff, 0, 0);
}
__device__ __forceinline__ u32x4 lds4(uint32_t off) { return *(const u32x4*)(smem + off); }
constexpr int PD = 3, NG = 16, SLOT = NG * 1024;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const u32x4* in, const char* w, float* out, int panels, int wbytes) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4 b[16];
  for (int i = 0; i < 16; ++i) b[i] = in[(lane + 64 * i + 17 * wave) & 1023];
  // prime slots 0 and 1
  for (int g = wave; g < NG; g += 8) { blds(w, lane * 16, g * 1024, g * 1024); blds(w, lane * 16, SLOT + g * 1024, SLOT + g * 1024); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint32_t off0 = 0, off1 = SLOT, off2 = 2 * SLOT;
  uint32_t gsrc = 2 * SLOT;                       // byte offset of the panel two ahead in the weight buffer
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = lds4(off0 + lane * 16 + i * 1024);
  for (int pnl = 0; pnl < panels; ++pnl) {
    const uint32_t p = off0 + lane * 16, pn = off1 + lane * 16;
    const char* src = w + gsrc;
    const uint32_t dst = off2;
    const bool issuer3 = (pnl & 7) == wave;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int s = gi % (PD + 1);
      acc = MFMA(r[s], b[gi], acc);
      __builtin_amdgcn_sched_barrier(0);
      const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
      if (nb < NG) r[sp] = lds4(p + nb * 1024);
      if (gi == 0 && MODE != 8) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 4) { blds(src, lane * 16, (2 * wave) * 1024, dst + (2 * wave) * 1024); blds(src, lane * 16, (2 * wave + 1) * 1024, dst + (2 * wave + 1) * 1024); }
      }
      if (MODE == 0 && (gi == 1 || gi == 2)) blds(src, lane * 16, (2 * wave + gi - 1) * 1024, dst + (2 * wave + gi - 1) * 1024);
      if (MODE == 2 && gi >= 1 && gi <= 4 && wave < 4) blds(src, lane * 16, (4 * wave + gi - 1) * 1024, dst + (4 * wave + gi - 1) * 1024);
      if (MODE == 3 && issuer3) blds(src, lane * 16, gi * 1024, dst + gi * 1024);
      if (MODE == 5 && (gi == 12 || gi == 13)) blds(src, lane * 16, (2 * wave + gi - 12) * 1024, dst + (2 * wave + gi - 12) * 1024);
      if (MODE == 6) {
        if ((gi == 1 || gi == 2) && wave < 4) blds(src, lane * 16, (2 * wave + gi - 1) * 1024, dst + (2 * wave + gi - 1) * 1024);
        if ((gi == 7 || gi == 8) && wave >= 4) blds(src, lane * 16, (2 * wave + gi - 7) * 1024, dst + (2 * wave + gi - 7) * 1024);
      }
      if (nb >= NG) r[sp] = lds4(pn + (nb - NG) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the kernel's epilogue would go here; the accumulator chain just runs on)
    const uint32_t t = off0; off0 = off1; off1 = off2; off2 = t;
    gsrc += SLOT; if (gsrc + SLOT > (uint32_t)wbytes) gsrc = 0;
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// round 5, "what comes next": FOUR waves (one per SIMD), each 64 samples = two 32-sample column blocks: every 1 KiB fragment feeds two
// MFMAs (two accumulator chains), half the LDS fragment traffic per MFMA; 4 pieces per wave and panel, same ring, same barrier.
__global__ __launch_bounds__(256, 1) void k2(const u32x4* in, const char* w, float* out, int panels, int wbytes) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4 b0[16], b1[16];
  for (int i = 0; i < 16; ++i) { b0[i] = in[(lane + 64 * i + 17 * wave) & 1023]; b1[i] = in[(lane + 64 * i + 31 * wave + 300) & 1023]; }
  for (int g = wave; g < NG; g += 4) { blds(w, lane * 16, g * 1024, g * 1024); blds(w, lane * 16, SLOT + g * 1024, SLOT + g * 1024); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint32_t off0 = 0, off1 = SLOT, off2 = 2 * SLOT, gsrc = 2 * SLOT;
  f32x16 a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = lds4(off0 + lane * 16 + i * 1024);
  for (int pnl = 0; pnl < panels; ++pnl) {
    const uint32_t p = off0 + lane * 16, pn = off1 + lane * 16;
    const char* src = w + gsrc;
    const uint32_t dst = off2;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int s = gi % (PD + 1);
      a0 = MFMA(r[s], b0[gi], a0);
      __builtin_amdgcn_sched_barrier(0);
      const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
      if (nb < NG) r[sp] = lds4(p + nb * 1024);
      if (gi == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (nb >= NG) r[sp] = lds4(pn + (nb - NG) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      a1 = MFMA(r[s], b1[gi], a1);
      __builtin_amdgcn_sched_barrier(0);
      if (gi >= 1 && gi <= 4) blds(src, lane * 16, (4 * wave + gi - 1) * 1024, dst + (4 * wave + gi - 1) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    const uint32_t t = off0; off0 = off1; off1 = off2; off2 = t;
    gsrc += SLOT; if (gsrc + SLOT > (uint32_t)wbytes) gsrc = 0;
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void run2(const u32x4* in, const char* w, float* out, int wbytes) {
  const int panels = 20000, launches = 12;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SLOT);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k2, dim3(256), dim3(256), 3 * SLOT, 0, in, w, out, panels, wbytes);
  (void)hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k2, dim3(256), dim3(256), 3 * SLOT, 0, in, w, out, panels, wbytes);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= launches;
  const double tf = (double)panels * NG * 8 * 256 * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;     // same tile: 4 waves x 2 blocks = 8 column blocks
  printf("four waves x two column blocks per fragment (4 pieces per wave, gaps 1-4)          %8.3f ms  %6.0f TFLOP/s (%4.1f %% of 2516)  ns/panel %.1f\n", ms, tf, tf / 25.16,
         ms * 1e6 / panels);
  fflush(stdout);
}

template <int MODE>
void run(const char* what, const u32x4* in, const char* w, float* out, int wbytes) {
  const int panels = 20000, launches = 12;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SLOT);
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 3 * SLOT, 0, in, w, out, panels, wbytes);
  (void)hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 3 * SLOT, 0, in, w, out, panels, wbytes);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= launches;
  const double tf = (double)panels * NG * 8 * 256 * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  printf("mode %d  %-62s %8.3f ms  %6.0f TFLOP/s (%4.1f %% of 2516)  ns/panel %.1f (16 MFMAs x 2 waves = 427 ns @ 2.4 GHz)\n", MODE, what, ms, tf, tf / 25.16,
         ms * 1e6 / panels);
  fflush(stdout);
}

int main() {
  const int wbytes = 1536 * 1024;                 // one network's worth of packed weights, L2-resident
  std::vector<unsigned> h(wbytes / 4);
  unsigned long long s = 0x9e3779b97f4a7c15ull;
  for (auto& v : h) {
    unsigned x = 0;
    for (int half = 0; half < 2; ++half) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned r = (unsigned)(s >> 33);
      x |= (((r & 1) << 15) | ((118 + ((r >> 1) % 9)) << 7) | ((r >> 8) & 0x7f)) << (16 * half);
    }
    v = x;
  }
  char* w; u32x4* in; float* out;
  (void)hipMalloc(&w, wbytes); (void)hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
  (void)hipMalloc(&in, 1024 * 16); (void)hipMemcpy(in, h.data(), 1024 * 16, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("every wave 2 pieces, gaps 1-2 (shipped)", in, w, out, wbytes);
    run<1>("no DMA", in, w, out, wbytes);
    run<2>("waves 0-3 only, 4 pieces each", in, w, out, wbytes);
    run<3>("one wave per panel issues all 16", in, w, out, wbytes);
    run<4>("every wave 2 pieces, burst behind the barrier", in, w, out, wbytes);
    run<5>("every wave 2 pieces, gaps 12-13", in, w, out, wbytes);
    run<6>("waves 4-7 six gaps behind waves 0-3", in, w, out, wbytes);
    run<8>("no DMA, no barrier", in, w, out, wbytes);
    run2(in, w, out, wbytes);
  }
  return 0;
}
