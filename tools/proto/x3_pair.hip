// micro-benchmark (round 5, "what comes next"): the three-product hidden-layer tile loop with ONE wave per SIMD (the shipped
// organisation of csrc/mf_bf16.hpp's mma_tile_x: four waves, each 32 samples x the whole K range, (hi, lo) operands of 256 features
// in 128 registers) against the K range split over a PAIR of waves on the same SIMD (eight waves, each 32 samples x half the K
// range, 64 operand registers; the partner's partial sums of a tile cross through LDS, the tile's owner adds them, runs the
// epilogue and stores the rows) -- with and without the four 16-byte row stores per tile and lane that the training forward / the
// dX chains issue (HBM write-back behind them, the VM counter shared with the LDS-DMA pieces: DESIGN.md, "the training step's
// memory-bound kernels").  Everything else as in the kernels: 32 groups of 1 KiB per weight panel (16 k-steps x (hi, lo)), three
// MFMAs per k-step, a 3-slot LDS ring fed by LDS-DMA two panels ahead, a fragment ring of depth 3, one barrier per panel,
// ~80 VALU of epilogue per tile in the MFMA gaps, random bf16 operands.  Timing only.
//   MODE 0  one wave per SIMD, no row stores        MODE 1  one wave per SIMD, row stores (vmcnt(4) at the panel barrier)
//   MODE 2  wave pairs, no row stores               MODE 3  wave pairs, row stores
//   MODE 4 / 5  wave pairs + a NINTH wave that issues every LDS-DMA piece and alone waits for them: the compute waves never wait on
//               the VM counter, their row stores stay in flight as long as they take (without / with row stores)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/x3_pair.hip -o build/proto/x3_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ void blds(const char* base, uint32_t lane16, uint32_t soff, uint32_t lds_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, (int)lane16, (int)soff, 0, 0);
}
__device__ __forceinline__ u32x4 lds4(uint32_t off) { return *(const u32x4*)(smem + off); }
constexpr int PD = 3, NGP = 32, SLOT = NGP * 1024;          // groups per panel: 16 k-steps x (hi, lo)
constexpr uint32_t kXch = 3 * SLOT;                          // exchange area of the pair modes: 2 parities x 4 pairs x 4 KiB

// KS: k-steps this wave multiplies per panel (16: the whole K range; 8: its half).  A wave's groups of a panel: g0 .. g0 + 2 KS - 1.
// CLUSTER: the four row stores of a tile back to back in its last gap instead of one per gap
// NOMF: no MFMAs (the fragments are still read); NODMA: no LDS-DMA pieces (the ring keeps its first contents)
template <int MODE, bool CLUSTER = false, bool NOMF = false, bool NODMA = false>
__global__ __launch_bounds__((MODE >= 4 ? 576 : MODE >= 2 ? 512 : 256), 1) void k(const u32x4* in, const char* w, float* out, char* rows, long long row_bytes,
                                                                   long long rows_total, int panels, int wbytes) {
  constexpr bool PAIR = MODE >= 2, STORES = MODE & 1, DMAW = MODE >= 4;
  constexpr int NW = PAIR ? 8 : 4, KS = PAIR ? 8 : 16, NG = 2 * KS, NPIECE = NGP / NW;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = PAIR ? wave >> 2 : 0, pair = wave & 3;
  u32x4 bh[KS], bl[KS];
  for (int i = 0; i < KS; ++i) { bh[i] = in[(lane + 64 * i + 17 * wave) & 1023]; bl[i] = in[(lane + 64 * i + 29 * wave + 512) & 1023]; }
  if (wave < NW) for (int g = wave; g < NGP; g += NW) { blds(w, lane * 16, g * 1024, g * 1024); blds(w, lane * 16, SLOT + g * 1024, SLOT + g * 1024); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint32_t off0 = 0, off1 = SLOT, off2 = 2 * SLOT, gsrc = 2 * SLOT;
  if (DMAW && wave == NW) {                                     // the DMA wave: per panel, wait for its pieces of the next panel, barrier, the panel two ahead
    for (int pnl = 0; pnl < panels; ++pnl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* src = w + gsrc;
#pragma unroll
      for (int g = 0; g < NGP; ++g) blds(src, lane * 16, g * 1024, off2 + g * 1024);
      const uint32_t t = off0; off0 = off1; off1 = off2; off2 = t;
      gsrc += SLOT; if (gsrc + SLOT > (uint32_t)wbytes) gsrc = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const uint32_t g0 = half * NG * 1024 + lane * 16;            // this wave's first group of a panel
  // row stores: a lane owns a sample row (lane & 31), 16 bytes per store at 32 q + 16 (lane >> 5) of the tile's 128-byte row piece
  const long long rows_per_wg = rows_total / gridDim.x / 128 * 128;
  char* rbase = rows + ((long long)blockIdx.x * rows_per_wg + pair * 32 + (lane & 31)) * row_bytes + 16 * (lane >> 5);
  long long roff = 0;                                           // advances by one 128-byte piece per tile, by 128 rows per layer of 72 tiles
  f32x16 acc, pend;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; pend[i] = 0.f; }
  float sink = 0.f;
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = lds4(off0 + g0 + i * 1024);
  int tcount = 0;
  for (int pnl = 0; pnl < panels; ++pnl) {
    const uint32_t p = off0 + g0, pn = off1 + g0;
    const char* src = w + gsrc;
    const uint32_t dst = off2;
    const bool owner = !PAIR || ((pnl >> 2) & 1) == half;       // the owner of a tile alternates every four tiles (the feature halves of a layer)
    const uint32_t xo = kXch + (uint32_t)(pnl & 1) * 16384u + (uint32_t)pair * 4096u + (uint32_t)lane * 16u;
    int m = 0;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int s = gi % (PD + 1), ks = gi >> 1;
      if (!NOMF) acc = MFMA(r[s], bh[ks], acc); else acc[gi & 15] += __builtin_bit_cast(float, r[s][0] ^ bh[ks][1]);
      __builtin_amdgcn_sched_barrier(0);
      const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
      if (nb < NG) r[sp] = lds4(p + nb * 1024);
      if (gi == 0) {
        // the panel barrier: this wave's pieces of the next panel have landed; STORES: the four row stores behind them stay in flight
        const bool stored = STORES && (!PAIR || (pnl > 0 && (((pnl - 1) >> 2) & 1) == half)) && pnl > 1;   // (this wave closed the last panel with four stores)
        if (!DMAW) { if (stored) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (PAIR) {                                             // the partner's partial sums of the tile this wave owned LAST panel
          const bool owned_prev = pnl > 0 && (((pnl - 1) >> 2) & 1) == half;
          if (owned_prev) {
            const uint32_t xp = kXch + (uint32_t)((pnl - 1) & 1) * 16384u + (uint32_t)pair * 4096u + (uint32_t)lane * 16u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const u32x4 v = lds4(xp + q * 1024);
#pragma unroll
              for (int i = 0; i < 4; ++i) pend[4 * q + i] += __builtin_bit_cast(float, v[i]);
            }
          }
        }
      }
      if (nb >= NG) r[sp] = lds4(pn + (nb - NG) * 1024);
      // the pending tile's epilogue in the gaps (owner only): ~5 VALU per gap over 16 gaps, the four row stores in the last four
      auto gap = [&](int mm) {
        const bool owned_prev = !PAIR || (pnl > 0 && (((pnl - 1) >> 2) & 1) == half);
        if (!owned_prev) return;
        constexpr int NM = 3 * KS;
        const int step0 = 16 * mm / NM, step1 = 16 * (mm + 1) / NM;
        for (int st = step0; st < step1; ++st) {
          float v0 = pend[st], v1 = pend[(st + 5) & 15];
          v0 = __builtin_fmaxf(v0, 0.f); v1 = __builtin_fmaxf(v1, 0.f);
          const float d = v0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & 0xffff0000u);
          sink += d * v1;
        }
        if (STORES && !CLUSTER && mm >= NM - 4) {
          const int q = mm - (NM - 4);
          f32x4 v = {pend[4 * q], pend[4 * q + 1], pend[4 * q + 2], pend[4 * q + 3]};
          *(f32x4*)(rbase + roff + 32 * q) = v;
        }
        if (STORES && CLUSTER && mm == NM - 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {pend[4 * q], pend[4 * q + 1], pend[4 * q + 2], pend[4 * q + 3]};
            *(f32x4*)(rbase + roff + 32 * q) = v;
          }
        }
      };
      gap(m++);
      __builtin_amdgcn_sched_barrier(0);
      if (!(gi & 1)) {                                          // hi group: its second product, the LDS-DMA pieces behind it
        if (!NOMF) acc = MFMA(r[s], bl[ks], acc);
        __builtin_amdgcn_sched_barrier(0);
        const int j = gi >> 1;                                  // slot index 0 .. KS-1
        if (!DMAW && !NODMA && j * NPIECE / KS != (j + 1) * NPIECE / KS) {
          const int kpc = j * NPIECE / KS;
          blds(src, lane * 16, (NPIECE * wave + kpc) * 1024, dst + (NPIECE * wave + kpc) * 1024);
        }
        gap(m++);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // tile done: the owner keeps its accumulators as the pending tile, the partner hands its partial sums over
    if (PAIR && !owner) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(unsigned, acc[4 * q + i]);
        *(u32x4*)(smem + xo + q * 1024) = v;
      }
    } else {
      pend = acc;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // the row piece of the next tile: 72 tiles of 128 bytes fill a 9 216-byte row, then the next 128 rows
    ++tcount;                                                   // (both waves of a pair: the tiles they own interleave in the same rows)
    roff += 128;
    if (tcount % 72 == 0) { roff += 128 * row_bytes - 72 * 128; if (roff + 128 * row_bytes > rows_per_wg * row_bytes) roff = 0; }
    (void)owner;
    const uint32_t t = off0; off0 = off1; off1 = off2; off2 = t;
    gsrc += SLOT; if (gsrc + SLOT > (uint32_t)wbytes) gsrc = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = sink;
  for (int i = 0; i < 16; ++i) s += pend[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool CLUSTER = false, bool NOMF = false, bool NODMA = false>
void run(const char* what, const u32x4* in, const char* w, float* out, char* rows, long long row_bytes, long long rows_total, int wbytes) {
  const int panels = 20000, launches = 6;
  constexpr int threads = MODE >= 4 ? 576 : MODE >= 2 ? 512 : 256;
  const int lds = 3 * SLOT + 32768;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)(k<MODE, CLUSTER, NOMF, NODMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, CLUSTER, NOMF, NODMA>), dim3(256), dim3(threads), lds, 0, in, w, out, rows, row_bytes, rows_total, panels, wbytes);
  (void)hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k<MODE, CLUSTER, NOMF, NODMA>), dim3(256), dim3(threads), lds, 0, in, w, out, rows, row_bytes, rows_total, panels, wbytes);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= launches;
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  const double tf = (double)panels * 48 * 4 * 256 * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;   // 48 MFMAs x 4 column blocks per panel and workgroup
  const double gb = (MODE & 1) ? (double)panels * 256 * 128 * 128 / 1e9 : 0.0;                // 128 rows x 128 bytes per tile
  printf("mode %d  %-44s %8.3f ms  %6.0f TFLOP/s (%4.1f %% of 2516)  ns/panel %6.1f   row stores %.2f GB = %.2f TB/s\n", MODE, what, ms, tf, tf / 25.16,
         ms * 1e6 / panels, gb, gb / ms);
  fflush(stdout);
}

int main() {
  const int wbytes = 2048 * 1024;
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
  char* w; u32x4* in; float* out; char* rows;
  const long long row_bytes = 9728, rows_total = 256LL * 128 * 12;       // 3.8 GB of dump rows: 1 536 rows per workgroup
  (void)hipMalloc(&w, wbytes); (void)hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
  (void)hipMalloc(&in, 1024 * 16); (void)hipMemcpy(in, h.data(), 1024 * 16, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 576 * 4);
  if (hipMalloc(&rows, rows_total * row_bytes) != hipSuccess) { printf("no memory for the rows\n"); return 1; }
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("one wave per SIMD, no row stores", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<1>("one wave per SIMD, four row stores per tile", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<2>("wave pairs (K split), no row stores", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<3>("wave pairs (K split), four row stores per tile", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<4>("wave pairs + DMA wave, no row stores", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<5>("wave pairs + DMA wave, four row stores per tile", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<1, true>("one wave per SIMD, the four stores back to back", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<5, true>("wave pairs + DMA wave, stores back to back", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<1, false, true>("one wave per SIMD, stores, NO MFMAs", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<1, false, false, true>("one wave per SIMD, stores, NO LDS-DMA", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<1, false, true, true>("one wave per SIMD, stores, neither", in, w, out, rows, row_bytes, rows_total, wbytes);
    run<0, false, true>("one wave per SIMD, no stores, NO MFMAs", in, w, out, rows, row_bytes, rows_total, wbytes);
  }
  return 0;
}
