// micro-benchmark (round 5): what the HBM gives the ROW patterns of the training step -- the activation dump of the training forward
// (csrc/mf_bf16.hpp dump_store: a lane owns one sample row and stores 16 bytes of it per instruction, 32 contiguous bytes per
// sample and instruction), the gradient rows of the dX chains (same shape) and the operand reads of mf_weight_grads (1 KiB of a
// 9.5 KiB row per sample and item).  No arithmetic: these are the ceilings of the access patterns themselves, against which the
// 2.1-2.2 TB/s of the dumping kernels and the 3-3.7 TB/s of the weight-gradient kernels (profiles/r05_train_*) are read.
//
//   write patterns, a wave owns 32 rows, every "layer" is 1 KiB of the row:
//     rows16    16 B per lane at row[lane & 31] + 32 (lane >> 5)-th half      (the shipped store: C/D order of a 32 x 32 tile)
//     rows128   8 lanes cover 128 B of one row, 8 rows per instruction        (what an LDS transpose of one tile would give)
//     rows1k    64 lanes cover the layer's whole KiB of ONE row per instruction
//     planes16  rows16 into layer-major planes [layer][sample][256] (row pitch 1 KiB)
//     linear    the wave's 32 rows as one contiguous stream (no row structure at all)
//   read patterns, mf_weight_grads<true>'s: per 16-sample stage a thread loads 8 x 8 B (two features of eight rows), a wave
//   covers 512 B of each row; G and X rows, DEPTH stages in flight:
//     rd_rows   row pitch 9 728 B (the dump's), rd_planes  pitch 1 KiB (layer-major planes)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/hbm_patterns.hip -o build/proto/hbm_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kLayers = 9;

template <bool NT> __device__ __forceinline__ void st16(char* p, f32x4 v) {
  if (NT) __builtin_nontemporal_store(v, (f32x4*)p); else *(f32x4*)p = v;
}

// MODE 0 rows16, 1 rows128, 2 rows1k, 3 planes16, 4 linear
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void wr(char* base, long long rows, long long pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, s = lane & 31;
  const f32x4 v = {1.f + lane, 2.f, 3.f + wave, 4.f};
  const long long groups = rows / 32;
  for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
    const long long r0 = g * 32;
    if (MODE == 0) {
      char* row = base + (r0 + s) * pitch + 16 * h;
      for (int l = 0; l < kLayers; ++l)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) st16<NT>(row + l * 1024 + t * 128 + q * 32, v);
    } else if (MODE == 1) {
      char* row = base + (r0 + (lane >> 3)) * pitch + 16 * (lane & 7);
      for (int l = 0; l < kLayers; ++l)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) st16<NT>(row + 8 * i * pitch + l * 1024 + t * 128, v);
    } else if (MODE == 2) {
      char* row = base + r0 * pitch + 16 * lane;
      for (int l = 0; l < kLayers; ++l)
#pragma unroll 8
        for (int r = 0; r < 32; ++r) st16<NT>(row + r * pitch + l * 1024, v);
    } else if (MODE == 3) {
      for (int l = 0; l < kLayers; ++l) {
        char* row = base + ((long long)l * rows + r0 + s) * 1024 + 16 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) st16<NT>(row + t * 128 + q * 32, v);
      }
    } else {
      char* p = base + r0 * pitch + 16 * lane;
      const int n = (int)(32 * pitch / 1024);
#pragma unroll 8
      for (int i = 0; i < n; ++i) st16<NT>(p + i * 1024, v);
    }
  }
}

// mf_weight_grads<true>'s operand reads: items = column blocks of 1 KiB; workgroup b streams a contiguous range of rows of one item
template <int DEPTH>
__global__ __launch_bounds__(512) void rd(const char* gbase, const char* xbase, long long rows, long long pitch, int items, float* out) {
  const int u = threadIdx.x, opnd = u >> 8, o = (u >> 7) & 1, fp = u & 127;
  const int item = blockIdx.x % items;
  const long long per = rows / (gridDim.x / items) / 16 * 16;
  const long long r0 = (long long)(blockIdx.x / items) * per;
  const char* p = (opnd ? xbase : gbase) + (r0 + 8 * o) * pitch + item * 1024 + fp * 8;
  f32x2 buf[DEPTH][8];
  const int stages = (int)(per / 16);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[d][i] = *(const f32x2*)(p + ((long long)d * 16 + i) * pitch);
  float acc = 0.f;
  for (int st = 0; st < stages; st += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += buf[d][i][0] * buf[d][i][1];
      const long long nx = st + d + DEPTH;
      if (nx < stages)
#pragma unroll
        for (int i = 0; i < 8; ++i) buf[d][i] = *(const f32x2*)(p + (nx * 16 + i) * pitch);
      __syncthreads();                                   // the kernel's one barrier per stage
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <class F> static double time_ms(F&& f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const long long rows = 393216, pitch = 9728;           // one MoCo pass of the joint step: 2 048 rays x 192 samples, (D + 1) W + W / 2 floats
  const size_t bytes = (size_t)rows * pitch;
  char *buf, *buf2; float* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&buf2, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, bytes)); CK(hipMemset(buf2, 0, bytes));
  const double wbytes = (double)rows * kLayers * 1024;
  printf("rows %lld  pitch %lld B  written per launch %.2f GB (%d layers x 1 KiB per row)\n", rows, pitch, wbytes / 1e9, kLayers);
  for (int grid : {256, 512, 1024}) {
#define W(MODE, NT, name) { double ms = time_ms([&] { wr<MODE, NT><<<grid, 256>>>(buf, rows, pitch); }); \
    printf("write %-10s %s grid %4d : %7.3f ms  %6.2f TB/s\n", name, NT ? "nt" : "  ", grid, ms, (MODE == 4 ? (double)bytes : wbytes) / ms / 1e9); }
    W(0, false, "rows16") W(0, true, "rows16") W(1, false, "rows128") W(1, true, "rows128") W(2, false, "rows1k") W(2, true, "rows1k")
    W(3, false, "planes16") W(3, true, "planes16") W(4, false, "linear") W(4, true, "linear")
  }
  const int items = 8;
  const double rbytes = 2.0 * rows / 16 * 16 * 1024 * items;
  for (int grid : {256, 512}) {
#define R(DEPTH, P, name) { double ms = time_ms([&] { rd<DEPTH><<<grid, 512>>>(buf, buf2, (P) == 1024 ? rows * 9 : rows, P, (P) == 1024 ? 1 : items, out); }); \
    printf("read  %-10s depth %d grid %4d : %7.3f ms  %6.2f TB/s\n", name, DEPTH, grid, ms, ((P) == 1024 ? 2.0 * rows * 9 * 1024 : rbytes) / ms / 1e9); }
    R(2, pitch, "rd_rows") R(4, pitch, "rd_rows") R(2, 1024LL, "rd_planes") R(4, 1024LL, "rd_planes")
  }
  return 0;
}
