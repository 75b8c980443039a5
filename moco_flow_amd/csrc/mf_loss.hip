// mf_loss.hip -- the additive pieces of the training losses of one render_rays call, in two small launches instead
// of ~20 elementwise / reduction launches, a mask compaction and a host sync:
//   MSELoss over both passes                       models/losses.py:4-14
//   consensus means over the alpha >= 0.01 mask    models/rendering.py:306-314 + trainer/trainer_moco_flow.py:317-328
// Output = 12 doubles, one (sum, count) pair per term the reference averages separately (the layout of
// moco_flow_amd/dist.py::loss_partials): [mse_c | mse_f | local_c | local_f | global_c | global_f].
// The mean-only caller divides; the multi-GPU caller all-reduces the 96 bytes first.  Deterministic: fixed-order
// partial sums through `scratch`, no atomics.
#include "mf_host.hpp"

namespace mf {

constexpr int kLossBlocks = 256;
constexpr int kLossThreads = 256;
constexpr int kLossSlots = 12;       // per pass: sq, cnt_masked, loc_masked, glob_masked, loc_all, glob_all

struct LossPass {
  const float* rgb; const float* alphas; const float* dl; const float* dg;
  long long n_pix;    // N * 3 (0 if rgb missing)
  long long n_samp;   // N * S (0 if no consensus planes)
};
struct LossParams {
  LossPass pass[2];
  const float* target;
  double* scratch;    // (kLossBlocks, kLossSlots)
  double* out;        // (12)
};

__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__global__ __launch_bounds__(kLossThreads) void loss_partials_kernel(LossParams p) {
  __shared__ double red[kLossThreads / 64][kLossSlots];
  double acc[kLossSlots];
#pragma unroll
  for (int k = 0; k < kLossSlots; ++k) acc[k] = 0.0;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nth = (long long)gridDim.x * blockDim.x;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const LossPass& P = p.pass[q];
    for (long long i = tid; i < P.n_pix; i += nth) {
      const double d = (double)P.rgb[i] - (double)p.target[i];
      acc[6 * q + 0] += d * d;
    }
    for (long long i = tid; i < P.n_samp; i += nth) {
      const bool m = P.alphas[i] >= 0.01f;                       // rendering.py:306
      const double l = P.dl ? (double)P.dl[i] : 0.0, g = P.dg ? (double)P.dg[i] : 0.0;
      acc[6 * q + 1] += m ? 1.0 : 0.0;
      acc[6 * q + 2] += m ? l : 0.0;
      acc[6 * q + 3] += m ? g : 0.0;
      acc[6 * q + 4] += l;
      acc[6 * q + 5] += g;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kLossSlots; ++k) {
    const double s = wave_sum_d(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < kLossSlots) {
    double s = 0.0;
    for (int w = 0; w < kLossThreads / 64; ++w) s += red[w][threadIdx.x];
    p.scratch[(long long)blockIdx.x * kLossSlots + threadIdx.x] = s;
  }
}

// One workgroup of kLossBlocks threads: thread b holds block b's 12 partials (independent loads), then a fixed tree --
// xor-shuffles inside a wave, the waves' sums in wave order -- so the result is the same in every run.  (Twelve
// threads walking the blocks one dependent load at a time took 24 us, longer than the partials kernel itself.)
__global__ __launch_bounds__(kLossBlocks) void loss_finish_kernel(LossParams p, int n_blocks) {
  __shared__ double tot[kLossSlots];
  __shared__ double red[kLossBlocks / 64][kLossSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[kLossSlots];
#pragma unroll
  for (int k = 0; k < kLossSlots; ++k) v[k] = (int)threadIdx.x < n_blocks ? p.scratch[(long long)threadIdx.x * kLossSlots + k] : 0.0;
#pragma unroll
  for (int k = 0; k < kLossSlots; ++k) {
    const double s = wave_sum_d(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < kLossSlots) {
    double s = 0.0;
    for (int w = 0; w < kLossBlocks / 64; ++w) s += red[w][threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 0; q < 2; ++q) {
      const LossPass& P = p.pass[q];
      p.out[2 * q] = tot[6 * q];
      p.out[2 * q + 1] = (double)P.n_pix;
      // mask with no element set -> all elements (rendering.py:307-308)
      const bool none = tot[6 * q + 1] == 0.0;
      const double cnt = none ? (double)P.n_samp : tot[6 * q + 1];
      p.out[4 + 2 * q] = P.dl ? (none ? tot[6 * q + 4] : tot[6 * q + 2]) : 0.0;
      p.out[4 + 2 * q + 1] = P.dl ? cnt : 0.0;
      p.out[8 + 2 * q] = P.dg ? (none ? tot[6 * q + 5] : tot[6 * q + 3]) : 0.0;
      p.out[8 + 2 * q + 1] = P.dg ? cnt : 0.0;
    }
  }
}

}  // namespace mf

using namespace mf;

extern "C" int64_t mf_loss_partials_scratch_bytes(void) { return (int64_t)kLossBlocks * kLossSlots * sizeof(double); }

extern "C" int32_t mf_loss_partials(const mf_loss_pass* coarse, const mf_loss_pass* fine, const float* target, int64_t n_rays,
                                    double* out12, void* scratch, void* stream) {
  if (!coarse || !out12 || !scratch || n_rays < 0) return fail(MF_E_INVALID, "mf_loss_partials: null argument");
  LossParams p{};
  const mf_loss_pass* src[2] = {coarse, fine};
  long long work = 0;
  for (int q = 0; q < 2; ++q) {
    if (!src[q]) continue;
    const mf_loss_pass& s = *src[q];
    if (s.rgb && !target) return fail(MF_E_INVALID, "mf_loss_partials: rgb without a target");
    if ((s.disp_local || s.disp_global) && (!s.alphas || s.n_samples < 1))
      return fail(MF_E_INVALID, "mf_loss_partials: consensus planes need alphas and n_samples");
    p.pass[q].rgb = s.rgb; p.pass[q].alphas = s.alphas; p.pass[q].dl = s.disp_local; p.pass[q].dg = s.disp_global;
    p.pass[q].n_pix = s.rgb ? n_rays * 3 : 0;
    p.pass[q].n_samp = (s.disp_local || s.disp_global) ? n_rays * (long long)s.n_samples : 0;
    work += p.pass[q].n_pix + p.pass[q].n_samp;
  }
  p.target = target;
  p.scratch = static_cast<double*>(scratch);
  p.out = out12;
  int blocks = (int)((work + kLossThreads * 8 - 1) / (kLossThreads * 8));
  blocks = blocks < 1 ? 1 : (blocks > kLossBlocks ? kLossBlocks : blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(loss_partials_kernel, dim3(blocks), dim3(kLossThreads), 0, st, p);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kLossBlocks), 0, st, p, blocks);
  return check_launch("mf_loss_partials");
}
