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
  float* means;       // (6) sum / count of each pair as fp32 (0 / 0 = nan, like torch.mean of an empty vector), or null
};

// Sum over the 64 lanes, returned wave-uniform, on the DPP path (row shifts + row broadcasts, both halves of the double moved
// together: ~20 instructions).  The six __shfl_xor steps it replaces are twelve ds_bpermute round trips per value, and both
// kernels reduce TWELVE values (round 4, under rocprofv3: partials kernel 9.9 -> 8.3 us, finish kernel 4.9 -> 4.5 at C5's shard).
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
// (row_bcast:15 / :31 exist on the GFX9 family only -- gfx90a / gfx942 / gfx950; this library is built for gfx950 alone, but the
//  Makefile's ARCH is overridable, so any other target takes the xor-shuffle tree: the sums are float64, the order is free)
__device__ inline double wave_sum_d(double v) {
#if !(defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__)) && defined(__HIP_DEVICE_COMPILE__)
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
  v += dpp_f64<0x111, 0xf>(v);                   // row_shr:1, 2, 4, 8 -- lane 15 of every row holds the row's sum
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);                   // row_bcast:15 -> rows 1, 3
  v += dpp_f64<0x143, 0xc>(v);                   // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's sum
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
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
// wave_sum_d (DPP row shifts / broadcasts) inside a wave, the waves' sums in wave order -- so the result is the same in every run.  (Twelve
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
    if (p.means)
      for (int k = 0; k < 6; ++k) p.means[k] = (float)(p.out[2 * k] / p.out[2 * k + 1]);
  }
}

// ---- backward of the partial sums (the training-mode loss epilogue) ----
// g12 = dL / d out12 (only the six sums carry a gradient).  Seeds, written whole (zeros where the mask is off) into the
// buffers the composite / NoF backward nodes consume:
//   g_rgb[i]        = g12[2q]     * 2 (rgb[i] - target[i])                              MSELoss, models/losses.py:4-14
//   g_recon[p][c]   = g12[4+2q] * m_p * (-sign(x_p[c] - recon_p[c])) / 3                 mean_c |x - recon|, rendering.py:310-314
// with x = o + d z the observation-space point (rendering.py:262-263) and m the consensus mask alpha >= 0.01, all-true
// when empty (:306-308) -- known from the forward's count (count == N S means every point was counted).
struct LossGradPass {
  const float* rgb; float* g_rgb;
  const float* alphas; const float* rays; long long ray_stride; const float* z;
  const float* rl; float* g_rl; const float* rg; float* g_rg;
  long long n_pix, n_samp; int S;
};
struct LossGradParams {
  LossGradPass pass[2];
  const float* target; const double* out12; const double* g12;
};

__global__ __launch_bounds__(kLossThreads) void loss_partials_backward_kernel(LossGradParams p) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nth = (long long)gridDim.x * blockDim.x;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const LossGradPass& P = p.pass[q];
    if (P.g_rgb) {
      const float g = (float)p.g12[2 * q];
      for (long long i = tid; i < P.n_pix; i += nth) P.g_rgb[i] = g * (2.f * (P.rgb[i] - p.target[i]));
    }
    if (P.g_rl || P.g_rg) {
      const float gl = (float)(p.g12[4 + 2 * q] / 3.0), gg = (float)(p.g12[8 + 2 * q] / 3.0);
      const double cnt = P.g_rl ? p.out12[4 + 2 * q + 1] : p.out12[8 + 2 * q + 1];
      const bool all = cnt == (double)P.n_samp;
      for (long long i = tid; i < P.n_samp; i += nth) {
        const long long ray = i / P.S;
        const float* rp = P.rays + ray * P.ray_stride;
        const float z = P.z[i];
        const bool m = all || P.alphas[i] >= 0.01f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float x = rp[c] + rp[3 + c] * z;
          if (P.g_rl) {
            const float d = x - P.rl[i * 3 + c];
            P.g_rl[i * 3 + c] = m ? (d > 0.f ? -gl : (d < 0.f ? gl : 0.f)) : 0.f;
          }
          if (P.g_rg) {
            const float d = x - P.rg[i * 3 + c];
            P.g_rg[i * 3 + c] = m ? (d > 0.f ? -gg : (d < 0.f ? gg : 0.f)) : 0.f;
          }
        }
      }
    }
  }
}

}  // namespace mf

using namespace mf;

extern "C" int32_t mf_loss_partials_backward(const mf_loss_grad_pass* coarse, const mf_loss_grad_pass* fine, const float* target,
                                             int64_t n_rays, const double* out12, const double* g12, void* stream) {
  if (!coarse || !out12 || !g12 || n_rays < 0) return fail(MF_E_INVALID, "mf_loss_partials_backward: null argument");
  if (n_rays == 0) return MF_OK;
  LossGradParams p{};
  const mf_loss_grad_pass* src[2] = {coarse, fine};
  long long work = 0;
  for (int q = 0; q < 2; ++q) {
    if (!src[q]) continue;
    const mf_loss_grad_pass& s = *src[q];
    if (s.g_rgb && (!s.rgb || !target)) return fail(MF_E_INVALID, "mf_loss_partials_backward: g_rgb without rgb / target");
    const bool cons = s.g_recon_local || s.g_recon_global;
    if (cons && (!s.alphas || !s.rays || !s.z_vals || s.n_samples < 1 || s.ray_stride < 6 || (s.g_recon_local && !s.recon_local) ||
                 (s.g_recon_global && !s.recon_global)))
      return fail(MF_E_INVALID, "mf_loss_partials_backward: consensus seeds need alphas, rays, z_vals, n_samples and the reconstructed points");
    LossGradPass& P = p.pass[q];
    P.rgb = s.rgb; P.g_rgb = s.g_rgb; P.alphas = s.alphas; P.rays = s.rays; P.ray_stride = s.ray_stride; P.z = s.z_vals;
    P.rl = s.recon_local; P.g_rl = s.g_recon_local; P.rg = s.recon_global; P.g_rg = s.g_recon_global;
    P.n_pix = s.g_rgb ? n_rays * 3 : 0;
    P.S = s.n_samples;
    P.n_samp = cons ? n_rays * (long long)s.n_samples : 0;
    work += P.n_pix + P.n_samp;
  }
  p.target = target; p.out12 = out12; p.g12 = g12;
  if (work == 0) return MF_OK;
  int blocks = (int)((work + kLossThreads * 4 - 1) / (kLossThreads * 4));
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(loss_partials_backward_kernel, dim3(blocks), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_loss_partials_backward");
}

extern "C" int64_t mf_loss_partials_scratch_bytes(void) { return (int64_t)kLossBlocks * kLossSlots * sizeof(double); }

extern "C" int32_t mf_loss_partials(const mf_loss_pass* coarse, const mf_loss_pass* fine, const float* target, int64_t n_rays,
                                    double* out12, float* means6, void* scratch, void* stream) {
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
  p.means = means6;
  int blocks = (int)((work + kLossThreads * 8 - 1) / (kLossThreads * 8));
  blocks = blocks < 1 ? 1 : (blocks > kLossBlocks ? kLossBlocks : blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(loss_partials_kernel, dim3(blocks), dim3(kLossThreads), 0, st, p);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kLossBlocks), 0, st, p, blocks);
  return check_launch("mf_loss_partials");
}
