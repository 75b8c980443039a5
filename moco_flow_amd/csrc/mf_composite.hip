// mf_composite.hip -- backward of the alpha-composite (models/rendering.py:157-192) on per-sample planes.
//
// Replaces the ~40-node autograd graph torch records for nerf_inference's compositing under
// loss.backward() (deltas, activation, exp, cumprod, weighted sums; trainer/base.py:188-197).  One wave
// per ray, lanes over samples, two sweeps through LDS:
//   forward sweep  (ascending chunks of 64): alpha_i, T_i (wave product scan + carry), w_i = alpha_i T_i and
//                  g_w_i = g_rgb.(c_i - bg) + g_depth z_i + g_opacity;
//   backward sweep (descending): suffix_i = sum_{k>i} g_w_k w_k (wave suffix scan + carry),
//                  g_alpha_i = g_w_i T_i - suffix_i / (1 - alpha_i + 1e-10),
//                  g_sigma_i = g_alpha_i * delta_i exp(-delta_i a_i) * act'(sigma_i + noise_i),   g_c_i = w_i g_rgb.
// Nothing flows to z_vals (rendering.py:323 detaches the resampled depths; the coarse depths are data).
#include "mf_host.hpp"
#include "mf_core.hpp"

namespace mf {

struct CompBwdParams {
  const float* rays; long long ray_stride; long long n_rays;
  int S;
  const float* z_vals;       // (N,S)
  const float* rgbsigma;     // (N*S,4)
  const float* noise;        // (N,S) or null (pre-scaled by noise_std)
  int activation;
  const float* bg;           // (N,3) or null
  const float* g_rgb;        // (N,3) or null
  const float* g_depth;      // (N) or null
  const float* g_opacity;    // (N) or null
  float* g_out;              // (N*S,4)
};

MF_D float wave_scan_mul_incl(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= d) v *= o;
  }
  return v;
}
MF_D float wave_suffix_sum_incl(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(v, d, 64);
    if (lane + d < 64) v += o;
  }
  return v;
}

constexpr int kCompWaves = 4;

__global__ __launch_bounds__(64 * kCompWaves) void composite_backward_kernel(CompBwdParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * kCompWaves + wave;
  if (ray >= p.n_rays) return;
  const int S = p.S;
  float* L = reinterpret_cast<float*>(smem) + (size_t)wave * 5 * S;
  float *lw = L, *lgt = L + S, *lgw = L + 2 * S, *lfac = L + 3 * S, *lpt = L + 4 * S;
  const float* rp = p.rays + ray * p.ray_stride;
  const float dnorm = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  float gr[3] = {0.f, 0.f, 0.f}, bgc[3] = {0.f, 0.f, 0.f};
  if (p.g_rgb) { gr[0] = p.g_rgb[ray * 3]; gr[1] = p.g_rgb[ray * 3 + 1]; gr[2] = p.g_rgb[ray * 3 + 2]; }
  if (p.bg) { bgc[0] = p.bg[ray * 3]; bgc[1] = p.bg[ray * 3 + 1]; bgc[2] = p.bg[ray * 3 + 2]; }
  const float gd = p.g_depth ? p.g_depth[ray] : 0.f, go = p.g_opacity ? p.g_opacity[ray] : 0.f;
  const float* zr = p.z_vals + ray * S;
  float carry = 1.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool v = i < S;
    const int ii = v ? i : S - 1;
    const float4 s4 = *reinterpret_cast<const float4*>(p.rgbsigma + (ray * S + ii) * 4);
    const float z = zr[ii];
    const float znext = zr[ii + 1 < S ? ii + 1 : ii];
    float delta = (ii == S - 1) ? 1e10f : znext - z;
    delta = delta * dnorm;
    float sg = s4.w;
    if (p.noise) sg = sg + p.noise[ray * S + ii];
    float a, da;                                   // activation and its derivative (torch's backward formulas)
    if (p.activation == MF_ACT_RELU) {
      a = fmaxf(sg, 0.f);
      da = sg > 0.f ? 1.f : 0.f;
    } else {
      a = sg > 20.f ? sg : log1pf(expf(sg));
      const float ez = expf(sg);
      da = sg > 20.f ? 1.f : ez / (ez + 1.f);
    }
    const float e = expf(-delta * a);
    float alpha = 1.f - e;
    if (!v) alpha = 0.f;
    const float pt = v ? (1.f - alpha) + 1e-10f : 1.f;
    const float incl = wave_scan_mul_incl(pt, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
    const float T = carry * excl;
    const float w = alpha * T;
    carry = carry * __shfl(incl, 63, 64);
    if (v) {
      const float gw = gr[0] * (s4.x - bgc[0]) + gr[1] * (s4.y - bgc[1]) + gr[2] * (s4.z - bgc[2]) + gd * z + go;
      lw[i] = w;
      lgt[i] = gw * T;
      lgw[i] = gw * w;
      lfac[i] = delta * e * da;
      lpt[i] = pt;
    }
  }
  float tail = 0.f;                                // sum of g_w_k w_k over the chunks behind this one
  const int last = ((S - 1) / 64) * 64;
  for (int base = last; base >= 0; base -= 64) {
    const int i = base + lane;
    const bool v = i < S;
    const float gww = v ? lgw[i] : 0.f;
    const float incl = wave_suffix_sum_incl(gww, lane);
    const float suffix = (incl - gww) + tail;
    tail = tail + __shfl(incl, 0, 64);
    if (v) {
      const float w = lw[i];
      const float g_alpha = lgt[i] - suffix / lpt[i];
      *reinterpret_cast<float4*>(p.g_out + (ray * S + i) * 4) = make_float4(w * gr[0], w * gr[1], w * gr[2], g_alpha * lfac[i]);
    }
  }
}

}  // namespace mf

using namespace mf;

extern "C" int32_t mf_composite_backward(const float* rays, int64_t ray_stride, int64_t n_rays, int32_t S,
                                         const float* z_vals, const float* rgbsigma, const float* noise,
                                         int32_t activation, const float* background, const float* g_rgb,
                                         const float* g_depth, const float* g_opacity, float* g_rgbsigma, void* stream) {
  if (n_rays < 0 || (n_rays > 0 && (!rays || !z_vals || !rgbsigma || !g_rgbsigma)))
    return fail(MF_E_INVALID, "mf_composite_backward: null argument");
  if (activation != MF_ACT_RELU && activation != MF_ACT_SOFTPLUS)
    return fail(MF_E_INVALID, "mf_composite_backward: activation %d", activation);
  if (S < 1 || S > 2048) return fail(MF_E_UNSUPPORTED, "mf_composite_backward: S=%d (1..2048)", S);
  if (n_rays == 0) return MF_OK;
  CompBwdParams p{rays, ray_stride, n_rays, S, z_vals, rgbsigma, noise, activation, background, g_rgb, g_depth, g_opacity, g_rgbsigma};
  const size_t lds = (size_t)kCompWaves * 5 * S * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(composite_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_composite_backward: cannot reserve %zu bytes of LDS", lds);
  const unsigned grid = (unsigned)((n_rays + kCompWaves - 1) / kCompWaves);
  hipLaunchKernelGGL(composite_backward_kernel, dim3(grid), dim3(64 * kCompWaves), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_composite_backward");
}
