// mf_sample.hip -- the per-ray bookkeeping kernels around the fused pass:
//   mf_sample_pdf_merge : sample_pdf (models/rendering.py:5-46) + cat + sort (:321-326)
//   mf_compact_mask     : consensus mask compaction (models/rendering.py:306-314, 365-373)
// One wavefront per ray.  Index parity notes: the CDF is accumulated
// left-to-right in fp32 exactly like torch.cumsum on CPU (each lane re-adds its own prefix
// sequentially, so the result does not depend on a scan tree); searchsorted(right=True) is an
// exact integer count of cdf[k] <= u.
#include "mf_host.hpp"
#include "mf_core.hpp"

namespace mf {

struct PdfParams {
  const float* z;                          // (N, nb+1) coarse depths, or NULL when bins_in is given
  const float* bins_in;                    // (N, nb) explicit bins, or NULL (mid-points of z)
  const float* w; long long w_stride;      // first of the nb-1 weights of a ray, row stride
  long long n_rays; int nb, M;
  const float* u; long long u_stride;      // u_stride 0: one shared row (deterministic linspace)
  const float* cdf_in;                     // optional (N, nb): skip pdf/cdf (index-parity tests)
  float* z_out; int* inds_out; float* z_new_out;
  uint32_t per_wave_floats;
  float eps;                               // rendering.py:5 `eps` (1e-5 in every call the reference makes)
};

// Ascending bitonic sort of 64*E floats held E per lane (element lane*E + q) and the store of the first T of them.
// A compare-exchange evaluates ONE predicate on both sides of a pair ("the upper element is smaller"), so the two lanes of
// a pair always agree and no value is duplicated or lost, whatever the inputs (NaNs included).
template <int E>
MF_D void sort_store(const float* zall, float* zo, int T, int lane) {
  constexpr int P = 64 * E;
  float v[E];
#pragma unroll
  for (int q = 0; q < E; ++q) v[q] = lane * E + q < T ? zall[lane * E + q] : __builtin_inff();
#pragma unroll
  for (int k = 2; k <= P; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      if (j >= E) {                                        // partner: the same slot of lane ^ (j / E)
        const bool lower = (lane & (j / E)) == 0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const float o = __shfl_xor(v[q], j / E);
          const bool up = ((lane * E + q) & k) == 0;
          const float a = lower ? v[q] : o, b = lower ? o : v[q];   // a: the pair's lower index
          const bool swap = up ? b < a : a < b;
          v[q] = swap ? o : v[q];
        }
      } else {                                             // partner: slot q ^ j of this lane
#pragma unroll
        for (int q = 0; q < E; ++q) {
          if (q & j) continue;
          const bool up = ((lane * E + q) & k) == 0;
          const float a = v[q], b = v[q ^ j];
          const bool swap = up ? b < a : a < b;
          v[q] = swap ? b : a;
          v[q ^ j] = swap ? a : b;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (lane * E + q < T) zo[lane * E + q] = v[q];
}

__global__ __launch_bounds__(256) void sample_pdf_merge_kernel(PdfParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * 4 + wave;
  if (ray >= p.n_rays) return;
  const int nb = p.nb, M = p.M, S = nb + 1, nw = nb - 1, T = S + M;
  // every array starts on a 16-byte boundary: the sums below read them four floats at a time
  const int T4 = (T + 3) & ~3, nb4 = (nb + 3) & ~3;
  float* zall = sm + (size_t)wave * p.per_wave_floats;   // [T] : coarse z then new samples
  float* bins = zall + T4;                                // [nb]
  float* cdf = bins + nb4;                                // [nb]
  float* pdf = cdf + nb4;                                 // [nw]
  if (p.z) {
    const float* zr = p.z + ray * S;
    for (int i = lane; i < S; i += 64) zall[i] = zr[i];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nb; i += 64) bins[i] = 0.5f * (zall[i] + zall[i + 1]);          // :321
  } else {
    for (int i = lane; i < nb; i += 64) bins[i] = p.bins_in[ray * nb + i];
  }
  if (p.cdf_in) {
    for (int i = lane; i < nb; i += 64) cdf[i] = p.cdf_in[ray * nb + i];
  } else {
    const float* wr = p.w + ray * p.w_stride;                                                // weights[:, 1:-1]
    for (int i = lane; i < nw; i += 64) pdf[i] = wr[i] + p.eps;                              // :20
    __builtin_amdgcn_wave_barrier();
    // :21 torch.sum -- its association order is backend/ISA specific even inside the reference;
    // here: plain left-to-right fp32 (every lane recomputes it, LDS broadcast reads)
    float tot = 0.f;
    for (int i = 0; i < nw; i += 4) {                      // (same left-to-right order, four elements per LDS read)
      const f32x4 v = *reinterpret_cast<const f32x4*>(pdf + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) tot = i + j < nw ? tot + v[j] : tot;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nw; i += 64) pdf[i] = pdf[i] / tot;
    __builtin_amdgcn_wave_barrier();
    // cdf[k] = pdf[0] + ... + pdf[k-1], summed left to right (:22-23)
    for (int k0 = 0; k0 < nb; k0 += 64) {
      const int k = k0 + lane;
      const int kmax = k0 + 63 < nb ? k0 + 63 : nb - 1;    // the block's longest sum (uniform trip count)
      float c = 0.f;
      for (int i = 0; i < kmax; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pdf + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = i + j < k ? c + v[j] : c;
      }
      if (k < nb) cdf[k] = c;
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int m = lane; m < M; m += 64) {
    const float u = p.u[ray * p.u_stride + m];
    // inds = searchsorted(cdf, u, right=True) = #{k : cdf[k] <= u}   (cdf non-decreasing)
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int inds = lo;
    const int below = inds - 1 > 0 ? inds - 1 : 0;                                          // :34
    const int above = inds < nw ? inds : nw;                                                // :35
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bins[below], b1 = bins[above];
    float denom = c1 - c0;
    if (denom < p.eps) denom = 1.f;                                                          // :41-42
    const float s = b0 + (u - c0) / denom * (b1 - b0);                                       // :45
    zall[S + m] = s;
    if (p.inds_out) p.inds_out[ray * M + m] = inds;
    if (p.z_new_out) p.z_new_out[ray * M + m] = s;
  }
  if (!p.z_out) return;
  __builtin_amdgcn_wave_barrier();
  // torch.sort of the S+M depths (:326).  Only the values leave, so equal depths are interchangeable and any comparison
  // sort gives torch's output bit for bit: a bitonic network over the wave's registers (E depths per lane, padded with
  // +inf), 64E >= T.  (The rank sort this replaces -- rank = #{x < v} + #{x == v, earlier}, T compares per depth -- was
  // 35 of the launch's 41 us at T = 192; it stays as the path for T > 1024.)
  float* zo = p.z_out + ray * T;
  if (T <= 128) return sort_store<2>(zall, zo, T, lane);
  if (T <= 256) return sort_store<4>(zall, zo, T, lane);
  if (T <= 512) return sort_store<8>(zall, zo, T, lane);
  if (T <= 1024) return sort_store<16>(zall, zo, T, lane);
  for (int k = lane; k < T; k += 64) {
    const float v = zall[k];
    int rank = 0;
    for (int i = 0; i < T; ++i) {
      const float x = zall[i];
      rank += (x < v || (x == v && i < k)) ? 1 : 0;
    }
    zo[rank] = v;
  }
}

// ------------------------------------------------------------------ mask compaction
struct CompactParams {
  const float* alphas; const float* va; const float* vb;
  long long n_rays; int S;
  float* oa; float* ob;
  long long* count;
  long long* offs;     // scratch: n_rays + 1
};

__global__ __launch_bounds__(256) void compact_count_kernel(CompactParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * 4 + wave;
  if (ray >= p.n_rays) return;
  int c = 0;
  for (int base = 0; base < p.S; base += 64) {
    const int i = base + lane;
    const bool m = i < p.S && p.alphas[ray * p.S + i] >= 0.01f;                              // :306
    c += __popcll(__ballot(m));
  }
  if (lane == 0) p.offs[ray] = c;
}

// exclusive scan of the per-ray counts by one workgroup; offs[n_rays] = total
__global__ __launch_bounds__(1024) void compact_scan_kernel(CompactParams p) {
  __shared__ long long part[1024];
  __shared__ long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (long long base = 0; base < p.n_rays; base += 1024) {
    const long long i = base + threadIdx.x;
    const long long v = i < p.n_rays ? p.offs[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const long long o = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += o;
      __syncthreads();
    }
    const long long incl = part[threadIdx.x];
    const long long carry = carry_s;
    if (i < p.n_rays) p.offs[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long total = carry_s;
    p.offs[p.n_rays] = total;
    // no sample passes the threshold: the reference falls back to an all-true mask (:307-308)
    *p.count = total == 0 ? p.n_rays * (long long)p.S : total;
  }
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(CompactParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * 4 + wave;
  if (ray >= p.n_rays) return;
  const bool all = p.offs[p.n_rays] == 0;
  long long o = all ? ray * p.S : p.offs[ray];
  for (int base = 0; base < p.S; base += 64) {
    const int i = base + lane;
    const bool m = i < p.S && (all || p.alphas[ray * p.S + i] >= 0.01f);
    const unsigned long long b = __ballot(m);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (m) {
      if (p.oa) p.oa[o + before] = p.va[ray * p.S + i];
      if (p.ob) p.ob[o + before] = p.vb[ray * p.S + i];
    }
    o += __popcll(b);
  }
}

}  // namespace mf

using namespace mf;

extern "C" int32_t mf_sample_pdf_eps(const float* bins, const float* z_coarse, const float* weights, int64_t w_stride,
                                     int64_t n_rays, int32_t n_bins, int32_t M, const float* u, int64_t u_stride,
                                     const float* cdf_in, float* z_new_out, int32_t* inds_out, float* z_sorted_out,
                                     float eps, void* stream);

extern "C" int32_t mf_sample_pdf(const float* bins, const float* z_coarse, const float* weights, int64_t w_stride,
                                 int64_t n_rays, int32_t n_bins, int32_t M, const float* u, int64_t u_stride,
                                 const float* cdf_in, float* z_new_out, int32_t* inds_out, float* z_sorted_out,
                                 void* stream) {
  return mf_sample_pdf_eps(bins, z_coarse, weights, w_stride, n_rays, n_bins, M, u, u_stride, cdf_in, z_new_out, inds_out,
                           z_sorted_out, 1e-5f, stream);
}

extern "C" int32_t mf_sample_pdf_eps(const float* bins, const float* z_coarse, const float* weights, int64_t w_stride,
                                     int64_t n_rays, int32_t n_bins, int32_t M, const float* u, int64_t u_stride,
                                     const float* cdf_in, float* z_new_out, int32_t* inds_out, float* z_sorted_out,
                                     float eps, void* stream) {
  if (n_rays < 0 || n_bins < 2 || M < 1) return fail(MF_E_INVALID, "mf_sample_pdf: n_rays=%lld n_bins=%d M=%d", (long long)n_rays, n_bins, M);
  if (n_rays == 0) return MF_OK;
  if ((!bins && !z_coarse) || (!weights && !cdf_in) || !u) return fail(MF_E_INVALID, "mf_sample_pdf: null argument");
  if (z_sorted_out && !z_coarse) return fail(MF_E_INVALID, "mf_sample_pdf: the sorted merge needs z_coarse");
  PdfParams p{bins ? nullptr : z_coarse, bins, weights, w_stride, n_rays, n_bins, M, u, u_stride, cdf_in,
              z_sorted_out, inds_out, z_new_out, 0, eps};
  if (bins && z_sorted_out) return fail(MF_E_INVALID, "mf_sample_pdf: give either explicit bins or z_coarse (+merge)");
  const int S = n_bins + 1;
  const int r4 = 3;
  p.per_wave_floats = (uint32_t)(((S + M + r4) & ~r4) + 2 * ((n_bins + r4) & ~r4) + ((n_bins - 1 + r4) & ~r4));   // 16-byte aligned arrays
  const size_t lds = (size_t)p.per_wave_floats * 4 * 4;
  if (lds > 64 * 1024) return fail(MF_E_UNSUPPORTED, "mf_sample_pdf: n_bins+M=%d too large", S + M);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), lds,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("mf_sample_pdf");
}

extern "C" int32_t mf_sample_pdf_merge(const float* z_coarse, const float* weights, int64_t n_rays, int32_t S,
                                       int32_t M, const float* u, float* z_out, int32_t* inds_out,
                                       float* z_new_out, void* stream) {
  if (S < 3) return fail(MF_E_INVALID, "mf_sample_pdf_merge: S=%d", S);
  if (!weights || !z_out) return fail(MF_E_INVALID, "mf_sample_pdf_merge: null argument");
  return mf_sample_pdf(nullptr, z_coarse, weights + 1, S, n_rays, S - 1, M, u, u ? M : 0, nullptr, z_new_out,
                       inds_out, z_out, stream);
}

extern "C" int64_t mf_compact_scratch_bytes(int64_t n_rays) { return (n_rays + 2) * 8; }

extern "C" int32_t mf_compact_mask(const float* alphas, const float* vals_a, const float* vals_b, int64_t n_rays,
                                   int32_t S, float* out_a, float* out_b, int64_t* count, void* scratch, void* stream) {
  if (n_rays < 0 || S < 1) return fail(MF_E_INVALID, "mf_compact_mask: n_rays=%lld S=%d", (long long)n_rays, S);
  if (!count || !scratch) return fail(MF_E_INVALID, "mf_compact_mask: count / scratch missing");
  if (n_rays > 0 && (!alphas || (out_a && !vals_a) || (out_b && !vals_b))) return fail(MF_E_INVALID, "mf_compact_mask: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  CompactParams p{alphas, vals_a, vals_b, n_rays, S, out_a, out_b, reinterpret_cast<long long*>(count),
                  static_cast<long long*>(scratch)};
  const unsigned blocks = (unsigned)((n_rays + 3) / 4);
  if (n_rays > 0) hipLaunchKernelGGL(compact_count_kernel, dim3(blocks), dim3(256), 0, st, p);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, p);
  if (n_rays > 0) hipLaunchKernelGGL(compact_scatter_kernel, dim3(blocks), dim3(256), 0, st, p);
  return check_launch("mf_compact_mask");
}
