// mf_forward.hip -- module-level entry points (the trainers call the networks directly,
// outside render_rays: trainer_moco_flow.py:146-187, 508-526; trainer_nerf.py:231,240):
//   mf_embedding_forward : Embedding.forward   models/embedding.py:30-47
//   mf_nerf_forward      : NeRF.forward        models/nerf.py:61-102
//   mf_nof_forward       : NoF.forward         models/nof.py:55-85
// The two network kernels run the same register-resident MFMA core as the fused render pass;
// only the prologue (embedded inputs are read from memory instead of being computed) and the
// epilogue (raw rgb/sigma rows are stored instead of being composited) differ.
#include "mf_host.hpp"
#include "mf_layout.hpp"
#include "mf_nets.hpp"

namespace mf {

// ------------------------------------------------------------------ Embedding.forward
struct EmbFwdParams {
  mf_embedding e;
  const float* x;
  float* out;
  long long B;
  long long out_stride;   // floats per output row (>= the embedding's width; the columns past it are written 0)
  int repeat;             // output row b embeds input row b / repeat
};

// one thread per (output row, unit): unit u < C copies x_u; C <= u < C + C F evaluates ONE sincos for (frequency, channel)
// and writes both of its columns; the remaining units write the zero padding up to out_stride
__global__ void embedding_forward_kernel(EmbFwdParams p) {
  const int C = p.e.in_channels, F = p.e.n_freqs;
  const int OC = C * (2 * F + 1);
  const int units = C + C * F + (int)(p.out_stride - OC);
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.B * units) return;
  const long long b = idx / units;
  const int u = (int)(idx - b * units);
  const float* x = p.x + (b / p.repeat) * C;
  float* o = p.out + b * p.out_stride;
  if (u < C) {
    o[u] = x[u];
  } else if (u < C + C * F) {
    const int f = (u - C) / C, c = (u - C) % C;
    float sn, cs;
    sincosf(p.e.freq[f] * x[c], &sn, &cs);
    o[C + (2 * f) * C + c] = p.e.weight[f] * sn;
    o[C + (2 * f + 1) * C + c] = p.e.weight[f] * cs;
  } else {
    o[OC + (u - C - C * F)] = 0.f;
  }
}

// ------------------------------------------------------------------ NeRF.forward
struct NerfFwdParams {
  NetDev net;
  const float* in;
  long long in_stride, B;
  int sigma_only, extra_kind, extra_cols, xyz_cols;
  float* out;
  float* dump; long long dump_stride;          // DUMP: per-sample layer outputs [h_0 .. h_{D-1} | final | extra], as mf_render_pass dumps them
  uint32_t ring_off, buf_bytes;
};

template <bool DUMP>
__global__ __launch_bounds__(kThreads, 2) void nerf_forward_kernel(NerfFwdParams p) {
  const LaneId id;
  NetDev net = p.net;
  load_resident(net, id);
  Stream st;
  CarryT<kPD> carry;
  st.ring = p.ring_off;
  st.buf_bytes = p.buf_bytes;
  st.dbg = 0;
  st.keep2 = 0;
  start_program(net, st, carry, id);                    // also drains the resident-block DMA
  const long long ntiles = (p.B + kTile - 1) / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b = tile * kTile + id.wave * kWaveSamples + id.j;
    const bool valid = b < p.B;
    const float* row = p.in + (valid ? b : p.B - 1) * p.in_stride;
    float embx[kStepsNerfXyz], ext[kStepsExtraMax];
#pragma unroll
    for (int e = 0; e < kStepsNerfXyz; ++e) {
      const int f = sel4(id.g, emb_feature(kEmbNerfXyz, 0, e, 0), emb_feature(kEmbNerfXyz, 1, e, 0),
                         emb_feature(kEmbNerfXyz, 2, e, 0), emb_feature(kEmbNerfXyz, 3, e, 0));
      embx[e] = (f >= 0 && f < p.xyz_cols) ? row[f] : 0.f;      // (features beyond in_channels_xyz have no input column)
    }
#pragma unroll
    for (int e = 0; e < kStepsExtraMax; ++e) {
      int f = -1;
      if (p.extra_kind == kEmbDir)
        f = sel4(id.g, emb_feature(kEmbDir, 0, e, 0), emb_feature(kEmbDir, 1, e, 0), emb_feature(kEmbDir, 2, e, 0),
                 emb_feature(kEmbDir, 3, e, 0));
      else if (p.extra_kind == kEmbInd)
        f = sel4(id.g, emb_feature(kEmbInd, 0, e, 0), emb_feature(kEmbInd, 1, e, 0), emb_feature(kEmbInd, 2, e, 0),
                 emb_feature(kEmbInd, 3, e, 0));
      ext[e] = (!p.sigma_only && f >= 0 && f < p.extra_cols) ? row[p.xyz_cols + f] : 0.f;
    }
    float sigma, rgb[3] = {0.f, 0.f, 0.f};
    float* dump_row = nullptr;
    if constexpr (DUMP) {
      if (valid) dump_row = p.dump + b * p.dump_stride;
    }
    nerf_eval<16, DUMP>(net, embx, ext, p.sigma_only != 0, st, carry, id, follow_of(net), sigma, rgb, dump_row);
    // (the sample index is rebuilt from an opaque lane index: the 64-bit output address is then formed here instead of
    //  being carried -- with DUMP: spilled -- across the MFMA section)
    int jo = id.j;
    asm volatile("" : "+v"(jo));
    const long long bo = tile * kTile + id.wave * kWaveSamples + jo;
    if (bo < p.B && id.g == 0) {
      if (p.sigma_only) p.out[bo] = sigma;
      else *reinterpret_cast<float4*>(p.out + bo * 4) = make_float4(rgb[0], rgb[1], rgb[2], sigma);
    }
  }
  wait_vm0();   // the stream runs two panels ahead: drain the LDS-DMA before the workgroup retires
}

// ------------------------------------------------------------------ NoF.forward
struct NofFwdParams {
  NetDev net;
  const float* in;
  long long in_stride, B;
  const float* xyz;
  float* out;
  uint32_t ring_off, buf_bytes;
  int xyz_cols, in_cols;     // in_channels_xyz, in_channels_xyz + extra_feat_dim: the columns an input row has
};

template <int NK>
__global__ __launch_bounds__(kThreads, 2) void nof_forward_kernel(NofFwdParams p) {
  const LaneId id;
  NetDev net = p.net;
  load_resident(net, id);
  Stream st;
  CarryT<kPD> carry;
  st.ring = p.ring_off;
  st.buf_bytes = p.buf_bytes;
  st.dbg = 0;
  st.keep2 = 0;
  start_program(net, st, carry, id);
  const long long ntiles = (p.B + kTile - 1) / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b = tile * kTile + id.wave * kWaveSamples + id.j;
    const bool valid = b < p.B;
    const long long bb = valid ? b : p.B - 1;
    const float* row = p.in + bb * p.in_stride;
    float emb[kStepsNofIn];
#pragma unroll
    for (int e = 0; e < kStepsNofIn; ++e) {
      const int f = sel4(id.g, emb_feature(kEmbNofIn, 0, e, p.xyz_cols), emb_feature(kEmbNofIn, 1, e, p.xyz_cols),
                         emb_feature(kEmbNofIn, 2, e, p.xyz_cols), emb_feature(kEmbNofIn, 3, e, p.xyz_cols));
      emb[e] = (f >= 0 && f < p.in_cols) ? row[f] : 0.f;
    }
    const float xyz[3] = {p.xyz[bb * 3 + 0], p.xyz[bb * 3 + 1], p.xyz[bb * 3 + 2]};
    float o[3];
    nof_eval<false, NK>(net, emb, xyz, st, carry, id, follow_of(net), o);
    if (valid && id.g == 0) {
      p.out[b * 3 + 0] = o[0];
      p.out[b * 3 + 1] = o[1];
      p.out[b * 3 + 2] = o[2];
    }
  }
  wait_vm0();
}

// ------------------------------------------------------------------ fused point queries
// sigma (and the canonical position) of free points: the whole of forward_nof + embed + pad +
// NeRF(sigma_only) that trainer_moco_flow.py:146-187 / 500-526 (visualize_mesh lattice, SMPL-point
// losses) spell with five module calls and three padded temporaries per 10 000-point chunk, as one
// launch over all points: xyz -> [bw NoF(ind)] -> encode in registers -> NeRF trunk -> sigma.
struct PointsParams {
  NetDev nerf, nof;
  EmbParams exyz, nxyz, nind;
  const float* xyz;        // (B,3)
  const float* ind;        // (B,) per-point image index, or null -> ind_scalar
  float ind_scalar;
  long long B;
  float* sigma;            // (B,)  raw sigma (no activation)
  float* canon;            // (B,3) or null: the point after the backward flow
  uint32_t ring_off, buf_bytes;
};

template <bool NOF>
__global__ __launch_bounds__(kThreads, 2) void points_kernel(PointsParams p) {
  const LaneId id;
  load_resident(p.nerf, id);
  if (NOF) load_resident(p.nof, id);
  Stream st;
  CarryT<kPD> carry;
  st.ring = p.ring_off;
  st.buf_bytes = p.buf_bytes;
  st.dbg = 0;
  st.keep2 = 0;
  const NextLayer prog_first = NOF ? follow_of(p.nof) : follow_of(p.nerf);
  if (NOF) start_program(p.nof, st, carry, id);
  else start_program(p.nerf, st, carry, id);
  const long long ntiles = (p.B + kTile - 1) / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b = tile * kTile + id.wave * kWaveSamples + id.j;
    const bool valid = b < p.B;
    const long long bb = valid ? b : p.B - 1;
    float x[3] = {p.xyz[bb * 3 + 0], p.xyz[bb * 3 + 1], p.xyz[bb * 3 + 2]};
    if (NOF) {
      const float ind = p.ind ? p.ind[bb] : p.ind_scalar;
      float emb[kStepsNofIn], out[3];
      nof_embed(emb, x, ind, p.nxyz, p.nind, id.g);
      nof_eval<>(p.nof, emb, x, st, carry, id, follow_of(p.nerf), out);
      x[0] = out[0]; x[1] = out[1]; x[2] = out[2];
      if (valid && id.g == 0 && p.canon) {
        p.canon[b * 3 + 0] = x[0]; p.canon[b * 3 + 1] = x[1]; p.canon[b * 3 + 2] = x[2];
      }
    }
    float embx[kStepsNerfXyz], ext[kStepsExtraMax];
    emb_eval<3, 10>(embx, x, p.exyz, id.g);
#pragma unroll
    for (int e = BlkXyz10::SLOTS; e < kStepsNerfXyz; ++e) embx[e] = 0.f;
#pragma unroll
    for (int e = 0; e < kStepsExtraMax; ++e) ext[e] = 0.f;
    float sigma, rgb[3];
    nerf_eval<16>(p.nerf, embx, ext, true, st, carry, id, prog_first, sigma, rgb);
    if (valid && id.g == 0) p.sigma[b] = sigma;
  }
  wait_vm0();
}

int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

}  // namespace mf

using namespace mf;

static int32_t embedding_forward(const char* who, const mf_embedding* e, const float* x, int64_t B, int32_t repeat,
                                 float* out, int64_t out_stride, void* stream) {
  if (!e || (B > 0 && (!x || !out))) return fail(MF_E_INVALID, "%s: null argument", who);
  if (e->in_channels < 1 || e->n_freqs < 0 || e->n_freqs > MF_MAX_FREQS)
    return fail(MF_E_INVALID, "%s: in_channels=%d n_freqs=%d out of range", who, e->in_channels, e->n_freqs);
  const long long width = (long long)e->in_channels * (2 * e->n_freqs + 1);
  if (out_stride <= 0) out_stride = width;
  if (B < 0 || repeat < 1 || out_stride < width)
    return fail(MF_E_INVALID, "%s: B=%lld repeat=%d out_stride=%lld (embedding width %lld)", who, (long long)B, repeat,
                (long long)out_stride, width);
  if (B == 0) return MF_OK;
  EmbFwdParams p{*e, x, out, B, out_stride, repeat};
  const long long total = B * (e->in_channels * (1 + e->n_freqs) + (out_stride - width));
  hipLaunchKernelGGL(embedding_forward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch(who);
}

extern "C" int32_t mf_embedding_forward(const mf_embedding* e, const float* x, int64_t B, float* out, void* stream) {
  return embedding_forward("mf_embedding_forward", e, x, B, 1, out, 0, stream);
}

extern "C" int32_t mf_embedding_forward_rows(const mf_embedding* e, const float* x, int64_t B, int32_t repeat, float* out,
                                             int64_t out_stride, void* stream) {
  return embedding_forward("mf_embedding_forward_rows", e, x, B, repeat, out, out_stride, stream);
}

static int32_t nerf_forward_launch(const char* who, const mf_nerf_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                                   int64_t B, int32_t sigma_only, float* out, float* dump, int64_t dump_stride, void* stream) {
  if (!d || !packed || (B > 0 && (!inputs || !out))) return fail(MF_E_INVALID, "%s: null argument", who);
  NerfFwdParams p{};
  if (!nerf_layout(*d, p.net.L)) return fail(MF_E_UNSUPPORTED, "%s: unsupported NeRF configuration", who);
  if (p.net.L.NK != 16) return fail(MF_E_UNSUPPORTED, "%s: only W=256 is built", who);
  if (dump && dump_stride < (int64_t)p.net.L.n_trunk * p.net.L.W + p.net.L.W / 2)
    return fail(MF_E_INVALID, "%s: dump_stride %lld too small", who, (long long)dump_stride);
  if (B == 0) return MF_OK;
  p.net.packed = static_cast<const char*>(packed);
  p.net.res_lds = 0;
  p.in = inputs; p.in_stride = in_stride; p.B = B; p.sigma_only = sigma_only; p.out = out;
  p.dump = dump; p.dump_stride = dump_stride;
  p.extra_kind = d->extra_feat_type == MF_EXTRA_DIR ? kEmbDir : (d->extra_feat_type == MF_EXTRA_IND ? kEmbInd : kEmbNone);
  p.extra_cols = d->extra_feat_type == MF_EXTRA_NONE ? 0 : d->extra_feat_dim;
  p.xyz_cols = d->in_channels_xyz;
  p.ring_off = (uint32_t)p.net.L.res_bytes;
  p.buf_bytes = (uint32_t)p.net.L.max_groups * kGroupBytes;
  const size_t lds = p.ring_off + 3 * (size_t)p.buf_bytes;
  const void* fn = dump ? reinterpret_cast<const void*>(nerf_forward_kernel<true>) : reinterpret_cast<const void*>(nerf_forward_kernel<false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
  const long long ntiles = (B + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  if (dump) hipLaunchKernelGGL(nerf_forward_kernel<true>, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(nerf_forward_kernel<false>, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  return check_launch(who);
}

extern "C" int32_t mf_nerf_forward(const mf_nerf_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                                   int64_t B, int32_t sigma_only, float* out, void* stream) {
  return nerf_forward_launch("mf_nerf_forward", d, packed, inputs, in_stride, B, sigma_only, out, nullptr, 0, stream);
}

extern "C" int32_t mf_nerf_forward_dump(const mf_nerf_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                                        int64_t B, float* out, float* dump_acts, int64_t dump_stride, void* stream) {
  if (B > 0 && !dump_acts) return fail(MF_E_INVALID, "mf_nerf_forward_dump: null dump buffer");
  return nerf_forward_launch("mf_nerf_forward_dump", d, packed, inputs, in_stride, B, 0, out, dump_acts, dump_stride, stream);
}

extern "C" int32_t mf_nof_forward(const mf_nof_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                                  const float* xyz, int64_t B, float* out, void* stream) {
  if (!d || !packed || (B > 0 && (!inputs || !xyz || !out))) return fail(MF_E_INVALID, "mf_nof_forward: null argument");
  NofFwdParams p{};
  if (!nof_layout(*d, p.net.L, 0, true)) return fail(MF_E_UNSUPPORTED, "mf_nof_forward: unsupported NoF configuration");
  if (B == 0) return MF_OK;
  p.net.packed = static_cast<const char*>(packed);
  p.net.res_lds = 0;
  p.in = inputs; p.in_stride = in_stride; p.B = B; p.xyz = xyz; p.out = out;
  p.xyz_cols = d->in_channels_xyz; p.in_cols = d->in_channels_xyz + d->extra_feat_dim;
  p.ring_off = (uint32_t)p.net.L.res_bytes;
  p.buf_bytes = (uint32_t)p.net.L.max_groups * kGroupBytes;
  const size_t lds = p.ring_off + 3 * (size_t)p.buf_bytes;
  void (*kern)(NofFwdParams) = p.net.L.NK == 16 ? nof_forward_kernel<16> : nof_forward_kernel<8>;     // W = 256: the bare NoF() default
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nof_forward: cannot reserve %zu bytes of LDS", lds);
  const long long ntiles = (B + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nof_forward");
}

static void emb_to_params(const mf_embedding& e, EmbParams& o) {
  for (int k = 0; k < 16; ++k) {
    o.freq[k] = k < e.n_freqs ? e.freq[k] : 0.f;
    o.weight[k] = k < e.n_freqs ? e.weight[k] : 0.f;
  }
}

namespace mf {
int points_sigma_bf16(int prec, const mf_nerf_desc* nerf, const void* nerf_packed, const mf_embedding* emb_xyz, const mf_nof_desc* nof,
                      const void* nof_packed, const mf_embedding* nof_emb_xyz, const mf_embedding* nof_emb_ind, const float* xyz,
                      const float* ind, float ind_scalar, int64_t B, float* sigma, float* canon, void* workspace,
                      int64_t workspace_bytes, hipStream_t st);   // mf_render_bf16.hip
int64_t points_workspace_bytes_bf16(const mf_nof_desc* nof, int per_point_ind, int64_t B);
}

extern "C" int64_t mf_points_sigma_workspace_bytes(int32_t precision, const mf_nof_desc* nof, int32_t per_point_ind, int64_t B) {
  return precision != MF_PREC_F32 ? points_workspace_bytes_bf16(nof, per_point_ind, B) : 0;
}

extern "C" int32_t mf_points_sigma(const mf_nerf_desc* nerf, const void* nerf_packed, const mf_embedding* emb_xyz,
                                   const mf_nof_desc* nof, const void* nof_packed, const mf_embedding* nof_emb_xyz,
                                   const mf_embedding* nof_emb_ind, const float* xyz, const float* ind,
                                   float ind_scalar, int64_t B, float* sigma, float* canon, void* stream) {
  return mf_points_sigma_p(MF_PREC_F32, nerf, nerf_packed, emb_xyz, nof, nof_packed, nof_emb_xyz, nof_emb_ind, xyz, ind, ind_scalar, B,
                           sigma, canon, nullptr, 0, stream);
}

extern "C" int32_t mf_points_sigma_p(int32_t precision, const mf_nerf_desc* nerf, const void* nerf_packed, const mf_embedding* emb_xyz,
                                     const mf_nof_desc* nof, const void* nof_packed, const mf_embedding* nof_emb_xyz,
                                     const mf_embedding* nof_emb_ind, const float* xyz, const float* ind,
                                     float ind_scalar, int64_t B, float* sigma, float* canon, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (!nerf || !nerf_packed || !emb_xyz || (B > 0 && (!xyz || !sigma)))
    return fail(MF_E_INVALID, "mf_points_sigma: null argument");
  if (precision < MF_PREC_F32 || precision > MF_PREC_BF16X3) return fail(MF_E_INVALID, "mf_points_sigma: precision %d", precision);
  if (precision != MF_PREC_F32) {
    if (emb_xyz->in_channels != 3 || emb_xyz->n_freqs > 10)
      return fail(MF_E_UNSUPPORTED, "mf_points_sigma: xyz embedding must have 3 channels and <= 10 frequencies");
    if (nof) {
      if (!nof_packed || !nof_emb_xyz || !nof_emb_ind) return fail(MF_E_INVALID, "mf_points_sigma: NoF arguments missing");
      if (nof_emb_xyz->in_channels != 3 || nof_emb_xyz->n_freqs > 5 || nof_emb_ind->in_channels != 1 || nof_emb_ind->n_freqs > 16)
        return fail(MF_E_UNSUPPORTED, "mf_points_sigma: NoF embeddings must be xyz(3, <=5 freqs) and ind(1, <=16 freqs)");
    }
    if (B == 0) return MF_OK;
    return points_sigma_bf16(precision, nerf, nerf_packed, emb_xyz, nof, nof_packed, nof_emb_xyz, nof_emb_ind, xyz, ind, ind_scalar, B, sigma, canon,
                             workspace, workspace_bytes, static_cast<hipStream_t>(stream));
  }
  PointsParams p{};
  if (!nerf_layout(*nerf, p.nerf.L) || p.nerf.L.NK != 16)
    return fail(MF_E_UNSUPPORTED, "mf_points_sigma: unsupported NeRF configuration");
  if (emb_xyz->in_channels != 3 || emb_xyz->n_freqs > 10)
    return fail(MF_E_UNSUPPORTED, "mf_points_sigma: xyz embedding must have 3 channels and <= 10 frequencies");
  if (B == 0) return MF_OK;
  uint32_t lds = 0;
  p.nerf.packed = static_cast<const char*>(nerf_packed);
  p.nerf.res_lds = lds; lds += (uint32_t)p.nerf.L.res_bytes;
  int max_groups = p.nerf.L.max_groups;
  emb_to_params(*emb_xyz, p.exyz);
  if (nof) {
    if (!nof_packed || !nof_emb_xyz || !nof_emb_ind) return fail(MF_E_INVALID, "mf_points_sigma: NoF arguments missing");
    if (!nof_layout(*nof, p.nof.L)) return fail(MF_E_UNSUPPORTED, "mf_points_sigma: unsupported NoF configuration");
    if (nof_emb_xyz->in_channels != 3 || nof_emb_xyz->n_freqs > 5 || nof_emb_ind->in_channels != 1 || nof_emb_ind->n_freqs > 16)
      return fail(MF_E_UNSUPPORTED, "mf_points_sigma: NoF embeddings must be xyz(3, <=5 freqs) and ind(1, <=16 freqs)");
    p.nof.packed = static_cast<const char*>(nof_packed);
    p.nof.res_lds = lds; lds += (uint32_t)p.nof.L.res_bytes;
    if (p.nof.L.max_groups > max_groups) max_groups = p.nof.L.max_groups;
    emb_to_params(*nof_emb_xyz, p.nxyz);
    emb_to_params(*nof_emb_ind, p.nind);
  }
  p.xyz = xyz; p.ind = ind; p.ind_scalar = ind_scalar; p.B = B; p.sigma = sigma; p.canon = canon;
  p.ring_off = lds;
  p.buf_bytes = (uint32_t)max_groups * kGroupBytes;
  lds += 3 * p.buf_bytes;
  const void* fn = nof ? reinterpret_cast<const void*>(points_kernel<true>) : reinterpret_cast<const void*>(points_kernel<false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_points_sigma: cannot reserve %u bytes of LDS", lds);
  const long long ntiles = (B + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nof) hipLaunchKernelGGL(points_kernel<true>, dim3(grid), dim3(kThreads), lds, st, p);
  else hipLaunchKernelGGL(points_kernel<false>, dim3(grid), dim3(kThreads), lds, st, p);
  return check_launch("mf_points_sigma");
}
