// mf_nofgrad.hip -- backward of one neural-motion-flow evaluation  out = NoF(embed(pts), embed(ind), pts)
// (models/rendering.py:49-83 + models/nof.py:69-82), the unit the consensus chains of render_rays are
// made of (rendering.py:262-314: bw, fw o bw, fw o bw o fw o bw).  Replaces the autograd graph torch
// records for it under loss.backward() (trainer_moco_flow.py:317-328, trainer/base.py:188-197).
//
//   mf_nof_points_dump : the forward of one evaluation on free points, storing what the backward reads:
//       post-ReLU layer outputs, the raw head T, and the embedded input in natural column order.
//   mf_nof_backward    : d_out (P,3) -> d_pts (P,3) + every pre-activation gradient, one launch:
//       transform backward (flow: identity; quaternion: forward-mode partials of the restated kornia
//       formulas, PARITY UNPINNED as in the forward) -> head^T on the VALU -> the trunk's W^T chain on
//       v_mfma_f32_16x16x4_f32 through a transposed fragment stream (same core as the NeRF backward) ->
//       the embedded-input gradient of layer 0 and of the skip layer as ONE K = 2W contraction ->
//       sin/cos chain rule back to the point.
// Weight gradients are then mf_weight_grads items on (dump, gradient buffer).
#include "mf_host.hpp"
#include "mf_layout.hpp"
#include "mf_nets.hpp"
#include "mf_bwd.hpp"
#include "mf_nofbwd.hpp"
#include <cstdlib>

namespace mf {

int device_cus();   // mf_forward.hip

constexpr int kNofW = 128;
constexpr int kNofEmbCols = 80;       // embedded input row of the dump: [xyz 33 | ind 33 | 0 x 14]
constexpr int kNofHeadPad = 16;       // head slot of the dump / gradient rows: T (9 | 3), zero padded

// ------------------------------------------------------------------ forward with dump
struct NofDumpParams {
  NetDev net;
  EmbParams exyz, eind;
  const float* pts;          // (P,3)
  const float* inputs;       // (P, in_stride) pre-embedded [xyz 33 | ind 33] rows (module-level call), or null
  long long in_stride;
  const float* ind;          // per ray: ind[ray * ind_stride]   (unused with `inputs`)
  long long ind_stride;
  long long P;
  int S;                     // samples per ray: sample s belongs to ray s / S
  float* out;                // (P,3)
  float* acts;               // (P, stride): [h_1 .. h_D | T pad 16]
  long long stride;
  float* emb;                // (P, 80)
  uint32_t ring_off, buf_bytes;
};

__global__ __launch_bounds__(kThreads, 2) void nof_points_dump_kernel(NofDumpParams p) {
  const LaneId id;
  const NetDev net = p.net;
  load_resident(net, id);
  Stream st;
  CarryT<kPD> carry;
  st.ring = p.ring_off;
  st.buf_bytes = p.buf_bytes;
  st.dbg = 0;
  st.keep2 = 0;
  start_program(net, st, carry, id);
  const int D = net.L.n_trunk;
  const long long ntiles = (p.P + kTile - 1) / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long s = tile * kTile + id.wave * kWaveSamples + id.j;
    const bool valid = s < p.P;
    const long long ss = valid ? s : p.P - 1;
    const float x[3] = {p.pts[ss * 3 + 0], p.pts[ss * 3 + 1], p.pts[ss * 3 + 2]};
    float emb[kStepsNofIn];
    if (p.inputs) {
      const float* row = p.inputs + ss * p.in_stride;
#pragma unroll
      for (int e = 0; e < kStepsNofIn; ++e) {
        const int f = sel4(id.g, emb_feature(kEmbNofIn, 0, e, 33), emb_feature(kEmbNofIn, 1, e, 33),
                           emb_feature(kEmbNofIn, 2, e, 33), emb_feature(kEmbNofIn, 3, e, 33));
        emb[e] = f >= 0 ? row[f] : 0.f;
      }
    } else {
      const float indv = p.ind[(ss / p.S) * p.ind_stride];
      nof_embed(emb, x, indv, p.exyz, p.eind, id.g);
    }
    if (valid && p.emb) {
      float* erow = p.emb + s * kNofEmbCols;
#pragma unroll
      for (int e = 0; e < kStepsNofIn; ++e) {
        const int f = sel4(id.g, emb_feature(kEmbNofIn, 0, e, 33), emb_feature(kEmbNofIn, 1, e, 33),
                           emb_feature(kEmbNofIn, 2, e, 33), emb_feature(kEmbNofIn, 3, e, 33));
        if (f >= 0) erow[f] = emb[e];
      }
      if (id.g == 0)
        for (int c = 66; c < kNofEmbCols; ++c) erow[c] = 0.f;
    }
    float* drow = valid ? p.acts + s * p.stride : nullptr;
    f32x4 act[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) act[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.keep2 = 0;
    // (rows with room behind T also get the layers' ReLU bit rows: nof_eval's `masks`, mf_nets.hpp)
    const bool masks = p.stride >= (long long)D * kNofW + kNofHeadPad + 4 * D;
    for (int l = 0; l < D; ++l)
      trunk_layer<8, kStepsNofIn, true>(net, l, act, emb, st, carry, id,
                                               l == D - 1 ? follow_of(net) : next_trunk(net, l + 1),
                                               drow ? drow + l * kNofW : nullptr,
                                               (drow && masks) ? reinterpret_cast<unsigned*>(drow + D * kNofW + kNofHeadPad) + 4 * l : nullptr);
    const uint32_t wo = net.res_lds + net.L.off_head_w * 4, bo = net.res_lds + net.L.off_head_b * 4;
    float T[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o[3];
    if (net.L.n_head == 9) {
      valu_head(act, wo, kNofW, bo, id.g, T);
      quat_transform(T, x, o);
    } else {
      float T3[3];
      valu_head(act, wo, kNofW, bo, id.g, T3);
#pragma unroll
      for (int c = 0; c < 3; ++c) { T[c] = T3[c]; o[c] = T3[c] + x[c]; }
    }
    if (valid && id.g == 0) {
      p.out[s * 3 + 0] = o[0]; p.out[s * 3 + 1] = o[1]; p.out[s * 3 + 2] = o[2];
      float4* tr = reinterpret_cast<float4*>(drow + (long long)D * kNofW);
      tr[0] = make_float4(T[0], T[1], T[2], T[3]);
      tr[1] = make_float4(T[4], T[5], T[6], T[7]);
      tr[2] = make_float4(T[8], 0.f, 0.f, 0.f);
      tr[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  wait_vm0();
}

// ------------------------------------------------------------------ backward "network"
// layer i < D-1 : trunk layer l = D-1-i transposed (hidden columns only), W -> W, ReLU mask h_l
// layer D-1     : [W_0[:, :64]^T | W_skip[:, :64]^T]  (K = W or 2W -> 64 embedded columns)
inline bool nof_bwd_layout(const mf_nof_desc& d, NetLayout& L, int& skip) {
  NetLayout F;
  if (!nof_layout(d, F, 0)) return false;
  skip = -1;
  for (int l = 1; l < d.D; ++l)
    if ((d.skip_mask >> l) & 1u) {
      if (skip >= 0) return false;               // one skip layer at most (every reference config)
      skip = l;
    }
  L = NetLayout{};
  L.W = F.W; L.NK = F.NK; L.NP = F.NP;
  L.n_trunk = d.D;
  L.n_head = d.use_quat ? 9 : 3;
  int off = 0;
  L.off_bias_trunk = off; off += L.W;            // shared all-zero bias vector
  L.off_head_w = off; off += L.n_head * L.W;     // nof_encoding_final.weight, natural order
  L.res_bytes = round_up((int64_t)off * 4, kGroupBytes);
  const int k_emb = skip >= 0 ? 2 * L.NK : L.NK; // k-quads of the last layer
  L.max_groups = 2 * k_emb;
  L.panel_bytes = ((int64_t)(d.D - 1) * L.NP * 2 * L.NK + 2 * 2 * k_emb) * kGroupBytes;
  return true;
}

struct NofBwdPackJob {
  const float* W[MF_MAX_LAYERS];       // forward trunk weights
  int ld[MF_MAX_LAYERS];
  int D, skip, NK, NP;
  const float* head_w;
  int n_head_w, off_head_w, res_floats;
  float* res;
  float* panels;
  long long total_groups;
};

__global__ void pack_nof_bwd_kernel(NofBwdPackJob job) {
  const long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx < job.res_floats) {
    const int o = (int)gidx - job.off_head_w;
    job.res[gidx] = (o >= 0 && o < job.n_head_w) ? job.head_w[o] : 0.f;
  }
  const long long grp = gidx >> 6;
  if (grp >= job.total_groups) return;
  const int lane = (int)(gidx & 63);
  const int i = lane & 15, g = lane >> 4;
  const int per_layer = job.NP * 2 * job.NK;            // groups of a W -> W layer
  const long long chain = (long long)(job.D - 1) * per_layer;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  float* pv = &v.x;
  if (grp < chain) {
    const int li = (int)(grp / per_layer), local = (int)(grp % per_layer);
    const int l = job.D - 1 - li;                        // forward layer
    const int P = local / (2 * job.NK), gi = local % (2 * job.NK);
    const int b = gi >> 1, half = gi & 1;
    const int n = 32 * P + 16 * half + i;
    const int col0 = l == job.skip ? job.ld[l] - 16 * job.NK : 0;
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * b + 4 * g + r;
      pv[r] = job.W[l][(long long)k * job.ld[l] + col0 + n];
    }
  } else {
    const int kq = job.skip >= 0 ? 2 * job.NK : job.NK;
    const int local = (int)(grp - chain);
    const int P = local / (2 * kq), gi = local % (2 * kq);
    const int b = gi >> 1, half = gi & 1;
    const int n = 32 * P + 16 * half + i;                // embedded column 0..63
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * b + 4 * g + r;                  // [d z_0 ; d z_skip]
      const int l = k < 16 * job.NK ? 0 : job.skip;
      const int kk = k < 16 * job.NK ? k : k - 16 * job.NK;
      pv[r] = job.W[l][(long long)kk * job.ld[l] + n];
    }
  }
  reinterpret_cast<float4*>(job.panels)[gidx] = v;
}

// ------------------------------------------------------------------ the backward kernel
struct NofBwdParams {
  NetDev net;
  EmbParams exyz;
  int D, skip;
  long long P, stride;
  const float* pts;        // (P,3) input points of the evaluation
  const float* acts;       // (P,stride) dump of mf_nof_points_dump
  const float* g_out;      // (P,3)
  float* gpre;             // (round_up(P,128), stride): [d z_0 .. d z_{D-1} | d T pad 16]
  float* g_pts;            // (P,3) or null
  uint32_t ring_off, buf_bytes;
  int dbg;
};

__global__ __launch_bounds__(kThreads, 2) void nof_backward_kernel(NofBwdParams p) {
  const LaneId id;
  const NetDev net = p.net;
  load_resident(net, id);
  Stream st;
  CarryT<kPD> carry;
  st.ring = p.ring_off;
  st.buf_bytes = p.buf_bytes;
  st.dbg = MF_TIMING_FLAGS ? p.dbg : 0;
  st.keep2 = 0;
  const uint32_t zero_bias = net.res_lds + net.L.off_bias_trunk * 4;
  const uint32_t headw = net.res_lds + net.L.off_head_w * 4;
  const char* first = net.packed + net.L.res_bytes;
  const int D = p.D, NH = net.L.n_head;
  const int g_chain = 2 * 8, g_emb = p.skip >= 0 ? 2 * 16 : 2 * 8;
  st.start(first, D > 1 ? g_chain : g_emb, id);
  carry.load(st.slot_off(0) + id.lane * 16, zero_bias, id.g);
  const float nosig[kBwdSigSteps] = {0.f, 0.f, 0.f, 0.f};
  const long long ntiles = (p.P + kTile - 1) / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long s = tile * kTile + id.wave * kWaveSamples + id.j;
    const bool valid = s < p.P;
    const long long ss = valid ? s : p.P - 1;
    const float x[3] = {p.pts[ss * 3 + 0], p.pts[ss * 3 + 1], p.pts[ss * 3 + 2]};
    const float go[3] = {p.g_out[ss * 3 + 0], p.g_out[ss * 3 + 1], p.g_out[ss * 3 + 2]};
    const float* arow = p.acts + ss * p.stride;
    float* grow = p.gpre + s * p.stride;          // rows up to round_up(P,128) exist
    float T[9], dT[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dx[3];
    {
      const float4* tr = reinterpret_cast<const float4*>(arow + (long long)D * kNofW);
      const float4 t0 = tr[0], t1 = tr[1], t2 = tr[2];
      T[0] = t0.x; T[1] = t0.y; T[2] = t0.z; T[3] = t0.w; T[4] = t1.x; T[5] = t1.y; T[6] = t1.z; T[7] = t1.w; T[8] = t2.x;
    }
    if (NH == 9) {
      quat_transform_backward(T, x, go, dT, dx);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { dT[c] = go[c]; dx[c] = go[c]; }
    }
    {
      float4* tg = reinterpret_cast<float4*>(grow + (long long)D * kNofW);     // all four lane groups write the same values
      tg[0] = make_float4(dT[0], dT[1], dT[2], dT[3]);
      tg[1] = make_float4(dT[4], dT[5], dT[6], dT[7]);
      tg[2] = make_float4(dT[8], 0.f, 0.f, 0.f);
      tg[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // d z_{D-1} = (W_head^T d T) * [h_D > 0]
    f32x4 a[8], b[8], cat[16];
    {
      const float* hrow = arow + (long long)(D - 1) * kNofW;
      float* ghrow = grow + (long long)(D - 1) * kNofW;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          if (c < NH) {
            const f32x4 w = lds_f4(headw + (c * kNofW + 16 * t + 4 * id.g) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(w[r], dT[c], acc[r]);
          }
        }
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(hrow + 16 * t + 4 * id.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[t][r] = h4[r] > 0.f ? acc[r] : 0.f;
        *reinterpret_cast<f32x4*>(ghrow + 16 * t + 4 * id.g) = a[t];
      }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) cat[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // chain: d z_{l-1} = (W_l[:, hidden]^T d z_l) * [h_l > 0],  l = D-1 .. 1
    for (int l = D - 1; l >= 1; --l) {
      if (l == p.skip) {
#pragma unroll
        for (int t = 0; t < 8; ++t) cat[8 + t] = a[t];
      }
      NextLayer nx;
      nx.groups = l > 1 ? g_chain : g_emb;
      nx.jump = nullptr;
      nx.bias_off = zero_bias;
      bwd_layer<2, 8, 4, true, true>(a, nosig, b, g_chain, zero_bias, st, carry, id, nx,
                                     arow + (long long)(l - 1) * kNofW, grow + (long long)(l - 1) * kNofW);
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = b[t];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) cat[t] = a[t];
    // embedded-input gradient: columns 0..63 of  W_0^T d z_0 (+ W_skip[:, :66]^T d z_skip)
    f32x4 ge[4];
    {
      NextLayer nx;
      nx.groups = D > 1 ? g_chain : g_emb;
      nx.jump = first;
      nx.bias_off = zero_bias;
      if (p.skip >= 0) bwd_layer<2, 16, 2, false, false>(cat, nosig, ge, g_emb, zero_bias, st, carry, id, nx, nullptr, nullptr);
      else bwd_layer<2, 8, 2, false, false>(a, nosig, ge, g_emb, zero_bias, st, carry, id, nx, nullptr, nullptr);
    }
    // sin/cos chain rule (embedding.py:42-46): column f = 16 t + 4 g + r of [x | sin f0 x | cos f0 x | ...]
    float ex[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 3; ++t) {          // columns 0..47 cover the 33 xyz columns
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * id.g + r;
        const float gv = ge[t][r];
        float contrib = 0.f;
        int comp = 0;
        if (f < 3) {
          contrib = gv;
          comp = f;
        } else if (f < 33) {
          const int k = (f - 3) / 6, rem = (f - 3) % 6;
          comp = rem % 3;
          const bool is_cos = rem >= 3;
          float fr = p.exyz.freq[0], w = p.exyz.weight[0];
#pragma unroll
          for (int q = 1; q < 5; ++q) {
            fr = k == q ? p.exyz.freq[q] : fr;
            w = k == q ? p.exyz.weight[q] : w;
          }
          const float xc = comp == 0 ? x[0] : (comp == 1 ? x[1] : x[2]);
          float sn, cs;
          sincosf(fr * xc, &sn, &cs);
          contrib = w * fr * (is_cos ? -sn : cs) * gv;
        }
        ex[0] += comp == 0 ? contrib : 0.f;
        ex[1] += comp == 1 ? contrib : 0.f;
        ex[2] += comp == 2 ? contrib : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[c] = xgroup_sum(ex[c]);
    if (valid && id.g == 0 && p.g_pts) {
      p.g_pts[s * 3 + 0] = dx[0] + ex[0];
      p.g_pts[s * 3 + 1] = dx[1] + ex[1];
      p.g_pts[s * 3 + 2] = dx[2] + ex[2];
    }
  }
  wait_vm0();
}

static void emb_params(const mf_embedding& e, EmbParams& o) {
  for (int k = 0; k < 16; ++k) {
    o.freq[k] = k < e.n_freqs ? e.freq[k] : 0.f;
    o.weight[k] = k < e.n_freqs ? e.weight[k] : 0.f;
  }
}

}  // namespace mf

using namespace mf;

extern "C" int32_t mf_nof_points_dump(const mf_nof_desc* d, const void* packed, const mf_embedding* emb_xyz,
                                      const mf_embedding* emb_ind, const float* pts, const float* ind,
                                      int64_t ind_stride, int32_t S, int64_t P, float* out, float* acts,
                                      int64_t stride, float* emb, void* stream) {
  if (!d || !packed || !emb_xyz || !emb_ind || (P > 0 && (!pts || !ind || !out || !acts || !emb)))
    return fail(MF_E_INVALID, "mf_nof_points_dump: null argument");
  NofDumpParams p{};
  if (!nof_layout(*d, p.net.L)) return fail(MF_E_UNSUPPORTED, "mf_nof_points_dump: unsupported NoF configuration");
  if (emb_xyz->in_channels != 3 || emb_xyz->n_freqs > 5 || emb_ind->in_channels != 1 || emb_ind->n_freqs != 16)
    return fail(MF_E_UNSUPPORTED, "mf_nof_points_dump: NoF embeddings must be xyz(3, <=5 freqs) and ind(1, 16 freqs)");
  if (S < 1 || stride < (int64_t)d->D * kNofW + kNofHeadPad || (stride & 3))
    return fail(MF_E_INVALID, "mf_nof_points_dump: S=%d / stride=%lld invalid", S, (long long)stride);
  if (P == 0) return MF_OK;
  p.net.packed = static_cast<const char*>(packed);
  p.net.res_lds = 0;
  emb_params(*emb_xyz, p.exyz);
  emb_params(*emb_ind, p.eind);
  p.pts = pts; p.inputs = nullptr; p.in_stride = 0; p.ind = ind; p.ind_stride = ind_stride; p.P = P; p.S = S; p.out = out; p.acts = acts; p.stride = stride; p.emb = emb;
  p.ring_off = (uint32_t)p.net.L.res_bytes;
  p.buf_bytes = (uint32_t)p.net.L.max_groups * kGroupBytes;
  const size_t lds = p.ring_off + 3 * (size_t)p.buf_bytes;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(nof_points_dump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nof_points_dump: cannot reserve %zu bytes of LDS", lds);
  const long long ntiles = (P + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  hipLaunchKernelGGL(nof_points_dump_kernel, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nof_points_dump");
}

extern "C" int32_t mf_nof_forward_dump(const mf_nof_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                                       const float* xyz, int64_t B, float* out, float* acts, int64_t stride, void* stream) {
  if (!d || !packed || (B > 0 && (!inputs || !xyz || !out || !acts)))
    return fail(MF_E_INVALID, "mf_nof_forward_dump: null argument");
  NofDumpParams p{};
  if (!nof_layout(*d, p.net.L)) return fail(MF_E_UNSUPPORTED, "mf_nof_forward_dump: unsupported NoF configuration");
  if (stride < (int64_t)d->D * kNofW + kNofHeadPad || (stride & 3))
    return fail(MF_E_INVALID, "mf_nof_forward_dump: stride=%lld invalid", (long long)stride);
  if (B == 0) return MF_OK;
  p.net.packed = static_cast<const char*>(packed);
  p.net.res_lds = 0;
  p.pts = xyz; p.inputs = inputs; p.in_stride = in_stride; p.ind = nullptr; p.ind_stride = 0; p.P = B; p.S = 1;
  p.out = out; p.acts = acts; p.stride = stride; p.emb = nullptr;
  p.ring_off = (uint32_t)p.net.L.res_bytes;
  p.buf_bytes = (uint32_t)p.net.L.max_groups * kGroupBytes;
  const size_t lds = p.ring_off + 3 * (size_t)p.buf_bytes;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(nof_points_dump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nof_forward_dump: cannot reserve %zu bytes of LDS", lds);
  const long long ntiles = (B + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  hipLaunchKernelGGL(nof_points_dump_kernel, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nof_forward_dump");
}

extern "C" int64_t mf_nof_bwd_packed_bytes(const mf_nof_desc* d) {
  NetLayout L;
  int skip;
  if (!d || !nof_bwd_layout(*d, L, skip)) { fail(MF_E_UNSUPPORTED, "mf_nof_bwd_packed_bytes: unsupported NoF configuration"); return 0; }
  return L.res_bytes + L.panel_bytes;
}

extern "C" int32_t mf_nof_pack_bwd(const mf_nof_desc* d, void* packed, void* stream) {
  NetLayout L;
  int skip;
  if (!d || !packed) return fail(MF_E_INVALID, "mf_nof_pack_bwd: null argument");
  if (!nof_bwd_layout(*d, L, skip)) return fail(MF_E_UNSUPPORTED, "mf_nof_pack_bwd: unsupported NoF configuration");
  NofBwdPackJob job{};
  const int cin = d->in_channels_xyz + d->extra_feat_dim;
  for (int l = 0; l < d->D; ++l) {
    if (!d->trunk_w[l]) return fail(MF_E_INVALID, "mf_nof_pack_bwd: missing weight pointer for layer %d", l);
    job.W[l] = d->trunk_w[l];
    job.ld[l] = (l == 0 ? cin : L.W) + ((l > 0 && l == skip) ? cin : 0);
  }
  if (!d->head_w) return fail(MF_E_INVALID, "mf_nof_pack_bwd: missing head weight");
  job.D = d->D; job.skip = skip; job.NK = L.NK; job.NP = L.NP;
  job.head_w = d->head_w;
  job.n_head_w = L.n_head * L.W;
  job.off_head_w = L.off_head_w;
  job.res_floats = (int)(L.res_bytes / 4);
  job.res = static_cast<float*>(packed);
  job.panels = reinterpret_cast<float*>(static_cast<char*>(packed) + L.res_bytes);
  job.total_groups = L.panel_bytes / kGroupBytes;
  const long long slots = job.total_groups * 64 > job.res_floats ? job.total_groups * 64 : job.res_floats;
  hipLaunchKernelGGL(pack_nof_bwd_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), job);
  return check_launch("mf_nof_pack_bwd");
}

extern "C" int32_t mf_nof_backward(const mf_nof_desc* d, const void* packed_bwd, const mf_embedding* emb_xyz, int64_t P,
                                   const float* pts, const float* acts, int64_t stride, const float* g_out,
                                   float* gpre, float* g_pts, void* stream) {
  if (!d || !packed_bwd || !emb_xyz || (P > 0 && (!pts || !acts || !g_out || !gpre)))
    return fail(MF_E_INVALID, "mf_nof_backward: null argument");
  NofBwdParams p{};
  if (!nof_bwd_layout(*d, p.net.L, p.skip)) return fail(MF_E_UNSUPPORTED, "mf_nof_backward: unsupported NoF configuration");
  if (emb_xyz->in_channels != 3 || emb_xyz->n_freqs > 5)
    return fail(MF_E_UNSUPPORTED, "mf_nof_backward: xyz embedding must have 3 channels and <= 5 frequencies");
  if (stride < (int64_t)d->D * kNofW + kNofHeadPad || (stride & 3))
    return fail(MF_E_INVALID, "mf_nof_backward: stride %lld invalid", (long long)stride);
  if (P == 0) return MF_OK;
  p.net.packed = static_cast<const char*>(packed_bwd);
  p.net.res_lds = 0;
  emb_params(*emb_xyz, p.exyz);
  p.D = d->D; p.P = P; p.stride = stride;
  p.pts = pts; p.acts = acts; p.g_out = g_out; p.gpre = gpre; p.g_pts = g_pts;
  p.ring_off = (uint32_t)p.net.L.res_bytes;
  p.buf_bytes = (uint32_t)p.net.L.max_groups * kGroupBytes;
  p.dbg = 0;
  if (const char* e = getenv("MF_DEBUG_FLAGS")) p.dbg = atoi(e);   // timing ablations only
  const size_t lds = p.ring_off + 3 * (size_t)p.buf_bytes;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(nof_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nof_backward: cannot reserve %zu bytes of LDS", lds);
  const long long ntiles = (P + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  hipLaunchKernelGGL(nof_backward_kernel, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nof_backward");
}
