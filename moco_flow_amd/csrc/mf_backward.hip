// mf_backward.hip -- input-gradient chain of the canonical NeRF over the training forward's dump.
//
// What it replaces: the autograd backward of models/nerf.py:78-102 that torch records for
// `loss.backward()` (trainer/base.py:188-197).  Per ray-sample, given dL/d[rgb, sigma]:
//     d_o   = d_rgb * rgb * (1 - rgb)                         (nn.Sigmoid, nerf.py:57-59)
//     d_e   = (W_rgb^T d_o)              * [e   > 0]          (extra_encoding ReLU, nerf.py:98)
//     d_g   =  W_e[:, :W]^T d_e                               (xyz_encoding_final has no activation)
//     d_z_{D-1} = (W_f^T d_g + w_sigma d_sigma) * [h_D > 0]
//     d_z_{l-1} = (W_l[:, hidden]^T d_z_l)      * [h_l > 0]   l = D-1 .. 1
// i.e. the same register-resident transposed MLP as the forward, run on W^T: the nine W-wide
// contractions go to v_mfma_f32_16x16x4_f32 through the fragment stream (packed from the transposed
// weights by mf_nerf_pack_bwd), the ReLU masks come from the forward's activation dump, and every
// pre-activation gradient is stored in the dump's own layout [d_z_0 .. d_z_{D-1} | d_g | d_e] so that
// the weight gradients are plain library GEMMs  dW_l = d_z_l^T h_l  on (dump, this buffer).
// d_sigma enters the xyz_encoding_final^T layer as one extra k-step (a 4-step "embedded" block whose
// only live slot carries d_sigma against the packed sigma.weight row).
#include "mf_host.hpp"
#include "mf_layout.hpp"
#include "mf_nets.hpp"
#include "mf_bwd.hpp"
#include <cstdlib>

namespace mf {

// Backward "network" of a NeRF(D, W=256): layer 0 = extra_encoding^T (K = W/2), layer 1 =
// xyz_encoding_final^T (+ the d_sigma block), layers 2..D = trunk layers D-1..1 transposed.
// Behind them, the embedded-input gradient (ABI v9): layer D+1 = xyz_encoding_1[:, :64]^T (64 output rows = embedded
// features, K = W) and, with one skip layer, layer D+2 = that layer's embedded columns transposed:
//     d emb = W_0[:, :63]^T d_z_0 + W_skip[:, :63]^T d_z_skip.
// (More than one skip layer: not built; n_head = number of these layers, 0 = g_emb unsupported.)
inline int bwd_skip_layer(const mf_nerf_desc& d) {      // the single skip layer, 0 = none, -1 = several
  int s = 0;
  for (int l = 1; l < d.D; ++l)
    if ((d.skip_mask >> l) & 1u) { if (s) return -1; s = l; }
  return s;
}
inline bool nerf_bwd_layout(const mf_nerf_desc& d, NetLayout& L) {
  NetLayout F;
  if (!nerf_layout(d, F, 0) || F.W != 256) return false;
  L = NetLayout{};
  L.W = F.W; L.NK = F.NK; L.NP = F.NP;
  L.n_trunk = d.D + 1;
  L.n_head = bwd_skip_layer(d) < 0 ? 0 : (bwd_skip_layer(d) > 0 ? 2 : 1);
  L.emb_steps = kBwdSigSteps;
  L.emb_mask = 2u;
  L.relu_mask = 0;
  L.extra_steps = -1;
  int off = 0;
  L.off_bias_trunk = off; off += L.W;            // one shared all-zero bias vector
  L.off_rgb_w = off; off += 3 * (L.W / 2);       // rgb.0.weight, natural order (VALU prologue)
  L.res_bytes = round_up((int64_t)off * 4, kGroupBytes);
  L.max_groups = 2 * (L.NK + 1);
  L.panel_bytes = ((int64_t)(L.NK + 2 * (L.NK + 1) + 2 * L.NK * (d.D - 1)) * L.NP +
                   (int64_t)L.n_head * 2 * L.NK * 2 /* 64 rows = 2 panels */) * kGroupBytes;
  return true;
}
MF_HD int bwd_groups(const NetLayout& L, int layer) {
  return layer == 0 ? L.NK : (layer == 1 ? 2 * (L.NK + 1) : 2 * L.NK);
}

int device_cus();   // mf_forward.hip

// ------------------------------------------------------------------ packing (transposed fragment stream)
struct BwdPackJob {
  const float* W[MF_MAX_LAYERS + 3];   // forward weight feeding backward layer i
  int ld[MF_MAX_LAYERS + 3];           // its row length (forward in-features)
  int col0[MF_MAX_LAYERS + 3];         // first hidden column
  int groups[MF_MAX_LAYERS + 3];
  long long g0[MF_MAX_LAYERS + 4];
  int ncols[MF_MAX_LAYERS + 3];        // forward input columns present (output rows beyond are zero): emb layers
  int n_layers, NP;
  const float* sigma_w;
  const float* rgb_w;
  int res_floats, off_rgb_w, n_rgb_w;
  float* res;
  float* panels;
  long long total_groups;
};

__global__ void pack_bwd_kernel(BwdPackJob job) {
  const long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx < job.res_floats) {
    const int o = (int)gidx - job.off_rgb_w;
    job.res[gidx] = (o >= 0 && o < job.n_rgb_w) ? job.rgb_w[o] : 0.f;
  }
  const long long grp = gidx >> 6;
  if (grp >= job.total_groups) return;
  const int lane = (int)(gidx & 63);
  int li = 0;
  while (li + 1 < job.n_layers && grp >= job.g0[li + 1]) ++li;
  const long long local = grp - job.g0[li];
  const int P = (int)(local / job.groups[li]), gi = (int)(local % job.groups[li]);
  const int b = gi >> 1, half = gi & 1;
  const int i = lane & 15, g = lane >> 4;
  const int n = 32 * P + 16 * half + i;            // output feature of the backward layer = forward input column
  const int sig = li == 1 ? 1 : 0;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sig && b == 0) {
    if (g == 0) v.x = job.sigma_w[n];              // step 0 of lane group 0 carries d_sigma
  } else {
    const int bh = b - sig;
    float* pv = &v.x;
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * bh + 4 * g + r;           // forward output row
      pv[r] = n < job.ncols[li] ? job.W[li][(long long)k * job.ld[li] + job.col0[li] + n] : 0.f;
    }
  }
  reinterpret_cast<float4*>(job.panels)[gidx] = v;
}

// ------------------------------------------------------------------ the kernel
struct BwdParams {
  NetDev net;
  int D;
  long long P, stride;
  const float* g_out;      // (P,4)  dL/d[rgb, sigma]
  const float* acts;       // (P,stride) forward dump
  const float* rgbsigma;   // (P,4)
  float* gpre;             // (round_up(P,128), stride)
  float* ghead;            // (P,4)  [d rgb pre-sigmoid, d sigma]
  float* g_emb;            // (P,64) dL/d embedded input (natural column order, column 63 = 0), or null
  int skip;                // the skip layer (0 = none)
  uint32_t ring_off, buf_bytes;
  int dbg;
};

__global__ __launch_bounds__(kThreads, 2) void nerf_backward_kernel(BwdParams p) {
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
  const char* first = net.packed + net.L.res_bytes;
  st.start(first, bwd_groups(net.L, 0), id);
  carry.load(st.slot_off(0) + id.lane * 16, zero_bias, id.g);
  const int D = p.D, W = net.L.W;
  const uint32_t rgbw = net.res_lds + net.L.off_rgb_w * 4;
  const int last_layer = p.g_emb ? D + net.L.n_head : D;      // the embedded-input layers run only when asked for
  auto next_of = [&](int layer) {                 // the layer after `layer` in program order
    NextLayer f;
    const int nl = layer + 1 <= last_layer ? layer + 1 : 0;
    f.groups = nl > D ? 2 * net.L.NK : bwd_groups(net.L, nl);
    f.jump = nl == 0 ? first : nullptr;
    f.bias_off = zero_bias;
    return f;
  };
  const long long ntiles = (p.P + kTile - 1) / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long s = tile * kTile + id.wave * kWaveSamples + id.j;
    const bool valid = s < p.P;
    const long long ss = valid ? s : p.P - 1;
    const float4 go = *reinterpret_cast<const float4*>(p.g_out + ss * 4);
    const float4 rs = *reinterpret_cast<const float4*>(p.rgbsigma + ss * 4);
    const float d0 = go.x * rs.x * (1.f - rs.x), d1 = go.y * rs.y * (1.f - rs.y), d2 = go.z * rs.z * (1.f - rs.z);
    if (valid && id.g == 0) *reinterpret_cast<float4*>(p.ghead + s * 4) = make_float4(d0, d1, d2, go.w);
    const float* arow = p.acts + ss * p.stride;
    float* grow = p.gpre + s * p.stride;           // rows up to round_up(P,128) exist
    // d_e = (W_rgb^T d_o) * [e > 0], in the B-operand layout (k-tile t, lane group g: features 16t+4g+r)
    f32x4 de[8];
    {
      const float* erow = arow + (long long)(D + 1) * W;
      float* gerow = grow + (long long)(D + 1) * W;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f32x4 w0 = lds_f4(rgbw + (0 * (W / 2) + 16 * t + 4 * id.g) * 4);
        const f32x4 w1 = lds_f4(rgbw + (1 * (W / 2) + 16 * t + 4 * id.g) * 4);
        const f32x4 w2 = lds_f4(rgbw + (2 * (W / 2) + 16 * t + 4 * id.g) * 4);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(erow + 16 * t + 4 * id.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_fmaf(w2[r], d2, __builtin_fmaf(w1[r], d1, w0[r] * d0));
          de[t][r] = e4[r] > 0.f ? v : 0.f;
        }
        *reinterpret_cast<f32x4*>(gerow + 16 * t + 4 * id.g) = de[t];
      }
    }
    const float nosig[kBwdSigSteps] = {0.f, 0.f, 0.f, 0.f};
    const float sig[kBwdSigSteps] = {id.g == 0 ? go.w : 0.f, 0.f, 0.f, 0.f};
    f32x4 a[16], b[16];
    // layer 0: d_g = W_e[:, :W]^T d_e          (no activation on xyz_encoding_final)
    bwd_layer<2, 8, 8, false, true>(de, nosig, a, bwd_groups(net.L, 0), zero_bias, st, carry, id, next_of(0), nullptr,
                           grow + (long long)D * W);
    // layer 1: d_z_{D-1} = (W_f^T d_g + w_sigma d_sigma) * [h_D > 0]
    bwd_layer<3, 16, 8, true, true>(a, sig, b, bwd_groups(net.L, 1), zero_bias, st, carry, id, next_of(1),
                           arow + (long long)(D - 1) * W, grow + (long long)(D - 1) * W);
    // layers 2..D: d_z_{l-1} = (W_l^T d_z_l) * [h_l > 0],  l = D-1 .. 1
    for (int i = 2; i <= D; ++i) {
      const int l = D + 1 - i;
      bwd_layer<2, 16, 8, true, true>(b, nosig, a, bwd_groups(net.L, i), zero_bias, st, carry, id, next_of(i),
                             arow + (long long)(l - 1) * W, grow + (long long)(l - 1) * W);
#pragma unroll
      for (int t = 0; t < 16; ++t) b[t] = a[t];
    }
    if (p.g_emb) {
      // d emb = W_0[:, :64]^T d_z_0 (+ W_skip[:, :64]^T d_z_skip): 64 output rows = 2 panels, K = W each
      f32x4 ge[4];
      bwd_layer<2, 16, 2, false, false>(b, nosig, ge, 2 * net.L.NK, zero_bias, st, carry, id, next_of(D + 1), nullptr, nullptr);
      if (p.skip > 0) {
        // d_z_skip was stored by this very lane several layers ago (its hooks' vmcnt waits retired the stores)
        const float* zrow = grow + (long long)p.skip * W;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          a[2 * t] = *reinterpret_cast<const f32x4*>(zrow + 32 * t + 4 * id.g);
          a[2 * t + 1] = *reinterpret_cast<const f32x4*>(zrow + 32 * t + 16 + 4 * id.g);
        }
        f32x4 g2[4];
        bwd_layer<2, 16, 2, false, false>(a, nosig, g2, 2 * net.L.NK, zero_bias, st, carry, id, next_of(D + 2), nullptr, nullptr);
#pragma unroll
        for (int t = 0; t < 4; ++t) ge[t] += g2[t];
      }
      if (valid) {
        float* er = p.g_emb + s * 64;
        *reinterpret_cast<f32x4*>(er + 4 * id.g) = ge[0];
        *reinterpret_cast<f32x4*>(er + 16 + 4 * id.g) = ge[1];
        *reinterpret_cast<f32x4*>(er + 32 + 4 * id.g) = ge[2];
        *reinterpret_cast<f32x4*>(er + 48 + 4 * id.g) = ge[3];
      }
    }
  }
  wait_vm0();
}

// embedding.py:42-46 differentiated: g_x[c] = g_emb[c] + sum_k f_k (emb[cos_kc] g_emb[sin_kc] - emb[sin_kc] g_emb[cos_kc]),
// emb = the embedded input itself (its sin / cos columns already carry the per-frequency weight w_k).
struct EmbBwdParams { const float* g_emb; long long g_stride; const float* emb; long long e_stride; long long P; int C, F; float freq[16]; float* g_x; };
// A workgroup takes 64 samples: their gradient and embedding rows go through LDS with row-contiguous (coalesced) loads --
// one thread per sample walking its own two rows column by column re-fetched every cache line 16 times (rocprofv3: 6x the
// algorithmic HBM bytes, 177 us for 393 k samples) -- then one thread per (sample, component) does the sum; odd LDS
// pitch: conflict-free.
constexpr int kEmbBwdSamples = 64;
__global__ __launch_bounds__(256) void embed_backward_kernel(EmbBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float eb_sm[];
  const int ncols = p.C * (2 * p.F + 1), pitch = ncols | 1;
  float* gs = eb_sm;
  float* es = eb_sm + kEmbBwdSamples * pitch;
  const long long s0 = (long long)blockIdx.x * kEmbBwdSamples;
  const int ns = (int)((p.P - s0) < kEmbBwdSamples ? (p.P - s0) : kEmbBwdSamples);
  for (int i = threadIdx.x; i < ns * ncols; i += blockDim.x) {
    const int r = i / ncols, c = i - r * ncols;
    gs[r * pitch + c] = p.g_emb[(s0 + r) * p.g_stride + c];
    es[r * pitch + c] = p.emb[(s0 + r) * p.e_stride + c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ns * p.C; i += blockDim.x) {
    const int r = i / p.C, c = i - r * p.C;
    const float* g = gs + r * pitch;
    const float* e = es + r * pitch;
    float acc = g[c];
    for (int k = 0; k < p.F; ++k) {
      const int is = p.C + 2 * p.C * k + c, ic = is + p.C;
      acc += p.freq[k] * (e[ic] * g[is] - e[is] * g[ic]);
    }
    p.g_x[(s0 + r) * p.C + c] = acc;
  }
}

}  // namespace mf

using namespace mf;

extern "C" int64_t mf_nerf_bwd_packed_bytes(const mf_nerf_desc* d) {
  NetLayout L;
  if (!d || !nerf_bwd_layout(*d, L)) { fail(MF_E_UNSUPPORTED, "mf_nerf_bwd_packed_bytes: unsupported NeRF configuration"); return 0; }
  return L.res_bytes + L.panel_bytes;
}

extern "C" int32_t mf_nerf_pack_bwd(const mf_nerf_desc* d, void* packed, void* stream) {
  NetLayout L;
  if (!d || !packed) return fail(MF_E_INVALID, "mf_nerf_pack_bwd: null argument");
  if (!nerf_bwd_layout(*d, L)) return fail(MF_E_UNSUPPORTED, "mf_nerf_pack_bwd: unsupported NeRF configuration (W=%d D=%d)", d->W, d->D);
  BwdPackJob job{};
  const int ext = d->extra_feat_type == MF_EXTRA_NONE ? 0 : d->extra_feat_dim;
  job.n_layers = d->D + 1;
  job.NP = L.NP;
  long long g0 = 0;
  for (int i = 0; i <= d->D; ++i) {
    if (i == 0) { job.W[i] = d->extra_w; job.ld[i] = L.W + ext; job.col0[i] = 0; }
    else if (i == 1) { job.W[i] = d->final_w; job.ld[i] = L.W; job.col0[i] = 0; }
    else {
      const int l = d->D + 1 - i;
      const bool skip = (d->skip_mask >> l) & 1u;
      job.W[i] = d->trunk_w[l];
      job.ld[i] = L.W + (skip ? d->in_channels_xyz : 0);
      job.col0[i] = skip ? d->in_channels_xyz : 0;
    }
    if (!job.W[i]) return fail(MF_E_INVALID, "mf_nerf_pack_bwd: missing weight pointer (backward layer %d)", i);
    job.groups[i] = bwd_groups(L, i);
    job.g0[i] = g0;
    g0 += (long long)job.groups[i] * L.NP;
  }
  for (int i = 0; i <= d->D; ++i) job.ncols[i] = 1 << 30;
  for (int e = 0; e < L.n_head; ++e) {                      // embedded-input gradient layers
    const int i = d->D + 1 + e, l = e == 0 ? 0 : bwd_skip_layer(*d);
    job.W[i] = d->trunk_w[l];
    if (!job.W[i]) return fail(MF_E_INVALID, "mf_nerf_pack_bwd: missing weight pointer (layer %d)", l);
    job.ld[i] = (l == 0 ? 0 : L.W) + d->in_channels_xyz;
    job.col0[i] = 0;
    job.ncols[i] = d->in_channels_xyz;
    job.groups[i] = 2 * L.NK;
    job.g0[i] = g0;
    g0 += (long long)job.groups[i] * 2;
  }
  job.n_layers = d->D + 1 + L.n_head;
  job.g0[job.n_layers] = g0;
  if (!d->sigma_w || !d->rgb_w) return fail(MF_E_INVALID, "mf_nerf_pack_bwd: missing sigma / rgb weight");
  job.sigma_w = d->sigma_w;
  job.rgb_w = d->rgb_w;
  job.res_floats = (int)(L.res_bytes / 4);
  job.off_rgb_w = L.off_rgb_w;
  job.n_rgb_w = 3 * (L.W / 2);
  job.res = static_cast<float*>(packed);
  job.panels = reinterpret_cast<float*>(static_cast<char*>(packed) + L.res_bytes);
  job.total_groups = g0;
  if (g0 * kGroupBytes != L.panel_bytes) return fail(MF_E_INVALID, "mf_nerf_pack_bwd: layout mismatch");
  const long long slots = g0 * 64;
  hipLaunchKernelGGL(pack_bwd_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), job);
  return check_launch("mf_nerf_pack_bwd");
}

extern "C" int32_t mf_nerf_backward(const mf_nerf_desc* d, const void* packed_bwd, int64_t P, const float* g_out,
                                    const float* acts, int64_t stride, const float* rgbsigma, float* gpre,
                                    float* ghead, void* stream) {
  return mf_nerf_backward_x(d, packed_bwd, P, g_out, acts, stride, rgbsigma, gpre, ghead, nullptr, stream);
}

extern "C" int32_t mf_embedding_backward(const mf_embedding* e, const float* g_emb, int64_t g_stride, const float* emb,
                                         int64_t e_stride, int64_t P, float* g_x, void* stream) {
  if (!e || P < 0 || (P > 0 && (!g_emb || !emb || !g_x))) return fail(MF_E_INVALID, "mf_embedding_backward: null argument");
  if (e->in_channels < 1 || e->n_freqs < 0 || e->n_freqs > MF_MAX_FREQS) return fail(MF_E_INVALID, "mf_embedding_backward: bad embedding");
  const int width = e->in_channels * (2 * e->n_freqs + 1);
  if (g_stride < width || e_stride < width) return fail(MF_E_INVALID, "mf_embedding_backward: strides shorter than %d", width);
  if (P == 0) return MF_OK;
  EmbBwdParams p{};
  p.g_emb = g_emb; p.g_stride = g_stride; p.emb = emb; p.e_stride = e_stride; p.P = P; p.C = e->in_channels; p.F = e->n_freqs;
  for (int k = 0; k < e->n_freqs; ++k) p.freq[k] = e->weight[k] != 0.f ? e->freq[k] : 0.f;   // muted frequency: emb columns are 0 anyway
  p.g_x = g_x;
  const int ncols = p.C * (2 * p.F + 1);
  const size_t lds = (size_t)2 * kEmbBwdSamples * (ncols | 1) * sizeof(float);
  // wide embeddings (in_channels (2 N_freqs + 1) > 127, e.g. Embedding(4, 16)) need more than the 64 KiB a launch gets by
  // default: raise the kernel's limit up to the CU's 160 KiB, refuse beyond
  if (lds > 160 * 1024) return fail(MF_E_UNSUPPORTED, "mf_embedding_backward: %d embedded columns exceed the staged width (312)", ncols);
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(embed_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_embedding_backward: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(embed_backward_kernel, dim3((unsigned)((P + kEmbBwdSamples - 1) / kEmbBwdSamples)), dim3(256), lds,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("mf_embedding_backward");
}

extern "C" int32_t mf_nerf_backward_x(const mf_nerf_desc* d, const void* packed_bwd, int64_t P, const float* g_out,
                                      const float* acts, int64_t stride, const float* rgbsigma, float* gpre,
                                      float* ghead, float* g_emb, void* stream) {
  if (!d || !packed_bwd || (P > 0 && (!g_out || !acts || !rgbsigma || !gpre || !ghead)))
    return fail(MF_E_INVALID, "mf_nerf_backward: null argument");
  BwdParams p{};
  if (!nerf_bwd_layout(*d, p.net.L)) return fail(MF_E_UNSUPPORTED, "mf_nerf_backward: unsupported NeRF configuration");
  if (g_emb && p.net.L.n_head == 0) return fail(MF_E_UNSUPPORTED, "mf_nerf_backward: the embedded-input gradient is built for at most one skip layer");
  p.g_emb = g_emb;
  p.skip = bwd_skip_layer(*d) > 0 ? bwd_skip_layer(*d) : 0;
  if (stride < (int64_t)(d->D + 1) * d->W + d->W / 2 || (stride & 3))
    return fail(MF_E_INVALID, "mf_nerf_backward: stride %lld too small or not a multiple of 4", (long long)stride);
  if (P == 0) return MF_OK;
  p.net.packed = static_cast<const char*>(packed_bwd);
  p.net.res_lds = 0;
  p.D = d->D; p.P = P; p.stride = stride;
  p.g_out = g_out; p.acts = acts; p.rgbsigma = rgbsigma; p.gpre = gpre; p.ghead = ghead;
  p.ring_off = (uint32_t)p.net.L.res_bytes;
  p.buf_bytes = (uint32_t)p.net.L.max_groups * kGroupBytes;
  p.dbg = 0;
  if (const char* e = getenv("MF_DEBUG_FLAGS")) p.dbg = atoi(e);   // timing ablations only
  const size_t lds = p.ring_off + 3 * (size_t)p.buf_bytes;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nerf_backward: cannot reserve %zu bytes of LDS", lds);
  const long long ntiles = (P + kTile - 1) / kTile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  hipLaunchKernelGGL(nerf_backward_kernel, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nerf_backward");
}
