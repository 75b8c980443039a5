// mf_pack.hip -- re-order nn.Linear parameters into the MFMA fragment stream.
//
// Replaces nothing in the reference (it keeps (out,in) row-major tensors and calls addmm,
// models/nerf.py:84-99, models/nof.py:70-75); this is the layout transform the fused kernels
// need.  It is a pure permutation + zero padding: every packed float is either one source
// weight or 0.  A panel is 32 output rows; its groups alternate between its two 16-row tiles.
// Group (tile half, k-quad q) holds, for lane (i = lane&15, g = lane>>4) and r = 0..3,
// W[32P + 16*half + i][col(step = 4q + r, g)]  -- see mf_core.hpp for the step -> column maps.
#include <hip/hip_fp16.h>
#include "mf_host.hpp"
#include "mf_layout.hpp"

namespace mf {

struct PackRegion {          // one trunk/extra layer's panels
  const float* W;            // (n_out, n_in)
  int n_in;
  int tiles;                 // panels (n_out / 32)
  int groups;                // groups per panel (2 per k-quad)
  int emb_steps;             // fp32: MFMA k-steps taken from the embedded-input block; bf16: its 16-slot k-steps (0 if none)
  int emb_first;             // 1: emb steps precede hidden steps (trunk); 0: follow them (extra)
  int emb_kind;
  int emb_col0;              // column of embedded feature 0 in W
  int emb_cols;              // embedded columns present in W (features >= this are zero pad)
  int hid_steps;             // fp32: hidden k-steps 4*NK (= W/4); bf16: unused (see hid_batches)
  int hid_batches;           // hidden k-quads (fp32) / 16-k steps (bf16) per tile row: NK; 0 if none
  int bf16;                  // groups hold 8 bf16 per lane (32x32x16 A fragments) instead of 4 fp32
  int hid_col0;              // column of hidden feature 0 in W
  int xyz_cols;
  int n_rows;                // rows present in W (bf16 head panel: 3|9 of its 32; 0 = all)
  int row_terms;             // bf16 head panels of the fast mode (NetLayout::head_tiles): tile row c < n_rows = bf16(W[c]), row row_terms + c =
                             // bf16(W[c] - hi); row_terms = 8 (NeRF sigma / rgb: <= 4 rows) or 16 (NoF head: 3 | 9 rows); 0 = off
  int half;                  // the split terms are IEEE halves of wscale * w (NetLayout::half) instead of bf16
  float wscale;
  int hid_split;             // bf16: groups per hidden k-step ks: 1 = plain, 2 = (hi, lo), 3 = (hi, mid, lo)
  int emb_split;             // bf16: groups per embedded k-step, likewise
  long long dst_group0;      // first group index (in 1 KiB units) within the panel area
};

struct PackJob {
  PackRegion reg[MF_MAX_LAYERS + 3];
  int n_regions;
  long long total_groups;
  float* panels;             // packed + res_bytes
  float* poison;             // half-pair layouts: resident head-bias rows 27 (+ 4 = 31), set to NaN when a weight saturates; else null
};

__device__ inline unsigned short bf16_rne(float x) {        // round to nearest even (inf / nan pass through)
  const unsigned u = __float_as_uint(x);
  const unsigned rnd = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(((u & 0x7f800000u) == 0x7f800000u ? u : rnd) >> 16);
}

__global__ void pack_panels_kernel(PackJob job) {
  const long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 slot
  const long long grp = gidx >> 6;
  if (grp >= job.total_groups) return;
  const int lane = (int)(gidx & 63);
  int ri = 0;
  while (ri + 1 < job.n_regions && grp >= job.reg[ri + 1].dst_group0) ++ri;
  const PackRegion& R = job.reg[ri];
  const long long local = grp - R.dst_group0;
  const int P = (int)(local / R.groups), gi = (int)(local % R.groups);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  float* pv = &v.x;
  if (R.bf16) {
    // bf16 layout (mf_bf16.hpp): panel = ONE 32-row tile, group = A fragment of v_mfma_f32_32x32x16_bf16:
    // lane (i = lane&31, h = lane>>5) holds 8 bf16 = W[32P + i][col(k-step, slot 8h + e)], e = 0..7.
    // Embedded k-step ks is emb_split groups (1: bf16(w); 2: + lo = bf16(w - hi); 3: hi, mid, lo); hidden k-step ks covers
    // features 16 ks + hid_perm2(h, e), hid_split groups likewise.
    const int i = lane & 31, h = lane >> 5;
    int srow = 32 * P + i, rterm = 0;
    bool zero_row = R.n_rows && srow >= R.n_rows;
    if (R.row_terms) {                                       // (hi rows at 0 .., lo rows at 8 ..: both in lane half 0's accumulators)
      zero_row = !(srow < R.n_rows || (srow >= R.row_terms && srow < R.row_terms + R.n_rows));
      if (srow >= R.row_terms) { rterm = 1; srow -= R.row_terms; }
    }
    const float* row = R.W + (long long)(zero_row ? 0 : srow) * R.n_in;
    const int eg = R.emb_split * R.emb_steps;                // groups of the embedded block
    const int hg = R.hid_split * R.hid_batches;              // groups of the hidden block
    const int ge = R.emb_first ? gi : gi - hg;
    const int gh = R.emb_first ? gi - eg : gi;
    unsigned short h8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // term t of a weight: t = 0: bf16(w); 1: bf16(w - hi); 2: bf16(w - hi - mid)   (exact fp32 subtractions)
    const bool half = R.half != 0;
    const float ws = R.wscale;
    bool sat = false;
    auto term = [half, ws, &sat](float w, int t) {
      if (half) {                                           // hi = half(ws w) (saturated: no inf in the stream), lo = half(ws w - hi)
        // |ws w| beyond the half range (|w| >= 2047 at ws = 2^5: no trained NoF is near it) cannot be represented: the weight is
        // saturated AND the network's poison slots are set (job.poison, below) -- every point the network evaluates then comes out
        // NaN.  Round 5 saturated silently: finite WRONG results with no diagnostic; a host-side check would cost a
        // device sync at every re-pack, i.e. at every optimizer step.
        w = w * ws;
        if (!(fabsf(w) <= 65504.f)) sat = true;
        w = fminf(fmaxf(w, -65504.f), 65504.f);
        __half hb = __float2half_rn(w);
        if (t > 0) hb = __float2half_rn(w - __half2float(hb));
        return __half_as_ushort(hb);
      }
      unsigned short b = bf16_rne(w);
      for (int k = 0; k < t; ++k) {
        w -= __uint_as_float((unsigned)b << 16);
        b = bf16_rne(w);
      }
      return b;
    };
    if (zero_row) {
      // zero row of a partial tile
    } else if (ge >= 0 && ge < eg) {
      const int ks = ge / R.emb_split, t = ge % R.emb_split;
      for (int e = 0; e < 8; ++e) {
        const int f = emb_feature2(R.emb_kind, h, 8 * ks + e, R.xyz_cols);
        h8[e] = term((f >= 0 && f < R.emb_cols) ? row[R.emb_col0 + f] : 0.f, t);
      }
    } else if (gh >= 0 && gh < hg) {
      const int ks = gh / R.hid_split, t = R.row_terms ? rterm : gh % R.hid_split;
      for (int e = 0; e < 8; ++e) h8[e] = term(row[R.hid_col0 + 16 * ks + hid_perm2(h, e)], t);
    }
    unsigned* pu = reinterpret_cast<unsigned*>(&v.x);
    for (int w = 0; w < 4; ++w) pu[w] = (unsigned)h8[2 * w] | ((unsigned)h8[2 * w + 1] << 16);
    reinterpret_cast<float4*>(job.panels)[gidx] = v;
    // poison: rows 27 / 31 of the head's 32-row bias vector (the two lane halves' last junk rows: zero weight rows, so their
    // accumulators ARE this value) -- nof_eval_x3 adds that accumulator to every output point: + 0 or + NaN
    if (sat && job.poison) { job.poison[0] = __int_as_float(0x7fc00000); job.poison[4] = __int_as_float(0x7fc00000); }
    return;
  }
  const int b = gi >> 1, half = gi & 1;                 // batch within the panel, tile half
  const int i = lane & 15, g = lane >> 4;
  const int n = 32 * P + 16 * half + i;
  const int eb = R.emb_steps / 4;                       // embedded-input batches (fp32 k-quads)
  const int be = R.emb_first ? b : b - R.hid_batches;   // index within the embedded block
  const int bh = R.emb_first ? b - eb : b;              // index within the hidden block
  const float* row = R.W + (long long)n * R.n_in;
  if (be >= 0 && be < eb) {
    for (int r = 0; r < 4; ++r) {
      const int f = emb_feature(R.emb_kind, g, 4 * be + r, R.xyz_cols);
      pv[r] = (f >= 0 && f < R.emb_cols) ? row[R.emb_col0 + f] : 0.f;
    }
  } else if (bh >= 0 && bh < R.hid_batches) {
    for (int r = 0; r < 4; ++r) pv[r] = row[R.hid_col0 + 16 * bh + 4 * g + r];
  }
  reinterpret_cast<float4*>(job.panels)[gidx] = v;
}

struct ResCopy { const float* src; int dst_off; int n; float scale = 1.f; };
struct ResJob { ResCopy c[2 * MF_MAX_LAYERS + 8]; int n; float* res; int total; };

__global__ void pack_resident_kernel(ResJob job) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= job.total) return;
  float v = 0.f;
  for (int k = 0; k < job.n; ++k) {
    const int o = idx - job.c[k].dst_off;
    if (o >= 0 && o < job.c[k].n) v = job.c[k].src[o] * job.c[k].scale;
  }
  job.res[idx] = v;
}

// bf16 NoF: the fp32 image-index columns of the layers that consume the embedded input, TRANSPOSED:
// [embedded layer][column (kNofIndCols)][row] -- what nof_raybias_kernel contracts with emb(ind) once per ray, one thread
// per row reading its column entries coalesced
struct IndJob { const float* W[MF_MAX_LAYERS]; int n_in[MF_MAX_LAYERS]; int n_layers, rows, col0, cols; float* dst; float scale; };

__global__ void pack_ind_kernel(IndJob job) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_layer = job.rows * kNofIndCols;
  if (idx >= job.n_layers * per_layer) return;
  const int e = idx / per_layer, r = (idx % per_layer) / kNofIndCols, c = idx % kNofIndCols;
  job.dst[(e * kNofIndCols + c) * job.rows + r] = c < job.cols ? job.scale * job.W[e][(long long)r * job.n_in[e] + job.col0 + c] : 0.f;
}

static int launch_pack(const PackJob& job, const ResJob& rj, hipStream_t st) {
  const int rb = (rj.total + 255) / 256;
  hipLaunchKernelGGL(pack_resident_kernel, dim3(rb), dim3(256), 0, st, rj);
  const long long slots = job.total_groups * 64;
  const int pb = (int)((slots + 255) / 256);
  hipLaunchKernelGGL(pack_panels_kernel, dim3(pb), dim3(256), 0, st, job);
  return check_launch("mf_pack");
}

}  // namespace mf

using namespace mf;

extern "C" int64_t mf_nerf_packed_bytes_p(const mf_nerf_desc* d, int32_t precision) {
  NetLayout L;
  if (!d || (precision < MF_PREC_F32 || precision > MF_PREC_BF16X3) || !nerf_layout(*d, L, precision)) { fail(MF_E_UNSUPPORTED, "mf_nerf_packed_bytes: unsupported NeRF configuration"); return 0; }
  return L.res_bytes + L.panel_bytes;
}

extern "C" int64_t mf_nof_packed_bytes_p(const mf_nof_desc* d, int32_t precision) {
  NetLayout L;
  if (!d || (precision < MF_PREC_F32 || precision > MF_PREC_BF16X3) || !nof_layout(*d, L, precision, precision == MF_PREC_F32)) { fail(MF_E_UNSUPPORTED, "mf_nof_packed_bytes: unsupported NoF configuration"); return 0; }
  return L.res_bytes + L.panel_bytes + L.ind_bytes;
}

extern "C" int32_t mf_nerf_pack_p(const mf_nerf_desc* d, int32_t precision, void* packed, void* stream) {
  NetLayout L;
  if (!d || !packed) return fail(MF_E_INVALID, "mf_nerf_pack: null argument");
  if (precision < MF_PREC_F32 || precision > MF_PREC_BF16X3) return fail(MF_E_INVALID, "mf_nerf_pack: precision %d", precision);
  if (!nerf_layout(*d, L, precision)) return fail(MF_E_UNSUPPORTED, "mf_nerf_pack: unsupported NeRF configuration "
                                       "(W=%d D=%d in_channels_xyz=%d)", d->W, d->D, d->in_channels_xyz);
  PackJob job{};
  ResJob rj{};
  long long g0 = 0;
  int nr = 0;
  auto head_region = [&](const float* W, int rows, int groups) {       // NetLayout::head_tiles: one 32-row tile, (hi, lo) ROW pairs
    PackRegion& R = job.reg[nr++];
    R.W = W;
    R.n_in = groups * 16;
    R.tiles = 1;
    R.groups = groups;
    R.emb_first = 1;
    R.emb_kind = kEmbNone;
    R.hid_batches = groups;
    R.bf16 = 1;
    R.n_rows = rows;
    R.row_terms = 8;
    R.hid_split = 1;
    R.emb_split = 1;
    R.dst_group0 = g0;
    g0 += groups;
  };
  for (int l = 0; l < L.n_trunk; ++l) {
    if (l == L.n_trunk - 1 && L.head_tiles) {
      if (!d->sigma_w) return fail(MF_E_INVALID, "mf_nerf_pack: missing sigma weight");
      head_region(d->sigma_w, 1, nerf_sigma_groups(L));
    }
    PackRegion& R = job.reg[nr++];
    const bool has_emb = (L.emb_mask >> l) & 1;
    R.W = l < d->D ? d->trunk_w[l] : d->final_w;
    if (!R.W) return fail(MF_E_INVALID, "mf_nerf_pack: missing weight pointer for layer %d", l);
    R.n_in = (has_emb ? d->in_channels_xyz : 0) + (l > 0 ? L.W : 0);
    R.tiles = L.NP;
    R.groups = trunk_groups(L, l);
    R.emb_steps = has_emb ? L.emb_steps : 0;
    R.emb_first = 1;
    R.emb_kind = kEmbNerfXyz;
    R.emb_col0 = 0;
    R.emb_cols = d->in_channels_xyz;
    R.hid_steps = l > 0 ? L.NK * 4 : 0;
    R.hid_batches = l > 0 ? hidden_batches(L) : 0;
    R.bf16 = L.bf16;
    R.emb_split = L.emb_split ? L.terms : 1;
    R.hid_split = ((L.hsplit_mask >> l) & 1) ? L.terms : 1;
    R.hid_col0 = has_emb ? d->in_channels_xyz : 0;
    R.xyz_cols = d->in_channels_xyz;
    R.dst_group0 = g0;
    g0 += (long long)R.groups * R.tiles;
    const float* b = l < d->D ? d->trunk_b[l] : d->final_b;
    rj.c[rj.n++] = ResCopy{b, L.off_bias_trunk + l * L.W, L.W};
  }
  {
    PackRegion& R = job.reg[nr++];
    const int ext = d->extra_feat_type == MF_EXTRA_NONE ? 0 : d->extra_feat_dim;
    R.W = d->extra_w;
    if (!R.W) return fail(MF_E_INVALID, "mf_nerf_pack: missing extra_encoding weight");
    R.n_in = L.W + ext;
    R.tiles = L.NP / 2;
    R.groups = extra_groups(L);
    R.emb_steps = L.extra_steps;
    R.emb_first = 0;
    R.emb_kind = d->extra_feat_type == MF_EXTRA_DIR ? kEmbDir : (d->extra_feat_type == MF_EXTRA_IND ? kEmbInd : kEmbNone);
    R.emb_col0 = L.W;
    R.emb_cols = ext;
    R.hid_steps = L.NK * 4;
    R.hid_batches = hidden_batches(L);
    R.bf16 = L.bf16;
    R.emb_split = L.emb_split ? L.terms : 1;
    R.hid_split = ((L.hsplit_mask >> L.n_trunk) & 1) ? L.terms : 1;
    R.hid_col0 = 0;
    R.xyz_cols = 0;
    R.dst_group0 = g0;
    g0 += (long long)R.groups * R.tiles;
    rj.c[rj.n++] = ResCopy{d->extra_b, L.off_bias_extra, L.W / 2};
  }
  if (L.head_tiles) {
    if (!d->rgb_w) return fail(MF_E_INVALID, "mf_nerf_pack: missing rgb weight");
    head_region(d->rgb_w, 3, nerf_rgb_groups(L));
  }
  rj.c[rj.n++] = ResCopy{d->sigma_w, L.off_head_w, L.W};
  rj.c[rj.n++] = ResCopy{d->sigma_b, L.off_head_b, 1};
  rj.c[rj.n++] = ResCopy{d->rgb_w, L.off_rgb_w, 3 * (L.W / 2)};
  rj.c[rj.n++] = ResCopy{d->rgb_b, L.off_rgb_b, 3};
  for (int k = 0; k < rj.n; ++k)
    if (!rj.c[k].src) return fail(MF_E_INVALID, "mf_nerf_pack: missing bias/head pointer (%d)", k);
  job.n_regions = nr;
  job.total_groups = g0;
  job.panels = reinterpret_cast<float*>(static_cast<char*>(packed) + L.res_bytes);
  rj.res = static_cast<float*>(packed);
  rj.total = (int)(L.res_bytes / 4);
  if (g0 * kGroupBytes != L.panel_bytes) return fail(MF_E_INVALID, "mf_nerf_pack: layout mismatch");
  return launch_pack(job, rj, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mf_nof_pack_p(const mf_nof_desc* d, int32_t precision, void* packed, void* stream) {
  NetLayout L;
  if (!d || !packed) return fail(MF_E_INVALID, "mf_nof_pack: null argument");
  if (precision < MF_PREC_F32 || precision > MF_PREC_BF16X3) return fail(MF_E_INVALID, "mf_nof_pack: precision %d", precision);
  if (!nof_layout(*d, L, precision, precision == MF_PREC_F32)) return fail(MF_E_UNSUPPORTED, "mf_nof_pack: unsupported NoF configuration "
                                      "(W=%d D=%d in_channels_xyz=%d extra_feat_dim=%d)", d->W, d->D,
                                      d->in_channels_xyz, d->extra_feat_dim);
  PackJob job{};
  ResJob rj{};
  long long g0 = 0;
  int nr = 0;
  const int cin = d->in_channels_xyz + d->extra_feat_dim;
  // NetLayout::half (the NoF under MF_PREC_BF16X3): weights at 2^kNofHalfSW, biases at the accumulators' 2^(kNofHalfSA + kNofHalfSW)
  const float wscale = L.half ? (float)(1 << kNofHalfSW) : 1.f, bscale = L.half ? (float)(1 << (kNofHalfSA + kNofHalfSW)) : 1.f;
  for (int l = 0; l < L.n_trunk; ++l) {
    PackRegion& R = job.reg[nr++];
    const bool has_emb = (L.emb_mask >> l) & 1;
    R.W = d->trunk_w[l];
    if (!R.W || !d->trunk_b[l]) return fail(MF_E_INVALID, "mf_nof_pack: missing parameter pointer for layer %d", l);
    R.n_in = (has_emb ? cin : 0) + (l > 0 ? L.W : 0);
    R.tiles = L.NP;
    R.groups = trunk_groups(L, l);
    R.emb_steps = has_emb ? L.emb_steps : 0;
    R.emb_first = 1;
    R.emb_kind = kEmbNofIn;
    R.emb_col0 = 0;
    R.emb_cols = cin;
    R.hid_steps = l > 0 ? L.NK * 4 : 0;
    R.hid_batches = l > 0 ? hidden_batches(L) : 0;
    R.bf16 = L.bf16;
    R.emb_split = L.emb_split ? L.terms : 1;
    R.hid_split = ((L.hsplit_mask >> l) & 1) ? L.terms : 1;
    R.half = L.half;
    R.wscale = wscale;
    R.hid_col0 = has_emb ? cin : 0;
    R.xyz_cols = d->in_channels_xyz;
    R.dst_group0 = g0;
    g0 += (long long)R.groups * R.tiles;
    rj.c[rj.n++] = ResCopy{d->trunk_b[l], L.off_bias_trunk + l * L.W, L.W, bscale};
  }
  if (!d->head_w || !d->head_b) return fail(MF_E_INVALID, "mf_nof_pack: missing head parameters");
  if (L.bf16) {
    PackRegion& R = job.reg[nr++];
    R.W = d->head_w;
    R.n_in = L.W;
    R.tiles = 1;
    R.groups = head_groups(L);
    R.emb_first = 1;
    R.emb_kind = kEmbNone;
    R.hid_batches = L.NK;
    R.bf16 = 1;
    R.n_rows = L.n_head;
    R.hid_split = L.head_tiles ? 1 : L.terms;
    R.row_terms = L.head_tiles ? 16 : 0;
    R.emb_split = 1;
    R.half = L.half;
    R.wscale = wscale;
    R.dst_group0 = g0;
    g0 += R.groups;
  }
  rj.c[rj.n++] = ResCopy{d->head_w, L.off_head_w, L.n_head * L.W};
  rj.c[rj.n++] = ResCopy{d->head_b, L.off_head_b, L.n_head, bscale};
  job.n_regions = nr;
  job.total_groups = g0;
  job.panels = reinterpret_cast<float*>(static_cast<char*>(packed) + L.res_bytes);
  job.poison = L.half ? static_cast<float*>(packed) + L.off_head_b + 27 : nullptr;       // (the resident kernel zero-fills them first)
  rj.res = static_cast<float*>(packed);
  rj.total = (int)(L.res_bytes / 4);
  if (g0 * kGroupBytes != L.panel_bytes) return fail(MF_E_INVALID, "mf_nof_pack: layout mismatch");
  if (L.bf16) {
    IndJob ij{};
    for (int l = 0; l < L.n_trunk; ++l)
      if ((L.emb_mask >> l) & 1) {
        ij.W[ij.n_layers] = d->trunk_w[l];
        ij.n_in[ij.n_layers] = cin + (l > 0 ? L.W : 0);
        ++ij.n_layers;
      }
    ij.rows = L.W; ij.col0 = d->in_channels_xyz; ij.cols = d->extra_feat_dim;
    ij.scale = bscale;                    // (the per-ray index bias b + W_ind emb(ind) comes out at the accumulators' scale)
    ij.dst = reinterpret_cast<float*>(static_cast<char*>(packed) + L.res_bytes + L.panel_bytes);
    const int total = ij.n_layers * ij.rows * kNofIndCols;
    hipLaunchKernelGGL(pack_ind_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), ij);
  }
  return launch_pack(job, rj, static_cast<hipStream_t>(stream));
}

extern "C" int64_t mf_nerf_packed_bytes(const mf_nerf_desc* d) { return mf_nerf_packed_bytes_p(d, MF_PREC_F32); }
extern "C" int64_t mf_nof_packed_bytes(const mf_nof_desc* d) { return mf_nof_packed_bytes_p(d, MF_PREC_F32); }
extern "C" int32_t mf_nerf_pack(const mf_nerf_desc* d, void* packed, void* stream) { return mf_nerf_pack_p(d, MF_PREC_F32, packed, stream); }
extern "C" int32_t mf_nof_pack(const mf_nof_desc* d, void* packed, void* stream) { return mf_nof_pack_p(d, MF_PREC_F32, packed, stream); }
