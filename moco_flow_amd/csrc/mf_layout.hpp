// mf_layout.hpp -- packed-buffer layouts derived from the C-ABI descriptors (host + device).
#pragma once
#include "../../include/mocoflow_hip.h"
#include "mf_core.hpp"

namespace mf {

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// returns false (layout zeroed) for configurations the kernels do not implement
// `bf16` = the MF_PREC_* value: 0 fp32, 1 bf16, 2 bf16x3 (bf16 layout with more (hi, lo) split ranges, mf_bf16.hpp)
inline bool nerf_layout(const mf_nerf_desc& d, NetLayout& L, int bf16 = 0) {
  L = NetLayout{};
  const bool x3 = bf16 == MF_PREC_BF16X3;
  L.bf16 = bf16 ? 1 : 0;
  if (d.W != 256 && d.W != 128) return false;
  if (d.D < 2 || d.D + 1 > MF_MAX_LAYERS) return false;
  // the embedded xyz block has slots for 3 channels x <= 10 frequencies = 63 features; a narrower in_channels_xyz (the
  // reference's default is 33, models/nerf.py:6-12) leaves the upper features without a weight column (packed as zeros), a
  // 64th column only ever sees the reference's zero padding (rendering.py:127-129)
  if (d.in_channels_xyz < 3 || d.in_channels_xyz > 64) return false;
  if (d.extra_feat_type != MF_EXTRA_NONE && (d.extra_feat_dim < 1 || d.extra_feat_dim > 32)) return false;
  if (d.skip_mask & 1u) return false;                 // layer 0 already takes the input
  if (d.skip_mask >> d.D) return false;
  L.W = d.W;
  L.NK = d.W / 16;
  L.NP = d.W / 32;
  L.n_trunk = d.D + 1;                                // + xyz_encoding_final (no ReLU)
  if (bf16 && d.W != 256) return false;
  L.emb_steps = bf16 ? kKsNerfXyz : kStepsNerfXyz;     // bf16: 16-slot k-steps (mf_bf16.hpp); fp32: 4-k MFMA steps
  L.emb_split = x3 ? 1 : 0;                            // bf16: the NeRF's encodings are plain bf16 operands (mf_bf16.hpp); x3: split
  L.hsplit_mask = x3 ? ((1u << (d.D + 2)) - 1u) & ~1u : 0u;   // x3: every hidden range (trunk, final, extra_encoding) as (hi, lo) pairs
  L.terms = 2;
  L.emb_mask = 1u | d.skip_mask;
  L.relu_mask = (1u << d.D) - 1u;
  switch (d.extra_feat_type) {
    case MF_EXTRA_NONE: L.extra_steps = 0; break;
    case MF_EXTRA_IND: L.extra_steps = bf16 ? kKsInd : kStepsInd; if (d.extra_feat_dim < 1) return false; break;
    case MF_EXTRA_DIR: L.extra_steps = bf16 ? kKsDir : kStepsDir; if (d.extra_feat_dim < 3) return false; break;
    default: return false;
  }
  int off = 0;
  L.off_bias_trunk = off; off += L.n_trunk * L.W;
  L.off_bias_extra = off; off += L.W / 2;
  L.off_head_w = off; off += L.W;
  L.off_head_b = off; off += 4;
  L.off_rgb_w = off; off += 3 * (L.W / 2);
  L.off_rgb_b = off; off += 4;
  L.n_head = 1;
  L.head_tiles = bf16 == MF_PREC_BF16 ? 1 : 0;        // fast bf16 mode: sigma / rgb heads on the matrix pipe (NetLayout::head_tiles)
  L.res_bytes = round_up((int64_t)off * 4, kGroupBytes);
  int64_t groups = 0;
  L.max_groups = 0;
  // stream order: trunk layers 0 .. D-1, [sigma head panel], xyz_encoding_final, extra_encoding, [rgb head panel]
  for (int l = 0; l < L.n_trunk; ++l) {
    if (l == L.n_trunk - 1) groups += nerf_sigma_groups(L);
    const int g = trunk_groups(L, l);
    groups += (int64_t)g * L.NP;
    if ((x3 ? panel_cap(g) : g) > L.max_groups) L.max_groups = x3 ? panel_cap(g) : g;
  }
  const int ge = extra_groups(L);
  groups += (int64_t)ge * (L.NP / 2);      // (W/2)-wide layer: NP/2 panels in either layout
  if ((x3 ? panel_cap(ge) : ge) > L.max_groups) L.max_groups = x3 ? panel_cap(ge) : ge;
  groups += nerf_rgb_groups(L);
  L.panel_bytes = groups * kGroupBytes;
  return true;
}

// wide_ok: W = 256 too (fp32 only) -- the reference's bare `NoF()` (models/nof.py:7-15: D = 8, W = 256, skips = [4]); built for the
// module-level forward alone (mf_nof_forward and its packer): the fused passes keep three networks' resident blocks beside the ring
// and have no room for 256-wide NoFs, every other entry point keeps rejecting them
inline bool nof_layout(const mf_nof_desc& d, NetLayout& L, int bf16 = 0, bool wide_ok = false) {
  L = NetLayout{};
  const bool x3 = bf16 == MF_PREC_BF16X3;
  L.bf16 = bf16 ? 1 : 0;
  if (d.W != 128 && !(wide_ok && !bf16 && d.W == 256)) return false;
  if (d.D < 2 || d.D > MF_MAX_LAYERS) return false;
  // slots for 3 x <= 5 xyz frequencies (33 features) and 1 x <= 16 index frequencies (33); narrower blocks leave the upper
  // features without a column (models/nof.py:7-15: in_channels_xyz = 33, extra_feat_dim = 0 by default)
  if (d.in_channels_xyz < 3 || d.in_channels_xyz > 33 || d.extra_feat_dim < 0 || d.extra_feat_dim > 33) return false;
  if ((d.skip_mask & 1u) || (d.skip_mask >> d.D)) return false;
  L.W = d.W;
  L.NK = d.W / 16;
  L.NP = d.W / 32;
  L.n_trunk = d.D;
  L.emb_steps = bf16 ? kKsNofXyz : kStepsNofIn;         // bf16: xyz block only, the image-index block is a per-ray bias
  L.emb_split = bf16 ? 1 : 0;                          // bf16: the NoF's embedded input keeps 16 mantissa bits
  L.hsplit_mask = x3 ? ((1u << d.D) - 1u) & ~1u : 0u;  // x3: every hidden range split too
  L.terms = x3 ? kNofTermsX3 : 2;                      // x3: kNofTermsX3 terms per split operand (mf_core.hpp)
  L.half = (x3 && kNofHalfX3) ? 1 : 0;                 // x3: IEEE-half (hi, lo) pairs, scaled (mf_core.hpp)
  L.head_tiles = bf16 == MF_PREC_BF16 ? 1 : 0;         // fast mode: the head panel's terms as tile rows (NetLayout::head_tiles)
  L.emb_mask = 1u | d.skip_mask;
  L.relu_mask = (1u << d.D) - 1u;
  L.extra_steps = -1;
  L.n_head = d.use_quat ? 9 : 3;
  int off = 0;
  L.off_bias_trunk = off; off += L.n_trunk * L.W;
  L.off_head_w = off; off += L.n_head * L.W;
  L.off_head_b = off; off += bf16 ? 32 : 12;          // bf16: the head is an MFMA tile, its bias a 32-row vector
  L.res_bytes = round_up((int64_t)off * 4, kGroupBytes);
  int64_t groups = 0;
  L.max_groups = 0;
  for (int l = 0; l < L.n_trunk; ++l) {
    const int g = trunk_groups(L, l);
    groups += (int64_t)g * L.NP;
    if ((x3 ? panel_cap(g) : g) > L.max_groups) L.max_groups = x3 ? panel_cap(g) : g;
  }
  if (bf16) {                                          // head panel: one 32-row tile, hidden k-steps as (hi, lo[, ...]) groups
    groups += head_groups(L);
    if (head_groups(L) > L.max_groups) L.max_groups = head_groups(L);
  }
  L.panel_bytes = groups * kGroupBytes;
  L.n_emb_layers = __builtin_popcount(L.emb_mask);
  L.ind_bytes = bf16 ? round_up((int64_t)L.n_emb_layers * L.W * kNofIndCols * 4, kGroupBytes) : 0;
  return true;
}

}  // namespace mf
