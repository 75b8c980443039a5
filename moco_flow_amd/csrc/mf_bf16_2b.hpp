// mf_bf16_2b.hpp -- the fast bf16 mode with TWO 32-sample column blocks per wave (round 6).
//
// mf_bf16.hpp's tiling reads one fresh 1 KiB A fragment (32 weight rows x 16 k) from LDS per MFMA, and the two waves of a
// SIMD -- same weights, different samples -- each read their own copy: at the matrix pipe's full rate that IS the LDS's
// 128 B / cycle / CU.  Here a wave owns 64 samples as two column blocks and every fragment feeds two MFMAs (block 0, block 1):
// half the LDS fragment traffic per sample, one wave per SIMD (workgroup = 4 waves, still 256 samples per pass over the weight
// stream, the same panels, the same arithmetic in the same order -- results are bit-identical to the one-block kernels').
// Priced in isolation in round 5 (tools/proto/dma_mix.hip, profiles/r05_dma_mix_two_blocks.txt: the hidden-layer panel loop with
// its LDS-DMA ring and barrier 56.6 -> 60.0 % of the nominal bf16 peak).  Registers: 2 x (64 in + 64 out) of activations, two
// accumulator sets in flight per block, ~400 of the 512 a lone wave has (hipcc parks what does not fit the 256 VGPRs in AGPRs).
// The per-sample VALU code (ray -> point -> encodings, heads, sample buffer) runs once per block.
#pragma once
#include "mf_bf16.hpp"

namespace mf {
namespace bf {

constexpr int kWaves2 = 4;                  // waves per workgroup, one per SIMD
constexpr int kBlockSamples = 2 * kWaveSamples;    // samples per wave (two column blocks)
using Stream2 = StreamT<kWaves2>;

// One output tile for both column blocks.  Per weight group: MFMA block 0, [fragment prefetch, the panel hook, epilogue steps],
// MFMA block 1, [an LDS-DMA piece of the panel two ahead -- a gap without a fragment read --, epilogue steps]; a SPLIT embedded
// k-step's hi group is followed by the two Whi * xlo products.  `acc*` leave as the tile's accumulators; the first MFMA of each
// block takes `init*` (the bias / per-ray bias vectors, read during the previous tile) as its C operand.
// gap(m): m-th MFMA slot of the tile (the caller's deferred work: the previous tile's epilogue, the next tile's bias read).
template <int NGE, int KHID, bool SPLIT>
MF_D constexpr int tile2_slots() { return 2 * ((SPLIT ? 2 : 1) * NGE + KHID) + (SPLIT ? 2 * NGE : 0); }

template <int NGE, int KHID, bool EMB_FIRST, bool SPLIT, class Hook, class Piece, class Gap>
MF_D void mma_tile2(Carry& carry, const u32x4* hid0, const u32x4* hid1, const u32x4* xhi0, const u32x4* xhi1, const u32x4* xlo0,
                    const u32x4* xlo1, uint32_t p, uint32_t pn, const f32x16& init0, const f32x16& init1, f32x16& acc0,
                    f32x16& acc1, Hook&& hook, Piece&& piece, Gap&& gap) {
  constexpr int NEG = (SPLIT ? 2 : 1) * NGE;            // groups of the embedded block
  constexpr int NG = NEG + KHID;
  static_assert(NG > PD, "panel shorter than the fragment pipeline");
  static_assert(NG >= 4, "panel too short for the DMA pieces");
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = carry.w[i];
  int m = 0;
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int s = gi % (PD + 1);
    const int ge = EMB_FIRST ? gi : gi - KHID;            // index within the embedded groups
    const bool emb = ge >= 0 && ge < NEG;
    const int ks = emb ? (SPLIT ? ge >> 1 : ge) : (EMB_FIRST ? gi - NEG : gi);
    const u32x4& b0 = emb ? xhi0[ks] : hid0[ks];
    const u32x4& b1 = emb ? xhi1[ks] : hid1[ks];
    if (gi == 0) acc0 = MF_MFMA32(r[s], b0, init0);
    else acc0 = MF_MFMA32(r[s], b0, acc0);
    __builtin_amdgcn_sched_barrier(0);
    const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
    if (nb < NG) r[sp] = lds_u4(p + nb * kGroupBytes);
    if (gi == 0) hook();
    if (nb >= NG) r[sp] = lds_u4(pn + (nb - NG) * kGroupBytes);
    gap(m++);
    __builtin_amdgcn_sched_barrier(0);
    if (gi == 0) acc1 = MF_MFMA32(r[s], b1, init1);
    else acc1 = MF_MFMA32(r[s], b1, acc1);
    __builtin_amdgcn_sched_barrier(0);
    piece(gi);                                            // (gi = 0: behind the panel hook of the first gap)
    gap(m++);
    __builtin_amdgcn_sched_barrier(0);
    if (SPLIT && emb && !(ge & 1)) {                      // Whi * xlo of both blocks
      acc0 = MF_MFMA32(r[s], xlo0[ge >> 1], acc0);
      __builtin_amdgcn_sched_barrier(0);
      gap(m++);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = MF_MFMA32(r[s], xlo1[ge >> 1], acc1);
      __builtin_amdgcn_sched_barrier(0);
      gap(m++);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < PD; ++i) carry.w[i] = r[(NG + i) % (PD + 1)];
}

// the pending tile's epilogue in slot m of an NM-slot tile: 16 steps (8 per block), none in slot 0 (the panel barrier) and none
// in the last two (the next tile's bias reads)
MF_D constexpr int epi2_lo(int m, int nm) { const int e = nm - 3; return m < 1 ? 0 : (m - 1 >= e ? 16 : 16 * (m - 1) / e); }
MF_D constexpr int epi2_hi(int m, int nm) { const int e = nm - 3; return m < 1 ? 0 : (m >= e ? 16 : 16 * m / e); }

// One trunk layer on both blocks: out[b] <- relu?(W_l [emb[b] ; act[b]] + bias).  MODE / TPP / RBT as trunk_layer_m; a per-ray
// bias (LdsRayBias) is per block (the two blocks' samples sit on different rays).
template <int KH, int NGE, int MODE, bool SPLIT, class RBT = NoRayBias, int TPP = 1>
MF_D void trunk_layer_m2(const Net& net, int layer, bool relu, const u32x4 (&act)[2][KH], u32x4 (&out)[2][KH], const u32x4 (&xhi)[2][NGE],
                         const u32x4 (&xlo)[2][NGE], Stream2& st, Carry& carry, const Lane& id, const Next& nxt, const RBT (&rb)[2]) {
  constexpr bool RB = __is_same(RBT, LdsRayBias) && (MODE & 1);
  static_assert(__is_same(RBT, LdsRayBias) || __is_same(RBT, NoRayBias), "two-block layers: static or LDS-staged bias");
  constexpr int NT = KH / 2;
  constexpr int NP = NT / TPP;
  static_assert(NT % TPP == 0, "tiles per panel");
  constexpr int groups = ((MODE & 1) ? (SPLIT ? 2 : 1) * NGE : 0) + ((MODE & 2) ? KH : 0);      // of one tile
  constexpr int pgroups = TPP * groups;
  constexpr int NM = tile2_slots<(MODE & 1) ? NGE : 0, (MODE & 2) ? KH : 0, SPLIT>();
  constexpr int kInitSlot = NM - 2;                         // (earlier -- the middle of the tile -- holds 16-32 registers longer: 12-28 spilled, and measured nothing)
  const unsigned lo = relu ? 0u : 0x80008000u;
  uint32_t bias0 = net.res_lds + layer * (16 * KH) * 4, bias1 = bias0;
  if constexpr (RB) {
    const uint32_t el = (uint32_t)__builtin_popcount(net.emb_mask & ((1u << layer) - 1u)) * (16 * KH) * 4;
    bias0 = rb[0].lane_off + el;
    bias1 = rb[1].lane_off + el;
  }
  // (the init vectors are dead behind a tile's first group: the next tile's are read into the same registers near its end)
  f32x16 init0 = bias_acc(bias0, id.h), init1 = {};
  if constexpr (RB) init1 = bias_acc(bias1, id.h);
  f32x16 acc0 = {}, acc1 = {}, pend0 = {}, pend1 = {};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pi = t / TPP, sub = t % TPP;
    const bool second = sub != 0;
    const uint32_t p = st.slot_off(0) + id.lane * 16 + sub * groups * kGroupBytes;
    const uint32_t pn = sub == TPP - 1 ? st.slot_off(1) + id.lane * 16 : p + groups * kGroupBytes;
    auto hook = [&]() {
      if (second) return;
      st.sync(pi + 2 < NP ? pgroups : (pi == NP - 2 ? nxt.groups : nxt.groups2),
              pi == NP - 2 ? nxt.jump : (pi == NP - 1 ? nxt.jump2 : nullptr), id, true, pi + 2 < NP ? pgroups : -1);
    };
    // this wave's pieces of the panel two ahead (<= 8: a panel is <= 32 groups over 4 waves): one per group of this tile, the
    // panel's later tiles carry on where a short first tile stops
    constexpr int ppt = groups < Stream2::kPieces ? groups : Stream2::kPieces;
    auto piece = [&](int k) { if (k < ppt && sub * ppt + k < Stream2::kPieces) st.piece(sub * ppt + k, id); };
    const int tp = t > 0 ? t - 1 : 0;
    auto gap = [&](int m) __attribute__((always_inline)) {
      if (t > 0) {
#pragma unroll
        for (int u = epi2_lo(m, NM); u < epi2_hi(m, NM); ++u) {
          if (u & 1) epi_pair(pend1, u >> 1, lo, out[1][2 * tp], out[1][2 * tp + 1]);
          else epi_pair(pend0, u >> 1, lo, out[0][2 * tp], out[0][2 * tp + 1]);
        }
      }
      // the next tile's init vectors, into the registers the tile's first group freed
      if (t + 1 < NT) {
        if (m == kInitSlot) init0 = bias_acc(bias0 + 32 * (t + 1) * 4, id.h);
        if (RB && m == kInitSlot + 1) init1 = bias_acc(bias1 + 32 * (t + 1) * 4, id.h);
      }
    };
    mma_tile2<(MODE & 1) ? NGE : 0, (MODE & 2) ? KH : 0, true, SPLIT>(carry, act[0], act[1], xhi[0], xhi[1], xlo[0], xlo[1], p, pn,
                                                                    init0, RB ? init1 : init0, acc0, acc1, hook, piece, gap);
    pend0 = acc0; pend1 = acc1;
    if (sub == TPP - 1) st.advance();
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    epi_pair(pend0, u, lo, out[0][2 * (NT - 1)], out[0][2 * (NT - 1) + 1]);
    epi_pair(pend1, u, lo, out[1][2 * (NT - 1)], out[1][2 * (NT - 1) + 1]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int KH, int NGE, bool SPLIT, class RBT = NoRayBias, int TPP0 = 1, int TPPH = 1, int TPPS = 1>
MF_D void trunk_layer2(const Net& net, int layer, bool relu, const u32x4 (&act)[2][KH], u32x4 (&out)[2][KH], const u32x4 (&xhi)[2][NGE],
                       const u32x4 (&xlo)[2][NGE], Stream2& st, Carry& carry, const Lane& id, const Next& nxt, const RBT (&rb)[2]) {
  const int has_emb = (net.emb_mask >> layer) & 1;
  if (layer == 0) trunk_layer_m2<KH, NGE, 1, SPLIT, RBT, TPP0>(net, layer, relu, act, out, xhi, xlo, st, carry, id, nxt, rb);
  else if (has_emb) trunk_layer_m2<KH, NGE, 3, SPLIT, RBT, TPPS>(net, layer, relu, act, out, xhi, xlo, st, carry, id, nxt, rb);
  else trunk_layer_m2<KH, NGE, 2, SPLIT, RBT, TPPH>(net, layer, relu, act, out, xhi, xlo, st, carry, id, nxt, rb);
}

// the D trunk layers, in pairs a -> b -> a; returns with the last layer's output in `a`
template <int KH, int NGE, bool SPLIT, int TPP0 = 1, int TPPH = 1, int TPPS = 1, class NextOf, class RBT, class AfterFirst>
MF_D void trunk2(const Net& net, int D, u32x4 (&a)[2][KH], const u32x4 (&xhi)[2][NGE], const u32x4 (&xlo)[2][NGE], Stream2& st, Carry& carry,
                 const Lane& id, NextOf&& next_of, const RBT (&rb)[2], AfterFirst&& after_first) {
  u32x4 b[2][KH];
  int l = 0;
  for (; l + 1 < D; l += 2) {
    trunk_layer2<KH, NGE, SPLIT, RBT, TPP0, TPPH, TPPS>(net, l, true, a, b, xhi, xlo, st, carry, id, next_of(l), rb);
    if (l == 0) after_first();
    st.tl.stamp(10 + l, id);
    trunk_layer2<KH, NGE, SPLIT, RBT, TPP0, TPPH, TPPS>(net, l + 1, true, b, a, xhi, xlo, st, carry, id, next_of(l + 1), rb);
    st.tl.stamp(11 + l, id);
  }
  if (l < D) {
    trunk_layer2<KH, NGE, SPLIT, RBT, TPP0, TPPH, TPPS>(net, l, true, a, b, xhi, xlo, st, carry, id, next_of(l), rb);
    if (l == 0) after_first();
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int t = 0; t < KH; ++t) a[k][t] = b[k][t];
  }
}

// head panel on both blocks (TERMS / ZERO as head_tile)
template <int KHID, int TERMS, bool ZERO, class Hook, class Piece>
MF_D void head_tile2(Carry& carry, const u32x4* hid0, const u32x4* hid1, uint32_t p, uint32_t pn, uint32_t bias_off, int h, f32x16& acc0,
                     f32x16& acc1, Hook&& hook, Piece&& piece) {
  constexpr int NG = TERMS * KHID;
  static_assert(NG > PD && NG >= 4, "head panel too short");
  f32x16 init;
  if constexpr (ZERO) {
#pragma unroll
    for (int i = 0; i < 16; ++i) init[i] = 0.f;
  } else {
    init = bias_acc(bias_off, h);
  }
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = carry.w[i];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int s = gi % (PD + 1), ks = TERMS == 2 ? gi >> 1 : gi;
    if (gi == 0) acc0 = MF_MFMA32(r[s], hid0[ks], init);
    else acc0 = MF_MFMA32(r[s], hid0[ks], acc0);
    __builtin_amdgcn_sched_barrier(0);
    const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
    if (nb < NG) r[sp] = lds_u4(p + nb * kGroupBytes);
    if (gi == 0) hook();
    if (nb >= NG) r[sp] = lds_u4(pn + (nb - NG) * kGroupBytes);
    __builtin_amdgcn_sched_barrier(0);
    if (gi == 0) acc1 = MF_MFMA32(r[s], hid1[ks], init);
    else acc1 = MF_MFMA32(r[s], hid1[ks], acc1);
    __builtin_amdgcn_sched_barrier(0);
    piece(gi);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < PD; ++i) carry.w[i] = r[(NG + i) % (PD + 1)];
}

// extra_encoding (nerf.py:98) on both blocks
template <int NGX>
MF_D void extra_layer2(const Net& net, const u32x4 (&act)[2][16], const u32x4 (&ex)[2][kKsExtraMax], u32x4 (&out)[2][8], Stream2& st,
                       Carry& carry, const Lane& id, const Next& nxt) {
  constexpr int NT = 4;
  constexpr int groups = 16 + NGX;
  constexpr int NM = tile2_slots<NGX, 16, false>();
  const uint32_t bias_off = net.res_lds + (net.D + 1) * 256 * 4;
  f32x16 init = bias_acc(bias_off, id.h), acc0 = {}, acc1 = {}, pend0 = {}, pend1 = {};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const uint32_t p = st.slot_off(0) + id.lane * 16;
    const uint32_t pn = st.slot_off(1) + id.lane * 16;
    auto hook = [&]() {
      st.sync(t + 2 < NT ? groups : (t == NT - 2 ? nxt.groups : nxt.groups2),
              t == NT - 2 ? nxt.jump : (t == NT - 1 ? nxt.jump2 : nullptr), id, true, t + 2 < NT ? groups : -1);
    };
    auto piece = [&](int k) { if (k < Stream2::kPieces) st.piece(k, id); };
    const int tp = t > 0 ? t - 1 : 0;
    auto gap = [&](int m) __attribute__((always_inline)) {
      if (t > 0) {
#pragma unroll
        for (int u = epi2_lo(m, NM); u < epi2_hi(m, NM); ++u) {
          if (u & 1) epi_pair(pend1, u >> 1, 0u, out[1][2 * tp], out[1][2 * tp + 1]);
          else epi_pair(pend0, u >> 1, 0u, out[0][2 * tp], out[0][2 * tp + 1]);
        }
      }
      if (m == NM - 2 && t + 1 < NT) init = bias_acc(bias_off + 32 * (t + 1) * 4, id.h);
    };
    mma_tile2<NGX, 16, false, false>(carry, act[0], act[1], ex[0], ex[1], ex[0], ex[1], p, pn, init, init, acc0, acc1, hook, piece, gap);
    pend0 = acc0; pend1 = acc1;
    st.advance();
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    epi_pair(pend0, u, 0u, out[0][2 * (NT - 1)], out[0][2 * (NT - 1) + 1]);
    epi_pair(pend1, u, 0u, out[1][2 * (NT - 1)], out[1][2 * (NT - 1) + 1]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Canonical NeRF (W = 256) on this wave's 2 x 32 samples (nerf_eval's program: trunk, sigma head panel, xyz_encoding_final,
// extra_encoding, rgb head panel).  `make_extra(b, ex)` builds block b's extra operands right before extra_encoding.
template <class MakeExtra>
MF_D void nerf_eval2(const Net& net, const u32x4 (&xe)[2][kKsNerfXyz], MakeExtra&& make_extra, bool sigma_only, Stream2& st,
                     Carry& carry, const Lane& id, const Next& follow, float (&sigma)[2], float (&rgb)[2][3]) {
  u32x4 act[2][16];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) act[b][t][i] = 0;
  const int D = net.D;
  const NoRayBias norb[2] = {};
  const Next sg_next = sigma_only ? Next{16, nullptr, follow.groups, follow.jump} : Next{16, nullptr, 16, nullptr};
  trunk2<16, kKsNerfXyz, false, kNerfTpp0, kNerfTppH, kNerfTppS>(net, D, act, xe, xe, st, carry, id, [&](int l) {
    return l == D - 1 ? sg_next : next_trunk_bf<16, kKsNerfXyz, false, 0, kNerfTppH, kNerfTppS>(net, l + 1);
  }, norb, [] {});
  const uint32_t r_sigma_w = net.res_lds + ((D + 1) * 256 + 128) * 4;
  {
    const uint32_t p = st.slot_off(0) + id.lane * 16, pn = st.slot_off(1) + id.lane * 16;
    auto hook = [&]() {
      if (sigma_only) st.sync(follow.groups2, follow.jump2, id);
      else st.sync(16, nullptr, id, true, 16);
    };
    auto piece = [&](int k) { if (k < Stream2::kPieces) st.piece(k, id); };
    f32x16 a0, a1;
    head_tile2<16, 1, true>(carry, act[0], act[1], p, pn, 0u, id.h, a0, a1, hook, piece);
    st.advance();
    const float sb = lds_f(r_sigma_w + 256 * 4);
    sigma[0] = a0[0] + a0[4] + sb;
    sigma[1] = a1[0] + a1[4] + sb;
  }
  st.tl.stamp(30, id);
  if (sigma_only) return;
  const int xg = 16 + net.aux;
  const Next ex{xg, nullptr, xg, nullptr};
  u32x4 fin[2][16];
  trunk_layer_m2<16, kKsNerfXyz, 2, false, NoRayBias, kNerfTppH>(net, D, false, act, fin, xe, xe, st, carry, id, ex, norb);
  st.tl.stamp(31, id);
  u32x4 e[2][8], eo[2][kKsExtraMax];
  make_extra(0, eo[0]);
  make_extra(1, eo[1]);
  st.tl.stamp(32, id);
  const Next rg_next{8, nullptr, follow.groups, follow.jump};
  if (net.aux == 2) extra_layer2<2>(net, fin, eo, e, st, carry, id, rg_next);
  else if (net.aux == 1) extra_layer2<1>(net, fin, eo, e, st, carry, id, rg_next);
  else extra_layer2<0>(net, fin, eo, e, st, carry, id, rg_next);
  st.tl.stamp(33, id);
  {
    const uint32_t p = st.slot_off(0) + id.lane * 16, pn = st.slot_off(1) + id.lane * 16;
    auto hook = [&]() { st.sync(follow.groups2, follow.jump2, id); };
    auto piece = [&](int k) { if (k < Stream2::kPieces) st.piece(k, id); };
    f32x16 a0, a1;
    head_tile2<8, 1, true>(carry, e[0], e[1], p, pn, 0u, id.h, a0, a1, hook, piece);
    st.advance();
    const uint32_t r_rgb_b = r_sigma_w + (256 + 4 + 384) * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float bc = lds_f(r_rgb_b + 4 * c);
      rgb[0][c] = 1.f / (1.f + expf(-(a0[c] + a0[4 + c] + bc)));       // nn.Sigmoid, nerf.py:57-59
      rgb[1][c] = 1.f / (1.f + expf(-(a1[c] + a1[4 + c] + bc)));
    }
  }
}

// Neural motion flow (W = 128) on this wave's 2 x 32 samples (nof_eval's program)
template <class AfterFirst>
MF_D void nof_eval2(const Net& net, const u32x4 (&xhi)[2][kKsNofXyz], const u32x4 (&xlo)[2][kKsNofXyz], const float (&xyz)[2][3],
                    Stream2& st, Carry& carry, const Lane& id, const Next& follow, float (&out)[2][3], const LdsRayBias (&rb)[2],
                    AfterFirst&& after_first) {
  u32x4 act[2][8];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) act[b][t][i] = 0;
  const int D = net.D;
  const Next hd{8, nullptr, follow.groups, follow.jump};
  trunk2<8, kKsNofXyz, true, kNofTpp0, kNofTppH, kNofTppS>(net, D, act, xhi, xlo, st, carry, id,
      [&](int l) { return l == D - 1 ? hd : next_trunk_np<8, kKsNofXyz, true, kNofTppH, kNofTppS>(net, l + 1, D, hd); }, rb, after_first);
  f32x16 a[2];
  {
    const uint32_t p = st.slot_off(0) + id.lane * 16, pn = st.slot_off(1) + id.lane * 16;
    auto hook = [&]() { st.sync(follow.groups2, follow.jump2, id); };
    auto piece = [&](int k) { if (k < Stream2::kPieces) st.piece(k, id); };
    head_tile2<8, 1, false>(carry, act[0], act[1], p, pn, net.res_lds + (D + net.aux) * 128 * 4, id.h, a[0], a[1], hook, piece);
    st.advance();
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float own[5], oth[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { own[i] = a[b][i] + a[b][8 + i]; oth[i] = __shfl_xor(own[i], 32, 64); }
    if (net.aux == 9) {
      float T[9];
#pragma unroll
      for (int i = 0; i < 4; ++i) { T[i] = id.h ? oth[i] : own[i]; T[4 + i] = id.h ? own[i] : oth[i]; }
      T[8] = id.h ? oth[4] : own[4];
      quat_transform<true>(T, xyz[b], out[b]);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[b][c] = (id.h ? oth[c] : own[c]) + xyz[b][c];
    }
  }
}

}  // namespace bf
}  // namespace mf
