// mf_nets.hpp -- the two networks of the hot path on top of the fused MLP core.
//   nerf_eval : models/nerf.py:78-102   (trunk, sigma head, final, extra, rgb)
//   nof_eval  : models/nof.py:69-82     (trunk, 3|9 head, quaternion transform)
#pragma once
#include "mf_core.hpp"

namespace mf {

MF_D const char* first_panel(const NetDev& n) { return n.packed + n.L.res_bytes; }
MF_D int first_groups(const NetDev& n) { return trunk_groups(n.L, 0); }

// Where the program continues after a network: the first layer of the next network evaluated.
MF_D NextLayer follow_of(const NetDev& n) {
  NextLayer f;
  f.groups = trunk_groups(n.L, 0);
  f.jump = n.packed + n.L.res_bytes;
  f.bias_off = n.res_lds + n.L.off_bias_trunk * 4;
  return f;
}
MF_D NextLayer next_trunk(const NetDev& n, int layer) {   // trunk layer `layer` of the same network
  NextLayer f;
  f.groups = trunk_groups(n.L, layer);
  f.jump = nullptr;
  f.bias_off = n.res_lds + (n.L.off_bias_trunk + layer * n.L.W) * 4;
  return f;
}

// Copy a network's resident block (biases + VALU head weights) global -> LDS.
MF_D void load_resident(const NetDev& n, const LaneId& id) {
  const int groups = (int)(n.L.res_bytes / kGroupBytes);
  for (int g = id.wave; g < groups; g += kWaves) blds16(n.packed, id.lane * 16, g * kGroupBytes, n.res_lds + g * kGroupBytes);
}

// Prime the stream and the carry at the first panel of network `n` (kernel start).
template <int PD>
MF_D void start_program(const NetDev& n, Stream& st, CarryT<PD>& carry, const LaneId& id) {
  st.start(first_panel(n), first_groups(n), id);
  carry.load(st.slot_off(0) + id.lane * 16, n.res_lds + n.L.off_bias_trunk * 4, id.g);
}

// extra_encoding (nerf.py:98): (W/2) outputs from [final(W) ; extra block], ReLU.
template <int NK, bool DUMP = false>
MF_D void extra_layer(const NetDev& net, const f32x4 (&act)[NK],
                      const float (&ext)[kStepsExtraMax], f32x4 (&out)[NK / 2],
                      Stream& st, CarryT<kPD>& carry, const LaneId& id, const NextLayer& nxt,
                      float* dump_row = nullptr, unsigned* mask_row = nullptr) {
  constexpr int NPO = NK / 4;                      // panels of the (W/2)-wide layer
  constexpr int QH = NK;                           // hidden batches in front of the extra block
  const int groups = extra_groups(net.L);
  const int qe = net.L.extra_steps / 4;
  const uint32_t bias_off = net.res_lds + net.L.off_bias_extra * 4;
  const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  static_assert(NPO % 4 == 0, "mask words collect four panels");
  unsigned macc = 0;
#pragma unroll
  for (int t = 0; t < NPO; ++t) {
    const uint32_t p = st.slot_off(0) + id.lane * 16;
    const uint32_t pn = st.slot_off(1) + id.lane * 16;
    const uint32_t nb = (t + 1 < NPO) ? bias_off + 32 * (t + 1) * 4 : nxt.bias_off;
    auto hook = [&](int ph) { st.template sync_and_dma<DUMP>(t + 2 < NPO ? groups : nxt.groups, t == NPO - 2 ? nxt.jump : nullptr, id, ph); };
    // hidden part through the common path (kept linear: lo = -inf), then the <= 2 extra k-quads (fp32)
    f32x4 E, O;
    out_pair<2, NK, 4>(carry, act, dummy, p, pn, nb, id.g, id.wave < kWaves / 2 && !(st.dbg & 64), hook,
                             -__builtin_inff(), E, O);
#pragma unroll
    for (int q = 0; q < kStepsExtraMax / 4; ++q) {
      if (q < qe) {
        const f32x4 wE = lds_f4(p + (2 * (QH + q)) * kGroupBytes);
        const f32x4 wO = lds_f4(p + (2 * (QH + q) + 1) * kGroupBytes);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          E = MF_MFMA(wE[r], ext[4 * q + r], E);
          O = MF_MFMA(wO[r], ext[4 * q + r], O);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      E[i] = fmaxf(E[i], 0.f);
      O[i] = fmaxf(O[i], 0.f);
    }
    out[2 * t] = E;
    out[2 * t + 1] = O;
    if constexpr (DUMP) {
      if (dump_row) {
        *reinterpret_cast<f32x4*>(dump_row + 32 * t + 4 * id.g) = E;
        *reinterpret_cast<f32x4*>(dump_row + 32 * t + 16 + 4 * id.g) = O;
      }
      const bool wm = __ballot(mask_row != nullptr) != 0ull;
      const bool put = wm && relu_mask_put(macc, mask_row, dump_row && mask_row, t, id.g, E, O);
      st.keep2 = __ballot(dump_row != nullptr) != 0ull ? (put ? 3 : 2) : 0;   // the wave issued the stores
    }
    st.advance();
  }
}

// Canonical NeRF on this wave's 16 samples.  `follow`: the first layer of whatever the panel
// program evaluates after this network (the stream jumps there behind the last panel used).
// DUMP (training forward): `dump_row` = this lane's sample row [h_0 .. h_{D-1} | final | extra] (nullptr: skip).
template <int NK, bool DUMP = false>
MF_D void nerf_eval(const NetDev& net, const float (&embx)[kStepsNerfXyz], const float (&ext)[kStepsExtraMax],
                    bool sigma_only, Stream& st, CarryT<kPD>& carry, const LaneId& id,
                    const NextLayer& follow, float& sigma, float (&rgb)[3], float* dump_row = nullptr,
                    unsigned* mask_row = nullptr) {
  f32x4 act[NK];
#pragma unroll
  for (int t = 0; t < NK; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) act[t][i] = 0;
  const int D = net.L.n_trunk - 1;
  for (int l = 0; l < D; ++l) {
    const bool last = sigma_only && l == D - 1;
    trunk_layer<NK, kStepsNerfXyz, DUMP>(net, l, act, embx, st, carry, id, last ? follow : next_trunk(net, l + 1),
                                               dump_row ? dump_row + l * net.L.W : nullptr, mask_row ? mask_row + l * 8 : nullptr);
    st.tl.stamp(10 + l, id);
  }
  float sg[1];
  valu_head(act, net.res_lds + net.L.off_head_w * 4, net.L.W, net.res_lds + net.L.off_head_b * 4, id.g, sg);
  sigma = sg[0];
  st.tl.stamp(30, id);
  if (sigma_only) return;
  NextLayer ex;
  ex.groups = extra_groups(net.L);
  ex.jump = nullptr;
  ex.bias_off = net.res_lds + net.L.off_bias_extra * 4;
  trunk_layer<NK, kStepsNerfXyz, DUMP>(net, D, act, embx, st, carry, id, ex,          // xyz_encoding_final
                                             dump_row ? dump_row + D * net.L.W : nullptr);
  st.tl.stamp(31, id);
  f32x4 e[NK / 2];
  extra_layer<NK, DUMP>(net, act, ext, e, st, carry, id, follow, dump_row ? dump_row + (D + 1) * net.L.W : nullptr,
                        mask_row ? mask_row + (D + 1) * 8 : nullptr);
  st.tl.stamp(33, id);
  float o[3];
  valu_head(e, net.res_lds + net.L.off_rgb_w * 4, net.L.W / 2, net.res_lds + net.L.off_rgb_b * 4, id.g, o);
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[c] = 1.f / (1.f + expf(-o[c]));   // nn.Sigmoid, nerf.py:57-59
}

// kornia 0.6.5 quaternion_log_to_exp + quaternion_to_rotation_matrix as restated in
// oracle/kornia_restated.py (PARITY UNPINNED), then nof.py:80:
//   out = (xyz - s) R + s + t      ((xyz - s) as a ROW vector)
// FAST (the bf16 fast mode only): the same formulas on the hardware's approximate units -- v_sqrt_f32 / v_rcp_f32 (1 ulp) and
// v_sin_f32 / v_cos_f32 on the angle in revolutions (abs error ~1e-6: the angle is the norm of a head output, well
// under a revolution) instead of OCML's exact sincosf (~110 instructions), sqrtf and seven IEEE divisions; the point
// moves by ~1e-6 relative, under what the bf16 operands around it cost.  ~2 % of a C3 tile, ~4 % of C3g's.
template <bool FAST = false>
MF_D void quat_transform(const float (&T)[9], const float (&xyz)[3], float (&out)[3]) {
  const float vx = T[0], vy = T[1], vz = T[2];
  float qx, qy, qz, qw;
  if constexpr (FAST) {
    const float n = fmaxf(__builtin_amdgcn_sqrtf(vx * vx + vy * vy + vz * vz), 1e-8f);
    const float rev = n * 0.15915494309189535f;
    const float s_n = __builtin_amdgcn_sinf(rev) * __builtin_amdgcn_rcpf(n);
    qx = vx * s_n; qy = vy * s_n; qz = vz * s_n; qw = __builtin_amdgcn_cosf(rev);
    const float rq = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz + qw * qw), 1e-12f));
    qx *= rq; qy *= rq; qz *= rq; qw *= rq;
  } else {
    float n = sqrtf(vx * vx + vy * vy + vz * vz);
    n = fmaxf(n, 1e-8f);
    float sn, cn;
    sincosf(n, &sn, &cn);
    qx = vx * sn / n; qy = vy * sn / n; qz = vz * sn / n; qw = cn;
    const float qn = fmaxf(sqrtf(qx * qx + qy * qy + qz * qz + qw * qw), 1e-12f);
    qx /= qn; qy /= qn; qz /= qn; qw /= qn;
  }
  const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const float txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const float R00 = 1.f - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
  const float R10 = txy + twz, R11 = 1.f - (txx + tzz), R12 = tyz - twx;
  const float R20 = txz - twy, R21 = tyz + twx, R22 = 1.f - (txx + tyy);
  const float px = xyz[0] - T[3], py = xyz[1] - T[4], pz = xyz[2] - T[5];
  out[0] = (px * R00 + py * R10 + pz * R20) + T[3] + T[6];
  out[1] = (px * R01 + py * R11 + pz * R21) + T[4] + T[7];
  out[2] = (px * R02 + py * R12 + pz * R22) + T[5] + T[8];
}

// Neural motion flow on this wave's 16 samples; emb = [xyz block ; ind block] (kStepsNofIn).
// DUMP (training forward): `drow` = this lane's sample row [h_1 .. h_D | T (9 | 3) zero-padded to 16] (nullptr: skip),
// the layout mf_nof_backward / mf_weight_grads read (mf_nofgrad.hip).
// `masks` (ABI v15, uniform): the row has room behind T for the layers' ReLU BIT rows -- 4 words per layer at float offset
// D W + 16 (trunk_layer's mask_row: byte 4 t + g = this lane's eight outputs of panel t) -- what mf_nof_backward3 reads
// instead of the 512 bytes of activations per layer.
// NK = W / 16: 8 everywhere but the module-level forward of the reference's bare NoF() (W = 256: NK = 16, no dump).
template <bool DUMP = false, int NK = 8>
MF_D void nof_eval(const NetDev& net, const float (&emb)[kStepsNofIn], const float (&xyz)[3], Stream& st,
                   CarryT<kPD>& carry, const LaneId& id, const NextLayer& follow, float (&out)[3],
                   float* drow = nullptr, bool masks = false) {
  static_assert(!DUMP || NK == 8, "the dump layout (bit rows: four words per layer) is the 128-wide one");
  f32x4 act[NK];
#pragma unroll
  for (int t = 0; t < NK; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) act[t][i] = 0;
  const int D = net.L.n_trunk;
  for (int l = 0; l < D; ++l) {
    const bool last = l == D - 1;
    trunk_layer<NK, kStepsNofIn, DUMP>(net, l, act, emb, st, carry, id, last ? follow : next_trunk(net, l + 1),
                                       drow ? drow + l * net.L.W : nullptr,
                                       (DUMP && drow && masks) ? reinterpret_cast<unsigned*>(drow + D * net.L.W + 16) + 4 * l : nullptr);
  }
  const uint32_t wo = net.res_lds + net.L.off_head_w * 4, bo = net.res_lds + net.L.off_head_b * 4;
  float T[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (net.L.n_head == 9) {
    valu_head(act, wo, net.L.W, bo, id.g, T);
    quat_transform(T, xyz, out);
  } else {
    float T3[3];
    valu_head(act, wo, net.L.W, bo, id.g, T3);
#pragma unroll
    for (int c = 0; c < 3; ++c) { T[c] = T3[c]; out[c] = T3[c] + xyz[c]; }
  }
  if constexpr (DUMP) {
    if (drow && id.g == 0) {
      float4* tr = reinterpret_cast<float4*>(drow + D * net.L.W);
      tr[0] = make_float4(T[0], T[1], T[2], T[3]);
      tr[1] = make_float4(T[4], T[5], T[6], T[7]);
      tr[2] = make_float4(T[8], 0.f, 0.f, 0.f);
      tr[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// NoF input block from a point and an image index (rendering.py:70-75)
MF_D void nof_embed(float (&emb)[kStepsNofIn], const float (&xyz)[3], float ind, const EmbParams& exyz,
                    const EmbParams& eind, int g) {
  emb_eval<3, 5>(emb, xyz, exyz, g);
  const float iv[1] = {ind};
  emb_eval<1, 16>(emb + BlkXyz5::SLOTS, iv, eind, g);
#pragma unroll
  for (int e = BlkXyz5::SLOTS + BlkInd16::SLOTS; e < kStepsNofIn; ++e) emb[e] = 0.f;
}

MF_D void nof_embed_lds(float (&emb)[kStepsNofIn], const float (&xyz)[3], float ind, uint32_t par_xyz, uint32_t par_ind, int g) {
  emb_eval_lds<3, 5>(emb, xyz, par_xyz, g);
  const float iv[1] = {ind};
  emb_eval_lds<1, 16>(emb + BlkXyz5::SLOTS, iv, par_ind, g);
#pragma unroll
  for (int e = BlkXyz5::SLOTS + BlkInd16::SLOTS; e < kStepsNofIn; ++e) emb[e] = 0.f;
}

}  // namespace mf
