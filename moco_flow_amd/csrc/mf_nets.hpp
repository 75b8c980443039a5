// mf_nets.hpp -- the two networks of the hot path on top of the fused MLP core.
//   nerf_eval : models/nerf.py:78-102   (trunk, sigma head, final, extra, rgb)
//   nof_eval  : models/nof.py:69-82     (trunk, 3|9 head, quaternion transform)
#pragma once
#include "mf_core.hpp"

namespace mf {


MF_D const char* first_panel(const NetDev& n) { return n.packed + n.L.res_bytes; }
MF_D int first_groups(const NetDev& n) { return trunk_groups(n.L, 0); }

// Copy a network's resident block (biases + VALU head weights) global -> LDS.
MF_D void load_resident(const NetDev& n, const LaneId& id) {
  const int groups = (int)(n.L.res_bytes / kGroupBytes);
  for (int g = id.wave; g < groups; g += kWaves) glds16(n.packed + g * kGroupBytes + id.lane * 16, n.res_lds + g * kGroupBytes);
}

// extra_encoding (nerf.py:98): (W/2) outputs from [final(W) ; extra block], ReLU.
template <int NT>
MF_D void extra_layer(const NetDev& net, const f32x16 (&act)[NT], const float (&ext)[kStepsExtraMax],
                      f32x16 (&out)[NT / 2], Stream& st, const LaneId& id, int next_groups, const char* jump) {
  const int groups = extra_groups(net.L);
  const int ge = net.L.extra_steps / 4;
  const uint32_t bias_off = net.res_lds + net.L.off_bias_extra * 4;
  const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT / 2; ++t) {
    if (t == NT / 2 - 1) {
      if (jump) st.gnext = jump;
      st.prefetch(next_groups, id);
    } else {
      st.prefetch(groups, id);
    }
    const uint32_t p = st.cur_off() + id.lane * 16;
    f32x16 acc = bias_tile(bias_off, t, id.h);
    acc = out_tile<2, NT, 4>(acc, act, dummy, p);
#pragma unroll
    for (int g = 0; g < kStepsExtraMax / 4; ++g) {
      if (g < ge) {
        const f32x4 w = lds_f4(p + (NT * 4 + g) * kGroupBytes);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = MF_MFMA(w[r], ext[4 * g + r], acc);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[t][i] = fmaxf(acc[i], 0.f);
    st.flip();
  }
}

// Canonical NeRF on this wave's 32 samples.  `follow`/`follow_groups`: the program's next
// panel after this network (always given; the stream jumps there after the last panel used).
template <int NT>
MF_D void nerf_eval(const NetDev& net, const float (&embx)[kStepsNerfXyz], const float (&ext)[kStepsExtraMax],
                    bool sigma_only, Stream& st, const LaneId& id, const char* follow, int follow_groups,
                    float& sigma, float (&rgb)[3]) {
  f32x16 act[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) act[t][i] = 0.f;
  const int D = net.L.n_trunk - 1;
  for (int l = 0; l < D; ++l) {
    const bool last = sigma_only && l == D - 1;
    trunk_layer<NT, kStepsNerfXyz>(net, l, act, embx, st, id, last ? follow_groups : trunk_groups(net.L, l + 1),
                                   last ? follow : nullptr);
  }
  float sg[1];
  valu_head<NT, 1>(act, net.res_lds + net.L.off_head_w * 4, net.L.W, net.res_lds + net.L.off_head_b * 4, id.h, sg);
  sigma = sg[0];
  if (sigma_only) return;
  trunk_layer<NT, kStepsNerfXyz>(net, D, act, embx, st, id, extra_groups(net.L), nullptr);   // xyz_encoding_final
  f32x16 e[NT / 2];
  extra_layer<NT>(net, act, ext, e, st, id, follow_groups, follow);
  float o[3];
  valu_head<NT / 2, 3>(e, net.res_lds + net.L.off_rgb_w * 4, net.L.W / 2, net.res_lds + net.L.off_rgb_b * 4, id.h, o);
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[c] = 1.f / (1.f + expf(-o[c]));   // nn.Sigmoid, nerf.py:57-59
}

// kornia 0.6.5 quaternion_log_to_exp + quaternion_to_rotation_matrix as restated in
// oracle/kornia_restated.py (PARITY UNPINNED), then nof.py:80:
//   out = (xyz - s) R + s + t      ((xyz - s) as a ROW vector)
MF_D void quat_transform(const float (&T)[9], const float (&xyz)[3], float (&out)[3]) {
  const float vx = T[0], vy = T[1], vz = T[2];
  float n = sqrtf(vx * vx + vy * vy + vz * vz);
  n = fmaxf(n, 1e-8f);
  float sn, cn;
  sincosf(n, &sn, &cn);
  float qx = vx * sn / n, qy = vy * sn / n, qz = vz * sn / n, qw = cn;
  const float qn = fmaxf(sqrtf(qx * qx + qy * qy + qz * qz + qw * qw), 1e-12f);
  qx /= qn; qy /= qn; qz /= qn; qw /= qn;
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

// Neural motion flow on this wave's 32 samples; emb = [xyz block ; ind block] (kStepsNofIn).
MF_D void nof_eval(const NetDev& net, const float (&emb)[kStepsNofIn], const float (&xyz)[3], Stream& st,
                   const LaneId& id, const char* follow, int follow_groups, float (&out)[3]) {
  constexpr int NT = 4;
  f32x16 act[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) act[t][i] = 0.f;
  const int D = net.L.n_trunk;
  for (int l = 0; l < D; ++l) {
    const bool last = l == D - 1;
    trunk_layer<NT, kStepsNofIn>(net, l, act, emb, st, id, last ? follow_groups : trunk_groups(net.L, l + 1),
                                 last ? follow : nullptr);
  }
  const uint32_t wo = net.res_lds + net.L.off_head_w * 4, bo = net.res_lds + net.L.off_head_b * 4;
  if (net.L.n_head == 9) {
    float T[9];
    valu_head<NT, 9>(act, wo, net.L.W, bo, id.h, T);
    quat_transform(T, xyz, out);
  } else {
    float T[3];
    valu_head<NT, 3>(act, wo, net.L.W, bo, id.h, T);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = T[c] + xyz[c];
  }
}

// NoF input block from a point and an image index (rendering.py:70-75)
MF_D void nof_embed(float (&emb)[kStepsNofIn], const float (&xyz)[3], float ind, const EmbParams& exyz,
                    const EmbParams& eind, int h) {
  emb_eval<3, 5>(emb, xyz, exyz, h);
  const float iv[1] = {ind};
  emb_eval<1, 16>(emb + BlkXyz5::SLOTS, iv, eind, h);
#pragma unroll
  for (int e = BlkXyz5::SLOTS + BlkInd16::SLOTS; e < kStepsNofIn; ++e) emb[e] = 0.f;
}


}  // namespace mf
