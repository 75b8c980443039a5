// mf_nofbwd.hpp -- what the two backward kernels of a NoF evaluation share (mf_nofgrad.hip: fp32 MFMA; mf_nofgrad_bf16.hip:
// three bf16 products): the backward of the head's rigid transform (models/nof.py:75-82, kornia restated: PARITY UNPINNED like
// the forward, oracle/kornia_restated.py) by forward-mode partials.
#pragma once
#include "mf_nets.hpp"

namespace mf {

// ------------------------------------------------------------------ forward-mode partials (quaternion head)
struct D3 {                     // value + partials w.r.t. the three log-quaternion components
  float v, d[3];
};
MF_D D3 d3c(float c) { return D3{c, {0.f, 0.f, 0.f}}; }
MF_D D3 operator+(const D3& a, const D3& b) { return D3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
MF_D D3 operator-(const D3& a, const D3& b) { return D3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
MF_D D3 operator*(const D3& a, const D3& b) {
  return D3{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
MF_D D3 operator/(const D3& a, const D3& b) {
  const float q = a.v / b.v, ib = 1.f / b.v;
  return D3{q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
MF_D D3 d3_sqrt(const D3& a) {          // torch.norm backward: 0 at 0
  const float r = sqrtf(a.v), h = r > 0.f ? 0.5f / r : 0.f;
  return D3{r, {a.d[0] * h, a.d[1] * h, a.d[2] * h}};
}
MF_D D3 d3_clamp_min(const D3& a, float m) {   // torch.clamp(min=m): gradient passes where a >= m
  return a.v >= m ? a : d3c(m);
}
MF_D D3 d3_scale(const D3& a, float s, float ds) {   // a * s with ds = d s / d a.v ... helper for sin/cos
  return D3{s, {a.d[0] * ds, a.d[1] * ds, a.d[2] * ds}};
}

// R(v) of kornia 0.6.5 quaternion_log_to_exp + quaternion_to_rotation_matrix (as restated in
// quat_transform, mf_nets.hpp) with partials; R[3*i + j].
MF_D void quat_rotation_d3(const float (&v)[3], D3 (&R)[9]) {
  const D3 vx{v[0], {1.f, 0.f, 0.f}}, vy{v[1], {0.f, 1.f, 0.f}}, vz{v[2], {0.f, 0.f, 1.f}};
  const D3 n = d3_clamp_min(d3_sqrt(vx * vx + vy * vy + vz * vz), 1e-8f);
  float sn, cn;
  sincosf(n.v, &sn, &cn);
  const D3 s = d3_scale(n, sn, cn), c = d3_scale(n, cn, -sn);
  const D3 sn_n = s / n;
  D3 qx = vx * sn_n, qy = vy * sn_n, qz = vz * sn_n, qw = c;
  const D3 qn = d3_clamp_min(d3_sqrt(qx * qx + qy * qy + qz * qz + qw * qw), 1e-12f);
  qx = qx / qn; qy = qy / qn; qz = qz / qn; qw = qw / qn;
  const D3 two = d3c(2.f), one = d3c(1.f);
  const D3 tx = two * qx, ty = two * qy, tz = two * qz;
  const D3 twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const D3 txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const D3 tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = one - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = one - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = one - (txx + tyy);
}

// out = (x - s) R + s + t  (row vector, nof.py:80):  d T (9) and d x (3) from d out (3)
MF_D void quat_transform_backward(const float (&T)[9], const float (&x)[3], const float (&go)[3], float (&dT)[9],
                                  float (&dx)[3]) {
  const float v[3] = {T[0], T[1], T[2]};
  D3 R[9];
  quat_rotation_d3(v, R);
  const float p[3] = {x[0] - T[3], x[1] - T[4], x[2] - T[5]};
#pragma unroll
  for (int i = 0; i < 3; ++i) dx[i] = R[3 * i + 0].v * go[0] + R[3 * i + 1].v * go[1] + R[3 * i + 2].v * go[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) a += p[i] * go[j] * R[3 * i + j].d[k];
    dT[k] = a;                       // d v
    dT[3 + k] = go[k] - dx[k];       // d s
    dT[6 + k] = go[k];               // d t
  }
}


}  // namespace mf
