// mf_smpl.hip -- SMPL linear blend skinning and the per-vertex transforms the datasets turn into NoF supervision
// (SURVEY.md §8f row 4):
//   mf_smpl_lbs                 : SMPL.forward / SMPL.get_vertex_transformation   utils/smpl/smpl_model.py:96-186
//   mf_smpl_frame_transforms    : T_tgt @ inverse(T_src) per vertex               datasets/moco_flow_dataset.py:96-99
//   mf_apply_vertex_transforms  : cano = (trans[ind] @ [query, 1])[:3]            datasets/moco_flow_dataset.py:127-129
// All of it is small (V = 6890 vertices, 24 joints) and HBM / latency bound: the 17 MB of pose blend shapes are read
// once, coalesced (one wave per vertex row triple); everything else fits in L2.  fp32 like the reference.
#include "mf_host.hpp"

namespace mf {

struct SmplParams {
  int V, B, pose_is_rot;
  const float *v_template, *shapedirs, *posedirs, *jreg, *weights;
  int parent[24];                       // parent[0] unused
  const float *pose, *beta;
  float *v_shaped, *J, *G, *lrot;       // scratch: (B,V*3), (B,72), (B,24*16), (B,208)
  float *verts, *T;                     // outputs (either may be null)
};

// v_shaped = shapedirs[:, :, :10] @ beta + v_template                                  smpl_model.py:100-103
__global__ void smpl_shape_kernel(SmplParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= p.V * 3) return;
  const float* sd = p.shapedirs + (size_t)i * 10;
  const float* be = p.beta + b * 10;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 10; ++k) acc = __builtin_fmaf(sd[k], be[k], acc);
  p.v_shaped[(size_t)b * p.V * 3 + i] = acc + p.v_template[i];
}

// J = J_regressor @ v_shaped: one workgroup per (joint, component)                      smpl_model.py:105-108
__global__ __launch_bounds__(256) void smpl_joints_kernel(SmplParams p) {
  const int j = blockIdx.x / 3, c = blockIdx.x % 3, b = blockIdx.y;
  const float* jr = p.jreg + (size_t)j * p.V;
  const float* vs = p.v_shaped + (size_t)b * p.V * 3;
  float acc = 0.f;
  for (int v = threadIdx.x; v < p.V; v += 256) acc = __builtin_fmaf(jr[v], vs[v * 3 + c], acc);
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.J[b * 72 + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// rodrigues + kinematic chain + rest-pose removal: one wave per batch row          smpl_model.py:40-55, 110-135
__global__ __launch_bounds__(64) void smpl_pose_kernel(SmplParams p) {
  __shared__ float R[24][9], Jl[24][3], Gl[24][16], G[24][16];
  const int t = threadIdx.x, b = blockIdx.x;
  if (t < 24) {
    if (p.pose_is_rot) {
#pragma unroll
      for (int k = 0; k < 9; ++k) R[t][k] = p.pose[((size_t)b * 24 + t) * 9 + k];
    } else {
      const float x = p.pose[((size_t)b * 24 + t) * 3 + 0], y = p.pose[((size_t)b * 24 + t) * 3 + 1],
                  z = p.pose[((size_t)b * 24 + t) * 3 + 2];
      const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;           // norm of theta + 1e-8, :47
      const float n = sqrtf(ex * ex + ey * ey + ez * ez);
      const float nx = x / n, ny = y / n, nz = z / n;                        // theta / angle, :49
      const float half = n * 0.5f;
      float sn, cs;
      sincosf(half, &sn, &cs);
      float qw = cs, qx = sn * nx, qy = sn * ny, qz = sn * nz;
      const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);         // quat2mat normalises again, :25
      qw /= qn; qx /= qn; qy /= qn; qz /= qn;
      const float w2 = qw * qw, x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
      const float wx = qw * qx, wy = qw * qy, wz = qw * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
      R[t][0] = w2 + x2 - y2 - z2; R[t][1] = 2.f * xy - 2.f * wz;   R[t][2] = 2.f * wy + 2.f * xz;
      R[t][3] = 2.f * wz + 2.f * xy; R[t][4] = w2 - x2 + y2 - z2;   R[t][5] = 2.f * yz - 2.f * wx;
      R[t][6] = 2.f * xz - 2.f * wy; R[t][7] = 2.f * wx + 2.f * yz; R[t][8] = w2 - x2 - y2 + z2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) Jl[t][c] = p.J[b * 72 + t * 3 + c];
  }
  __syncthreads();
  if (t < 24) {
    if (t >= 1) {                                                             // lrotmin = R[1:] - I, :117-119
#pragma unroll
      for (int k = 0; k < 9; ++k) p.lrot[b * 208 + (t - 1) * 9 + k] = R[t][k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
    }
    const int par = t >= 1 ? p.parent[t] : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {                                             // G_ = [[R, J - J[parent]], [0 0 0 1]], :122-126
#pragma unroll
      for (int c = 0; c < 3; ++c) Gl[t][r * 4 + c] = R[t][r * 3 + c];
      Gl[t][r * 4 + 3] = t >= 1 ? Jl[t][r] - Jl[par][r] : Jl[t][r];
    }
    Gl[t][12] = 0.f; Gl[t][13] = 0.f; Gl[t][14] = 0.f; Gl[t][15] = 1.f;
  }
  __syncthreads();
  if (t < 16) G[0][t] = Gl[0][t];
  __syncthreads();
  for (int i = 1; i < 24; ++i) {                                              // G[i] = G[parent] @ G_[i], :127-129
    if (t < 16) {
      const int r = t >> 2, c = t & 3, par = p.parent[i];
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(G[par][r * 4 + k], Gl[i][k * 4 + c], acc);
      G[i][t] = acc;
    }
    __syncthreads();
  }
  // G - G @ [0 | (J, 0)]: only the last column changes, :131-135
  for (int e = t; e < 24 * 16; e += 64) {
    const int i = e >> 4, r = (e >> 2) & 3, c = e & 3;
    float v = G[i][e & 15];
    if (c == 3) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc = __builtin_fmaf(G[i][r * 4 + k], Jl[i][k], acc);
      v = v - acc;
    }
    p.G[(size_t)b * 384 + e] = v;
  }
}

// per vertex (one wave): pose blend shapes, blended transform, skinned position      smpl_model.py:120-121, 136-139
__global__ __launch_bounds__(256) void smpl_skin_kernel(SmplParams p) {
  __shared__ float sG[384], sL[208], sT[4][16], sP[4][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
  for (int e = threadIdx.x; e < 384; e += 256) sG[e] = p.G[(size_t)b * 384 + e];
  for (int e = threadIdx.x; e < 207; e += 256) sL[e] = p.lrot[b * 208 + e];
  __syncthreads();
  const int v = blockIdx.x * 4 + w;
  if (v >= p.V) return;
  if (p.verts) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pd = p.posedirs + ((size_t)v * 3 + c) * 207;
      float acc = 0.f;
      for (int k = lane; k < 207; k += 64) acc = __builtin_fmaf(pd[k], sL[k], acc);
      for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
      if (lane == 0) sP[w][c] = p.v_shaped[((size_t)b * p.V + v) * 3 + c] + acc;
    }
  }
  if (lane < 16) {
    const float* wt = p.weights + (size_t)v * 24;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 24; ++j) acc = __builtin_fmaf(wt[j], sG[j * 16 + lane], acc);
    sT[w][lane] = acc;
    if (p.T) p.T[((size_t)b * p.V + v) * 16 + lane] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (p.verts && lane < 3) {
    const float* Tr = &sT[w][lane * 4];
    float acc = Tr[0] * sP[w][0];
    acc = __builtin_fmaf(Tr[1], sP[w][1], acc);
    acc = __builtin_fmaf(Tr[2], sP[w][2], acc);
    p.verts[((size_t)b * p.V + v) * 3 + lane] = acc + Tr[3];
  }
}

// trans = T_tgt @ inverse(T_src), general 4x4 inverse by 2x2 minors                     moco_flow_dataset.py:96-99
__global__ void smpl_compose_kernel(const float* Ts, const float* Tt, long long V, float* out) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float a[16], t[16], inv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { a[e] = Ts[v * 16 + e]; t[e] = Tt[v * 16 + e]; }
  const float s0 = a[0] * a[5] - a[4] * a[1], s1 = a[0] * a[6] - a[4] * a[2], s2 = a[0] * a[7] - a[4] * a[3];
  const float s3 = a[1] * a[6] - a[5] * a[2], s4 = a[1] * a[7] - a[5] * a[3], s5 = a[2] * a[7] - a[6] * a[3];
  const float c5 = a[10] * a[15] - a[14] * a[11], c4 = a[9] * a[15] - a[13] * a[11], c3 = a[9] * a[14] - a[13] * a[10];
  const float c2 = a[8] * a[15] - a[12] * a[11], c1 = a[8] * a[14] - a[12] * a[10], c0 = a[8] * a[13] - a[12] * a[9];
  const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  const float id = 1.f / det;
  inv[0] = (a[5] * c5 - a[6] * c4 + a[7] * c3) * id;
  inv[1] = (-a[1] * c5 + a[2] * c4 - a[3] * c3) * id;
  inv[2] = (a[13] * s5 - a[14] * s4 + a[15] * s3) * id;
  inv[3] = (-a[9] * s5 + a[10] * s4 - a[11] * s3) * id;
  inv[4] = (-a[4] * c5 + a[6] * c2 - a[7] * c1) * id;
  inv[5] = (a[0] * c5 - a[2] * c2 + a[3] * c1) * id;
  inv[6] = (-a[12] * s5 + a[14] * s2 - a[15] * s1) * id;
  inv[7] = (a[8] * s5 - a[10] * s2 + a[11] * s1) * id;
  inv[8] = (a[4] * c4 - a[5] * c2 + a[7] * c0) * id;
  inv[9] = (-a[0] * c4 + a[1] * c2 - a[3] * c0) * id;
  inv[10] = (a[12] * s4 - a[13] * s2 + a[15] * s0) * id;
  inv[11] = (-a[8] * s4 + a[9] * s2 - a[11] * s0) * id;
  inv[12] = (-a[4] * c3 + a[5] * c1 - a[6] * c0) * id;
  inv[13] = (a[0] * c3 - a[1] * c1 + a[2] * c0) * id;
  inv[14] = (-a[12] * s3 + a[13] * s1 - a[14] * s0) * id;
  inv[15] = (a[8] * s3 - a[9] * s1 + a[10] * s0) * id;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(t[r * 4 + k], inv[k * 4 + c], acc);
      out[v * 16 + r * 4 + c] = acc;
    }
}

// cano = (trans[ind] @ [query, 1])[:3]                                               moco_flow_dataset.py:127-129
__global__ void smpl_apply_kernel(const float* trans, const long long* ind, long long V, const float* q, long long Q, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q) return;
  long long v = ind[i];
  v = v < 0 ? 0 : (v >= V ? V - 1 : v);
  const float* T = trans + v * 16;
  const float x = q[i * 3], y = q[i * 3 + 1], z = q[i * 3 + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = T[r * 4] * x;
    acc = __builtin_fmaf(T[r * 4 + 1], y, acc);
    acc = __builtin_fmaf(T[r * 4 + 2], z, acc);
    out[i * 3 + r] = acc + T[r * 4 + 3];
  }
}

}  // namespace mf

using namespace mf;

static int64_t smpl_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t mf_smpl_scratch_bytes(int64_t n_verts, int64_t B) {
  if (n_verts < 1 || B < 0) return 0;
  return smpl_align(B * n_verts * 3 * 4) + smpl_align(B * 72 * 4) + smpl_align(B * 384 * 4) + smpl_align(B * 208 * 4);
}

extern "C" int32_t mf_smpl_lbs(const mf_smpl_model* m, const float* pose, int32_t pose_is_rotmat, const float* betas, int64_t B,
                               float* verts, float* T, void* scratch, void* stream) {
  if (!m || m->n_verts < 1 || !m->v_template || !m->shapedirs || !m->posedirs || !m->j_regressor || !m->weights)
    return fail(MF_E_INVALID, "mf_smpl_lbs: incomplete model");
  for (int i = 1; i < 24; ++i)
    if (m->parent[i] < 0 || m->parent[i] >= i) return fail(MF_E_INVALID, "mf_smpl_lbs: parent[%d] = %d is not an earlier joint", i, m->parent[i]);
  if (B < 0 || B > 65535) return fail(MF_E_INVALID, "mf_smpl_lbs: batch %lld", (long long)B);
  if (B == 0 || (!verts && !T)) return MF_OK;
  if (!pose || !betas || !scratch) return fail(MF_E_INVALID, "mf_smpl_lbs: null argument");
  SmplParams p{};
  p.V = m->n_verts; p.B = (int)B; p.pose_is_rot = pose_is_rotmat;
  p.v_template = m->v_template; p.shapedirs = m->shapedirs; p.posedirs = m->posedirs; p.jreg = m->j_regressor; p.weights = m->weights;
  for (int i = 0; i < 24; ++i) p.parent[i] = m->parent[i];
  p.pose = pose; p.beta = betas; p.verts = verts; p.T = T;
  char* s = static_cast<char*>(scratch);
  p.v_shaped = reinterpret_cast<float*>(s); s += smpl_align(B * p.V * 3 * 4);
  p.J = reinterpret_cast<float*>(s); s += smpl_align(B * 72 * 4);
  p.G = reinterpret_cast<float*>(s); s += smpl_align(B * 384 * 4);
  p.lrot = reinterpret_cast<float*>(s);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(smpl_shape_kernel, dim3((p.V * 3 + 255) / 256, (unsigned)B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(smpl_joints_kernel, dim3(72, (unsigned)B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(smpl_pose_kernel, dim3((unsigned)B), dim3(64), 0, st, p);
  hipLaunchKernelGGL(smpl_skin_kernel, dim3((p.V + 3) / 4, (unsigned)B), dim3(256), 0, st, p);
  return check_launch("mf_smpl_lbs");
}

extern "C" int32_t mf_smpl_frame_transforms(const float* T_src, const float* T_tgt, int64_t V, float* trans, void* stream) {
  if (V < 0) return fail(MF_E_INVALID, "mf_smpl_frame_transforms: V=%lld", (long long)V);
  if (V == 0) return MF_OK;
  if (!T_src || !T_tgt || !trans) return fail(MF_E_INVALID, "mf_smpl_frame_transforms: null argument");
  hipLaunchKernelGGL(smpl_compose_kernel, dim3((unsigned)((V + 127) / 128)), dim3(128), 0, static_cast<hipStream_t>(stream), T_src, T_tgt,
                     (long long)V, trans);
  return check_launch("mf_smpl_frame_transforms");
}

extern "C" int32_t mf_apply_vertex_transforms(const float* trans, const int64_t* ind, int64_t V, const float* query, int64_t Q,
                                              float* cano, void* stream) {
  if (V < 1 || Q < 0) return fail(MF_E_INVALID, "mf_apply_vertex_transforms: V=%lld Q=%lld", (long long)V, (long long)Q);
  if (Q == 0) return MF_OK;
  if (!trans || !ind || !query || !cano) return fail(MF_E_INVALID, "mf_apply_vertex_transforms: null argument");
  hipLaunchKernelGGL(smpl_apply_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), trans,
                     reinterpret_cast<const long long*>(ind), (long long)V, query, (long long)Q, cano);
  return check_launch("mf_apply_vertex_transforms");
}
