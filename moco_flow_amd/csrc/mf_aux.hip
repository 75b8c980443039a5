// mf_aux.hip -- the producers on either side of the hot path:
//   mf_make_rays : Camera.make_rays / gen_ray_directions / gen_rays   utils/camera.py:29-81, 134-148
//   mf_image_compose : the foreground scatter-back of MoCoFlowTrainer.render / NeRFTrainer.render
//                      trainer/trainer_moco_flow.py:249-266, trainer/trainer_nerf.py:128-140
//   mf_valid_rays_mask : Camera.get_valid_rays_mask  utils/camera.py:119-132 (hull + fill of the projected AABB)
//   mf_knn1      : knn_cuda.KNN(k=1)  (vendored wheel docker/KNN_CUDA-0.2: knn_cuda/csrc/cuda/knn.cu:29-183)
#include "mf_host.hpp"

namespace mf {

struct RaysParams {
  int H, W;
  float fx, cx, cy;
  float R[9], t[3];
  int has_c2w;
  float nearv, farv, idx;
  float* out;
};

// one thread per pixel; row-major pixel order (row j, column i), 9 floats per ray
__global__ void make_rays_kernel(RaysParams p) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (long long)p.H * p.W) return;
  const int j = (int)(n / p.W), i = (int)(n - (long long)j * p.W);
  // camera.py:47-48: ((i - cx)/f0, -(j - cy)/f0, -1); both axes use focal[0]
  const float dx = ((float)i - p.cx) / p.fx;
  const float dy = -(((float)j - p.cy) / p.fx);
  const float dz = -1.f;
  float wx, wy, wz, ox = 0.f, oy = 0.f, oz = 0.f;
  if (p.has_c2w) {
    // camera.py:73: directions @ c2w[:, :3].T  (dot over the camera axes, in order)
    wx = dx * p.R[0] + dy * p.R[1] + dz * p.R[2];
    wy = dx * p.R[3] + dy * p.R[4] + dz * p.R[5];
    wz = dx * p.R[6] + dy * p.R[7] + dz * p.R[8];
    ox = p.t[0]; oy = p.t[1]; oz = p.t[2];
  } else {
    wx = dx; wy = dy; wz = dz;
  }
  const float nrm = sqrtf(wx * wx + wy * wy + wz * wz);      // camera.py:68/74
  float* o = p.out + n * 9;
  o[0] = ox; o[1] = oy; o[2] = oz;
  o[3] = wx / nrm; o[4] = wy / nrm; o[5] = wz / nrm;
  o[6] = p.nearv; o[7] = p.farv; o[8] = p.idx;
}

struct ComposeParams {
  const unsigned char* msk;     // (B) 0/1, or null: every ray was rendered
  const long long* rank;        // (B) index of ray b in the rendered arrays (valid where msk)
  long long B;
  const float* opacity;         // (M)
  const float* rgb;             // (M,3)
  const float* depth;           // (M)
  const float* background;      // (B,3)
  float* img;                   // (B,3)
  float* depth_out;             // (B)
};

// One thread per pixel.  trainer_moco_flow.py:252-263: img = 0, depth = 10; masked rays get depth 8;
// rays whose rendered opacity is > 0 take the rendered colour / depth; rays with foreground_mask == 0
// (not rendered, or rendered with opacity exactly 0) take the background colour.
__global__ void image_compose_kernel(ComposeParams p) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const bool m = p.msk ? p.msk[b] != 0 : true;
  float r = 0.f, g = 0.f, bl = 0.f, d = 10.f, fg = 0.f;
  long long k = 0;
  if (m) {
    k = p.rank ? p.rank[b] : b;
    fg = p.opacity[k];
    d = 8.f;
  }
  if (fg > 0.f) {
    r = p.rgb[k * 3]; g = p.rgb[k * 3 + 1]; bl = p.rgb[k * 3 + 2];
    d = p.depth[k];
  } else if (fg == 0.f) {
    r = p.background[b * 3]; g = p.background[b * 3 + 1]; bl = p.background[b * 3 + 2];
  }
  p.img[b * 3] = r; p.img[b * 3 + 1] = g; p.img[b * 3 + 2] = bl;
  p.depth_out[b] = d;
}

struct MaskParams {
  int H, W, n;
  int hx[8], hy[8];            // convex hull vertices (x = column, y = row), any orientation
  int ymin, ymax, imin;        // rows of the hull; first vertex with the smallest row
  // the hull's edges as cv2's Line() rasterises them: clipped to the image (clipLine, on the host), traversed left to right
  int lx1[8], ly1[8], lx2[8], ly2[8], lvalid[8];
  unsigned char* out;
};

// Camera.get_valid_rays_mask (utils/camera.py:119-132) = cv2.fillConvexPoly(mask, cv2.convexHull(pts), 255), one thread per
// pixel, restating OpenCV's algorithm (modules/imgproc/src/drawing.cpp, FillConvexPoly with line_type 8, shift 0) in
// closed form.  PARITY UNPINNED vs cv2 itself (absent from the image); oracle/cpu_ref.py::valid_rays_mask restates the same
// algorithm as the sequential loops OpenCV runs, and the two must agree bit for bit.  A pixel is set iff it lies on
//  (a) the OUTLINE: every hull edge drawn by Line() -- clipLine, then the 8-connected Bresenham walk of LineIterator from
//      the leftmost endpoint: after k steps along the major axis the minor coordinate has advanced
//      m_k = floor((2 dmin k + dmaj - 1) / (2 dmaj))   (err starts at dmaj - 2 dmin; a step is diagonal iff err < 0); or
//  (b) a SPAN: rows ymin .. min(ymax - 1, H - 1) -- the scan-line loop runs out of edges at the hull's last row, which is
//      covered by the outline only -- between the two edge chains that descend from the top vertex, each stepped in
//      16.16 fixed point: x(y) = (xa << 16) + dx (y - ya), dx = (((xb - xa) << 17) + (yb - ya)) / (2 (yb - ya)) (C
//      division, truncating), pixels (x_left + 0.5) >> 16 .. (x_right + 0.5) >> 16.
// (Round 2's rule -- exact intersections, no outline -- dropped the outline's runs along shallow edges.)
__device__ inline bool on_cv_line(int x, int y, int x1, int y1, int x2, int y2) {
  long long dx = (long long)x2 - x1, dy = (long long)y2 - y1;
  if (dx < 0) { dx = -dx; dy = -dy; const int tx = x1, ty = y1; x1 = x2; y1 = y2; x2 = tx; y2 = ty; }   // leftToRight
  const int sy = dy < 0 ? -1 : 1;
  if (dy < 0) dy = -dy;
  if (dy > dx) {                                 // the walk is along y, x advances by +1 on the diagonal steps
    const long long k = (long long)(y - y1) * sy;
    if (k < 0 || k > dy) return false;
    return x == x1 + (int)((2 * dx * k + dy - 1) / (2 * dy));
  }
  const long long k = (long long)x - x1;
  if (k < 0 || k > dx) return false;
  if (dx == 0) return y == y1;                   // a single point
  return y == y1 + sy * (int)((2 * dy * k + dx - 1) / (2 * dx));
}

__global__ void valid_mask_kernel(MaskParams p) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.H * p.W) return;
  const int y = (int)(idx / p.W), x = (int)(idx - (long long)y * p.W);
  bool in = false;
  const int n = p.n;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (e < (n > 1 ? n : 1) && p.lvalid[e] && on_cv_line(x, y, p.lx1[e], p.ly1[e], p.lx2[e], p.ly2[e])) in = true;
  const int last = p.ymax - 1 < p.H - 1 ? p.ymax - 1 : p.H - 1;
  if (!in && n >= 3 && y >= p.ymin && y <= last) {
    long long xe[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {                // the two chains from vertex imin: +1 and -1 through the hull
      int i0 = p.imin;
      long long xf = 0;
      for (int s = 0; s < n; ++s) {
        int i1 = c == 0 ? i0 + 1 : i0 - 1;
        i1 = i1 >= n ? i1 - n : (i1 < 0 ? i1 + n : i1);
        // (runtime indices into the by-value argument would put a private copy of it in scratch: select instead)
        int ax = 0, ay = 0, bx = 0, by = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (q == i0) { ax = p.hx[q]; ay = p.hy[q]; }
          if (q == i1) { bx = p.hx[q]; by = p.hy[q]; }
        }
        if (by > y) {                            // the edge that is active in row y (horizontal edges are passed over)
          const long long num = (((long long)(bx - ax)) << 17) + (by - ay);
          xf = (((long long)ax) << 16) + (num / (2LL * (by - ay))) * (y - ay);
          break;
        }
        i0 = i1;
      }
      xe[c] = xf;
    }
    const long long xl = xe[0] < xe[1] ? xe[0] : xe[1], xr = xe[0] < xe[1] ? xe[1] : xe[0];
    const long long x1 = (xl + 32768) >> 16, x2 = (xr + 32768) >> 16;
    in = x2 >= 0 && x1 < p.W && x >= x1 && x <= x2;
  }
  p.out[idx] = in ? 1 : 0;
}

// cv::clipLine (drawing.cpp) on the image rectangle [0, W-1] x [0, H-1]; false = nothing of the segment is inside
static bool cv_clip_line(long long W, long long H, long long& x1, long long& y1, long long& x2, long long& y2) {
  const long long right = W - 1, bottom = H - 1;
  if (W <= 0 || H <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (x2 - x1) / (y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (x2 - x1) / (y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (y2 - y1) / (x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (y2 - y1) / (x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

struct KnnParams {
  const float* ref; long long V;
  const float* query; long long Q;
  float* dist; long long* ind;
};

// k = 1 nearest reference point of each query: one thread per query, reference points staged
// through LDS in tiles; first minimum wins on ties (the wheel's insertion sort uses strict '<').
__global__ __launch_bounds__(256) void knn1_kernel(KnnParams p) {
  __shared__ float tile[1024 * 3];
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = q < p.Q;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (valid) { qx = p.query[q * 3]; qy = p.query[q * 3 + 1]; qz = p.query[q * 3 + 2]; }
  float best = __builtin_inff();
  long long besti = 0;
  for (long long base = 0; base < p.V; base += 1024) {
    const int n = (int)((p.V - base) < 1024 ? (p.V - base) : 1024);
    __syncthreads();
    for (int k = threadIdx.x; k < n * 3; k += blockDim.x) tile[k] = p.ref[base * 3 + k];
    __syncthreads();
    for (int v = 0; v < n; ++v) {
      const float ax = tile[v * 3] - qx, ay = tile[v * 3 + 1] - qy, az = tile[v * 3 + 2] - qz;
      const float d = __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax));   // knn.cu:75-79 (nvcc fmad)
      if (d < best) { best = d; besti = base + v; }
    }
  }
  if (valid) {
    p.dist[q] = sqrtf(best);                                  // knn.cu:178-183
    p.ind[q] = besti;                                         // 0-based (knn_cuda/__init__.py:45)
  }
}

// rendering.py:245-251: z_vals (N, S) from the rays' near / far and the S linspace steps, separately rounded
// multiplies and adds exactly as the torch expression (and as the fused pass computes them in registers)
// With perturb_rand (N, S): the stratified jitter of rendering.py:253-260 in the same launch --
//   mid = 0.5 (z[:, :-1] + z[:, 1:]); upper = [mid, z_last]; lower = [z_0, mid]; z = lower + (upper - lower) * (perturb * rand)
// -- every operation separately rounded in the torch expression's order (this unit is built with -ffp-contract=off).
__global__ void z_vals_kernel(const float* rays, long long ray_stride, long long n_rays, const float* z_steps, int S,
                              int use_disp, const float* perturb_rand, float perturb, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * S) return;
  const long long ray = i / S;
  const int si = (int)(i - ray * S);
  const float nearv = rays[ray * ray_stride + 6], farv = rays[ray * ray_stride + 7];
  auto zat = [&](int k) {
    const float t = z_steps[k];
    if (!use_disp) return nearv * (1.f - t) + farv * t;
    return 1.f / (1.f / nearv * (1.f - t) + 1.f / farv * t);
  };
  const float z = zat(si);
  if (!perturb_rand) { out[i] = z; return; }
  const float lower = si > 0 ? 0.5f * (zat(si - 1) + z) : z;
  const float upper = si + 1 < S ? 0.5f * (z + zat(si + 1)) : z;
  out[i] = lower + (upper - lower) * (perturb * perturb_rand[i]);
}

}  // namespace mf

using namespace mf;

extern "C" int32_t mf_z_vals(const float* rays, int64_t ray_stride, int64_t n_rays, const float* z_steps, int32_t n_samples,
                             int32_t use_disp, const float* perturb_rand, float perturb, float* z_out, void* stream) {
  if (n_rays < 0 || n_samples < 1 || ray_stride < 8) return fail(MF_E_INVALID, "mf_z_vals: n_rays=%lld n_samples=%d ray_stride=%lld",
                                                                  (long long)n_rays, n_samples, (long long)ray_stride);
  if (n_rays == 0) return MF_OK;
  if (!rays || !z_steps || !z_out) return fail(MF_E_INVALID, "mf_z_vals: null argument");
  const long long n = n_rays * n_samples;
  hipLaunchKernelGGL(z_vals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rays,
                     (long long)ray_stride, (long long)n_rays, z_steps, n_samples, use_disp, perturb_rand, perturb, z_out);
  return check_launch("mf_z_vals");
}

extern "C" int32_t mf_make_rays(int32_t H, int32_t W, float focal, float cx, float cy, const float* c2w_host,
                                float nearv, float farv, float idx, float* rays_out, void* stream) {
  if (H < 0 || W < 0 || focal == 0.f) return fail(MF_E_INVALID, "mf_make_rays: H=%d W=%d focal=%g", H, W, focal);
  if ((long long)H * W == 0) return MF_OK;
  if (!rays_out) return fail(MF_E_INVALID, "mf_make_rays: null output");
  RaysParams p{};
  p.H = H; p.W = W; p.fx = focal; p.cx = cx; p.cy = cy;
  p.has_c2w = c2w_host != nullptr;
  if (c2w_host) {
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) p.R[a * 3 + b] = c2w_host[a * 4 + b];
      p.t[a] = c2w_host[a * 4 + 3];
    }
  }
  p.nearv = nearv; p.farv = farv; p.idx = idx; p.out = rays_out;
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(make_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_make_rays");
}

extern "C" int32_t mf_valid_rays_mask(int32_t H, int32_t W, const int32_t* pts_xy_host, int32_t n_pts, uint8_t* mask_out,
                                      void* stream) {
  if (H < 0 || W < 0 || n_pts < 0 || n_pts > 64 || (n_pts > 0 && !pts_xy_host))
    return fail(MF_E_INVALID, "mf_valid_rays_mask: H=%d W=%d n_pts=%d", H, W, n_pts);
  if ((long long)H * W == 0) return MF_OK;
  if (!mask_out) return fail(MF_E_INVALID, "mf_valid_rays_mask: null output");
  // convex hull on the host (Andrew's monotone chain; <= 64 points, the reference passes the 8 AABB corners)
  long long px[64], py[64];
  int m = 0;
  for (int i = 0; i < n_pts; ++i) { px[m] = pts_xy_host[2 * i]; py[m] = pts_xy_host[2 * i + 1]; ++m; }
  for (int i = 1; i < m; ++i)                                    // insertion sort by (x, y)
    for (int j = i; j > 0 && (px[j] < px[j - 1] || (px[j] == px[j - 1] && py[j] < py[j - 1])); --j) {
      long long t = px[j]; px[j] = px[j - 1]; px[j - 1] = t;
      t = py[j]; py[j] = py[j - 1]; py[j - 1] = t;
    }
  int u = 0;
  for (int i = 0; i < m; ++i)                                    // unique
    if (i == 0 || px[i] != px[u - 1] || py[i] != py[u - 1]) { px[u] = px[i]; py[u] = py[i]; ++u; }
  m = u;
  long long hx[130], hy[130];
  int k = 0;
  auto cross = [&](int o, long long ax, long long ay, long long bx, long long by) {
    return (ax - hx[o]) * (by - hy[o]) - (ay - hy[o]) * (bx - hx[o]);
  };
  if (m <= 2) {
    for (int i = 0; i < m; ++i) { hx[k] = px[i]; hy[k] = py[i]; ++k; }
  } else {
    for (int i = 0; i < m; ++i) {
      while (k >= 2 && cross(k - 2, hx[k - 1], hy[k - 1], px[i], py[i]) <= 0) --k;
      hx[k] = px[i]; hy[k] = py[i]; ++k;
    }
    const int lower = k + 1;
    for (int i = m - 2; i >= 0; --i) {
      while (k >= lower && cross(k - 2, hx[k - 1], hy[k - 1], px[i], py[i]) <= 0) --k;
      hx[k] = px[i]; hy[k] = py[i]; ++k;
    }
    --k;                                                         // the last point repeats the first
  }
  if (k > 8) return fail(MF_E_UNSUPPORTED, "mf_valid_rays_mask: hull with %d vertices (max 8: the AABB corners)", k);
  MaskParams p{};
  p.H = H; p.W = W; p.n = k; p.out = mask_out;
  p.ymin = 1; p.ymax = 0;
  for (int i = 0; i < k; ++i) {
    p.hx[i] = (int)hx[i]; p.hy[i] = (int)hy[i];
    if (i == 0 || p.hy[i] < p.ymin) { p.ymin = p.hy[i]; p.imin = i; }
    if (i == 0 || p.hy[i] > p.ymax) p.ymax = p.hy[i];
  }
  for (int i = 0; i < (k > 1 ? k : (k == 1 ? 1 : 0)); ++i) {      // edge i: vertex i-1 -> vertex i (FillConvexPoly's Line() calls)
    const int j = (i + k - 1) % k;
    long long x1 = hx[j], y1 = hy[j], x2 = hx[i], y2 = hy[i];
    p.lvalid[i] = cv_clip_line(W, H, x1, y1, x2, y2) ? 1 : 0;
    p.lx1[i] = (int)x1; p.ly1[i] = (int)y1; p.lx2[i] = (int)x2; p.ly2[i] = (int)y2;
  }
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(valid_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_valid_rays_mask");
}

extern "C" int32_t mf_knn1(const float* ref, int64_t V, const float* query, int64_t Q, float* dist, int64_t* ind,
                           void* stream) {
  if (V < 1 || Q < 0) return fail(MF_E_INVALID, "mf_knn1: V=%lld Q=%lld", (long long)V, (long long)Q);
  if (Q == 0) return MF_OK;
  if (!ref || !query || !dist || !ind) return fail(MF_E_INVALID, "mf_knn1: null argument");
  KnnParams p{ref, V, query, Q, dist, reinterpret_cast<long long*>(ind)};
  hipLaunchKernelGGL(knn1_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_knn1");
}

extern "C" int32_t mf_image_compose(const uint8_t* rays_msk, const int64_t* rank, int64_t B, const float* opacity,
                                    const float* rgb, const float* depth, const float* background, float* img,
                                    float* depth_out, void* stream) {
  if (B < 0 || (B > 0 && (!opacity || !rgb || !depth || !background || !img || !depth_out)))
    return fail(MF_E_INVALID, "mf_image_compose: null argument");
  if (rays_msk && !rank) return fail(MF_E_INVALID, "mf_image_compose: a mask needs the rank array");
  if (B == 0) return MF_OK;
  ComposeParams p{rays_msk, reinterpret_cast<const long long*>(rank), B, opacity, rgb, depth, background, img, depth_out};
  hipLaunchKernelGGL(image_compose_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_image_compose");
}

// ---- mf_nof_embed_rows (ABI v16): the NoF's embedded input [emb_xyz(point) zero-padded to 33 | emb_ind(index of the point's ray) | 0]
// as (P, 80) rows in NATURAL column order -- the X operand of the 128 x 80 weight-gradient blocks of the NoF's embedded-input
// layers (models/rendering.py:70-75, models/embedding.py:42-47) -- for training forwards that do not write that plane themselves
// (MF_PREC_BF16X3: the index block is a per-ray bias there and never exists per sample).  64 threads per row: 15 evaluate one
// (frequency, component) pair each -- ONE sincosf, two columns --, 3 copy the raw point, 33 copy the ray's index block from the
// per-ray table `ind_emb` (the caller embeds the N indices once: the block is constant along a ray), 13 write the padding.
namespace mf {
struct NofEmbRowsParams { mf_embedding exyz; const float* pts; const float* ind_emb; int ind_width; int S; long long P; float* out; };

__global__ __launch_bounds__(256) void nof_embed_rows_kernel(const NofEmbRowsParams p) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.P) return;
  const int s = threadIdx.x & 63;
  float* o = p.out + row * 80;
  if (s < 15) {                                                           // [w_k sin(f_k x_c) | w_k cos(f_k x_c)], embedding.py:45: func(freq * x)
    const int k = s / 3, c = s % 3;
    float sn = 0.f, cs = 0.f;
    if (k < p.exyz.n_freqs) {
      sincosf(p.exyz.freq[k] * p.pts[row * 3 + c], &sn, &cs);
      sn *= p.exyz.weight[k]; cs *= p.exyz.weight[k];
    }
    o[3 + 6 * k + c] = sn;
    o[3 + 6 * k + 3 + c] = cs;
  } else if (s < 18) {
    o[s - 15] = p.pts[row * 3 + (s - 15)];
  } else if (s < 51) {
    const int c = s - 18;
    o[33 + c] = c < p.ind_width ? p.ind_emb[(row / p.S) * p.ind_width + c] : 0.f;
  } else {
    o[66 + (s - 51)] = 0.f;                                               // columns 66 .. 78
    if (s == 51) o[79] = 0.f;
  }
}
}  // namespace mf

extern "C" int32_t mf_nof_embed_rows(const mf_embedding* emb_xyz, const float* pts, const float* ind_emb, int32_t ind_width, int32_t S,
                                     int64_t P, float* out, void* stream) {
  if (!emb_xyz || P < 0 || S < 1 || ind_width < 0 || ind_width > 33 || (P > 0 && (!pts || !out || (ind_width > 0 && !ind_emb))))
    return fail(MF_E_INVALID, "mf_nof_embed_rows: bad argument");
  if (emb_xyz->in_channels != 3 || emb_xyz->n_freqs < 0 || emb_xyz->n_freqs > 5)
    return fail(MF_E_UNSUPPORTED, "mf_nof_embed_rows: xyz embedding 3 channels x <= 5 frequencies");
  if (P == 0) return MF_OK;
  NofEmbRowsParams p{*emb_xyz, pts, ind_emb, ind_width, S, P, out};
  hipLaunchKernelGGL(nof_embed_rows_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nof_embed_rows");
}
