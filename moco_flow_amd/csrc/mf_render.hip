// mf_render.hip -- one fused rendering pass of render_rays (models/rendering.py:195-375):
//
//   ray -> z (rendering.py:245-251) -> xyz (:262-263)
//       -> [bw NoF -> canonical xyz, fw NoF chains for the consensus terms (:270-286, :49-83)]
//       -> positional encoding (embedding.py:42-46) -> NeRF MLP (nerf.py:78-102)
//       -> sigma -> alpha -> exclusive transmittance scan -> rgb / depth / opacity (:157-192)
//
// in ONE kernel launch.  Nothing between the input rays and the output pixels is written to
// HBM except the optional (N,S) planes a caller asks for (weights/alphas feed sample_pdf and the
// consensus mask; disp_* are the per-sample consensus distances).
//
// Work decomposition: a workgroup (8 waves, two per SIMD, <= 256 registers each) takes a GROUP
// of G whole rays (G*S samples, a multiple of 128 whenever S allows), walks it in tiles of 128
// samples -- 16 per wave, one sample per lane&15, the four 16-lane groups holding the four
// k-quarters of every MFMA step -- and keeps each sample's (r,g,b,sigma,z) in LDS until the
// group's rays are composited by one wave per ray with a wavefront product-scan.  Workgroups
// are persistent (grid = #CUs) so the weight stream never drains between tiles.
#include <cstddef>
#include <cstdlib>

#include "mf_host.hpp"
#include "mf_layout.hpp"
#include "mf_nets.hpp"

namespace mf {

struct RenderParams {
  const float* rays; long long ray_stride; long long n_rays;
  const float* bg;
  int S;
  const float* z_vals; const float* z_steps; int use_disp;
  const float* noise;
  int activation, flags;
  NetDev nerf;
  int extra_type;
  NetDev bw, fw;
  float emb_par[4][32];            // [nerf xyz, nerf extra, nof xyz, nof ind] x (freq[16], weight[16]) -> LDS at par_off
  uint32_t par_off;
  float *rgb, *depth, *opacity, *weights, *alphas, *disp_local, *disp_global;
  int G;
  long long n_groups;
  uint32_t ring_off, buf_bytes, sbuf_off, zbuf_off;
  int dbg;
  float* dump_acts; long long dump_stride; float* dump_rgbsigma; float* dump_xyz;   // training forward
  unsigned* dump_mask; long long dump_mask_stride;                                  // ReLU bit mask rows (optional)
  float* dump_nof_acts; long long dump_nof_stride; float* dump_nof_emb; float* dump_nof_out;   // per chain step
  uint32_t nof_plane_pack;     // plane of step k = (pack >> 3k) & 7   (a packed scalar: no runtime index into the kernarg)
};

// inclusive product scan across the 64 lanes of a wave

template <bool MOCO, bool DUMP>
__global__ __launch_bounds__(kThreads, 2) void render_kernel(RenderParams p) {
  const LaneId id;
  const NetDev nerf = p.nerf;
#ifdef MF_TIMELINE
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  load_resident(nerf, id);
  if (MOCO) {
    load_resident(p.bw, id);
    if (p.flags & (MF_F_CHAIN_LOCAL | MF_F_CHAIN_GLOBAL)) load_resident(p.fw, id);
  }
  if (threadIdx.x < 128) {
    // embedding tables kernarg -> LDS through the kernarg segment pointer (a runtime index into the by-value struct
    // would make hipcc keep a scratch copy of all of `p`); published by start_program's barrier
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RenderParams, emb_par);
    *(float*)(smem + p.par_off + threadIdx.x * 4) = ((const __attribute__((address_space(4))) float*)ka)[threadIdx.x];
  }
  const uint32_t par_nerf_xyz = p.par_off, par_nerf_ext = p.par_off + 128, par_nof_xyz = p.par_off + 256,
                 par_nof_ind = p.par_off + 384;
  Stream st;
  CarryT<kPD> carry;
  st.ring = p.ring_off;
  st.buf_bytes = p.buf_bytes;
  st.dbg = MF_TIMING_FLAGS ? p.dbg : 0;
  st.keep2 = 0;
  st.tl.start(p.alphas, id);
  // the panel program of a tile: [bw NoF, fw NoF chains,] NeRF, then around again
  const NextLayer prog_first = MOCO ? follow_of(p.bw) : follow_of(nerf);
  if (MOCO) start_program(p.bw, st, carry, id);
  else start_program(nerf, st, carry, id);

  const int S = p.S;
  const bool sigma_only = p.flags & MF_F_SIGMA_ONLY;
  float4* sbuf = reinterpret_cast<float4*>(smem + p.sbuf_off);
  float* zbuf = reinterpret_cast<float*>(smem + p.zbuf_off);

  for (long long group = blockIdx.x; group < p.n_groups; group += gridDim.x) {
    const long long ray0 = group * p.G;
    const int nr = (int)((p.n_rays - ray0) < p.G ? (p.n_rays - ray0) : p.G);
    const int nsamp = nr * S;
    const int ntiles = (nsamp + kTile - 1) / kTile;

    for (int tile = 0; tile < ntiles; ++tile) {
      st.tl.stamp(1, id);
      // Per-sample bookkeeping (sample / ray indices, the ray's row pointer, the index columns) is NOT carried through the
      // tile: everything follows from the lane's column j and tile-uniform scalars, so each use site rebuilds what it
      // needs from an opaque copy of j (`where()`).  Held in registers from here, those values -- and the 64-bit
      // addresses hipcc derives from them ahead of time -- were what the MoCo training forward spilled (12 registers, 116
      // bytes of scratch per lane in round 2's build).  z and the observation-space point wait in the group's LDS
      // sample buffers (their slots are free until the tile's results are written).
      struct Where { int srel, sl, si; bool valid; long long ray; const float* rp; };
      auto where = [&]() {
        int jo = id.j;
        asm volatile("" : "+v"(jo));
        Where w;
        w.srel = tile * kTile + id.wave * kWaveSamples + jo;
        w.valid = w.srel < nsamp;
        w.sl = w.valid ? w.srel : nsamp - 1;
        const int rr = w.sl / S;
        w.si = w.sl - rr * S;
        w.ray = ray0 + rr;
        w.rp = p.rays + w.ray * p.ray_stride;
        return w;
      };
      float xin[3];                            // what the canonical NeRF sees
      {
        const Where w = where();
        const float* rp = w.rp;
        const float o[3] = {rp[0], rp[1], rp[2]};
        const float d[3] = {rp[3], rp[4], rp[5]};
        float z;
        if (p.z_vals) {
          z = p.z_vals[w.ray * S + w.si];
        } else {
          const float nearv = rp[6], farv = rp[7], t = p.z_steps[w.si];
          if (!p.use_disp) z = nearv * (1.f - t) + farv * t;                    // rendering.py:247
          else z = 1.f / (1.f / nearv * (1.f - t) + 1.f / farv * t);            // rendering.py:249
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) xin[c] = o[c] + d[c] * z;                    // rendering.py:262-263
        if (w.valid && id.g == 0) {
          zbuf[w.srel] = z;
          if (MOCO) sbuf[w.srel] = make_float4(xin[0], xin[1], xin[2], 0.f);
        }
      }
#ifdef MF_TIMELINE
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(xin[0]), "+v"(xin[1]), "+v"(xin[2]));
#endif
      st.tl.stamp(2, id);
      if (MOCO) {
        // chain program (rendering.py:270-282): step 0 bw(x,i) -> canon; local: fw(canon,i) -> recon;
        // global: fw(canon,j) -> a; bw(a,j) -> b; fw(b,i) -> chained recon.
        const bool loc = p.flags & MF_F_CHAIN_LOCAL, glob = p.flags & MF_F_CHAIN_GLOBAL;
        const int nsteps = 1 + (loc ? 1 : 0) + (glob ? 3 : 0);
        float canon[3] = {0.f, 0.f, 0.f}, cur[3] = {xin[0], xin[1], xin[2]};
        for (int step = 0; step < nsteps; ++step) {
          // role of this step: 0 = bw_i, 1 = local fw_i, 2 = fw_j, 3 = bw_j, 4 = final fw_i
          // (chain_global implies chain_local -- checked on the host -- so role == step)
          const int role = step;
          const bool use_fw = (role == 1 || role == 2 || role == 4);
          const NetDev net = use_fw ? p.fw : p.bw;
          const Where w = where();
          const float ind = w.rp[(role == 2 || role == 3) ? 9 : 8];
          if (role == 1 || role == 2) { cur[0] = canon[0]; cur[1] = canon[1]; cur[2] = canon[2]; }
          // what follows this evaluation in the panel program
          const bool last = step == nsteps - 1;
          const bool next_fw = (role + 1 == 1 || role + 1 == 2 || role + 1 == 4);
          const NextLayer follow = last ? follow_of(nerf) : (next_fw ? follow_of(p.fw) : follow_of(p.bw));
          float emb[kStepsNofIn], out[3];
          nof_embed_lds(emb, cur, ind, par_nof_xyz, par_nof_ind, id.g);
          float* nof_row = nullptr;
          if constexpr (DUMP) {
            if (w.valid && p.dump_nof_acts) {
              // training forward: what autograd.NofPoints' backward reads, per chain step (step-major planes)
              const long long nof_idx = (long long)((p.nof_plane_pack >> (3 * step)) & 7u) * p.n_rays * S + (w.ray * S + w.si);
              nof_row = p.dump_nof_acts + nof_idx * p.dump_nof_stride;
              // embedded input in the kernel's own slot order (column 20 g + e = slot e of lane group g: five 16-byte
              // stores per lane instead of twenty scattered dwords; mf_nof_emb_slot_features gives the column map)
              float4* e4 = reinterpret_cast<float4*>(p.dump_nof_emb + nof_idx * 80 + 20 * id.g);
#pragma unroll
              for (int q = 0; q < kStepsNofIn / 4; ++q) e4[q] = make_float4(emb[4 * q], emb[4 * q + 1], emb[4 * q + 2], emb[4 * q + 3]);
            }
            st.keep2 = 0;
          }
          nof_eval<DUMP>(net, emb, cur, st, carry, id, follow, out, nof_row,
                         DUMP && p.dump_nof_stride >= (long long)net.L.n_trunk * net.L.W + 16 + 4 * net.L.n_trunk);
          if constexpr (DUMP) {
            const Where v = where();
            if (v.valid && p.dump_nof_acts && id.g == 0) {
              const long long nof_idx = (long long)((p.nof_plane_pack >> (3 * step)) & 7u) * p.n_rays * S + (v.ray * S + v.si);
              float* q = p.dump_nof_out + nof_idx * 3;
              q[0] = out[0]; q[1] = out[1]; q[2] = out[2];
            }
          }
          if (role == 0) { canon[0] = out[0]; canon[1] = out[1]; canon[2] = out[2]; }
          if (role == 1 || role == 4) {
            const Where v = where();
            const float4 x4 = sbuf[v.sl];          // the observation-space point (own wave's write, or -- lanes past the
                                                   // group's last sample -- anything: their distances are never stored)
            const float dd = (fabsf(x4.x - out[0]) + fabsf(x4.y - out[1]) + fabsf(x4.z - out[2])) / 3.f;
            float* plane = role == 1 ? p.disp_local : p.disp_global;       // rendering.py:310-314, stored right away
            if (v.valid && id.g == 0 && plane) plane[v.ray * S + v.si] = dd;
          }
          cur[0] = out[0]; cur[1] = out[1]; cur[2] = out[2];
        }
        xin[0] = canon[0]; xin[1] = canon[1]; xin[2] = canon[2];
      }

      st.tl.stamp(3, id);
      float embx[kStepsNerfXyz], ext[kStepsExtraMax];
      if (!(MF_TIMING_FLAGS && (p.dbg & 4))) emb_eval_lds<3, 10>(embx, xin, par_nerf_xyz, id.g);
      else { for (int e = 0; e < kStepsNerfXyz; ++e) embx[e] = xin[e % 3]; }
#pragma unroll
      for (int e = BlkXyz10::SLOTS; e < kStepsNerfXyz; ++e) embx[e] = 0.f;
#pragma unroll
      for (int e = 0; e < kStepsExtraMax; ++e) ext[e] = 0.f;
      float* dump_row = nullptr;
      unsigned* mask_row = nullptr;
      {
        const Where w = where();
        if (!sigma_only) {
          if (p.extra_type == MF_EXTRA_DIR) {
            const float dd[3] = {w.rp[3], w.rp[4], w.rp[5]};
            emb_eval_lds<3, 4>(ext, dd, par_nerf_ext, id.g);                             // rendering.py:138-142
          } else if (p.extra_type == MF_EXTRA_IND) {
            const float iv[1] = {w.rp[8]};
            emb_eval_lds<1, 2>(ext, iv, par_nerf_ext, id.g);                             // rendering.py:133-137
          }
        }
        if constexpr (DUMP) {
          if (w.valid && p.dump_acts) dump_row = p.dump_acts + (w.ray * S + w.si) * p.dump_stride;
          // (round 4: the chain kernel writes them too -- one more spilled dword in its prologue, two scratch reloads per tile;
          //  the NeRF's dX chain under NoF then reads 32 bytes instead of 1 KiB per layer and sample)
          if (p.dump_mask) mask_row = p.dump_mask + (w.ray * S + w.si) * p.dump_mask_stride;   // (uniformly non-null when asked for)
        }
      }
      st.tl.stamp(4, id);
      float sigma, rgb[3] = {0.f, 0.f, 0.f};
      st.keep2 = 0;      // the first panel's barrier drains everything (see Stream::sync_and_dma)
      nerf_eval<16, DUMP>(nerf, embx, ext, sigma_only, st, carry, id, prog_first, sigma, rgb, dump_row, mask_row);
      {
        const Where w = where();
        if (w.valid && id.g == 0) {
          sbuf[w.srel] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
          if constexpr (DUMP) {
            const long long row = w.ray * S + w.si;
            if (p.dump_rgbsigma) *reinterpret_cast<float4*>(p.dump_rgbsigma + row * 4) = make_float4(rgb[0], rgb[1], rgb[2], sigma);
            if (p.dump_xyz) { float* q = p.dump_xyz + row * 3; q[0] = xin[0]; q[1] = xin[1]; q[2] = xin[2]; }
          }
        }
      }
      st.tl.stamp(5, id);
    }
    __syncthreads();
    st.tl.stamp(6, id);

    // ---- composite (rendering.py:157-192): one wave per ray, lanes over samples
    for (int rr = id.wave; rr < ((MF_TIMING_FLAGS && (p.dbg & 8)) ? 0 : nr); rr += kWaves) {
      const long long ray = ray0 + rr;
      const float* rp = p.rays + ray * p.ray_stride;
      const float dnorm = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);  // rendering.py:164
      float carry = 1.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_w = 0.f;
      for (int base = 0; base < S; base += 64) {
        // (opaque lane index: keeps hipcc from hoisting `plane + 4 lane` of every output plane out of the group loop
        //  as 64-bit per-lane addresses that then sit in -- or spill from -- registers across the MFMA section)
        int ln = id.lane;
        asm volatile("" : "+v"(ln));
        const int i = base + ln;
        const bool v = i < S;
        const int ii = v ? i : S - 1;
        const float4 s4 = sbuf[rr * S + ii];
        const float z = zbuf[rr * S + ii];
        const float znext = zbuf[rr * S + (ii + 1 < S ? ii + 1 : ii)];
        float delta = (ii == S - 1) ? 1e10f : znext - z;                       // :158-160
        delta = delta * dnorm;
        float sg = s4.w;
        if (p.noise) sg = sg + p.noise[ray * S + ii];                          // :166 (pre-scaled)
        float a;
        if (p.activation == MF_ACT_RELU) a = fmaxf(sg, 0.f);
        else a = sg > 20.f ? sg : log1pf(expf(sg));                            // nn.Softplus(beta=1, threshold=20)
        float alpha = 1.f - expf(-delta * a);                                  // :170/172
        if (!v) alpha = 0.f;
        const float pt = v ? (1.f - alpha) + 1e-10f : 1.f;                     // :176-177
        const float incl = wave_scan_mul_dpp(pt);
        const float excl = wave_shr1_dpp(1.f, incl);
        const float w = alpha * (carry * excl);                                // :178-179
        carry = carry * wave_last(incl);
        if (v) {
          if (p.weights) p.weights[ray * S + i] = w;
#ifndef MF_TIMELINE
          if (p.alphas) p.alphas[ray * S + i] = alpha;
#endif
          acc_w += w;
          acc_r += w * s4.x; acc_g += w * s4.y; acc_b += w * s4.z;
          acc_d += w * z;
        }
      }
      acc_w = wave_sum_dpp(acc_w);                                                 // :180
      if (!sigma_only) {
        acc_r = wave_sum_dpp(acc_r); acc_g = wave_sum_dpp(acc_g); acc_b = wave_sum_dpp(acc_b);   // :186
        acc_d = wave_sum_dpp(acc_d);                                               // :187
      }
      if (id.lane == 0) {
        if (p.opacity) p.opacity[ray] = acc_w;
        if (!sigma_only) {
          if (p.bg) {                                                          // :189-190
            const float k = 1.f - acc_w;
            acc_r = acc_r + p.bg[ray * 3 + 0] * k;
            acc_g = acc_g + p.bg[ray * 3 + 1] * k;
            acc_b = acc_b + p.bg[ray * 3 + 2] * k;
          }
          if (p.rgb) { p.rgb[ray * 3 + 0] = acc_r; p.rgb[ray * 3 + 1] = acc_g; p.rgb[ray * 3 + 2] = acc_b; }
          if (p.depth) p.depth[ray] = acc_d;
        }
      }
    }
    st.tl.stamp(7, id);
    __syncthreads();
    st.tl.stamp(8, id);
  }
  wait_vm0();   // the stream runs two panels ahead: drain the LDS-DMA before the workgroup retires
#ifdef MF_TIMELINE      // (timing builds only) every workgroup's start / end on the chip-wide 100 MHz clock, in alphas[2..3] of its last group
  if (threadIdx.x == 0 && p.alphas && blockIdx.x < p.n_groups) {
    const long long lastg = blockIdx.x + ((p.n_groups - 1 - blockIdx.x) / gridDim.x) * gridDim.x;
    float* o = p.alphas + lastg * p.G * p.S;
    o[2] = (float)(tl_rt0 & 0xFFFFFFull);
    o[3] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFFFull);
  }
#endif
}

static void to_table(const mf_embedding& e, float* o) {
  for (int k = 0; k < 16; ++k) {
    o[k] = k < e.n_freqs ? e.freq[k] : 0.f;
    o[16 + k] = k < e.n_freqs ? e.weight[k] : 0.f;
  }
}

int device_cus();   // mf_forward.hip
int render_pass_bf16(const mf_render_args* a, hipStream_t st, bool prepare_only);   // mf_render_bf16.hip
int64_t render_workspace_bytes_bf16(const mf_render_args* a);

}  // namespace mf

using namespace mf;

extern "C" int64_t mf_render_workspace_bytes(const mf_render_args* a) {
  if (!a || (a->precision != MF_PREC_BF16 && a->precision != MF_PREC_BF16X3) || a->n_rays <= 0) return 0;
  return render_workspace_bytes_bf16(a);
}

extern "C" int32_t mf_nof_emb_slot_features(int32_t* features80) {
  if (!features80) return fail(MF_E_INVALID, "mf_nof_emb_slot_features: null argument");
  for (int g = 0; g < 4; ++g)
    for (int e = 0; e < kStepsNofIn; ++e) features80[kStepsNofIn * g + e] = emb_feature(kEmbNofIn, g, e, 33);
  return MF_OK;
}

static int32_t render_entry(const mf_render_args* a, void* stream, bool prepare_only);
extern "C" int32_t mf_render_pass(const mf_render_args* a, void* stream) { return render_entry(a, stream, false); }
extern "C" int32_t mf_render_prepare(const mf_render_args* a, void* stream) { return render_entry(a, stream, true); }

static int32_t render_entry(const mf_render_args* a, void* stream, bool prepare_only) {
  if (!a || !a->nerf || !a->nerf_packed) return fail(MF_E_INVALID, "mf_render_pass: null argument");
  if (a->n_rays < 0 || a->n_samples < 1) return fail(MF_E_INVALID, "mf_render_pass: n_rays=%lld n_samples=%d",
                                                    (long long)a->n_rays, a->n_samples);
  if (a->n_rays == 0) return MF_OK;
  if (!a->rays || a->ray_stride < 9) return fail(MF_E_INVALID, "mf_render_pass: rays missing or ray_stride < 9");
  if (!a->z_vals && !a->z_steps) return fail(MF_E_INVALID, "mf_render_pass: need z_vals or z_steps");
  if (a->activation != MF_ACT_RELU && a->activation != MF_ACT_SOFTPLUS)
    return fail(MF_E_INVALID, "mf_render_pass: activation %d not supported", a->activation);
  if (a->precision < MF_PREC_F32 || a->precision > MF_PREC_BF16X3)
    return fail(MF_E_INVALID, "mf_render_pass: precision %d", a->precision);
  const int bf16 = a->precision != MF_PREC_F32;
  RenderParams p{};
  if (!nerf_layout(*a->nerf, p.nerf.L, 0)) return fail(MF_E_UNSUPPORTED, "mf_render_pass: unsupported NeRF configuration");
  if (p.nerf.L.NK != 16) return fail(MF_E_UNSUPPORTED, "mf_render_pass: only W=256 NeRF is built");
  const bool dump = a->dump_acts || a->dump_rgbsigma || a->dump_xyz || a->dump_nof_acts;
  if (dump && bf16 && a->precision != MF_PREC_BF16X3)
    return fail(MF_E_UNSUPPORTED, "mf_render_pass: the activation dump (training forward) exists in fp32 and in bf16x3");
  if (a->emb_xyz.in_channels != 3 || a->emb_xyz.n_freqs > 10)
    return fail(MF_E_UNSUPPORTED, "mf_render_pass: xyz embedding must have 3 channels and <= 10 frequencies");
  const bool sigma_only = a->flags & MF_F_SIGMA_ONLY;
  if (!sigma_only && a->nerf->extra_feat_type == MF_EXTRA_DIR &&
      (a->emb_extra.in_channels != 3 || a->emb_extra.n_freqs > 4 ||
       3 * (2 * a->emb_extra.n_freqs + 1) > a->nerf->extra_feat_dim))
    return fail(MF_E_UNSUPPORTED, "mf_render_pass: dir embedding must have 3 channels, <= 4 frequencies and fit extra_feat_dim");
  if (!sigma_only && a->nerf->extra_feat_type == MF_EXTRA_IND &&
      (a->emb_extra.in_channels != 1 || a->emb_extra.n_freqs > 2 ||
       (2 * a->emb_extra.n_freqs + 1) > a->nerf->extra_feat_dim))
    return fail(MF_E_UNSUPPORTED, "mf_render_pass: ind embedding must have 1 channel, <= 2 frequencies and fit extra_feat_dim");
  const bool moco = a->nof_bw != nullptr;
  const bool chains = a->flags & (MF_F_CHAIN_LOCAL | MF_F_CHAIN_GLOBAL);
  if (!moco && chains) return fail(MF_E_INVALID, "mf_render_pass: chain flags need NoF models");
  if ((a->flags & MF_F_CHAIN_GLOBAL) && !(a->flags & MF_F_CHAIN_LOCAL))
    return fail(MF_E_INVALID, "mf_render_pass: chain_global without chain_local (the reference raises UnboundLocalError, rendering.py:276-280)");
  if ((a->flags & MF_F_CHAIN_GLOBAL) && a->ray_stride < 10)
    return fail(MF_E_INVALID, "mf_render_pass: chain_global needs the chained image index column (ray_stride >= 10)");

  p.rays = a->rays; p.ray_stride = a->ray_stride; p.n_rays = a->n_rays; p.bg = a->background;
  p.S = a->n_samples; p.z_vals = a->z_vals; p.z_steps = a->z_steps; p.use_disp = a->use_disp;
  p.noise = a->noise; p.activation = a->activation; p.flags = a->flags;
  p.nerf.packed = static_cast<const char*>(a->nerf_packed);
  to_table(a->emb_xyz, p.emb_par[0]);
  to_table(a->emb_extra, p.emb_par[1]);
  p.extra_type = a->nerf->extra_feat_type;
  p.rgb = a->rgb; p.depth = a->depth; p.opacity = a->opacity; p.weights = a->weights; p.alphas = a->alphas;
  p.disp_local = a->disp_local; p.disp_global = a->disp_global;
  { const char* e = getenv("MF_DEBUG_FLAGS"); p.dbg = e ? atoi(e) : 0; }   // timing ablations only

  uint32_t lds = 0;
  p.nerf.res_lds = lds; lds += (uint32_t)p.nerf.L.res_bytes;
  int max_groups = p.nerf.L.max_groups;
  if (moco) {
    if (!a->nof_bw_packed) return fail(MF_E_INVALID, "mf_render_pass: nof_bw_packed missing");
    if (!nof_layout(*a->nof_bw, p.bw.L, 0)) return fail(MF_E_UNSUPPORTED, "mf_render_pass: unsupported backward NoF configuration");
    p.bw.packed = static_cast<const char*>(a->nof_bw_packed);
    p.bw.res_lds = lds; lds += (uint32_t)p.bw.L.res_bytes;
    if (p.bw.L.max_groups > max_groups) max_groups = p.bw.L.max_groups;
    if (chains) {
      if (!a->nof_fw || !a->nof_fw_packed) return fail(MF_E_INVALID, "mf_render_pass: chain flags need the forward NoF");
      if (!nof_layout(*a->nof_fw, p.fw.L, 0)) return fail(MF_E_UNSUPPORTED, "mf_render_pass: unsupported forward NoF configuration");
      p.fw.packed = static_cast<const char*>(a->nof_fw_packed);
      p.fw.res_lds = lds; lds += (uint32_t)p.fw.L.res_bytes;
      if (p.fw.L.max_groups > max_groups) max_groups = p.fw.L.max_groups;
    }
    if (a->nof_emb_xyz.in_channels != 3 || a->nof_emb_xyz.n_freqs > 5 || a->nof_emb_ind.in_channels != 1 ||
        a->nof_emb_ind.n_freqs > 16)
      return fail(MF_E_UNSUPPORTED, "mf_render_pass: NoF embeddings must be xyz(3, <=5 freqs) and ind(1, <=16 freqs)");
    to_table(a->nof_emb_xyz, p.emb_par[2]);
    to_table(a->nof_emb_ind, p.emb_par[3]);
  }
  if (bf16) return render_pass_bf16(a, static_cast<hipStream_t>(stream), prepare_only);     // validated above; own layout / launch
  if (prepare_only) return MF_OK;
  p.par_off = lds; lds += 512;
  p.ring_off = lds;
  p.buf_bytes = (uint32_t)max_groups * kGroupBytes;
  lds += 3 * p.buf_bytes;

  // rays per group: smallest G with G*S a multiple of the 128-sample tile, capped by the LDS left
  const uint32_t lds_cap = 160 * 1024;
  const int max_samples = (int)((lds_cap - lds) / 20);
  const int S = a->n_samples;
  if (S > max_samples) return fail(MF_E_UNSUPPORTED, "mf_render_pass: n_samples=%d exceeds the %d samples a workgroup can stage", S, max_samples);
  int G = 1;
  while ((G * S) % kTile != 0 && (G + 1) * S <= max_samples && G < 64) ++G;
  if ((G * S) % kTile != 0) {           // no exact fit: take as many rays as reduce the padding waste
    int best = 1; double best_eff = 0;
    for (int g = 1; g * S <= max_samples && g <= 64; ++g) {
      const int tiles = (g * S + kTile - 1) / kTile;
      const double eff = (double)(g * S) / (tiles * kTile);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = g; }
    }
    G = best;
  }
  // Several such ray sets per group (up to 8): the composite phase between two groups keeps at most G of the 8 waves
  // busy and costs two workgroup barriers (~4 k cycles per 128-sample tile at G = 2: tools/timeline.py), so it should
  // come once per several tiles -- as long as the CUs' shares stay what they were (same makespan in rays).
  {
    const long long cus = device_cus();
    auto makespan = [&](long long g) { const long long groups = (a->n_rays + g - 1) / g; return (groups + cus - 1) / cus * g; };
    const long long base = makespan(G);
    int best = 1;
    for (int c = 2; c <= 8; ++c)
      if ((long long)G * c * S <= max_samples && (long long)G * c <= 64 && makespan((long long)G * c) <= base) best = c;
    G *= best;
  }
  p.G = G;
  p.n_groups = (a->n_rays + G - 1) / G;
  p.sbuf_off = lds; lds += (uint32_t)(G * S) * 16;
  p.zbuf_off = lds; lds += (uint32_t)(G * S) * 4;
  lds = (lds + 15u) & ~15u;

  const int grid = (int)(p.n_groups < device_cus() ? p.n_groups : device_cus());
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->dump_acts && a->dump_stride < (int64_t)p.nerf.L.n_trunk * p.nerf.L.W + p.nerf.L.W / 2)
    return fail(MF_E_INVALID, "mf_render_pass: dump_stride %lld too small", (long long)a->dump_stride);
  p.dump_acts = a->dump_acts; p.dump_stride = a->dump_stride; p.dump_rgbsigma = a->dump_rgbsigma; p.dump_xyz = a->dump_xyz;
  if (a->dump_mask) {
    if (!a->dump_acts || a->dump_mask_stride < (int64_t)(p.nerf.L.n_trunk + 1) * 8)
      return fail(MF_E_INVALID, "mf_render_pass: dump_mask needs dump_acts and dump_mask_stride >= 8 (D + 2) words");
    p.dump_mask = a->dump_mask; p.dump_mask_stride = a->dump_mask_stride;
  }
  if (a->dump_nof_acts) {
    if (!moco || !a->dump_nof_emb || !a->dump_nof_out) return fail(MF_E_INVALID, "mf_render_pass: dump_nof_acts needs NoF models, dump_nof_emb and dump_nof_out");
    if (a->dump_nof_stride < (int64_t)p.bw.L.n_trunk * p.bw.L.W + 16 || (a->dump_nof_stride & 3))
      return fail(MF_E_INVALID, "mf_render_pass: dump_nof_stride %lld invalid", (long long)a->dump_nof_stride);
    if (a->nof_fw && (p.fw.L.n_trunk != p.bw.L.n_trunk || p.fw.L.W != p.bw.L.W))
      return fail(MF_E_UNSUPPORTED, "mf_render_pass: NoF dumps need bw and fw of the same depth and width");
    if (a->precision != MF_PREC_F32) return fail(MF_E_UNSUPPORTED, "mf_render_pass: NoF dumps are fp32 only");
  }
  p.dump_nof_acts = a->dump_nof_acts; p.dump_nof_stride = a->dump_nof_stride; p.dump_nof_emb = a->dump_nof_emb; p.dump_nof_out = a->dump_nof_out;
  if (a->dump_nof_acts) {
    const int nsteps = 1 + ((a->flags & MF_F_CHAIN_LOCAL) ? 1 : 0) + ((a->flags & MF_F_CHAIN_GLOBAL) ? 3 : 0);
    uint32_t seen = 0;
    for (int k = 0; k < nsteps; ++k) {
      const int pl = a->dump_nof_plane[k];
      if (pl < 0 || pl >= nsteps || ((seen >> pl) & 1u)) return fail(MF_E_INVALID, "mf_render_pass: dump_nof_plane must be a permutation of 0..%d", nsteps - 1);
      seen |= 1u << pl;
      p.nof_plane_pack |= (uint32_t)pl << (3 * k);
    }
  }
  void (*kern)(RenderParams) =
      dump ? (moco ? render_kernel<true, true> : render_kernel<false, true>)
           : (moco ? render_kernel<true, false> : render_kernel<false, false>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_render_pass: cannot reserve %u bytes of LDS", lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, p);
  return check_launch("mf_render_pass");
}
