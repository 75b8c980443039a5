// mf_render_bf16.hip -- the fused rendering pass of render_rays (models/rendering.py:195-375) with bf16 hidden
// GEMMs (MF_PREC_BF16, BASELINE configs C3-C5): same launch structure as mf_render.hip -- ray -> z -> xyz ->
// [NoF chains] -> positional encoding -> NeRF -> per-ray composite, one persistent launch, nothing per-sample
// written to HBM -- on the 32x32x16 bf16 core of mf_bf16.hpp: a wave owns 32 samples, a workgroup tile is 256
// samples, a weight panel is one 32-row tile of a layer.
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#include "mf_bf16.hpp"
#include "mf_bf16_2b.hpp"
#include "mf_host.hpp"
#include "mf_layout.hpp"

namespace mf {
namespace bf {

typedef const __attribute__((address_space(4))) char* typedef_kptr;

struct Params {
  const float* rays; long long ray_stride; long long n_rays;
  const float* bg;
  int S;
  const float* z_vals; const float* z_steps; int use_disp;
  const float* noise;
  int activation, flags;
  Net nerf, bw, fw;
  int extra_type;
  float emb_par[4][32];            // [nerf xyz, nerf extra, nof xyz, nof ind] x (freq[16], weight[16]) -> LDS
  float *rgb, *depth, *opacity, *weights, *alphas, *disp_local, *disp_global;
  int G;
  long long n_groups;
  uint32_t par_off, ring_off, buf_bytes, sbuf_off, zbuf_off;
  int pow2;                        // bit t: table t's frequencies are exactly 2^k (the logscale default)
  const float* raybias;            // MOCO: (n_rays, rb_combos, rb_layers, 128) per-ray NoF biases (nof_raybias_kernel)
  int rb_combos, rb_layers;
  uint32_t rb_off, rb_buf_bytes;   // LDS: two buffers of the current / next chain step's rows of the tile's rays
  float* dump_acts; long long dump_stride; float* dump_rgbsigma; float* dump_xyz;   // training forward (X3)
  unsigned* dump_mask; long long dump_mask_stride;                                  // ReLU bit rows of the NeRF's dump (optional)
  float* dump_nof_acts; long long dump_nof_stride; float* dump_nof_out;             // ... under NoF: per chain step (round 5)
  uint32_t nof_plane_pack;         // plane of step k = (pack >> 3k) & 7
};

// ---- per-ray bias of the NoF's embedded-input layers (models/rendering.py:73-75 + models/nof.py:69-73) ----
// The NoF's input is [emb(xyz) 33 | emb(ind) 33] and the image index is one number per ray, so in every layer l that
// consumes the embedded input  b_l + W_l[:, 33:66] emb(ind)  is constant along the ray.  nof_raybias_kernel evaluates it
// once per (entry, combination) in exact fp32 -- emb(ind) with OCML sincosf on the fp32-rounded argument 2^k ind, as
// embedding.py:45 does, 33 FMAs per row -- and the render kernel starts those layers' accumulators from it: the image
// index never enters the matrix pipe (2 of the 5 split k-steps of layer 0 and of the skip layer, 48 of 232 MFMAs per
// evaluation, and the 18 sin / cos of 2^15-sized arguments per lane are gone).
// Combination = (network, index column): render passes use bw(i), fw(i), fw(j), bw(j) (rendering.py:270-282).
struct RayBiasParams {
  const float* ind;                // entry e reads ind[e * ind_stride + col[c]]; null -> ind_scalar
  long long ind_stride;
  float ind_scalar;
  long long n_entries;
  int n_combos;
  const float* bias[4];            // per combination: the network's [layer][128] trunk biases (its resident block)
  const float* wind[4];            //                  its [embedded layer][kNofIndCols][128] index columns
  uint32_t emb_mask[4];
  int col[4];
  int n_layers;                    // embedded layers per network
  float freq[16], weight[16];      // the image-index embedding
  float* out;                      // (n_entries, n_combos, n_layers, 128)
};

constexpr int kRbEntries = 8;       // entries (rays) per workgroup of nof_raybias_kernel
constexpr int kFastBlocksDefault = 1; // column blocks per wave of the fast mode's render kernels (see render_pass_bf16)

// grid (ceil(n_entries / 8), n_combos), 256 threads.  Thread (embedded layer, row) first requests its 33 index-column
// weights + bias (coalesced: the packed block is [layer][column][row]); while they travel, threads (entry e, frequency k)
// do one exact sincosf each into the LDS copy of emb(ind) of the block's 8 entries; then every thread contracts its
// weights with the 8 embeddings (LDS broadcasts) and the 128-float rows go out coalesced.  (A latency chain of two
// dependent memory round trips and ~300 FMAs: short blocks, many of them.)
__global__ __launch_bounds__(256) void nof_raybias_kernel(const RayBiasParams p) {
  __shared__ float e[kRbEntries][kNofIndCols];
  const int c = blockIdx.y;
  const long long entry0 = (long long)blockIdx.x * kRbEntries;
  const int tid = threadIdx.x;
  // the by-value argument is indexed with constants only (a runtime index would put a private copy in scratch)
  const float* s_bias = c == 0 ? p.bias[0] : (c == 1 ? p.bias[1] : (c == 2 ? p.bias[2] : p.bias[3]));
  const float* s_wind = c == 0 ? p.wind[0] : (c == 1 ? p.wind[1] : (c == 2 ? p.wind[2] : p.wind[3]));
  const uint32_t s_mask = c == 0 ? p.emb_mask[0] : (c == 1 ? p.emb_mask[1] : (c == 2 ? p.emb_mask[2] : p.emb_mask[3]));
  const int col = c == 0 ? p.col[0] : (c == 1 ? p.col[1] : (c == 2 ? p.col[2] : p.col[3]));
  const int n_out = p.n_layers * 128;
  const int o = tid < n_out ? tid : n_out - 1;              // (layers beyond the first two: the loop below)
  const int el0 = o >> 7, row0 = o & 127;
  auto layer_of = [&](int el) {                             // the el-th set bit of emb_mask
    int layer = 0, seen = 0;
    for (uint32_t m = s_mask; m; m >>= 1, ++layer)
      if (m & 1u) { if (seen == el) break; ++seen; }
    return layer;
  };
  float w[33];
  float bias = s_bias[layer_of(el0) * 128 + row0];
#pragma unroll
  for (int k = 0; k < 33; ++k) w[k] = s_wind[(size_t)(el0 * kNofIndCols + k) * 128 + row0];
  if (tid < kRbEntries * 16) {
    const int le = tid >> 4, k = tid & 15;                  // column order of embedding.py:42-46 with C = 1
    const long long entry = entry0 + le < p.n_entries ? entry0 + le : p.n_entries - 1;
    const float ind = p.ind ? p.ind[entry * p.ind_stride + col] : p.ind_scalar;
    const typedef_kptr ka = (typedef_kptr)__builtin_amdgcn_kernarg_segment_ptr();
    const float fr = ((const __attribute__((address_space(4))) float*)(ka + offsetof(RayBiasParams, freq)))[k];
    const float wk = ((const __attribute__((address_space(4))) float*)(ka + offsetof(RayBiasParams, weight)))[k];
    float sn, cs;
    sincosf(fr * ind, &sn, &cs);
    e[le][1 + 2 * k] = wk * sn;
    e[le][2 + 2 * k] = wk * cs;
    if (k == 0) e[le][0] = ind;
    if (k < 3) e[le][33 + k] = 0.f;
  }
  __syncthreads();
  for (int ob = 0; ob < n_out; ob += 256) {
    const int oo = ob + tid;
    if (ob > 0 && oo < n_out) {                             // (more than two embedded layers: reload this thread's row)
      bias = s_bias[layer_of(oo >> 7) * 128 + (oo & 127)];
#pragma unroll
      for (int k = 0; k < 33; ++k) w[k] = s_wind[(size_t)((oo >> 7) * kNofIndCols + k) * 128 + (oo & 127)];
    }
    if (oo >= n_out) break;
    // (no loop vectorisation: it pairs two entries into v_pk_fma_f32, and this unit is kept free of packed-fp32 ops,
    //  csrc/Makefile)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int le = 0; le < kRbEntries && entry0 + le < p.n_entries; ++le) {
      float acc = bias;
#pragma unroll
      for (int k = 0; k < 33; ++k) acc = __builtin_fmaf(w[k], e[le][k], acc);
      p.out[((size_t)((entry0 + le) * p.n_combos + c) * p.n_layers + (oo >> 7)) * 128 + (oo & 127)] = acc;
    }
  }
}

// LDS-DMA of the per-ray bias rows one chain step of one tile needs: rays [ray_first, ray_first + n) of combination
// `combo`, entry = [embedded layer][128] floats, to `dst` (ray-major).  Pieces of 1 KiB round-robin over the waves; every
// panel barrier behind the issue publishes them (Stream::sync waits vmcnt(0) first), so the issue sits at least one panel
// in front of the first read and behind the last read of the buffer's previous content (see the call sites).
template <int NW = kWaves>
MF_D void stage_raybias(const float* table, int combos, int layers, long long ray_first, int n, int combo, uint32_t dst,
                        const Lane& id) {
  const uint32_t entry = (uint32_t)layers * 512u, chunk = (uint32_t)n * entry;
  const char* base = reinterpret_cast<const char*>(table) + ((size_t)ray_first * combos + combo) * entry;
  for (uint32_t q = id.wave; q * 1024u < chunk; q += NW) {
    uint32_t b = q * 1024u + id.lane * 16u;
    b = b < chunk ? b : chunk - 16u;                       // (tail lanes re-fetch the last 16 bytes into the padding)
    const uint32_t r = b / entry;
    blds16(base, r * (uint32_t)combos * entry + (b - r * entry), 0, dst + q * 1024u);
  }
}
template <int NW, class P>
MF_D void stage_raybias(const P& p, long long ray_first, int n, int combo, uint32_t dst, const Lane& id) {
  stage_raybias<NW>(p.raybias, p.rb_combos, p.rb_layers, ray_first, n, combo, dst, id);
}
// rays [first, first + n) touched by tile `tile` (TILE samples) of a group of `nr` rays
// rays of group `g` (G per group, the last one shorter): 32-bit scalar arithmetic (a 64-bit `<` is a VALU compare here)
MF_D int group_rays(long long n_rays, long long g, int G) {
  const long long rem = n_rays - g * G;
  return (rem >> 31) != 0 ? G : ((int)rem < G ? (int)rem : G);
}
template <int TILE>
MF_D void tile_rays(int tile, int nr, int S, int& first, int& n) {
  const int s0 = tile * TILE, s1 = (s0 + TILE < nr * S ? s0 + TILE : nr * S) - 1;
  first = s0 / S;
  n = s1 / S - first + 1;
}

// ---- composite (rendering.py:157-192) of a group's rays out of the LDS sample buffers: one wave per ray, lanes over samples
template <int NW>
MF_D void composite_group(const Params& p, const Lane& id, long long ray0, int nr, int S, bool sigma_only, const float4* sbuf, const float* zbuf) {
    for (int rr = id.wave; rr < nr; rr += NW) {
    const long long ray = ray0 + rr;
    const float* rp = p.rays + ray * p.ray_stride;
    const float dnorm = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);  // rendering.py:164
    float carry_t = 1.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_w = 0.f;
    for (int base = 0; base < S; base += 64) {
      // (the lane index is made opaque here: hipcc otherwise hoists `plane + 4 lane` of every output plane out of the
      //  whole group loop as 64-bit per-lane addresses and spills them across the MFMA section)
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
      const float w = alpha * (carry_t * excl);                              // :178-179
      carry_t = carry_t * wave_last(incl);
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
}

// X3: MF_PREC_BF16X3 (mf_bf16.hpp: every matrix product as a three-product split, heads on fp32 accumulators): 4 waves,
// one per SIMD, 128-sample tiles; else 8 waves, two per SIMD, 256-sample tiles
// DUMP (X3 && !MOCO): the training forward -- every sample's layer activations, (rgb, sigma) and input point are stored
// for the backward (mf_render_args.dump_*), as mf_render.hip's render_kernel<*, true> does in fp32.
template <bool MOCO, bool X3 = false, bool DUMP = false>
__global__ __launch_bounds__(X3 ? 256 : kThreads, X3 ? 1 : 2) void render_kernel_bf16(const Params p) {
  static_assert(!DUMP || X3, "the dump variants exist for the three-product passes");
  constexpr int NW = X3 ? 4 : kWaves;
  constexpr int TILE = NW * kWaveSamples;
  const Lane id;
#ifdef MF_TIMELINE
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  load_resident<NW>(p.nerf, id);
  if (MOCO) {
    load_resident<NW>(p.bw, id);
    if (p.flags & (MF_F_CHAIN_LOCAL | MF_F_CHAIN_GLOBAL)) load_resident<NW>(p.fw, id);
  }
  if (threadIdx.x < 128) {
    // the embedding tables go kernarg -> LDS through the kernarg segment pointer: indexing the by-value struct with
    // a runtime index would make hipcc keep a private (scratch) copy of all of `p`
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Params, emb_par);
    *(float*)(smem + p.par_off + threadIdx.x * 4) = ((const __attribute__((address_space(4))) float*)ka)[threadIdx.x];
  }
  const uint32_t par_nerf_xyz = p.par_off, par_nerf_ext = p.par_off + 128, par_nof_xyz = p.par_off + 256;
  StreamT<NW> st;
  st.tl.start(p.alphas, id);
  typename std::conditional<X3, CarryX, Carry>::type carry;
  constexpr int NP2 = X3 ? 1 : kNofTpp0;     // the fast mode's NoF layers stream several tiles per panel (mf_bf16.hpp),
  constexpr int NPH = X3 ? 0 : kNofTppH, NPS = X3 ? 0 : kNofTppS;
  constexpr int NF0 = X3 ? 1 : kNerfTpp0;    // its NeRF layer 0 four
  constexpr int TN = X3 ? kNofTermsX3 : 2;   // terms of the NoF's operands (x3: kNofTermsX3 terms, IEEE halves with kNofHalfX3: mf_core.hpp)
  const Next prog_first = MOCO ? first_of<8, kKsNofXyz, true, NP2, TN, NPH, NPS>(p.bw) : first_of<16, kKsNerfXyz, X3, NF0>(p.nerf);
  int seq = 0;                       // NoF evaluations done by this workgroup: evaluation number `seq` reads buffer seq & 1
  if (MOCO) {                        // rows of the first evaluation (first group, tile 0, bw at index i)
    const long long g0 = blockIdx.x;
    const int nr0 = group_rays(p.n_rays, g0, p.G);
    int f0, n0;
    tile_rays<TILE>(0, nr0, p.S, f0, n0);
    stage_raybias<NW>(p, g0 * p.G + f0, n0, 0, p.rb_off, id);
  }
  if (MOCO) start_program<8, kKsNofXyz, true, NP2, TN, NPH, NPS>(p.bw, st, carry, p.ring_off, p.buf_bytes, id);   // (its wait + barrier also publish the resident blocks / tables)
  else start_program<16, kKsNerfXyz, X3, NF0>(p.nerf, st, carry, p.ring_off, p.buf_bytes, id);

  const int S = p.S;
  const bool sigma_only = p.flags & MF_F_SIGMA_ONLY;
  float4* sbuf = reinterpret_cast<float4*>(smem + p.sbuf_off);
  float* zbuf = reinterpret_cast<float*>(smem + p.zbuf_off);

  for (long long group = blockIdx.x; group < p.n_groups; group += gridDim.x) {
    const long long ray0 = group * p.G;
    // (the rays left as a 32-bit scalar minimum: the 64-bit `rem < G` has no scalar compare on this target and became a
    //  v_cmp_lt_i64 against a VGPR copy of G that hipcc kept -- spilled -- across the whole group loop)
    const int nr = group_rays(p.n_rays, group, p.G);
    const int nsamp = nr * S;
    const int ntiles = (nsamp + TILE - 1) / TILE;

    for (int tile = 0; tile < ntiles; ++tile) {
      st.tl.stamp(1, id);
      // (the lane's sample slot re-derived per tile from the hardware lane count + the scalar wave index: as a loop invariant
      //  it -- and then the thread index it was rebuilt from -- was one more register held, i.e. spilled, through the MFMA section)
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      const int srel = tile * TILE + id.wave * kWaveSamples + (ln & 31);
      const bool valid = srel < nsamp;
      const int sl = valid ? srel : nsamp - 1;
      const int rr = sl / S;
      const int si = sl - rr * S;
      const long long ray = ray0 + rr;
      const float* rp = p.rays + ray * p.ray_stride;
      const float o[3] = {rp[0], rp[1], rp[2]};
      const float d[3] = {rp[3], rp[4], rp[5]};
      float z;
      if (p.z_vals) {
        z = p.z_vals[ray * S + si];
      } else {
        const float nearv = rp[6], farv = rp[7], t = p.z_steps[si];
        if (!p.use_disp) z = nearv * (1.f - t) + farv * t;                    // rendering.py:247
        else z = 1.f / (1.f / nearv * (1.f - t) + 1.f / farv * t);            // rendering.py:249
      }
      float x[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = o[c] + d[c] * z;                      // rendering.py:262-263

#ifdef MF_TIMELINE
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
#endif
      st.tl.stamp(2, id);
      float xin[3] = {x[0], x[1], x[2]};      // what the canonical NeRF sees
      if (MOCO) {
        // chain program (rendering.py:270-282): step 0 bw(x,i) -> canon; local: fw(canon,i) -> recon;
        // global: fw(canon,j) -> a; bw(a,j) -> b; fw(b,i) -> chained recon.
        const bool loc = p.flags & MF_F_CHAIN_LOCAL, glob = p.flags & MF_F_CHAIN_GLOBAL;
        const int nsteps = 1 + (loc ? 1 : 0) + (glob ? 3 : 0);
        float canon[3] = {0.f, 0.f, 0.f}, cur[3] = {x[0], x[1], x[2]};
        float dl = 0.f, dg = 0.f;
        for (int step = 0; step < nsteps; ++step) {
          // role of this step: 0 = bw_i, 1 = local fw_i, 2 = fw_j, 3 = bw_j, 4 = final fw_i
          const int role = step;
          const bool use_fw = (role == 1 || role == 2 || role == 4);
          const Net net = use_fw ? p.fw : p.bw;
          // per-ray bias of this (network, image index): rows bw(i), fw(i), fw(j), bw(j) of the table (the final fw(i) is
          // row 1), staged in LDS one evaluation ahead
          int tf, tn;
          tile_rays<TILE>(tile, nr, S, tf, tn);
          LdsRayBias rb{p.rb_off + (uint32_t)(seq & 1) * p.rb_buf_bytes + (uint32_t)(rr - tf) * (uint32_t)(p.rb_layers * 512)};
          ++seq;
          if (role == 1 || role == 2) { cur[0] = canon[0]; cur[1] = canon[1]; cur[2] = canon[2]; }
          const bool last = step == nsteps - 1;
          const bool next_fw = (role + 1 == 1 || role + 1 == 2 || role + 1 == 4);
          const Next follow = last ? first_of<16, kKsNerfXyz, X3, NF0>(p.nerf) : first_of<8, kKsNofXyz, true, NP2, TN, NPH, NPS>(next_fw ? p.fw : p.bw);
          u32x4 nhi[kKsNofXyz], nmid[TN == 3 ? kKsNofXyz : 1], nlo[kKsNofXyz];
          float out[3];
          if constexpr (X3) nof_embed_t<false, TN, kNofHalfX3>(nhi, reinterpret_cast<u32x4(&)[kKsNofXyz]>(nmid), nlo, cur, par_nof_xyz, id.h, p.pow2 & 4);
          else nof_embed<true>(nhi, nlo, cur, par_nof_xyz, id.h, p.pow2 & 4);
          auto stage_next = [&] {
            if (!last) stage_raybias<NW>(p, ray0 + tf, tn, role + 1 == 4 ? 1 : role + 1, p.rb_off + (uint32_t)(seq & 1) * p.rb_buf_bytes, id);
          };
          if constexpr (X3 && DUMP) {
            // training forward: what autograd.NofPointsDumped's backward reads, per chain step (step-major planes; the embedded
            // input is made from the points by mf_nof_embed_rows, the rows carry no ReLU bit words)
            const long long nof_idx = (long long)((p.nof_plane_pack >> (3 * step)) & 7u) * p.n_rays * S + (ray * S + si);
            const bool don = valid && p.dump_nof_acts != nullptr;
            // (a row with room behind T carries the layers' ReLU bit words, 4 per layer: what mf_nof_backward3 reads instead of the activations)
            float* nrow = p.dump_nof_acts + nof_idx * p.dump_nof_stride;
            const bool nmasks = p.dump_nof_stride >= (long long)net.D * 128 + 16 + 4 * net.D;
            RowDump nd{nrow + 4 * id.h, don, __ballot(don) != 0ull, reinterpret_cast<unsigned*>(nrow + net.D * 128 + 16), nmasks};
            dump_rows(nd, p.dump_nof_acts, nof_idx, p.dump_nof_stride, don, id.h);
            nof_eval_x3<TN, kNofHalfX3>(net, nhi, reinterpret_cast<const u32x4(&)[kKsNofXyz]>(nmid), nlo, cur, st, carry, id, follow, out, rb, stage_next, nd);
            if (don && id.h == 0) { float* q = p.dump_nof_out + nof_idx * 3; q[0] = out[0]; q[1] = out[1]; q[2] = out[2]; }
          } else if constexpr (X3) {
            nof_eval_x3<TN, kNofHalfX3>(net, nhi, reinterpret_cast<const u32x4(&)[kKsNofXyz]>(nmid), nlo, cur, st, carry, id, follow, out, rb, stage_next);
          } else {
            nof_eval(net, nhi, nlo, cur, st, carry, id, follow, out, rb, nullptr, stage_next);
          }
          if (role == 0) { canon[0] = out[0]; canon[1] = out[1]; canon[2] = out[2]; }
          if (role == 1) dl = (fabsf(x[0] - out[0]) + fabsf(x[1] - out[1]) + fabsf(x[2] - out[2])) / 3.f;
          if (role == 4) dg = (fabsf(x[0] - out[0]) + fabsf(x[1] - out[1]) + fabsf(x[2] - out[2])) / 3.f;
          cur[0] = out[0]; cur[1] = out[1]; cur[2] = out[2];
        }
        xin[0] = canon[0]; xin[1] = canon[1]; xin[2] = canon[2];
        if (valid && id.h == 0) {
          if (loc && p.disp_local) p.disp_local[ray * S + si] = dl;
          if (glob && p.disp_global) p.disp_global[ray * S + si] = dg;
        }
      }

      if (MOCO) {
        // rows of the NEXT tile's first evaluation (next tile of this group, or tile 0 of this workgroup's next group) into
        // the buffer the evaluation before the last one read: every wave is past that one (it has arrived at the last
        // evaluation's barriers), and the whole NeRF lies between this issue and the first read
        const bool more = tile + 1 < ntiles;
        const long long ng = more ? group : group + gridDim.x;
        if (ng < p.n_groups) {
          const int nnr = group_rays(p.n_rays, ng, p.G);
          int nf, nn;
          tile_rays<TILE>(more ? tile + 1 : 0, nnr, S, nf, nn);
          stage_raybias<NW>(p, ng * p.G + nf, nn, 0, p.rb_off + (uint32_t)(seq & 1) * p.rb_buf_bytes, id);
        }
      }
      st.tl.stamp(3, id);
      // The sample's depth goes to the group's LDS buffer NOW (nobody reads the slot before the group's barrier; one register less
      // across the NeRF) -- and carries the poison: NaN / inf in the point the NeRF sees (a NaN ray; a NoF whose weights left the half
      // range: mf_pack.hip's poison slot) must come OUT as NaN, but the integer-max ReLUs of these kernels turn a NaN with the sign
      // bit set into 0 and a poisoned network would render finite garbage.  A NaN depth makes the composite's delta, alpha and
      // every sum of the ray NaN.  (x * 0 is not folded: no fast-math in this unit.)
      if (valid && id.h == 0) zbuf[srel] = z + (xin[0] + xin[1] + xin[2]) * 0.f;
      float sigma, rgb[3] = {0.f, 0.f, 0.f};
      auto extra_slots = [&](float (&ext)[8 * kKsExtraMax]) {
#pragma unroll
        for (int e = 0; e < 8 * kKsExtraMax; ++e) ext[e] = 0.f;
        if (p.extra_type == MF_EXTRA_DIR) {
          const float dd[3] = {rp[3], rp[4], rp[5]};
          emb_eval<3, 4, true>(ext, dd, par_nerf_ext, id.h, p.pow2 & 2);                         // rendering.py:138-142
        } else if (p.extra_type == MF_EXTRA_IND) {
          const float iv[1] = {rp[8]};
          emb_eval<1, 2, true>(ext, iv, par_nerf_ext, id.h, p.pow2 & 2);                         // rendering.py:133-137
        }
      };
      if constexpr (X3) {
        u32x4 xh[kKsNerfXyz], xl[kKsNerfXyz];
        {
          float embx[B2Xyz10::SLOTS];
          jitter();
          // exact seeds + doubling chains (<= 2e-6): the transcendental unit's 2e-4 rad at 512 x would sit above the
          // 2^-16 of the split operands
          emb_eval<3, 10, false>(embx, xin, par_nerf_xyz, id.h, p.pow2 & 1);
          split_operands<kKsNerfXyz>(embx, B2Xyz10::SLOTS, xh, xl);
        }
        auto make_extra = [&](u32x4 (&eh)[kKsExtraMax], u32x4 (&el)[kKsExtraMax]) {
          float ext[8 * kKsExtraMax];
          extra_slots(ext);
          split_operands<kKsExtraMax>(ext, 8 * kKsExtraMax, eh, el);
        };
        st.tl.stamp(4, id);
        if constexpr (DUMP) {
          const long long row = ray * S + si;
          const bool don = valid && p.dump_acts != nullptr;
          RowDump dump{p.dump_acts + row * p.dump_stride + 4 * id.h, don, __ballot(don) != 0ull,
                       p.dump_mask ? p.dump_mask + row * p.dump_mask_stride : nullptr, p.dump_mask != nullptr};
          dump_rows(dump, p.dump_acts, row, p.dump_stride, don, id.h);
          nerf_eval_x3(p.nerf, xh, xl, make_extra, sigma_only, st, carry, id, prog_first, sigma, rgb, dump);
          if (valid && id.h == 0) {
            if (p.dump_rgbsigma) *reinterpret_cast<float4*>(p.dump_rgbsigma + row * 4) = make_float4(rgb[0], rgb[1], rgb[2], sigma);
            if (p.dump_xyz) { float* q = p.dump_xyz + row * 3; q[0] = xin[0]; q[1] = xin[1]; q[2] = xin[2]; }
          }
        } else {
          nerf_eval_x3(p.nerf, xh, xl, make_extra, sigma_only, st, carry, id, prog_first, sigma, rgb);
        }
      } else {
        u32x4 xe[kKsNerfXyz];
        {
          float embx[B2Xyz10::SLOTS];
          jitter();
          emb_eval<3, 10, true>(embx, xin, par_nerf_xyz, id.h, p.pow2 & 1);
          pack_operands<kKsNerfXyz>(embx, B2Xyz10::SLOTS, xe);
        }
        auto make_extra = [&](u32x4 (&eo)[kKsExtraMax]) {
          float ext[8 * kKsExtraMax];
          extra_slots(ext);
          pack_operands<kKsExtraMax>(ext, 8 * kKsExtraMax, eo);
        };
        st.tl.stamp(4, id);
        nerf_eval(p.nerf, xe, make_extra, sigma_only, st, carry, id, prog_first, sigma, rgb);
      }
      if (valid && id.h == 0) sbuf[srel] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
      st.tl.stamp(5, id);
    }
    __syncthreads();
    st.tl.stamp(6, id);

    composite_group<NW>(p, id, ray0, nr, S, sigma_only, sbuf, zbuf);
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

// The fast mode with two column blocks per wave (mf_bf16_2b.hpp): 4 waves, one per SIMD, 64 samples each -- 256-sample tiles as
// render_kernel_bf16<*, false>, the same panel program, the same arithmetic (bit-identical outputs); every weight fragment read
// from LDS feeds two MFMAs.  The per-sample code of a tile runs once per block b: sample slot wave * 64 + 32 b + (lane & 31).
template <bool MOCO>
__global__ __launch_bounds__(256, 1) void render_kernel_bf16_2b(const Params p) {
  constexpr int NW = kWaves2;
  constexpr int TILE = NW * kBlockSamples;
  const Lane id;
  load_resident<NW>(p.nerf, id);
  if (MOCO) {
    load_resident<NW>(p.bw, id);
    if (p.flags & (MF_F_CHAIN_LOCAL | MF_F_CHAIN_GLOBAL)) load_resident<NW>(p.fw, id);
  }
  if (threadIdx.x < 128) {
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Params, emb_par);
    *(float*)(smem + p.par_off + threadIdx.x * 4) = ((const __attribute__((address_space(4))) float*)ka)[threadIdx.x];
  }
  const uint32_t par_nerf_xyz = p.par_off, par_nerf_ext = p.par_off + 128, par_nof_xyz = p.par_off + 256;
  Stream2 st;
  st.tl.start(p.alphas, id);
  Carry carry;
  const Next prog_first = MOCO ? first_of<8, kKsNofXyz, true, kNofTpp0, 2, kNofTppH, kNofTppS>(p.bw) : first_of<16, kKsNerfXyz, false, kNerfTpp0>(p.nerf);
  int seq = 0;
  if (MOCO) {
    const long long g0 = blockIdx.x;
    const int nr0 = group_rays(p.n_rays, g0, p.G);
    int f0, n0;
    tile_rays<TILE>(0, nr0, p.S, f0, n0);
    stage_raybias<NW>(p, g0 * p.G + f0, n0, 0, p.rb_off, id);
  }
  if (MOCO) start_program<8, kKsNofXyz, true, kNofTpp0, 2, kNofTppH, kNofTppS>(p.bw, st, carry, p.ring_off, p.buf_bytes, id);
  else start_program<16, kKsNerfXyz, false, kNerfTpp0>(p.nerf, st, carry, p.ring_off, p.buf_bytes, id);

  const int S = p.S;
  const bool sigma_only = p.flags & MF_F_SIGMA_ONLY;
  float4* sbuf = reinterpret_cast<float4*>(smem + p.sbuf_off);
  float* zbuf = reinterpret_cast<float*>(smem + p.zbuf_off);

  for (long long group = blockIdx.x; group < p.n_groups; group += gridDim.x) {
    const long long ray0 = group * p.G;
    const int nr = group_rays(p.n_rays, group, p.G);
    const int nsamp = nr * S;
    const int ntiles = (nsamp + TILE - 1) / TILE;

    for (int tile = 0; tile < ntiles; ++tile) {
      st.tl.stamp(1, id);
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      int srel[2], rr[2], si[2];
      bool valid[2];
      const float* rp[2];
      float z[2], x[2][3], xin[2][3];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        srel[b] = tile * TILE + id.wave * kBlockSamples + b * kWaveSamples + (ln & 31);
        valid[b] = srel[b] < nsamp;
        const int sl = valid[b] ? srel[b] : nsamp - 1;
        rr[b] = sl / S;
        si[b] = sl - rr[b] * S;
        const long long ray = ray0 + rr[b];
        rp[b] = p.rays + ray * p.ray_stride;
        const float o[3] = {rp[b][0], rp[b][1], rp[b][2]};
        const float d[3] = {rp[b][3], rp[b][4], rp[b][5]};
        if (p.z_vals) {
          z[b] = p.z_vals[ray * S + si[b]];
        } else {
          const float nearv = rp[b][6], farv = rp[b][7], t = p.z_steps[si[b]];
          if (!p.use_disp) z[b] = nearv * (1.f - t) + farv * t;                    // rendering.py:247
          else z[b] = 1.f / (1.f / nearv * (1.f - t) + 1.f / farv * t);            // rendering.py:249
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[b][c] = o[c] + d[c] * z[b]; xin[b][c] = x[b][c]; }      // rendering.py:262-263
      }
      st.tl.stamp(2, id);
      if (MOCO) {
        // chain program (rendering.py:270-282), as render_kernel_bf16
        const bool loc = p.flags & MF_F_CHAIN_LOCAL, glob = p.flags & MF_F_CHAIN_GLOBAL;
        const int nsteps = 1 + (loc ? 1 : 0) + (glob ? 3 : 0);
        float canon[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, cur[2][3];
        float dl[2] = {0.f, 0.f}, dg[2] = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 3; ++c) cur[b][c] = x[b][c];
        for (int step = 0; step < nsteps; ++step) {
          const int role = step;
          const bool use_fw = (role == 1 || role == 2 || role == 4);
          const Net net = use_fw ? p.fw : p.bw;
          int tf, tn;
          tile_rays<TILE>(tile, nr, S, tf, tn);
          const uint32_t rbase = p.rb_off + (uint32_t)(seq & 1) * p.rb_buf_bytes;
          const LdsRayBias rb[2] = {LdsRayBias{rbase + (uint32_t)(rr[0] - tf) * (uint32_t)(p.rb_layers * 512)},
                                    LdsRayBias{rbase + (uint32_t)(rr[1] - tf) * (uint32_t)(p.rb_layers * 512)}};
          ++seq;
          if (role == 1 || role == 2) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int c = 0; c < 3; ++c) cur[b][c] = canon[b][c];
          }
          const bool last = step == nsteps - 1;
          const bool next_fw = (role + 1 == 1 || role + 1 == 2 || role + 1 == 4);
          const Next follow = last ? first_of<16, kKsNerfXyz, false, kNerfTpp0>(p.nerf)
                                   : first_of<8, kKsNofXyz, true, kNofTpp0, 2, kNofTppH, kNofTppS>(next_fw ? p.fw : p.bw);
          u32x4 nhi[2][kKsNofXyz], nlo[2][kKsNofXyz];
          float out[2][3];
          nof_embed<true>(nhi[0], nlo[0], cur[0], par_nof_xyz, id.h, p.pow2 & 4);
          nof_embed<true>(nhi[1], nlo[1], cur[1], par_nof_xyz, id.h, p.pow2 & 4);
          auto stage_next = [&] {
            if (!last) stage_raybias<NW>(p, ray0 + tf, tn, role + 1 == 4 ? 1 : role + 1, p.rb_off + (uint32_t)(seq & 1) * p.rb_buf_bytes, id);
          };
          nof_eval2(net, nhi, nlo, cur, st, carry, id, follow, out, rb, stage_next);
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (role == 0) { canon[b][0] = out[b][0]; canon[b][1] = out[b][1]; canon[b][2] = out[b][2]; }
            if (role == 1) dl[b] = (fabsf(x[b][0] - out[b][0]) + fabsf(x[b][1] - out[b][1]) + fabsf(x[b][2] - out[b][2])) / 3.f;
            if (role == 4) dg[b] = (fabsf(x[b][0] - out[b][0]) + fabsf(x[b][1] - out[b][1]) + fabsf(x[b][2] - out[b][2])) / 3.f;
            cur[b][0] = out[b][0]; cur[b][1] = out[b][1]; cur[b][2] = out[b][2];
          }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          xin[b][0] = canon[b][0]; xin[b][1] = canon[b][1]; xin[b][2] = canon[b][2];
          if (valid[b] && id.h == 0) {
            const long long idx = (ray0 + rr[b]) * S + si[b];
            if (loc && p.disp_local) p.disp_local[idx] = dl[b];
            if (glob && p.disp_global) p.disp_global[idx] = dg[b];
          }
        }
        // rows of the NEXT tile's first evaluation (see render_kernel_bf16)
        const bool more = tile + 1 < ntiles;
        const long long ng = more ? group : group + gridDim.x;
        if (ng < p.n_groups) {
          const int nnr = group_rays(p.n_rays, ng, p.G);
          int nf, nn;
          tile_rays<TILE>(more ? tile + 1 : 0, nnr, S, nf, nn);
          stage_raybias<NW>(p, ng * p.G + nf, nn, 0, p.rb_off + (uint32_t)(seq & 1) * p.rb_buf_bytes, id);
        }
      }
      st.tl.stamp(3, id);
#pragma unroll
      for (int b = 0; b < 2; ++b)          // (depth to LDS now, carrying the NaN poison: see render_kernel_bf16)
        if (valid[b] && id.h == 0) zbuf[srel[b]] = z[b] + (xin[b][0] + xin[b][1] + xin[b][2]) * 0.f;
      float sigma[2], rgb[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      u32x4 xe[2][kKsNerfXyz];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float embx[B2Xyz10::SLOTS];
        emb_eval<3, 10, true>(embx, xin[b], par_nerf_xyz, id.h, p.pow2 & 1);
        pack_operands<kKsNerfXyz>(embx, B2Xyz10::SLOTS, xe[b]);
      }
      auto make_extra = [&](int b, u32x4 (&eo)[kKsExtraMax]) {
        float ext[8 * kKsExtraMax];
#pragma unroll
        for (int e = 0; e < 8 * kKsExtraMax; ++e) ext[e] = 0.f;
        const float* q = b ? rp[1] : rp[0];
        if (p.extra_type == MF_EXTRA_DIR) {
          const float dd[3] = {q[3], q[4], q[5]};
          emb_eval<3, 4, true>(ext, dd, par_nerf_ext, id.h, p.pow2 & 2);                         // rendering.py:138-142
        } else if (p.extra_type == MF_EXTRA_IND) {
          const float iv[1] = {q[8]};
          emb_eval<1, 2, true>(ext, iv, par_nerf_ext, id.h, p.pow2 & 2);                         // rendering.py:133-137
        }
        pack_operands<kKsExtraMax>(ext, 8 * kKsExtraMax, eo);
      };
      st.tl.stamp(4, id);
      nerf_eval2(p.nerf, xe, make_extra, sigma_only, st, carry, id, prog_first, sigma, rgb);
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (valid[b] && id.h == 0) sbuf[srel[b]] = make_float4(rgb[b][0], rgb[b][1], rgb[b][2], sigma[b]);
      st.tl.stamp(5, id);
    }
    __syncthreads();
    composite_group<NW>(p, id, ray0, nr, S, sigma_only, sbuf, zbuf);
    __syncthreads();
  }
  wait_vm0();
}

// sigma (and the canonical position) of free points in bf16 mode: the lattice / SMPL-point query of mf_forward.hip's
// points_kernel (trainer_moco_flow.py:146-187, 500-526) on the 32x32x16 core -- xyz -> [bw NoF(ind)] -> encode -> NeRF
// trunk -> sigma, 256 points per workgroup tile, nothing else written.
struct PointsParamsBf {
  Net nerf, bw;
  float emb_par[4][32];            // [nerf xyz, -, nof xyz, nof ind] x (freq[16], weight[16]) -> LDS
  const float* xyz;                // (B,3)
  const float* ind;                // (B,) per-point image index, or null -> ind_scalar
  float ind_scalar;
  long long B;
  float* sigma;                    // (B,) raw sigma
  float* canon;                    // (B,3) or null
  uint32_t par_off, ring_off, buf_bytes;
  int pow2;
  const float* raybias;            // NOF: (B | 1, rb_layers, 128) per-point (ind given) or single (ind_scalar) NoF biases
  int rb_layers;
  uint32_t rb_off;                 // LDS: the single entry (ind_scalar)
};

// PERPT: an image index per point -- each lane fetches its own bias rows from the global table (register sets, RayBias);
// otherwise the one entry of ind_scalar sits in LDS for the whole launch.
// X3 (MF_PREC_BF16X3; not with PERPT): the three-product networks, 4 waves, 128 points per tile.
template <bool NOF, bool PERPT = false, bool X3 = false>
__global__ __launch_bounds__(X3 ? 256 : kThreads, X3 ? 1 : 2) void points_kernel_bf16(const PointsParamsBf p) {
  static_assert(!(X3 && PERPT), "per-point image indices: fast mode only");
  constexpr int NW = X3 ? 4 : kWaves;
  constexpr int TILE = NW * kWaveSamples;
  const Lane id;
  load_resident<NW>(p.nerf, id);
  if (NOF) load_resident<NW>(p.bw, id);
  if (NOF && !PERPT) stage_raybias<NW>(p.raybias, 1, p.rb_layers, 0, 1, 0, p.rb_off, id);
  if (threadIdx.x < 128) {
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PointsParamsBf, emb_par);
    *(float*)(smem + p.par_off + threadIdx.x * 4) = ((const __attribute__((address_space(4))) float*)ka)[threadIdx.x];
  }
  const uint32_t par_nerf_xyz = p.par_off, par_nof_xyz = p.par_off + 256;
  StreamT<NW> st;
  st.tl.start(nullptr, id);
  typename std::conditional<X3, CarryX, Carry>::type carry;
  constexpr int NP2 = X3 ? 1 : kNofTpp0, NF0 = X3 ? 1 : kNerfTpp0, NPH = X3 ? 0 : kNofTppH, NPS = X3 ? 0 : kNofTppS;
  constexpr int TN = X3 ? kNofTermsX3 : 2;
  const Next nerf_first = first_of<16, kKsNerfXyz, X3, NF0>(p.nerf);
  const Next prog_first = NOF ? first_of<8, kKsNofXyz, true, NP2, TN, NPH, NPS>(p.bw) : nerf_first;
  if (NOF) start_program<8, kKsNofXyz, true, NP2, TN, NPH, NPS>(p.bw, st, carry, p.ring_off, p.buf_bytes, id);
  else start_program<16, kKsNerfXyz, X3, NF0>(p.nerf, st, carry, p.ring_off, p.buf_bytes, id);
  const long long ntiles = (p.B + TILE - 1) / TILE;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long b = tile * TILE + id.wave * kWaveSamples + id.j;
    const bool valid = b < p.B;
    const long long bb = valid ? b : p.B - 1;
    float x[3] = {p.xyz[bb * 3 + 0], p.xyz[bb * 3 + 1], p.xyz[bb * 3 + 2]};
    if (NOF) {
      u32x4 nhi[kKsNofXyz], nmid[TN == 3 ? kKsNofXyz : 1], nlo[kKsNofXyz];
      float out[3];
      if constexpr (PERPT) {
        const float* rbp = p.raybias + (size_t)bb * (size_t)(p.rb_layers * 128) + 4 * id.h;
        RayBias rb;
        load_raybias(rb, rbp, 0);                      // in flight across the encoding
        nof_embed(nhi, nlo, x, par_nof_xyz, id.h, p.pow2 & 4);
        nof_eval(p.bw, nhi, nlo, x, st, carry, id, nerf_first, out, rb, rbp, [] {});
      } else {
        LdsRayBias rb{p.rb_off};
        if constexpr (X3) nof_embed_t<false, TN, kNofHalfX3>(nhi, reinterpret_cast<u32x4(&)[kKsNofXyz]>(nmid), nlo, x, par_nof_xyz, id.h, p.pow2 & 4);
        else nof_embed<true>(nhi, nlo, x, par_nof_xyz, id.h, p.pow2 & 4);
        if constexpr (X3) nof_eval_x3<TN, kNofHalfX3>(p.bw, nhi, reinterpret_cast<const u32x4(&)[kKsNofXyz]>(nmid), nlo, x, st, carry, id, nerf_first, out, rb, [] {});
        else nof_eval(p.bw, nhi, nlo, x, st, carry, id, nerf_first, out, rb, nullptr, [] {});
      }
      x[0] = out[0]; x[1] = out[1]; x[2] = out[2];
      if (valid && id.h == 0 && p.canon) {
        p.canon[b * 3 + 0] = x[0]; p.canon[b * 3 + 1] = x[1]; p.canon[b * 3 + 2] = x[2];
      }
    }
    float sigma, rgb[3] = {0.f, 0.f, 0.f};
    const float nanprop = (x[0] + x[1] + x[2]) * 0.f;      // (NaN / inf in -> NaN out: see render_kernel_bf16)
    if constexpr (X3) {
      u32x4 xh[kKsNerfXyz], xl[kKsNerfXyz];
      {
        float embx[B2Xyz10::SLOTS];
        emb_eval<3, 10, false>(embx, x, par_nerf_xyz, id.h, p.pow2 & 1);
        split_operands<kKsNerfXyz>(embx, B2Xyz10::SLOTS, xh, xl);
      }
      auto make_extra = [&](u32x4 (&eh)[kKsExtraMax], u32x4 (&el)[kKsExtraMax]) {   // (sigma only: never reached)
#pragma unroll
        for (int k = 0; k < kKsExtraMax; ++k) { eh[k] = u32x4{0u, 0u, 0u, 0u}; el[k] = u32x4{0u, 0u, 0u, 0u}; }
      };
      nerf_eval_x3(p.nerf, xh, xl, make_extra, true, st, carry, id, prog_first, sigma, rgb);
    } else {
      u32x4 xe[kKsNerfXyz];
      {
        float embx[B2Xyz10::SLOTS];
        emb_eval<3, 10, true>(embx, x, par_nerf_xyz, id.h, p.pow2 & 1);
        pack_operands<kKsNerfXyz>(embx, B2Xyz10::SLOTS, xe);
      }
      auto make_extra = [&](u32x4 (&eo)[kKsExtraMax]) {          // (sigma only: the extra block is never reached)
#pragma unroll
        for (int k = 0; k < kKsExtraMax; ++k) eo[k] = u32x4{0u, 0u, 0u, 0u};
      };
      nerf_eval(p.nerf, xe, make_extra, true, st, carry, id, prog_first, sigma, rgb);
    }
    if (valid && id.h == 0) p.sigma[b] = sigma + nanprop;
  }
  wait_vm0();
}

static bool emb_table(const mf_embedding& e, float* dst) {      // returns: frequencies are exactly 2^k
  bool pow2 = true;
  for (int k = 0; k < 16; ++k) {
    dst[k] = k < e.n_freqs ? e.freq[k] : 0.f;
    dst[16 + k] = k < e.n_freqs ? e.weight[k] : 0.f;
    if (k < e.n_freqs && e.freq[k] != (float)(1 << k)) pow2 = false;
  }
  return pow2;
}

// combination c of a ray-bias table: network (packed buffer + layout) and the index column it reads
// largest panel of a network in groups when its trunk tiles stream several per panel (the fast mode's <TPP0, TPPH, TPPS>:
// layer 0, hidden-only layers, skip layers; the NoF's head / the NeRF's extra_encoding tiles are one panel each)
static int fast_panel_groups(const NetLayout& L, bool is_nof, int tpp0, int tpph, int tpps) {
  int g = is_nof ? head_groups(L) : extra_groups(L);
  for (int l = 0; l < L.n_trunk; ++l) {
    const int tpp = l == 0 ? tpp0 : (((L.emb_mask >> l) & 1) ? tpps : tpph);
    if (tpp * trunk_groups(L, l) > g) g = tpp * trunk_groups(L, l);
  }
  return g;
}
static int nof_panel_groups(const NetLayout& L, bool x3) {
  return x3 ? fast_panel_groups(L, true, 1, 1, 1) : fast_panel_groups(L, true, bf::kNofTpp0, bf::kNofTppH, bf::kNofTppS);
}

static void raybias_combo(RayBiasParams& r, int c, const void* packed, const NetLayout& L, int col) {
  const char* base = static_cast<const char*>(packed);
  r.bias[c] = reinterpret_cast<const float*>(base) + L.off_bias_trunk;
  r.wind[c] = reinterpret_cast<const float*>(base + L.res_bytes + L.panel_bytes);
  r.emb_mask[c] = L.emb_mask;
  r.col[c] = col;
}

}  // namespace bf

int device_cus();   // mf_forward.hip

// combinations (network, index value) and embedded layers per network of a bf16 pass with NoF; 0 combos = no table
static void raybias_shape(const mf_render_args* a, int& combos, int& layers) {
  combos = layers = 0;
  if (!a->nof_bw) return;
  combos = (a->flags & MF_F_CHAIN_GLOBAL) ? 4 : ((a->flags & MF_F_CHAIN_LOCAL) ? 2 : 1);
  layers = __builtin_popcount(1u | a->nof_bw->skip_mask);
}

int64_t render_workspace_bytes_bf16(const mf_render_args* a) {
  int combos, layers;
  raybias_shape(a, combos, layers);
  return (int64_t)a->n_rays * combos * layers * 128 * 4;
}

// called by mf_render_pass / mf_render_prepare (mf_render.hip) after argument validation, precision == MF_PREC_BF16.
// prepare_only: fill the workspace (the per-ray NoF bias table) and return; else: the fused launch, which reads it.
int render_pass_bf16(const mf_render_args* a, hipStream_t st, bool prepare_only) {
  using namespace bf;
  Params p{};
  NetLayout Ln, Lb, Lf;
  const int prec = a->precision;                // MF_PREC_BF16 | MF_PREC_BF16X3
  const bool x3 = prec == MF_PREC_BF16X3;
  if (!nerf_layout(*a->nerf, Ln, prec)) return fail(MF_E_UNSUPPORTED, "mf_render_pass: unsupported NeRF configuration (bf16: W = 256)");
  const bool moco = a->nof_bw != nullptr;
  const bool chains = a->flags & (MF_F_CHAIN_LOCAL | MF_F_CHAIN_GLOBAL);
  p.rays = a->rays; p.ray_stride = a->ray_stride; p.n_rays = a->n_rays; p.bg = a->background;
  p.S = a->n_samples; p.z_vals = a->z_vals; p.z_steps = a->z_steps; p.use_disp = a->use_disp;
  p.noise = a->noise; p.activation = a->activation; p.flags = a->flags;
  p.extra_type = a->nerf->extra_feat_type;
  p.pow2 = (emb_table(a->emb_xyz, p.emb_par[0]) ? 1 : 0) | (emb_table(a->emb_extra, p.emb_par[1]) ? 2 : 0);
  p.rgb = a->rgb; p.depth = a->depth; p.opacity = a->opacity; p.weights = a->weights; p.alphas = a->alphas;
  p.disp_local = a->disp_local; p.disp_global = a->disp_global;

  const int tile_samples = x3 ? 4 * bf::kWaveSamples : bf::kTile;      // x3: 4 waves per workgroup
  uint32_t lds = 0;
  auto net_of = [&](const NetLayout& L, const void* packed, int D, int aux) {
    Net n;
    n.packed = static_cast<const char*>(packed);
    n.res_lds = lds;
    n.res_bytes = (uint32_t)L.res_bytes;
    n.D = D;
    n.emb_mask = L.emb_mask;
    n.aux = aux;
    lds += (uint32_t)L.res_bytes;
    return n;
  };
  p.nerf = net_of(Ln, a->nerf_packed, Ln.n_trunk - 1, Ln.extra_steps);
  int max_groups = x3 ? Ln.max_groups : fast_panel_groups(Ln, false, bf::kNerfTpp0, bf::kNerfTppH, bf::kNerfTppS);
  if (moco) {
    if (!nof_layout(*a->nof_bw, Lb, prec)) return fail(MF_E_UNSUPPORTED, "mf_render_pass: unsupported backward NoF configuration");
    p.bw = net_of(Lb, a->nof_bw_packed, Lb.n_trunk, Lb.n_head);
    if (bf::nof_panel_groups(Lb, x3) > max_groups) max_groups = bf::nof_panel_groups(Lb, x3);
    if (chains) {
      if (!nof_layout(*a->nof_fw, Lf, prec)) return fail(MF_E_UNSUPPORTED, "mf_render_pass: unsupported forward NoF configuration");
      p.fw = net_of(Lf, a->nof_fw_packed, Lf.n_trunk, Lf.n_head);
      if (bf::nof_panel_groups(Lf, x3) > max_groups) max_groups = bf::nof_panel_groups(Lf, x3);
    }
    p.pow2 |= (emb_table(a->nof_emb_xyz, p.emb_par[2]) ? 4 : 0) | (emb_table(a->nof_emb_ind, p.emb_par[3]) ? 8 : 0);
    // the per-ray bias table (image-index block of the NoFs' embedded-input layers), one small launch in front
    int combos, layers;
    raybias_shape(a, combos, layers);
    if (chains && __builtin_popcount(1u | a->nof_fw->skip_mask) != layers)
      return fail(MF_E_UNSUPPORTED, "mf_render_pass(bf16): backward and forward NoF must have the same number of skip layers");
    const int64_t need = render_workspace_bytes_bf16(a);
    if (need > 0 && (!a->workspace || a->workspace_bytes < need))
      return fail(MF_E_INVALID, "mf_render_pass(bf16): workspace of %lld bytes needed (mf_render_workspace_bytes), got %lld",
                  (long long)need, (long long)(a->workspace ? a->workspace_bytes : 0));
    if (a->n_rays > 0x7fffffffLL) return fail(MF_E_UNSUPPORTED, "mf_render_pass(bf16): too many rays for one launch");
    p.raybias = static_cast<const float*>(a->workspace);
    p.rb_combos = combos; p.rb_layers = layers;
    if (a->n_rays > 0 && prepare_only) {
      RayBiasParams r{};
      r.ind = a->rays; r.ind_stride = a->ray_stride; r.n_entries = a->n_rays; r.n_combos = combos; r.n_layers = layers;
      raybias_combo(r, 0, a->nof_bw_packed, Lb, 8);
      if (chains) {
        raybias_combo(r, 1, a->nof_fw_packed, Lf, 8);
        raybias_combo(r, 2, a->nof_fw_packed, Lf, 9);
        raybias_combo(r, 3, a->nof_bw_packed, Lb, 9);
      }
      for (int k = 0; k < 16; ++k) { r.freq[k] = p.emb_par[3][k]; r.weight[k] = p.emb_par[3][16 + k]; }
      r.out = static_cast<float*>(a->workspace);
      hipLaunchKernelGGL(nof_raybias_kernel, dim3((unsigned)((a->n_rays + kRbEntries - 1) / kRbEntries), combos), dim3(256), 0, st, r);
    }
  }
  if (prepare_only) return moco ? check_launch("mf_render_prepare") : MF_OK;
  p.par_off = lds; lds += 512;
  p.ring_off = lds;
  p.buf_bytes = (uint32_t)max_groups * kGroupBytes;
  lds += 3 * p.buf_bytes;
  if (moco) {
    // two buffers of per-ray bias rows: a tile of T samples touches at most ((T - 1) / S) + 2 rays
    const int r_max = (tile_samples - 1) / a->n_samples + 2;
    p.rb_buf_bytes = (uint32_t)round_up((int64_t)r_max * p.rb_layers * 512, 1024);
    p.rb_off = lds; lds += 2 * p.rb_buf_bytes;
    if (lds + 20u * (uint32_t)a->n_samples > 160u * 1024u)
      return fail(MF_E_UNSUPPORTED, "mf_render_pass(bf16): n_samples=%d leaves no room for the per-ray NoF bias rows of a tile "
                  "(%d rays); use MF_PREC_F32 for such short rays", a->n_samples, r_max);
  }

  // rays per group: smallest G with G*S a multiple of the tile (256 samples; x3: 128), capped by the LDS left
  const uint32_t lds_cap = 160 * 1024;
  const int max_samples = (int)((lds_cap - lds) / 20);
  const int S = a->n_samples;
  if (S > max_samples) return fail(MF_E_UNSUPPORTED, "mf_render_pass: n_samples=%d exceeds the %d samples a workgroup can stage", S, max_samples);
  int G = 1;
  while ((G * S) % tile_samples != 0 && (G + 1) * S <= max_samples && G < 64) ++G;
  if ((G * S) % tile_samples != 0) {        // no exact fit: take as many rays as reduce the padding waste
    int best = 1; double best_eff = 0;
    for (int g = 1; g * S <= max_samples && g <= 64; ++g) {
      const int tiles = (g * S + tile_samples - 1) / tile_samples;
      const double eff = (double)(g * S) / (tiles * tile_samples);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = g; }
    }
    G = best;
  }
  // Several such ray sets per group (up to 8), as in the fp32 pass (mf_render.hip): the composite phase between two groups
  // keeps at most one wave per ray busy and costs two workgroup barriers, so it comes once per several tiles -- as long as the
  // CUs' shares stay what they were (same makespan in rays).
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
  const bool dump = a->dump_acts || a->dump_rgbsigma || a->dump_xyz || a->dump_nof_acts;
  if (dump) {      // (validated by the caller: bf16x3)
    if (a->dump_acts && a->dump_stride < (int64_t)Ln.n_trunk * Ln.W + Ln.W / 2)
      return fail(MF_E_INVALID, "mf_render_pass: dump_stride %lld too small", (long long)a->dump_stride);
    if (a->dump_acts && ((a->dump_stride & 3) || (reinterpret_cast<uintptr_t>(a->dump_acts) & 15)))
      return fail(MF_E_INVALID, "mf_render_pass(bf16x3): dump_acts must be 16-byte aligned with a stride that is a multiple of 4 floats");
    p.dump_acts = a->dump_acts; p.dump_stride = a->dump_stride; p.dump_rgbsigma = a->dump_rgbsigma; p.dump_xyz = a->dump_xyz;
    if (a->dump_mask) {
      if (!a->dump_acts || a->dump_mask_stride < (int64_t)(Ln.n_trunk + 1) * 8)
        return fail(MF_E_INVALID, "mf_render_pass: dump_mask needs dump_acts and dump_mask_stride >= 8 (D + 2) words");
      p.dump_mask = a->dump_mask; p.dump_mask_stride = a->dump_mask_stride;
    }
    if (a->dump_nof_acts) {
      // the chain's evaluations: rows [h_1 .. h_D | T padded to 16] (no ReLU bit words, no embedded-input plane: mf_nof_embed_rows)
      if (!moco || !a->dump_nof_out) return fail(MF_E_INVALID, "mf_render_pass(bf16x3): dump_nof_acts needs NoF models and dump_nof_out");
      if (a->dump_nof_stride < (int64_t)Lb.n_trunk * Lb.W + 16 || (a->dump_nof_stride & 3) || (reinterpret_cast<uintptr_t>(a->dump_nof_acts) & 15))
        return fail(MF_E_INVALID, "mf_render_pass(bf16x3): dump_nof_acts must be 16-byte aligned with a stride >= D W + 16 that is a multiple of 4 floats");
      if (chains && (Lf.n_trunk != Lb.n_trunk)) return fail(MF_E_UNSUPPORTED, "mf_render_pass(bf16x3): the NoF dump needs both flows of one depth");
      const int nsteps = 1 + ((a->flags & MF_F_CHAIN_LOCAL) ? 1 : 0) + ((a->flags & MF_F_CHAIN_GLOBAL) ? 3 : 0);
      uint32_t seen = 0;
      for (int k = 0; k < nsteps; ++k) {
        const int pl = a->dump_nof_plane[k];
        if (pl < 0 || pl >= nsteps || ((seen >> pl) & 1u)) return fail(MF_E_INVALID, "mf_render_pass: dump_nof_plane must be a permutation of 0..%d", nsteps - 1);
        seen |= 1u << pl;
        p.nof_plane_pack |= (uint32_t)pl << (3 * k);
      }
      p.dump_nof_acts = a->dump_nof_acts; p.dump_nof_stride = a->dump_nof_stride; p.dump_nof_out = a->dump_nof_out;
    }
  }
  // the fast mode's kernels: one column block per wave (8 waves, two per SIMD: the default) or two (mf_bf16_2b.hpp: 4 waves, one
  // per SIMD, every weight fragment read from LDS feeds two MFMAs); same tiles, same LDS layout, bit-identical results.
  // MF_BF16_BLOCKS=1|2 selects the family.  Round 6 measured the two-block kernels at parity, not ahead (profiles/r06_two_blocks.txt:
  // half the LDS instructions, 13 % more cycles with a lone wave per SIMD, a higher clock -- C2 shape -0.5 %, C3 +0.7 %), so
  // they ship opt-in.
  // (read per call: a getenv is ~100 ns beside a >= 300 us pass, and a test can switch families inside one process)
  const char* be = getenv("MF_BF16_BLOCKS");
  const int blocks = be && be[0] == '1' ? 1 : (be && be[0] == '2' ? 2 : kFastBlocksDefault);
  const bool two = !x3 && blocks == 2;
  void (*kern)(const Params) = x3 ? (moco ? (dump ? render_kernel_bf16<true, true, true> : render_kernel_bf16<true, true>)
                                          : (dump ? render_kernel_bf16<false, true, true> : render_kernel_bf16<false, true>))
                                  : two ? (moco ? render_kernel_bf16_2b<true> : render_kernel_bf16_2b<false>)
                                        : (moco ? render_kernel_bf16<true, false> : render_kernel_bf16<false, false>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_render_pass: cannot reserve %u bytes of LDS", lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3((x3 || two) ? 256 : kThreads), lds, st, p);
  return check_launch("mf_render_pass(bf16)");
}

// called by mf_points_sigma_p (mf_forward.hip) after argument validation, precision == MF_PREC_BF16
int64_t points_workspace_bytes_bf16(const mf_nof_desc* nof, int per_point_ind, int64_t B) {
  if (!nof) return 0;
  return (per_point_ind ? B : 1) * (int64_t)__builtin_popcount(1u | nof->skip_mask) * 128 * 4;
}

int points_sigma_bf16(int prec, const mf_nerf_desc* nerf, const void* nerf_packed, const mf_embedding* emb_xyz, const mf_nof_desc* nof,
                      const void* nof_packed, const mf_embedding* nof_emb_xyz, const mf_embedding* nof_emb_ind, const float* xyz,
                      const float* ind, float ind_scalar, int64_t B, float* sigma, float* canon, void* workspace,
                      int64_t workspace_bytes, hipStream_t st) {
  using namespace bf;
  PointsParamsBf p{};
  NetLayout Ln, Lb;
  const bool x3 = prec == MF_PREC_BF16X3;
  if (x3 && nof && ind) return fail(MF_E_UNSUPPORTED, "mf_points_sigma(bf16x3): per-point image indices are not built (use MF_PREC_F32 or MF_PREC_BF16)");
  if (!nerf_layout(*nerf, Ln, prec)) return fail(MF_E_UNSUPPORTED, "mf_points_sigma: unsupported NeRF configuration (bf16: W = 256)");
  uint32_t lds = 0;
  auto net_of = [&](const NetLayout& L, const void* packed, int D, int aux) {
    Net n;
    n.packed = static_cast<const char*>(packed);
    n.res_lds = lds;
    n.res_bytes = (uint32_t)L.res_bytes;
    n.D = D;
    n.emb_mask = L.emb_mask;
    n.aux = aux;
    lds += (uint32_t)L.res_bytes;
    return n;
  };
  p.nerf = net_of(Ln, nerf_packed, Ln.n_trunk - 1, Ln.extra_steps);
  int max_groups = x3 ? Ln.max_groups : fast_panel_groups(Ln, false, bf::kNerfTpp0, bf::kNerfTppH, bf::kNerfTppS);
  p.pow2 = emb_table(*emb_xyz, p.emb_par[0]) ? 1 : 0;
  if (nof) {
    if (!nof_layout(*nof, Lb, prec)) return fail(MF_E_UNSUPPORTED, "mf_points_sigma: unsupported NoF configuration");
    p.bw = net_of(Lb, nof_packed, Lb.n_trunk, Lb.n_head);
    if (nof_panel_groups(Lb, x3) > max_groups) max_groups = nof_panel_groups(Lb, x3);
    p.pow2 |= (emb_table(*nof_emb_xyz, p.emb_par[2]) ? 4 : 0) | (emb_table(*nof_emb_ind, p.emb_par[3]) ? 8 : 0);
    // per-point (ind given) or single (ind_scalar) bias of the NoF's embedded-input layers, see nof_raybias_kernel
    const int64_t need = points_workspace_bytes_bf16(nof, ind != nullptr, B);
    if (!workspace || workspace_bytes < need)
      return fail(MF_E_INVALID, "mf_points_sigma(bf16): workspace of %lld bytes needed (mf_points_sigma_workspace_bytes), got %lld",
                  (long long)need, (long long)(workspace ? workspace_bytes : 0));
    const int64_t entries = ind ? B : 1;
    if (entries > 0x7fffffffLL) return fail(MF_E_UNSUPPORTED, "mf_points_sigma(bf16): too many points for one launch");
    p.raybias = static_cast<const float*>(workspace);
    p.rb_layers = Lb.n_emb_layers;
    if (B > 0) {
      RayBiasParams r{};
      r.ind = ind; r.ind_stride = 1; r.ind_scalar = ind_scalar; r.n_entries = entries; r.n_combos = 1; r.n_layers = Lb.n_emb_layers;
      raybias_combo(r, 0, nof_packed, Lb, 0);
      for (int k = 0; k < 16; ++k) { r.freq[k] = p.emb_par[3][k]; r.weight[k] = p.emb_par[3][16 + k]; }
      r.out = static_cast<float*>(workspace);
      hipLaunchKernelGGL(nof_raybias_kernel, dim3((unsigned)((entries + kRbEntries - 1) / kRbEntries), 1), dim3(256), 0, st, r);
    }
  }
  p.par_off = lds; lds += 512;
  p.ring_off = lds;
  p.buf_bytes = (uint32_t)max_groups * kGroupBytes;
  lds += 3 * p.buf_bytes;
  if (nof) { p.rb_off = lds; lds += (uint32_t)round_up((int64_t)Lb.n_emb_layers * 512, 1024); }
  p.xyz = xyz; p.ind = ind; p.ind_scalar = ind_scalar; p.B = B; p.sigma = sigma; p.canon = canon;
  const int tile = x3 ? 4 * bf::kWaveSamples : bf::kTile;
  const long long ntiles = (B + tile - 1) / tile;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  void (*kern)(const PointsParamsBf) = x3 ? (nof ? points_kernel_bf16<true, false, true> : points_kernel_bf16<false, false, true>)
                                          : (nof ? (ind ? points_kernel_bf16<true, true> : points_kernel_bf16<true, false>)
                                                 : points_kernel_bf16<false>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_points_sigma: cannot reserve %u bytes of LDS", lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(x3 ? 256 : kThreads), lds, st, p);
  return check_launch("mf_points_sigma(bf16)");
}

}  // namespace mf
