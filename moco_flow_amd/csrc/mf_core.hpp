// mf_core.hpp -- register-resident fused MLP core for gfx950 (MI355X, CDNA4).
//
// Design: the MLPs of models/nerf.py and models/nof.py are evaluated
// TRANSPOSED, H_out^T = W * H_in^T, on v_mfma_f32_32x32x2_f32:
//   A operand = a 32-row slice of the nn.Linear weight (rows = output features),
//   B operand = activations (columns = 32 ray-samples, one per lane&31),
//   C/D       = 32 output features x 32 samples, 16 fp32 per lane.
// The C/D register layout of that instruction (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5),
// col = lane&31) is exactly a valid B-operand layout for 16 further k-steps if the k
// order is permuted to  k(step=(q,r), half h) = r + 8q + 4h.  The weights are re-ordered
// once on the host side into that order ("fragment stream", mf_pack.hip), so a layer's
// output registers feed the next layer's MFMAs with no shuffle, no LDS round trip and no
// HBM traffic: activations never leave the register file.  One wave owns 32 samples and
// the full hidden vector (W/32 tiles x 16 regs); the four waves of a workgroup share the
// weight stream, which is DMA'd global->LDS (global_load_lds_dwordx4) one "panel" (the
// 32 output rows of one layer) ahead of the MFMAs, double buffered.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGroupBytes = 1024;   // one fragment group: 64 lanes x float4 = 4 k-steps
constexpr int kWaves = 4;           // waves per workgroup (one per SIMD)
constexpr int kTile = 128;          // samples per workgroup tile (32 per wave)

#define MF_HD __host__ __device__ __forceinline__
#define MF_D __device__ __forceinline__

// ------------------------------------------------------------------ embedding blocks
// A frequency embedding (models/embedding.py:42-46) of C components with at most F
// frequencies, split across the two 32-lane halves of a wave: half h owns the sin/cos
// pairs p = 2*pi + h (p -> frequency p / C, component p % C) and the raw components
// 2*ri + h.  SLOTS registers per lane; slot e of half h is reference column feature(h,e).
template <int C, int F>
struct EmbBlock {
  static constexpr int NPAIR = (C * F + 1) / 2;
  static constexpr int NRAW = (C + 1) / 2;
  static constexpr int SLOTS = 2 * NPAIR + NRAW;
  MF_HD static int feature(int h, int e) {
    if (e < 2 * NPAIR) {
      const int p = 2 * (e >> 1) + h;
      if (p >= C * F) return -1;
      const int f = p / C, c = p % C;
      return C + 2 * C * f + C * (e & 1) + c;
    }
    const int r = 2 * (e - 2 * NPAIR) + h;
    return r < C ? r : -1;
  }
};

constexpr int round4(int x) { return (x + 3) & ~3; }

// Input-slot maps of the three places an embedding enters a network.  `steps` is the
// number of MFMA k-steps (= registers per lane), a multiple of 4 (one fragment group).
enum EmbKind : int { kEmbNerfXyz = 0, kEmbNofIn = 1, kEmbDir = 2, kEmbInd = 3, kEmbNone = 4 };

using BlkXyz10 = EmbBlock<3, 10>;   // NeRF xyz, in_channels_xyz = 63
using BlkXyz5 = EmbBlock<3, 5>;     // NoF xyz, in_channels_xyz = 33
using BlkInd16 = EmbBlock<1, 16>;   // NoF ind, extra_feat_dim = 33
using BlkDir4 = EmbBlock<3, 4>;     // NeRF dir, 27
using BlkInd2 = EmbBlock<1, 2>;     // NeRF ind, 5

constexpr int kStepsNerfXyz = round4(BlkXyz10::SLOTS);                    // 32
constexpr int kStepsNofIn = round4(BlkXyz5::SLOTS + BlkInd16::SLOTS);     // 36
constexpr int kStepsDir = round4(BlkDir4::SLOTS);                         // 16
constexpr int kStepsInd = round4(BlkInd2::SLOTS);                         // 4
constexpr int kStepsExtraMax = kStepsDir;

// reference column (within the embedded input vector) of slot e of half h; -1 = zero pad.
// `xyz_cols` = the network's in_channels_xyz (NoF: the ind block starts there).
MF_HD int emb_feature(int kind, int h, int e, int xyz_cols) {
  switch (kind) {
    case kEmbNerfXyz:
      return e < BlkXyz10::SLOTS ? BlkXyz10::feature(h, e) : -1;
    case kEmbNofIn:
      if (e < BlkXyz5::SLOTS) return BlkXyz5::feature(h, e);
      if (e < BlkXyz5::SLOTS + BlkInd16::SLOTS) {
        const int f = BlkInd16::feature(h, e - BlkXyz5::SLOTS);
        return f < 0 ? -1 : xyz_cols + f;
      }
      return -1;
    case kEmbDir:
      return e < BlkDir4::SLOTS ? BlkDir4::feature(h, e) : -1;
    case kEmbInd:
      return e < BlkInd2::SLOTS ? BlkInd2::feature(h, e) : -1;
    default:
      return -1;
  }
}

// Embedding parameters as the kernels see them (uniform, lives in SGPRs / kernarg).
struct EmbParams {
  float freq[16];
  float weight[16];   // 0 beyond the module's N_freqs (== the reference's zero padding)
};

// ------------------------------------------------------------------ packed layouts
// A packed network = [resident block (biases, VALU head weights), padded to 1 KiB]
//                    [panels in program order], each panel = groups x 1 KiB.
struct NetLayout {
  int W, NT;               // hidden width, W/32
  int n_trunk;             // trunk layers streamed through the generic loop (NeRF: D+1 incl. final)
  int emb_steps;           // k-steps of the embedded-input block
  uint32_t emb_mask;       // trunk layers that consume the embedded input (layer 0 + skips)
  uint32_t relu_mask;      // trunk layers followed by ReLU
  int extra_steps;         // NeRF extra_encoding: k-steps of the extra block (0, 4, 16); -1 = no extra layer
  int64_t res_bytes;       // resident block size (multiple of 1 KiB)
  int64_t panel_bytes;     // all panels
  int max_groups;          // largest panel, in groups
  // resident block float offsets
  int off_bias_trunk;      // n_trunk * W
  int off_bias_extra;      // W/2
  int off_head_w;          // NeRF: sigma_w (W) ; NoF: head_w (n_head * W)
  int off_head_b;          // NeRF: sigma_b ; NoF: head_b
  int off_rgb_w;           // NeRF: 3 * W/2
  int off_rgb_b;
  int n_head;              // NoF: 9 | 3
};

MF_HD int trunk_groups(const NetLayout& L, int layer) {
  return (((L.emb_mask >> layer) & 1) ? L.emb_steps / 4 : 0) + (layer > 0 ? L.NT * 4 : 0);
}
MF_HD int extra_groups(const NetLayout& L) { return L.NT * 4 + L.extra_steps / 4; }

// ------------------------------------------------------------------ device helpers

extern __shared__ __attribute__((aligned(16))) char smem[];

MF_D void glds16(const char* g, uint32_t lds_off) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
}
MF_D void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

MF_D float xhalf_sum(float v) {   // v(lane) + v(lane ^ 32)
  return v + __shfl_xor(v, 32, 64);
}

struct LaneId {
  int lane, wave, j, h;
  MF_D LaneId() {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    j = lane & 31;
    h = lane >> 5;
  }
};

// Weight-panel stream: double-buffered LDS ring fed by LDS-DMA.
// Invariant between out-tiles: the current panel is complete and visible in buffer `cur`,
// no DMA in flight, every wave is past the barrier that ended the previous panel.
struct Stream {
  const char* gnext;      // global address of the next panel to fetch (wave-uniform)
  uint32_t ring;          // LDS byte offset of ring buffer 0
  uint32_t buf_bytes;     // bytes per ring buffer
  uint32_t cur;           // 0 / 1

  MF_D uint32_t cur_off() const { return ring + cur * buf_bytes; }
  // start fetching the next panel (`groups` KiB) into the other buffer
  MF_D void prefetch(int groups, const LaneId& id) {
    const uint32_t dst = ring + (cur ^ 1u) * buf_bytes;
    const char* g = gnext + id.lane * 16;
    for (int grp = id.wave; grp < groups; grp += kWaves) glds16(g + grp * kGroupBytes, dst + grp * kGroupBytes);
    gnext += (size_t)groups * kGroupBytes;
  }
  MF_D void flip() {
    wait_vm0();
    __syncthreads();
    cur ^= 1u;
  }
  // cold start: fetch the first panel of the program into buffer 0
  MF_D void start(const char* first, int groups, const LaneId& id) {
    cur = 1;
    gnext = first;
    prefetch(groups, id);
    flip();
  }
};

MF_D f32x4 lds_f4(uint32_t byte_off) { return *(const f32x4*)(smem + byte_off); }
MF_D float lds_f(uint32_t byte_off) { return *(const float*)(smem + byte_off); }

#define MF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// One output tile (32 features x 32 samples): acc += W_panel * [emb ; hidden].
// MODE: 1 = embedded input only, 2 = hidden only, 3 = both (skip layers, emb first).
// Fragment groups are fetched two ahead of the MFMAs that use them: the ds_reads for
// batch b+1 are issued right after the first MFMA of batch b (7 MFMAs = 448 cycles of
// cover), so the lgkmcnt wait in front of batch b+1 is free.
template <int MODE, int NT, int EMB>
MF_D f32x16 out_tile(f32x16 acc, const f32x16 (&hid)[NT], const float (&emb)[EMB], uint32_t panel_lane_off) {
  constexpr int GE = (MODE & 1) ? EMB / 4 : 0;
  constexpr int GH = (MODE & 2) ? NT * 4 : 0;
  constexpr int G = GE + GH;
  auto bop = [&](int g, int r) -> float {
    if (g < GE) return emb[4 * g + r];
    const int gh = g - GE;
    return hid[gh >> 2][4 * (gh & 3) + r];
  };
  f32x4 w0 = lds_f4(panel_lane_off);
  f32x4 w1 = (G > 1) ? lds_f4(panel_lane_off + kGroupBytes) : w0;
#pragma unroll
  for (int g = 0; g < G; g += 2) {
    f32x4 n0 = w0, n1 = w1;
    acc = MF_MFMA(w0[0], bop(g, 0), acc);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 2 < G) n0 = lds_f4(panel_lane_off + (g + 2) * kGroupBytes);
    if (g + 3 < G) n1 = lds_f4(panel_lane_off + (g + 3) * kGroupBytes);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 1; r < 4; ++r) acc = MF_MFMA(w0[r], bop(g, r), acc);
    if (g + 1 < G) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = MF_MFMA(w1[r], bop(g + 1, r), acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    w0 = n0;
    w1 = n1;
  }
  return acc;
}

// bias of output tile t in C/D register order: reg 4q+r <- bias[32t + 8q + 4h + r]
MF_D f32x16 bias_tile(uint32_t bias_byte_off, int t, int h) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = lds_f4(bias_byte_off + (32 * t + 8 * q + 4 * h) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[4 * q + r] = b[r];
  }
  return acc;
}

// Uniform per-network state handed to the device code (kernarg -> SGPRs).
struct NetDev {
  NetLayout L;
  const char* packed;      // global base of the packed buffer
  uint32_t res_lds;        // LDS byte offset where its resident block lives
};

// One trunk layer: out = act(W_l [emb;hid] + b_l) for all NT output tiles.
// `next_groups`/`jump`: size of the panel that follows this layer's last panel, and, if the
// program leaves this network's contiguous panel order there, its address.
template <int NT, int EMB>
MF_D void trunk_layer(const NetDev& net, int layer, f32x16 (&act)[NT], const float (&emb)[EMB],
                      Stream& st, const LaneId& id, int next_groups, const char* jump) {
  const int has_emb = (net.L.emb_mask >> layer) & 1;
  const int mode = (has_emb ? 1 : 0) | (layer > 0 ? 2 : 0);
  const int groups = trunk_groups(net.L, layer);
  const float lo = ((net.L.relu_mask >> layer) & 1) ? 0.f : -__builtin_inff();
  const uint32_t bias_off = net.res_lds + (net.L.off_bias_trunk + layer * net.L.W) * 4;
  f32x16 out[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t == NT - 1) {
      if (jump) st.gnext = jump;
      st.prefetch(next_groups, id);
    } else {
      st.prefetch(groups, id);
    }
    const uint32_t p = st.cur_off() + id.lane * 16;
    f32x16 acc = bias_tile(bias_off, t, id.h);
    if (mode == 2) acc = out_tile<2, NT, EMB>(acc, act, emb, p);
    else if (mode == 3) acc = out_tile<3, NT, EMB>(acc, act, emb, p);
    else acc = out_tile<1, NT, EMB>(acc, act, emb, p);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[t][i] = fmaxf(acc[i], lo);
    st.flip();
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) act[t] = out[t];
}

// VALU head: n_out dot products of the lane's half of the hidden vector with natural-order
// weight rows in LDS (broadcast ds_read_b128), summed across the two halves.  Every lane of
// a sample column ends up with the full sums.
template <int NT, int NOUT>
MF_D void valu_head(const f32x16 (&act)[NT], uint32_t w_byte_off, int row_floats, uint32_t b_byte_off,
                    int h, float (&out)[NOUT]) {
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = lds_f4(w_byte_off + (o * row_floats + 32 * t + 8 * q + 4 * h) * 4);
        s0 = __builtin_fmaf(w[0], act[t][4 * q + 0], s0);
        s1 = __builtin_fmaf(w[1], act[t][4 * q + 1], s1);
        s0 = __builtin_fmaf(w[2], act[t][4 * q + 2], s0);
        s1 = __builtin_fmaf(w[3], act[t][4 * q + 3], s1);
      }
    }
    out[o] = xhalf_sum(s0 + s1) + lds_f(b_byte_off + o * 4);
  }
}

// ------------------------------------------------------------------ embedding in registers
// dst[0..SLOTS) of block (C,F) for this lane-half.  arg = freq*x is rounded to fp32 before
// sin/cos exactly as `func(freq*x)` in embedding.py:45; weight multiplies the result.
template <int C, int F>
MF_D void emb_eval(float* dst, const float (&v)[C], const EmbParams& ep, int h) {
  using B = EmbBlock<C, F>;
#pragma unroll
  for (int pi = 0; pi < B::NPAIR; ++pi) {
    const int p0 = 2 * pi, p1 = 2 * pi + 1;
    const int f0 = p0 / C, c0 = p0 % C;
    const bool ok1 = p1 < C * F;
    const int f1 = ok1 ? p1 / C : f0, c1 = ok1 ? p1 % C : c0;
    const float x = h ? v[c1] : v[c0];
    const float fr = h ? ep.freq[f1] : ep.freq[f0];
    const float w = h ? (ok1 ? ep.weight[f1] : 0.f) : ep.weight[f0];
    float s = 0.f, c = 0.f;
    // weights are wave-uniform (kernarg): skip the transcendental when both halves are muted
    if (ep.weight[f0] != 0.f || (ok1 && ep.weight[f1] != 0.f)) sincosf(fr * x, &s, &c);
    dst[2 * pi] = w * s;
    dst[2 * pi + 1] = w * c;
  }
#pragma unroll
  for (int ri = 0; ri < B::NRAW; ++ri) {
    const int r0 = 2 * ri, r1 = 2 * ri + 1;
    dst[2 * B::NPAIR + ri] = h ? (r1 < C ? v[r1 < C ? r1 : 0] : 0.f) : v[r0];
  }
}


}  // namespace mf
