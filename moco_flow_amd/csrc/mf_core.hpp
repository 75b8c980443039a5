// mf_core.hpp -- register-resident fused MLP core for gfx950 (MI355X, CDNA4).
//
// Design.  The MLPs of models/nerf.py and models/nof.py are evaluated
// TRANSPOSED, H_out^T = W * H_in^T, on v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue):
//   A operand = 16 rows of the nn.Linear weight (rows = output features),
//   B operand = activations (columns = 16 ray-samples, one per lane&15),
//   C/D       = 16 output features x 16 samples, 4 fp32 per lane.
// The C/D register layout of that instruction (row = 4*(lane>>4) + reg, col = lane&15) is
// exactly a valid B-operand layout for 4 further k-steps if the k order inside a 16-wide
// k-tile is  k(step r, lane-group g) = 4g + r.  The weights are re-ordered once into that
// order ("fragment stream", mf_pack.hip), so a layer's output registers feed the next layer's
// MFMAs directly: no shuffle, no LDS round trip, no HBM traffic -- activations never leave
// the register file.
//
// One wave owns 16 samples and the whole hidden vector (W/16 k-tiles x 4 regs = 64 VGPRs for
// W = 256) and computes TWO 16-row output tiles at a time (two independent accumulator chains
// that share every B operand).  A workgroup is 8 waves = 128 samples, two waves per SIMD at
// <= 256 registers each: while one wave is in a prologue / epilogue / barrier, its SIMD
// partner keeps the matrix pipe busy.  The 8 waves share the weight stream, which is DMA'd
// global->LDS (buffer_load_dwordx4 ... lds, see blds16) in "panels" (32 output rows of one layer), running
// two panels ahead of the MFMAs in a 3-slot ring.
#pragma once
#include <hip/hip_runtime.h>
constexpr int kF32PD = 1;         // fp32 fragment prefetch distance in batches (2 measured 0.9-1.3 % slower)
#include <stdint.h>
constexpr int kF32DmaDelay = 8;   // batches (8 MFMAs each) between the late half's panel barrier and its LDS-DMA pieces
#ifndef MF_TIMING_FLAGS
#define MF_TIMING_FLAGS 0   // 1 (tools/build_ablate.sh): the kernels honour MF_DEBUG_FLAGS (timing ablations).
#endif                      // Production compiles the switches out: the tests on them cost 1.5 % of the C2 kernel.

namespace mf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kGroupBytes = 1024;   // one fragment group: 64 lanes x float4 = 16 rows x 16 k (4 MFMA steps)
constexpr int kWaves = 8;           // waves per workgroup (two per SIMD)
constexpr int kWaveSamples = 16;    // samples per wave
constexpr int kTile = 128;          // samples per workgroup tile
constexpr int kThreads = 512;

#define MF_HD __host__ __device__ __forceinline__
#define MF_D __device__ __forceinline__

constexpr int round4(int x) { return (x + 3) & ~3; }

// ------------------------------------------------------------------ embedding blocks
// A frequency embedding (models/embedding.py:42-46) of C components with at most F frequencies,
// split across the four 16-lane groups of a wave: group g owns the sin/cos pairs p = 4*pi + g
// (p -> frequency p / C, component p % C); the raw components ride as pseudo-pairs after the
// real ones (pseudo-pair q holds raw 2q, 2q+1).  SLOTS registers per lane; slot e of group g is
// reference column feature(g, e) (or -1: zero).
template <int C, int F>
struct EmbBlock {
  static constexpr int NPAIR = C * F;                       // real pairs
  static constexpr int NALL = NPAIR + (C + 1) / 2;          // + raw pseudo-pairs
  static constexpr int NPI = (NALL + 3) / 4;                // pair slots per lane group
  static constexpr int SLOTS = 2 * NPI;
  MF_HD static int feature(int g, int e) {
    const int p = 4 * (e >> 1) + g, sc = e & 1;
    if (p < NPAIR) {
      const int f = p / C, c = p % C;
      return C + 2 * C * f + C * sc + c;
    }
    const int raw = 2 * (p - NPAIR) + sc;
    return raw < C ? raw : -1;
  }
};

// Input-slot maps of the places an embedding enters a network.  `steps` = registers per lane
// = MFMA k-steps, a multiple of 4 (one fragment group per output tile).
enum EmbKind : int { kEmbNerfXyz = 0, kEmbNofIn = 1, kEmbDir = 2, kEmbInd = 3, kEmbNone = 4 };

using BlkXyz10 = EmbBlock<3, 10>;   // NeRF xyz, in_channels_xyz = 63 -> 16 slots
using BlkXyz5 = EmbBlock<3, 5>;     // NoF xyz, in_channels_xyz = 33  -> 10 slots
using BlkInd16 = EmbBlock<1, 16>;   // NoF ind, extra_feat_dim = 33   -> 10 slots
using BlkDir4 = EmbBlock<3, 4>;     // NeRF dir, 27                   -> 8 slots
using BlkInd2 = EmbBlock<1, 2>;     // NeRF ind, 5                    -> 2 slots

constexpr int kStepsNerfXyz = round4(BlkXyz10::SLOTS);                    // 16
constexpr int kStepsNofIn = round4(BlkXyz5::SLOTS + BlkInd16::SLOTS);     // 20
constexpr int kStepsDir = round4(BlkDir4::SLOTS);                         // 8
constexpr int kStepsInd = round4(BlkInd2::SLOTS);                         // 4
constexpr int kStepsExtraMax = kStepsDir;

// reference column (within the embedded input vector) of slot e of lane group g; -1 = zero pad.
// `xyz_cols` = the network's in_channels_xyz (NoF: the ind block starts there).
MF_HD int emb_feature(int kind, int g, int e, int xyz_cols) {
  switch (kind) {
    case kEmbNerfXyz:
      return e < BlkXyz10::SLOTS ? BlkXyz10::feature(g, e) : -1;
    case kEmbNofIn:
      if (e < BlkXyz5::SLOTS) {            // (a NoF narrower than 33 xyz columns: the block's upper features have no column --
        const int f = BlkXyz5::feature(g, e);      // the index block starts at xyz_cols)
        return f < xyz_cols ? f : -1;
      }
      if (e < BlkXyz5::SLOTS + BlkInd16::SLOTS) {
        const int f = BlkInd16::feature(g, e - BlkXyz5::SLOTS);
        return f < 0 ? -1 : xyz_cols + f;
      }
      return -1;
    case kEmbDir:
      return e < BlkDir4::SLOTS ? BlkDir4::feature(g, e) : -1;
    case kEmbInd:
      return e < BlkInd2::SLOTS ? BlkInd2::feature(g, e) : -1;
    default:
      return -1;
  }
}

// The same embeddings for the bf16 kernels' 32x32x16 tiles, where a sample column has TWO lanes (h = lane>>5):
// half h owns the pairs p = 2*pi + h; SLOTS per half, in k-steps of 8 slots per half (16 k per MFMA).
template <int C, int F>
struct EmbBlock2 {
  static constexpr int NPAIR = C * F;
  static constexpr int NALL = NPAIR + (C + 1) / 2;
  static constexpr int NPI = (NALL + 1) / 2;                // pairs per half
  static constexpr int SLOTS = 2 * NPI;
  MF_HD static int feature(int h, int e) {
    const int p = 2 * (e >> 1) + h, sc = e & 1;
    if (p < NPAIR) {
      const int f = p / C, c = p % C;
      return C + 2 * C * f + C * sc + c;
    }
    const int raw = 2 * (p - NPAIR) + sc;
    return (p < NALL && raw < C) ? raw : -1;
  }
};
using B2Xyz10 = EmbBlock2<3, 10>;   // 32 slots per half -> 4 k-steps
using B2Xyz5 = EmbBlock2<3, 5>;     // 18
using B2Dir4 = EmbBlock2<3, 4>;     // 14  -> 2 k-steps
using B2Ind2 = EmbBlock2<1, 2>;     // 4   -> 1 k-step
constexpr int kKsNerfXyz = (B2Xyz10::SLOTS + 7) / 8;                       // 4
// The bf16 NoF's matrix input is its xyz block only (3 k-steps): the image-index block (33 of the 66 input columns,
// models/rendering.py:73-75, models/nof.py:69-73) is constant along a ray, so  W[:, 33:66] * emb(ind) + b  of every
// layer that consumes the embedded input is a per-ray fp32 vector -- computed once per (ray, network, index value) by
// nof_raybias_kernel (mf_render_bf16.hip) from the fp32 columns kept behind the panels (kNofIndCols per row) and used
// as the accumulators' initial value.
constexpr int kKsNofXyz = (B2Xyz5::SLOTS + 7) / 8;                          // 3
constexpr int kNofIndCols = 36;                                             // 33 image-index columns + pad
constexpr int kKsDir = (B2Dir4::SLOTS + 7) / 8;                            // 2
constexpr int kKsInd = (B2Ind2::SLOTS + 7) / 8;                            // 1
constexpr int kKsExtraMax = kKsDir;

// reference column of slot e (0 .. 8*ksteps) of half h; -1 = zero pad
MF_HD int emb_feature2(int kind, int h, int e, int xyz_cols) {
  switch (kind) {
    case kEmbNerfXyz:
      return e < B2Xyz10::SLOTS ? B2Xyz10::feature(h, e) : -1;
    case kEmbNofIn: {    // xyz block only: the image-index block is a per-ray bias (kKsNofXyz)
      const int f = e < B2Xyz5::SLOTS ? B2Xyz5::feature(h, e) : -1;
      return f < xyz_cols ? f : -1;
    }
    case kEmbDir:
      return e < B2Dir4::SLOTS ? B2Dir4::feature(h, e) : -1;
    case kEmbInd:
      return e < B2Ind2::SLOTS ? B2Ind2::feature(h, e) : -1;
    default:
      return -1;
  }
}
// feature (within a 16-wide k-step) held by element e (0..7) of half h of a 32x32x16 operand: the order in which a
// finished 32x32 tile's accumulators sit in registers (row = (r&3) + 8*(r>>2) + 4*h, r = e or 8 + e)
MF_HD int hid_perm2(int h, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// Embedding parameters as the kernels see them (uniform, lives in SGPRs / kernarg).
struct EmbParams {
  float freq[16];
  float weight[16];   // 0 beyond the module's N_freqs (== the reference's zero padding)
};

// ------------------------------------------------------------------ packed layouts
// A packed network = [resident block (biases, VALU head weights), padded to 1 KiB]
//                    [panels in program order].
// A panel = 32 output rows of one layer = two 16-row tiles; its groups alternate between the
// two tiles: [tile0 k-quad 0][tile1 k-quad 0][tile0 k-quad 1] ...  (k-quad = 4 k-steps = 16 k).
struct NetLayout {
  int W, NK, NP;           // hidden width, k-tiles (W/16), panels per W-wide layer (W/32)
  int bf16;                // 1: the layout of the bf16 kernels (mf_bf16.hpp: 32x32x16 bf16 A fragments, one 32-row tile per panel)
  int n_trunk;             // trunk layers streamed through the generic loop (NeRF: D+1 incl. final)
  int emb_steps;           // k-steps of the embedded-input block
  int emb_split;           // bf16: every embedded k-step is TWO groups (hi = bf16(w), lo = bf16(w - hi)) -- the NoF
                           // (its output point feeds sin(512 x)); 0: plain bf16 like the hidden ranges -- the NeRF
  uint32_t emb_mask;       // trunk layers that consume the embedded input (layer 0 + skips)
  uint32_t relu_mask;      // trunk layers followed by ReLU
  int extra_steps;         // NeRF extra_encoding: k-steps of the extra block (0, 4, 8); -1 = no extra layer
  int64_t res_bytes;       // resident block size (multiple of 1 KiB)
  int64_t panel_bytes;     // all panels
  int64_t ind_bytes;       // bf16 NoF: fp32 image-index columns [embedded layer][kNofIndCols][row] behind the panels
  int n_emb_layers;        // layers that consume the embedded input (popcount of emb_mask)
  uint32_t hsplit_mask;    // MF_PREC_BF16X3: layers whose HIDDEN k-steps are (hi, lo) group pairs too (three products:
                           // activations and weights split) -- every layer of both networks; bit n_trunk = the NeRF's
                           // extra_encoding
  int max_groups;          // largest panel, in groups (x3: tiles of more than 32 groups stream as two panels, panel_cap)
  int half;                // 1: the split ranges are IEEE-half (hi, lo) pairs, weights 2^kNofHalfSW x, biases 2^(kNofHalfSA + kNofHalfSW) x
  int terms;               // bf16 terms of every SPLIT range of this network: 2 = (hi, lo), three products per k-step; 3 = (hi,
                           // mid, lo), six products down to 2^-24 -- the NoF under MF_PREC_BF16X3 (kNofTermsX3): its output point
                           // feeds sin(512 x) of the canonical encoding and needs fp32-class accuracy
  // resident block float offsets
  int off_bias_trunk;      // n_trunk * W
  int off_bias_extra;      // W/2
  int off_head_w;          // NeRF: sigma_w (W) ; NoF: head_w (n_head * W)
  int off_head_b;          // NeRF: sigma_b ; NoF: head_b
  int off_rgb_w;           // NeRF: 3 * W/2
  int off_rgb_b;
  int n_head;              // NoF: 9 | 3
  int head_tiles;          // fast bf16 mode (round 6).  NoF: the head panel's (hi, lo) terms are tile rows c and 16 + c of ONE group per
                           // k-step instead of group pairs.  NeRF: the sigma and rgb heads are matrix-pipe panels of the weight stream --
                           // ONE 32-row tile each whose rows c < n are bf16(w_c) and rows 8 + c are bf16(w_c - hi) (both land in
                           // lane half 0's accumulators: registers c and 4 + c), plain hidden k-steps: NK groups behind trunk layer
                           // D - 1 (sigma, in front of xyz_encoding_final), NK / 2 groups behind extra_encoding (rgb)
};

// fp32 -- batches of a panel: an embedded-input batch = one k-quad (16 k, 8 MFMAs); a hidden batch = one k-quad.
// Two 1 KiB groups per batch (one per 16-row tile of the 32-row panel).
// bf16 (mf_bf16.hpp) -- a panel is ONE 32-row tile; a group = 32 rows x 16 k of bf16 (one A fragment of
// v_mfma_f32_32x32x16_bf16); `emb_steps` / `extra_steps` count 16-slot k-steps of the embedded blocks, each stored
// as one group, or with `emb_split` as two (hi = bf16(w), lo = bf16(w - hi)); a W-wide hidden range is NK groups.
MF_HD int hidden_batches(const NetLayout& L) { return L.NK; }
MF_HD int trunk_batches(const NetLayout& L, int layer) {
  return (((L.emb_mask >> layer) & 1) ? L.emb_steps / 4 : 0) + (layer > 0 ? hidden_batches(L) : 0);
}
MF_HD int trunk_groups(const NetLayout& L, int layer) {
  if (L.bf16) return (((L.emb_mask >> layer) & 1) ? (L.emb_split ? L.terms : 1) * L.emb_steps : 0) +
                     (layer > 0 ? (((L.hsplit_mask >> layer) & 1) ? L.terms : 1) * L.NK : 0);
  return 2 * trunk_batches(L, layer);
}
// bf16 NoF: the 3|9-row head (nof.py:75-82) as one more panel behind the trunk: a 32-row tile (rows >= n_head zero)
// whose hidden k-steps are split (hi, lo) group pairs -- the head's weights keep 16 mantissa bits (x3: `terms` groups).
// (fast mode, round 6 -- NetLayout::head_tiles: ONE group per k-step, the (hi, lo) terms as ROWS c and 16 + c of the tile)
MF_HD int head_groups(const NetLayout& L) { return (L.head_tiles ? 1 : L.terms) * L.NK; }
MF_HD int nerf_sigma_groups(const NetLayout& L) { return L.head_tiles ? L.NK : 0; }
MF_HD int nerf_rgb_groups(const NetLayout& L) { return L.head_tiles ? L.NK / 2 : 0; }
MF_HD int extra_groups(const NetLayout& L) {
  if (L.bf16) return (((L.hsplit_mask >> L.n_trunk) & 1) ? L.terms : 1) * L.NK + (L.emb_split ? L.terms : 1) * L.extra_steps;
  return 2 * (hidden_batches(L) + L.extra_steps / 4);
}

// MF_PREC_BF16X3: a tile's groups stream as ONE panel up to 32 groups, as two halves beyond (the 3-slot LDS ring holds
// 32 KiB slots; the groups of a tile are contiguous in the packed buffer, so a panel boundary is only a barrier position)
MF_HD int panel_cap(int groups) { return groups > 32 ? (groups + 1) / 2 : groups; }

// bf16 terms of the NoF's operands under MF_PREC_BF16X3 (NetLayout::terms; host layout, packer and kernels agree on it)
// The NoF under MF_PREC_BF16X3.  Its output point feeds sin(512 x) of the canonical encoding, so its products need fp32-class
// operands.  Round 3: bf16 (hi, lo) pairs, three products: 16 significand bits, 1.5e-4 max-rel on the dense draw.  Round 4: bf16
// (hi, mid, lo) triples, SIX products: 24 bits, 1.4e-5 .. 3.1e-5, +12.6 % time on C3x.  Round 5: IEEE-half (hi, lo) pairs on
// v_mfma_f32_32x32x16_f16 (same rate as the bf16 instruction; the unit honours fp16 denormals, tools/proto/f16_denorm.hip):
// 22 significand bits in THREE products.  Both operands are carried at 2^5 x their value (exact: the accumulators are 2^10 x,
// biases packed 2^10 x, the epilogue un-scales) so that the lo terms of O(0.01 .. 1) values stay normal halves: priced in
// oracle/bf16_ref.py ("hsplit_g") at 1.2e-5 / 2.0e-5 max-rel on the two draws = the three-term split's; |x| < 2047 or the
// conversion overflows to inf (loudly: NaN rays).
constexpr int kNofTermsX3 = 2;
constexpr bool kNofHalfX3 = true;
constexpr int kNofHalfSA = 5, kNofHalfSW = 5;   // log2 of the activation / weight scale

// ------------------------------------------------------------------ device helpers
extern __shared__ __attribute__((aligned(16))) char smem[];

// LDS-DMA (global -> LDS without registers, 16 bytes per lane) through a buffer descriptor
// (buffer_load_dwordx4 ... offen lds): wave-uniform base address in SGPRs + ONE per-lane VGPR offset (lane * 16) + a
// scalar offset -- not global_load_lds with its 64-bit address per lane.  Measured on the bf16 trunk prototype (tools/proto/bf16_trunk2.hip, MI355X): 57 -> 74 %
// of the bf16 matrix peak from this change alone, for two reasons: (i) no v_lshl_add_u64 per piece and no 64 address
// pairs through the address path; (ii) global_load_lds is FLAT-encoded, and with a FLAT operation that may touch LDS
// pending, hipcc's waitcnt pass treats the LGKM counter as out of order and turns EVERY fragment wait into
// s_waitcnt lgkmcnt(0) -- draining the prefetched ds_read_b128s it was meant to overlap; with the MUBUF form it counts
// them (lgkmcnt(2..3)).  `base`, `soff`, `lds_off` must be wave-uniform.
MF_D void blds16(const char* base, uint32_t lane16, uint32_t soff, uint32_t lds_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, (int)lane16, (int)soff, 0, 0);
}
// the same with an instruction offset IMM (<= 4095; added to the global AND the LDS address): consecutive 1 KiB pieces of
// one wave share base, scalar offset and M0
template <int IMM>
MF_D void blds16_imm(const char* base, uint32_t lane16, uint32_t soff, uint32_t lds_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, (int)lane16, (int)soff, IMM, 0);
}
MF_D void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Race / hazard screen (-DMF_DBG_JITTER, tools/ab_lib.sh): a pseudo-random stall per wave in front of every panel barrier
// and embedding evaluation.  Results must stay bit-identical between runs (tools/stress_determinism.py).
#ifdef MF_DBG_JITTER
MF_D void jitter() {       // race screen: a pseudo-random stall per wave
  const unsigned t = (unsigned)__builtin_readcyclecounter();
  switch ((t >> 3) & 3u) {
    case 1: __builtin_amdgcn_s_sleep(3); break;
    case 2: __builtin_amdgcn_s_sleep(11); break;
    case 3: __builtin_amdgcn_s_sleep(40); break;
    default: break;
  }
}
#else
MF_D void jitter() {}
#endif

// sum over the four lane groups of a sample column (lanes j, j+16, j+32, j+48)
MF_D float xgroup_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// Wavefront product scan / sum on the DPP network (row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 and
// row_bcast:31): six VALU steps of a few cycles each instead of six ds_bpermute round trips (~100 cycles each) per
// scan or reduction -- the per-ray composite was 5.3 k cycles per 256-sample tile of the bf16 pass (4 % of the tile,
// during which no wave of the workgroup feeds the matrix pipe), tools/timeline.py.
template <int CTRL, int ROW_MASK>
MF_D float dpp_f(float identity, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
MF_D float wave_scan_mul_dpp(float v) {          // inclusive product scan over the 64 lanes
  v *= dpp_f<0x111, 0xf>(1.f, v);
  v *= dpp_f<0x112, 0xf>(1.f, v);
  v *= dpp_f<0x114, 0xf>(1.f, v);
  v *= dpp_f<0x118, 0xf>(1.f, v);
  v *= dpp_f<0x142, 0xa>(1.f, v);                // row_bcast:15 -> rows 1, 3
  v *= dpp_f<0x143, 0xc>(1.f, v);                // row_bcast:31 -> rows 2, 3
  return v;
}
MF_D float wave_shr1_dpp(float first, float v) { return dpp_f<0x138, 0xf>(first, v); }   // lane i <- lane i-1, lane 0 <- first
MF_D float wave_last(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
MF_D float wave_sum_dpp(float v) {               // sum over the 64 lanes, returned wave-uniform
  v += dpp_f<0x111, 0xf>(0.f, v);
  v += dpp_f<0x112, 0xf>(0.f, v);
  v += dpp_f<0x114, 0xf>(0.f, v);
  v += dpp_f<0x118, 0xf>(0.f, v);
  v += dpp_f<0x142, 0xa>(0.f, v);
  v += dpp_f<0x143, 0xc>(0.f, v);
  return wave_last(v);
}

struct LaneId {
  int lane, wave, j, g;
  MF_D LaneId() {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    j = lane & 15;
    g = lane >> 4;
  }
};

template <class T>
MF_D T sel4(int g, T a0, T a1, T a2, T a3) {
  const T lo = (g & 1) ? a1 : a0;
  const T hi = (g & 1) ? a3 : a2;
  return (g & 2) ? hi : lo;
}

// Phase timeline (-DMF_TIMELINE, tools/timeline.py): lane 0 of waves 0 and 4 of workgroup 0 store the shader
// clock at phase boundaries into the pass's `alphas` plane (which this build does not otherwise write).
struct Timeline {
#ifdef MF_TIMELINE
  float* buf; unsigned long long t0; int n; bool on;
  template <class Id> MF_D void start(float* b, const Id& id) {
    on = blockIdx.x == 0 && (id.wave == 0 || id.wave == 4);
    buf = b + (id.wave == 4 ? 512 : 0);
    n = 0;
    t0 = __builtin_amdgcn_s_memtime();
  }
  template <class Id> MF_D void stamp(int tag, const Id& id) {
    if (on && buf && n < 250) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (id.lane == 0) { buf[2 * n] = (float)tag; buf[2 * n + 1] = (float)(long long)(t - t0); }
      ++n;
    }
  }
#else
  template <class Id> MF_D void start(float*, const Id&) {}
  template <class Id> MF_D void stamp(int, const Id&) {}
#endif
};

// Weight-panel stream: a 3-slot LDS ring fed by LDS-DMA (blds16), running
// TWO panels ahead of the MFMAs.  While panel c is being multiplied, panel c+1 is already
// complete and visible (so its first fragments and bias can be pre-read during c's tail: no
// exposed LDS latency at a panel boundary) and panel c+2 is in flight.  One workgroup barrier
// per panel, placed behind the first MFMAs of the panel:
//   RAW: every wave waited vmcnt(0) for its own pieces of c+1 before arriving;
//   WAR: every wave has started c, hence finished reading c-1, whose slot c+2 overwrites.
struct Stream {
  Timeline tl;
  const char* gnext;      // global address of the panel two ahead of the one being computed
  uint32_t ring;          // LDS byte offset of slot 0
  uint32_t buf_bytes;     // bytes per slot
  uint32_t cur;           // slot (0..2) of the panel being computed
  int dbg;                // timing-ablation switches (MF_DEBUG_FLAGS; 0 in production)
  int keep2;              // KEEP2 instantiations only: VM stores this wave issued at the end of the panel (0, 2: the dump rows, 3: + the mask word)

  MF_D uint32_t slot_off(uint32_t k) const {
    uint32_t s = cur + k;
    s = s >= 3u ? s - 3u : s;
    return ring + s * buf_bytes;
  }
  MF_D void dma_to(uint32_t dst, int groups, const LaneId& id) {
    for (int grp = id.wave; grp < groups; grp += kWaves) blds16(gnext, id.lane * 16, grp * kGroupBytes, dst + grp * kGroupBytes);
    gnext += (size_t)groups * kGroupBytes;
  }
  // variant kept for A/B runs: only the "early" half of the workgroup (waves 4-7, one per SIMD)
  // issues the DMA (measured 1 % slower than all eight waves issuing, profiles/r01 notes)
  MF_D void dma_early_half(uint32_t dst, int groups, const LaneId& id) {
    if (id.wave >= kWaves / 2) {
      for (int grp = id.wave - kWaves / 2; grp < groups; grp += kWaves / 2)
        blds16(gnext, id.lane * 16, grp * kGroupBytes, dst + grp * kGroupBytes);
    }
    gnext += (size_t)groups * kGroupBytes;
  }
  // barrier of the panel + launch of the DMA for the panel two ahead.
  // KEEP2 (kernels that store two rows of a dump at the end of every panel: training forward,
  // backward chain): the wave's two youngest vector-memory operations are those stores; the VM counter
  // retires in issue order, so vmcnt(2) already guarantees the panel DMA issued before them has landed
  // and the stores stay in flight across the barrier instead of being drained at it.
  // `ph`: 0 = the barrier half only, 1 = the DMA half only, 2 = both.  The late half of the workgroup (waves 0-3, barrier
  // in the middle of their panel) issues its pieces kF32DmaDelay batches behind the barrier: straight behind it the
  // SIMD's other wave is issuing ITS pieces, and with both waves of a SIMD inside their 4-5 buffer_load ... lds at the
  // same time (60-100+ cycles of issue each) nobody feeds the matrix pipe.
  template <bool KEEP2 = false>
  MF_D void sync_and_dma(int groups, const char* jump, const LaneId& id, int ph = 2) {
    if (ph != 1) {
      jitter();
      if (KEEP2 && keep2 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (KEEP2 && keep2 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (!(dbg & 16)) wait_vm0();
      if (!(dbg & 1)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (ph != 0) {
      if (jump) gnext = jump;
      if (dbg & 32) dma_early_half(slot_off(2), groups, id);         // ablation: one issuing wave per SIMD
      else if (!(dbg & 2)) dma_to(slot_off(2), groups, id);
      else gnext += (size_t)groups * kGroupBytes;
    }
  }
  MF_D void advance() { cur = cur == 2u ? 0u : cur + 1u; }
  // cold start: the program's first two panels (same layer, `groups` each) into slots 0 and 1
  MF_D void start(const char* first, int groups, const LaneId& id) {
    cur = 0;
    gnext = first;
    dma_to(ring, groups, id);
    dma_to(ring + buf_bytes, groups, id);
    wait_vm0();
    __syncthreads();
  }
};

MF_D f32x4 lds_f4(uint32_t byte_off) { return *(const f32x4*)(smem + byte_off); }
MF_D float lds_f(uint32_t byte_off) { return *(const float*)(smem + byte_off); }

#define MF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Activation storage of one wave (16 samples): one f32x4 per 16-feature k-tile.
// Fragment prefetch distance in batches: a batch is 8 MFMAs (256 matrix cycles), one of them covers the LDS latency.
constexpr int kPD = kF32PD;

// What a panel needs before its first MFMAs, pre-read during the previous panel's tail: the
// fragment groups of its first PD batches (two tiles each) and the two tiles' bias in C/D order
// (reg r of lane group g <- bias[16*tile + 4g + r]).
template <int PD>
struct CarryT {
  f32x4 wE[PD], wO[PD], bE, bO;
  MF_D void load_bias(uint32_t bias_byte_off, int g) {
    bE = lds_f4(bias_byte_off + 16 * g);
    bO = lds_f4(bias_byte_off + 64 + 16 * g);
  }
  MF_D void load(uint32_t panel_lane_off, uint32_t bias_byte_off, int g) {   // cold start
#pragma unroll
    for (int i = 0; i < PD; ++i) {
      wE[i] = lds_f4(panel_lane_off + (2 * i) * kGroupBytes);
      wO[i] = lds_f4(panel_lane_off + (2 * i + 1) * kGroupBytes);
    }
    load_bias(bias_byte_off, g);
  }
};

// One panel = two 16-row output tiles (E: rows 0-15, O: rows 16-31) x 16 samples:
//   (E, O) = max(bias + W_panel * [emb ; hidden], lo)      (lo = 0: ReLU, -inf: linear)
// MODE: 1 = embedded input only, 2 = hidden only, 3 = both (skip layers, emb first).
// Per batch two ds_read_b128 (one group per tile) feed 8 fp32 MFMAs; the E
// and O chains alternate and share every B operand, so no MFMA directly follows its own
// predecessor and the ds_reads sit between independent MFMAs.  Groups are fetched PD batches
// ahead through a register ring that runs on into the NEXT panel's slot, so a panel boundary
// exposes no LDS latency.  `hook` is the panel's workgroup barrier (+ DMA of the panel two
// ahead).  It sits behind the FIRST batch's leading MFMA pair for the early half of the
// workgroup (waves 4-7) and in the MIDDLE of the panel for the late half (waves 0-3): since all
// eight waves meet at that barrier, the two waves that share a SIMD run half a panel out of
// phase, so one of them is always in MFMA-dense code while the other crosses a panel boundary
// (epilogue, branches, DMA issue).
template <int MODE, int NK, int EMB, class Hook>
MF_D void out_pair(CarryT<kPD>& carry, const f32x4 (&hid)[NK],
                   const float (&emb)[EMB], uint32_t panel_lane_off, uint32_t next_panel_lane_off,
                   uint32_t next_bias_off, int g, bool late, Hook&& hook, float lo, f32x4& outE, f32x4& outO,
                   bool late_prio = false) {
  // fp32: one batch (8 MFMAs) of prefetch is enough; this hand-shaped form of the loop below (two
  // named fragment registers instead of the ring) is what hipcc allocates best (12 % faster).
  constexpr int QE = (MODE & 1) ? EMB / 4 : 0;
  constexpr int QH = (MODE & 2) ? NK : 0;
  constexpr int Q = QE + QH;
  auto bop = [&](int q, int r) -> float {
    if (q < QE) return emb[4 * q + r];
    else return hid[q - QE][r];
  };
  constexpr int PDF = kPD;                 // 1 or 2 batches of fragment prefetch
  f32x4 E = carry.bE, O = carry.bO;
  f32x4 wE = carry.wE[0], wO = carry.wO[0];
  f32x4 xE = carry.wE[PDF - 1], xO = carry.wO[PDF - 1];   // second pipeline stage (PDF == 2)
  if (late_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    f32x4 nE = wE, nO = wO;
    E = MF_MFMA(wE[0], bop(q, 0), E);
    O = MF_MFMA(wO[0], bop(q, 0), O);
    __builtin_amdgcn_sched_barrier(0);
    const int nq = q + PDF;
    if (nq < Q) {
      nE = lds_f4(panel_lane_off + (2 * nq) * kGroupBytes);
      nO = lds_f4(panel_lane_off + (2 * nq + 1) * kGroupBytes);
    }
    {
      constexpr int QD = (Q / 2 + kF32DmaDelay < Q) ? Q / 2 + kF32DmaDelay : Q - 1;   // the late half's DMA batch
      if (q == 0 && !late) hook(2);
      if (QD == Q / 2) { if (q == Q / 2 && late) hook(2); }
      else {
        if (q == Q / 2 && late) hook(0);
        if (q == QD && late) hook(1);
      }
    }
    if (nq >= Q) {                                     // runs on into the next panel (after the barrier)
      nE = lds_f4(next_panel_lane_off + (2 * (nq - Q)) * kGroupBytes);
      nO = lds_f4(next_panel_lane_off + (2 * (nq - Q) + 1) * kGroupBytes);
    }
    if (q + 1 >= Q) carry.load_bias(next_bias_off, g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      E = MF_MFMA(wE[r], bop(q, r), E);
      O = MF_MFMA(wO[r], bop(q, r), O);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PDF == 2) {
      wE = xE; wO = xO;
      xE = nE; xO = nO;
    } else {
      wE = nE; wO = nO;
    }
  }
  carry.wE[0] = wE; carry.wO[0] = wO;
  if constexpr (PDF == 2) { carry.wE[1] = xE; carry.wO[1] = xO; }
  if (late_prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    outE[i] = fmaxf(E[i], lo);
    outO[i] = fmaxf(O[i], lo);
  }
}

// Uniform per-network state handed to the device code (kernarg -> SGPRs).
struct NetDev {
  NetLayout L;
  const char* packed;      // global base of the packed buffer
  uint32_t res_lds;        // LDS byte offset where its resident block lives
};

// The layer that follows in program order (possibly the first layer of another network).
struct NextLayer {
  int groups;              // panel size of that layer
  const char* jump;        // its first panel's address if the program leaves the contiguous order, else null
  uint32_t bias_off;       // LDS byte offset of its bias vector
};

// One trunk layer: act <- relu?(W_l [emb ; act] + b_l), NK/2 panels of two tiles.
// `dump_row` (training forward only, DUMP instantiations): this lane's sample row of the activation
// dump, already offset to this layer; the layer's post-activation outputs are stored there in
// natural feature order (the dW GEMMs of the backward read them).  nullptr: nothing stored.
// `mask_row` (ABI v13; layout of ABI v15): this lane's sample row of the ReLU bit mask, offset to this layer (8 words for a 256-wide
// layer, 4 for a 128-wide one).  A lane's eight outputs of panel t make one byte -- bit r = [output 32 t + 4 g + r > 0], bit
// 4 + r = [output 32 t + 16 + 4 g + r > 0] -- and the bytes of FOUR consecutive panels make one word: word 4 (t / 4) + g, byte
// t % 4.  One 4-byte store per lane and four panels (round 3 stored a byte per panel at byte 4 t + g: four times the store
// instructions, +0.44 ms per stage-1 forward launch); no cross-lane traffic in the forward; what the three-product backward
// chains read instead of the activations.  relu_mask_word / relu_mask_shift: where output f (0..31) of panel t sits.
// x > 0 as a bit in TWO instructions per element with the shift-or that places it: the float's bits as a signed integer clamped
// to [0, 1] (v_med3_i32: negative numbers, -0 and +0 give 0, every positive pattern >= 1 gives 1) -- the compare + select form
// took three (round 5: the bit words cost the fp32 training forward 0.16 ms of its 0.39 ms dump overhead per fine pass).
MF_D unsigned relu_bit(float x) {
  unsigned b;                 // (as C the clamp is canonicalised back into compare + select)
  asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(x));
  return b;
}
MF_D unsigned relu_mask_byte(const f32x4& E, const f32x4& O) {
  unsigned w = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) w |= relu_bit(E[r]) << r | relu_bit(O[r]) << (4 + r);
  return w;
}
MF_HD int relu_mask_word(int t, int f) { return 4 * (t >> 2) + (f < 16 ? f >> 2 : (f - 16) >> 2); }      // word of the layer's row
MF_HD int relu_mask_shift(int t, int f) { return 8 * (t & 3) + (f < 16 ? 0 : 4) + (f & 3); }               // bit within that word
// the x3 backward chains' side: tile t (the forward's panel t) as seen by lane half h of a 32-row tile -- rows 8 q + 4 h + i are
// outputs of lane groups g = h (q = 0, 2) and g = 2 + h (q = 1, 3): their two bytes as (byte of g = h) | (byte of g = 2 + h) << 8
MF_D unsigned relu_mask_pair(const unsigned* row, int t, int h) {
  const unsigned a = row[4 * (t >> 2) + h], b = row[4 * (t >> 2) + 2 + h];
  return ((a >> (8 * (t & 3))) & 0xffu) | (((b >> (8 * (t & 3))) & 0xffu) << 8);
}
// the writer's side: accumulate panel t's byte, store the word behind every fourth panel.  Returns: a store was issued.
MF_D bool relu_mask_put(unsigned& macc, unsigned* mask_row, bool ok, int t, int g, const f32x4& E, const f32x4& O) {
  macc = ((t & 3) == 0 ? 0u : macc) | relu_mask_byte(E, O) << (8 * (t & 3));
  if ((t & 3) != 3) return false;
  if (ok) mask_row[4 * (t >> 2) + g] = macc;
  return true;
}
template <int NK, int EMB, bool DUMP = false>
MF_D void trunk_layer(const NetDev& net, int layer, f32x4 (&act)[NK],
                      const float (&emb)[EMB], Stream& st, CarryT<kPD>& carry, const LaneId& id,
                      const NextLayer& nxt, float* dump_row = nullptr, unsigned* mask_row = nullptr) {
  constexpr int NP = NK / 2;
  const int has_emb = (net.L.emb_mask >> layer) & 1;
  const int mode = (has_emb ? 1 : 0) | (layer > 0 ? 2 : 0);
  const int groups = trunk_groups(net.L, layer);
  const float lo = ((net.L.relu_mask >> layer) & 1) ? 0.f : -__builtin_inff();
  const uint32_t bias_off = net.res_lds + (net.L.off_bias_trunk + layer * net.L.W) * 4;
  f32x4 out[NK];
  static_assert(NP % 4 == 0, "mask words collect four panels");
  unsigned macc = 0;
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    const uint32_t p = st.slot_off(0) + id.lane * 16;
    const uint32_t pn = st.slot_off(1) + id.lane * 16;
    const uint32_t nb = (t + 1 < NP) ? bias_off + 32 * (t + 1) * 4 : nxt.bias_off;
    // panel two ahead: same layer while t+2 < NP, else panel (t+2-NP) of the next layer
    auto hook = [&](int ph) { st.template sync_and_dma<DUMP>(t + 2 < NP ? groups : nxt.groups, t == NP - 2 ? nxt.jump : nullptr, id, ph); };
    const bool late = id.wave < kWaves / 2 && !(st.dbg & 64);
    f32x4 E, O;
    const bool prio = (st.dbg & 256) != 0;
    if (mode == 2) out_pair<2, NK, EMB>(carry, act, emb, p, pn, nb, id.g, late, hook, lo, E, O, prio);
    else if (mode == 3) out_pair<3, NK, EMB>(carry, act, emb, p, pn, nb, id.g, late, hook, lo, E, O, prio);
    else out_pair<1, NK, EMB>(carry, act, emb, p, pn, nb, id.g, late, hook, lo, E, O, prio);
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
#pragma unroll
  for (int t = 0; t < NK; ++t) act[t] = out[t];
}

// VALU head: NOUT dot products of the lane's quarter of the hidden vector with natural-order
// weight rows in LDS (broadcast ds_read_b128), summed across the four lane groups.  Every lane
// of a sample column ends up with the full sums.
template <int NK, int NOUT>
MF_D void valu_head(const f32x4 (&act)[NK], uint32_t w_byte_off, int /*row_floats == 16 NK at every call site*/,
                    uint32_t b_byte_off, int g, float (&out)[NOUT]) {
  // rows are 16 NK floats (compile time): every read is ONE per-lane base (w_byte_off + 16 g) plus an immediate; with a
  // runtime row length hipcc hoisted one address register per (row, tile) to the kernel entry and spilled them
  constexpr int row_floats = 16 * NK;
  const uint32_t wl = w_byte_off + 16 * g;
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NK; ++t) {
      const f32x4 w = lds_f4(wl + (o * row_floats + 16 * t) * 4);
      s0 = __builtin_fmaf(w[0], act[t][0], s0);
      s1 = __builtin_fmaf(w[1], act[t][1], s1);
      s0 = __builtin_fmaf(w[2], act[t][2], s0);
      s1 = __builtin_fmaf(w[3], act[t][3], s1);
    }
    out[o] = xgroup_sum(s0 + s1) + lds_f(b_byte_off + o * 4);
  }
}

// ------------------------------------------------------------------ embedding in registers
// dst[0..SLOTS) of block (C,F) for this lane group.  arg = freq*x is rounded to fp32 before
// sin/cos exactly as `func(freq*x)` in embedding.py:45; weight multiplies the result.
template <int C, int F>
MF_D void emb_eval(float* dst, const float (&v)[C], const EmbParams& ep, int g) {
  using B = EmbBlock<C, F>;
#pragma unroll
  for (int pi = 0; pi < B::NPI; ++pi) {
    // the four lane groups' pairs p = 4*pi + {0,1,2,3}: real pair -> (freq, comp); else raw pseudo-pair
    float xs[4], frs[4], ws[4], r0[4], r1[4];
    bool rl[4];
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = 4 * pi + k;
      const bool real = p < B::NPAIR;
      const int f = real ? p / C : 0, c = real ? p % C : 0;
      rl[k] = real;
      xs[k] = v[c];
      frs[k] = ep.freq[f];
      ws[k] = real ? ep.weight[f] : 0.f;
      const int raw = real ? 0 : 2 * (p - B::NPAIR);
      r0[k] = (!real && raw < C) ? v[raw < C ? raw : 0] : 0.f;
      r1[k] = (!real && raw + 1 < C) ? v[raw + 1 < C ? raw + 1 : 0] : 0.f;
      if (real) any_live |= ep.weight[f] != 0.f;       // wave-uniform (kernarg)
    }
    const float x = sel4(g, xs[0], xs[1], xs[2], xs[3]);
    const float fr = sel4(g, frs[0], frs[1], frs[2], frs[3]);
    const float w = sel4(g, ws[0], ws[1], ws[2], ws[3]);
    const bool real = sel4(g, rl[0], rl[1], rl[2], rl[3]);
    float s = 0.f, c = 0.f;
    if (any_live) sincosf(fr * x, &s, &c);   // skipped when every frequency of this slot is muted
    dst[2 * pi] = real ? w * s : sel4(g, r0[0], r0[1], r0[2], r0[3]);
    dst[2 * pi + 1] = real ? w * c : sel4(g, r1[0], r1[1], r1[2], r1[3]);
  }
}

// The same with the embedding's parameters in LDS (par_off: freq[16] then weight[16], floats) instead of the kernarg:
// four EmbParams are 128 scalars, which the fused pass cannot afford in SGPRs (they were spilled to VGPR lanes).
template <int C, int F>
MF_D void emb_eval_lds(float* dst, const float (&v)[C], uint32_t par_off, int g) {
  using B = EmbBlock<C, F>;
#pragma unroll
  for (int pi = 0; pi < B::NPI; ++pi) {
    float xs[4], r0[4], r1[4];
    int fs[4];
    bool rl[4];
    bool any_real = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = 4 * pi + k;
      const bool real = p < B::NPAIR;
      const int f = real ? p / C : 0, c = real ? p % C : 0;
      rl[k] = real;
      any_real |= real;
      xs[k] = v[c];
      fs[k] = f;
      const int raw = real ? 0 : 2 * (p - B::NPAIR);
      r0[k] = (!real && raw < C) ? v[raw < C ? raw : 0] : 0.f;
      r1[k] = (!real && raw + 1 < C) ? v[raw + 1 < C ? raw + 1 : 0] : 0.f;
    }
    if (!any_real) {                                     // (compile time) raw components only
      dst[2 * pi] = sel4(g, r0[0], r0[1], r0[2], r0[3]);
      dst[2 * pi + 1] = sel4(g, r1[0], r1[1], r1[2], r1[3]);
      continue;
    }
    const float x = sel4(g, xs[0], xs[1], xs[2], xs[3]);
    const int f = sel4(g, fs[0], fs[1], fs[2], fs[3]);
    const bool real = sel4(g, rl[0], rl[1], rl[2], rl[3]);
    const float fr = lds_f(par_off + 4 * f);
    const float w = real ? lds_f(par_off + 64 + 4 * f) : 0.f;
    bool live = false;                                   // wave-uniform: any of the four groups' frequencies un-muted
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (rl[k]) live |= lds_f(par_off + 64 + 4 * fs[k]) != 0.f;
    float s = 0.f, c = 0.f;
    if (__builtin_amdgcn_readfirstlane((int)live)) sincosf(fr * x, &s, &c);
    dst[2 * pi] = real ? w * s : sel4(g, r0[0], r0[1], r0[2], r0[3]);
    dst[2 * pi + 1] = real ? w * c : sel4(g, r1[0], r1[1], r1[2], r1[3]);
  }
}

}  // namespace mf
