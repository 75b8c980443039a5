// mf_bf16.hpp -- bf16 mode of the fused MLP core (BASELINE configs C3-C5) on v_mfma_f32_32x32x16_bf16.
//
// Same idea as mf_core.hpp -- the MLPs of models/nerf.py:78-102 and models/nof.py:69-82 evaluated TRANSPOSED,
// H_out^T = W * H_in^T, with the activations living in the register file from the embedding to the heads -- but
// shaped for the bf16 matrix pipe, which is 16x faster than the exact-fp32 one and therefore needs 4x more
// arithmetic per LDS byte, per barrier and per instruction issued than the fp32 tiling gives:
//   * A operand = 32 weight rows x 16 k (one ds_read_b128 per lane = one 1 KiB "group" per wave),
//     B operand = 16 k x 32 SAMPLES (a wave owns 32 samples, one per lane&31; the two lane halves h = lane>>5
//     hold the two k-octets of every step), C/D = 32 features x 32 samples, 16 fp32 per lane, 32 matrix cycles:
//     16 384 MACs per fragment read (8 192 in a 16x16x32 tiling with 16 samples per wave);
//   * the C/D layout (row = (r&3) + 8*(r>>2) + 4h, col = lane&31) is a B-operand layout for two 16-k steps of the
//     next layer if the k order inside a step is permuted to  slot 8h + e  ->  feature (e&3) + 8*(e>>2) + 4h.
//     The weights are packed in that order (mf_pack.hip), so a finished tile -- ReLU, RNE to bf16, pairs packed
//     into 8 dwords -- IS the next layer's operand: no shuffle, no LDS round trip;
//   * a panel = ONE 32-row tile x the layer's whole k range (16 groups for a 256-wide hidden range): 16 MFMAs of
//     32 cycles per wave between two barriers (the 16x16x32 tiling had 16 MFMAs of 16 cycles);
//   * a workgroup is 8 waves = 256 samples per pass over the weight stream (two waves per SIMD, <= 256 VGPRs).
// The NoF's embedded-input k ranges and its head use a two-term bf16 split of inputs and weights, x = hi + lo, three
// products hi*hi + hi*lo + lo*hi (16 mantissa bits): its output POINT feeds sin(512 x) of the canonical NeRF's
// encoding.  The NeRF's own encodings are plain bf16 operands like every hidden range (SPLIT = false): measured on the
// C2 shape the split bought 0.1 dB (61.15 vs 61.00 dB against the fp32 oracle -- the rounding of the 63 O(1) inputs is
// one more layer's worth of activation rounding) for 10 % more matrix instructions, and the pass is throttled by its own activity
// (DESIGN.md).  Accumulation, biases, the sigma / rgb / NoF heads and the composite stay fp32.
// The second half of the file is MF_PREC_BF16X3: the same tile loop with EVERY operand of both networks as such a split
// (three products per k-step, heads on the fp32 accumulators) -- fp32-class results at three bf16 matrix instructions per
// product -- in kernels of 4 waves, one per SIMD with the whole register file (StreamT<4>, layer_x, nerf_eval_x3, nof_eval_x3);
// csrc/mf_backward_bf16.hip runs the backward's input-gradient chain on it.
#pragma once
#include <type_traits>
#include "mf_nets.hpp"

namespace mf {
namespace bf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaveSamples = 32;     // samples per wave
constexpr int kTile = 256;           // samples per workgroup tile (8 waves)
constexpr int PD = 3;                // A-fragment prefetch distance in groups (2 / 4 / 5 measured: +-1 %)

#define MF_MFMA32(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

struct Lane {
  int lane, wave, j, h;
  MF_D Lane() {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    j = lane & 31;
    h = lane >> 5;
  }
};

MF_D u32x4 lds_u4(uint32_t byte_off) { return *(const u32x4*)(smem + byte_off); }

// Weight-panel stream (see mf_core.hpp Stream): 3-slot LDS ring fed by LDS-DMA, two panels ahead of the MFMAs,
// one workgroup barrier per panel.  NW = waves of the workgroup (8: the fast mode, two per SIMD; 4: MF_PREC_BF16X3, one
// per SIMD with the whole register file).
template <int NW>
struct StreamT {
  static constexpr int kNW = NW;
  static constexpr int kPieces = 32 / NW;      // 1 KiB pieces per wave of the largest panel (32 groups)
  Timeline tl;
  const char* gnext;      // global address of the panel two ahead of the one being computed
  uint32_t off0, off1, off2;   // LDS byte offsets of the slots holding the current panel, the next one, the one after
                               // (rotated by advance(): no modulo / multiply per panel)

  MF_D uint32_t slot_off(uint32_t k) const { return k == 0 ? off0 : (k == 1 ? off1 : off2); }
  // The panel hook, in two parts.  sync(): barrier of the panel, then the DMA source / destination of the panel two
  // ahead are latched.  piece(k): this wave's k-th 1 KiB piece of that panel (k = 0..kPieces-1: a panel is at most 32
  // groups); the pieces are issued one per MFMA gap behind the barrier instead of as a burst.
  const char* dsrc; uint32_t ddst; uint32_t pmask;
  // `sg` (sync's last argument): the groups of the panel two ahead when the CALLER knows them at compile time (the same
  // layer's next-but-one tile), -1 otherwise.  With sg a multiple of NW every wave issues exactly sg / NW pieces: after
  // inlining, `stat` / `nstat` are constants, piece(k) is either an unconditional LDS-DMA or nothing, and the per-piece
  // mask test + scalar branch (s_and_b64 vcc / s_cbranch_vccnz in front of every buffer_load ... lds, dead pieces included)
  // is gone -- what remains runtime are the two panels at a layer boundary.
  bool stat; int nstat;
  // KEEP: VM operations this wave has issued behind its last piece of the panel that must have landed (dump stores of
  // the training forward: the VM counter retires in order, so "all but the KEEP youngest" still covers every piece
  // without waiting for those stores' round trip).  A LOWER bound is safe; 0 waits for everything.
  // `keep_ok` (wave-uniform): the KEEP operations were really issued by this wave (a wave whose lanes all sit past the last
  // sample issues no dump stores: its youngest operations ARE pieces).
  template <int KEEP = 0>
  MF_D void sync(int groups, const char* jump, const Lane& id, bool keep_ok = true, int sg = -1) {
    // (MF_BF_ABL_*: timing-ablation builds only, tools/ab_lib.sh; results are garbage there)
    jitter();
    if constexpr (KEEP == 0) wait_vm0();     // this wave's pieces of the NEXT panel have landed
    else {
      if (keep_ok) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
      else wait_vm0();
    }
    __builtin_amdgcn_s_barrier();            // RAW: everybody's have; WAR: everybody left the previous panel
    asm volatile("" ::: "memory");
    if (jump) gnext = jump;
    // this wave's pieces: a BLOCK of consecutive groups (wave w: groups w per .. w per + per - 1), so that they share one
    // base / M0 and differ in the instruction offset only
    stat = sg >= 0 && sg % NW == 0;
    nstat = stat ? sg / NW : 0;
    const int per = stat ? nstat : (groups + NW - 1) / NW;
    const int first = id.wave * per;
    dsrc = gnext + first * kGroupBytes;
    ddst = off2 + first * kGroupBytes;
    const int mine = groups - first < per ? groups - first : per;
    pmask = stat ? 0u : (1u << (mine < 0 ? 0 : mine)) - 1u;
    gnext += (size_t)(stat ? sg : groups) * kGroupBytes;
  }
  MF_D void piece(int k, const Lane& id) {
    if (stat ? k < nstat : (bool)((pmask >> k) & 1u)) {
      const uint32_t hi = (uint32_t)(k >> 2) * (4 * kGroupBytes);
      switch (k & 3) {
        case 0: blds16_imm<0>(dsrc, id.lane * 16, hi, ddst + hi); break;
        case 1: blds16_imm<1024>(dsrc, id.lane * 16, hi, ddst + hi); break;
        case 2: blds16_imm<2048>(dsrc, id.lane * 16, hi, ddst + hi); break;
        default: blds16_imm<3072>(dsrc, id.lane * 16, hi, ddst + hi); break;
      }
    }
  }
  MF_D void advance() {
    const uint32_t t = off0;
    off0 = off1; off1 = off2; off2 = t;
  }
  MF_D void start(const char* first, int groups, int groups2, uint32_t ring, uint32_t buf_bytes, const Lane& id) {
    off0 = ring; off1 = ring + buf_bytes; off2 = ring + 2 * buf_bytes;
    for (int grp = id.wave; grp < groups; grp += NW) blds16(first, id.lane * 16, grp * kGroupBytes, off0 + grp * kGroupBytes);
    for (int grp = id.wave; grp < groups2; grp += NW) blds16(first, id.lane * 16, (groups + grp) * kGroupBytes, off1 + grp * kGroupBytes);
    gnext = first + (size_t)(groups + groups2) * kGroupBytes;
    wait_vm0();
    __syncthreads();
  }
};
using Stream = StreamT<kWaves>;
using Stream4 = StreamT<4>;

// Uniform per-network state of the bf16 kernels: SIX scalars.  (The fp32 kernels hand a whole NetLayout -- ~25 derived
// offsets per network -- through SGPRs; with three networks that alone overflowed the scalar file here and every
// access became a v_readlane from a spill register.)  Everything else follows from the compile-time shape (KH = W/16,
// EKS = embedded k-steps) and D:
//   resident block (floats): NeRF  [bias_trunk (D+1) W | bias_extra W/2 | sigma_w W | sigma_b 4 | rgb_w 3 W/2 | rgb_b 4]
//                            NoF   [bias_trunk D W | head_w n_head W | head_b 32]
struct Net {
  const char* packed;      // global base of the packed buffer
  uint32_t res_lds;        // LDS byte offset of its resident block
  uint32_t res_bytes;      // size of the resident block (panels start there)
  int D;                   // NeRF: trunk layers WITHOUT xyz_encoding_final; NoF: trunk layers
  uint32_t emb_mask;       // trunk layers that consume the embedded input (layer 0 + skips)
  int aux;                 // NeRF: k-steps of the extra block (0, 1, 2); NoF: head rows (3 | 9)
};
// HS (MF_PREC_BF16X3): the layers' HIDDEN k-steps are split too -- 0: none, 1: every layer.  T: bf16 terms (= groups per
// k-step) of a split range: 2 = (hi, lo), 3 = (hi, mid, lo) -- NetLayout::terms.
template <int KH, int EKS, bool SPLIT, int HS = 0, int T = 2>
MF_D int tgroups(const Net& n, int layer) {
  const bool hs = HS == 1;
  return (((n.emb_mask >> layer) & 1) ? (SPLIT ? T : 1) * EKS : 0) + (layer > 0 ? (hs ? T : 1) * KH : 0);
}

// What follows the layer being computed in the panel program: its first panel (`groups`, at `jump` if the program
// leaves the contiguous order there) and its second one (`groups2` / `jump2`: differ from the first when that layer
// is a single panel, like the NoF head).
struct Next {
  int groups; const char* jump;
  int groups2; const char* jump2;
};
// Tiles per panel (the fast mode): several 32-row tiles of a layer stream as ONE panel -- one barrier, one set of LDS-DMA pieces
// per wave -- where the ring slot holds them: template triples <TPP0, TPPH, TPPS> = layer 0 (embedded input only: 4-6 groups per
// tile), hidden-only layers, skip layers.
// A `Next` names the first panel of what follows and THE PANEL BEHIND IT, wherever that lives: a layer whose tiles all stream
// as one panel (KH / 2 == TPP) is followed, two panels on, by the first panel of the layer after it (`after` behind the last
// trunk layer: the NoF's head panel).
template <int KH, int EKS, bool SPLIT, int TPP = 1, int T = 2, int TPPH = 0, int TPPS = 0>
MF_D Next first_of(const Net& n) {       // layer 0: embedded input only, TPP tiles per panel
  const int g = TPP * (SPLIT ? T : 1) * EKS;
  if constexpr (TPPH > 0 && KH / 2 == TPP)      // (fast mode) a single-panel layer 0: behind it comes layer 1's first panel
    return Next{g, n.packed + n.res_bytes, (((n.emb_mask >> 1) & 1) ? TPPS : TPPH) * tgroups<KH, EKS, SPLIT, 0, T>(n, 1), nullptr};
  return Next{g, n.packed + n.res_bytes, g, nullptr};
}
template <int KH, int EKS, bool SPLIT, int HS = 0, int TPPH = 1, int TPPS = 1>
MF_D Next next_trunk_bf(const Net& n, int layer) {
  const int g = (((n.emb_mask >> layer) & 1) ? TPPS : TPPH) * tgroups<KH, EKS, SPLIT, HS>(n, layer);
  return Next{g, nullptr, g, nullptr};
}
// the same where layers may be single panels (the fast mode's NoF): trunk layer `layer` (>= 1) of D, `after` = what follows the trunk
template <int KH, int EKS, bool SPLIT, int TPPH, int TPPS>
MF_D Next next_trunk_np(const Net& n, int layer, int D, const Next& after) {
  const int tpp = ((n.emb_mask >> layer) & 1) ? TPPS : TPPH;
  const int g = tpp * tgroups<KH, EKS, SPLIT>(n, layer);
  if ((KH / 2) / tpp >= 2) return Next{g, nullptr, g, nullptr};
  if (layer + 1 < D) return Next{g, nullptr, (((n.emb_mask >> (layer + 1)) & 1) ? TPPS : TPPH) * tgroups<KH, EKS, SPLIT>(n, layer + 1), nullptr};
  return Next{g, nullptr, after.groups, after.jump};
}

template <int N>
struct CarryT {           // the first N fragments of the panel that follows, pre-read during the current one's tail
  static constexpr int kPD = N;
  u32x4 w[N];
  MF_D void load(uint32_t panel_lane_off) {
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = lds_u4(panel_lane_off + i * kGroupBytes);
  }
};
using Carry = CarryT<PD>;
constexpr int PDX = 3;               // the same distance in the MF_PREC_BF16X3 kernels (one wave per SIMD; 5 measured: 0 %)
using CarryX = CarryT<PDX>;

MF_D float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
MF_D float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// two floats -> one dword of two bf16 (RNE), ONE v_cvt_pk_bf16_f32
MF_D unsigned pack_bf16x2(float lo, float hi) {
  f32x2 v;
  v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// IEEE half, the NoF's operands under MF_PREC_BF16X3 (kNofHalfX3, mf_core.hpp): two floats -> one dword of two halves (RNE;
// |x| > 65504 -> inf), and back
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
MF_D unsigned pack_f16x2(float lo, float hi) {
  f32x2 v;
  v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
MF_D float hflo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
MF_D float hfhi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
// the matrix instruction of a tile: HF = v_mfma_f32_32x32x16_f16 (same shape, same rate, fp16 denormals honoured)
template <bool HF>
MF_D f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (HF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr float kHalfActScale = (float)(1 << kNofHalfSA), kHalfInvW = 1.f / (float)(1 << kNofHalfSW),
                kHalfInvAcc = 1.f / (float)(1 << (kNofHalfSA + kNofHalfSW));

// max(x, floor) on both bf16 halves of a dword through their int16 order: v_pk_max_i16.  floor = 0 is ReLU (every
// negative bf16, -0 included, has the sign bit set, i.e. is a negative int16); floor = 0x80008000 (int16 min) passes
// the value through.
MF_D unsigned pk_floor_bf16(unsigned x, unsigned floor) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), __builtin_bit_cast(i16x2, floor)));
}

// One output tile (32 features x the wave's 32 samples):
//     out = max(bias + W_tile * [emb ; hidden], floor)     EMB_FIRST (trunk layers: embedded input in front)
//     out = max(bias + W_tile * [hidden ; emb], floor)     !EMB_FIRST (NeRF extra_encoding)
// NGE = embedded 16-slot k-steps (SPLIT: each is the groups hi, lo and the MFMAs Whi*xhi, Whi*xlo, Wlo*xhi; else one
// group, one MFMA), KHID = hidden k-steps.
// The A fragments are fetched PD groups ahead through a register ring that runs on into the NEXT panel's slot.
// `hook` = the panel's barrier + DMA of the panel two ahead, behind the first group; its pieces go into the MFMA
// gaps that follow.
//
// Round 5 -- the tile as a PIPELINE STAGE.  Measured with the phase timeline (profiles/README.md, round 5): a 256-wide layer
// took 10.4 k cycles for 8 x 1 024 of matrix work, ~280 cycles per tile in which neither wave of a SIMD fed the pipe, and the
// ISA says what they were -- at every tile boundary a wave (i) waited for its last MFMA to retire (`s_nop 10`), (ii) ran the 16
// VALU of the epilogue (8 packed converts + 8 packed max), (iii) issued the four ds_reads that start the next tile's
// accumulators as its bias and waited `lgkmcnt(0)` for them, all in a row, and its SIMD partner, released by the same
// barrier, did the same at the same time.  Now:
//   * `acc` arrives INITIALISED (the bias, read into a second accumulator set during the previous tile -- `gap(NG - 2, NG)` --
//     while that tile's MFMAs still run), so the first MFMA of a tile waits for nothing;
//   * the epilogue of the PREVIOUS tile (`pend`) runs in this tile's MFMA gaps, one convert + max pair per gap (`gap(gi, NG)`,
//     gi >= 1: the caller's schedule), its accumulators retired long before;
// a tile boundary is then the panel barrier alone.  Costs 16 more live registers (two accumulator sets in flight).
// RB (the NoF's embedded-input layers, point queries): the accumulators start from `rb` = this lane's 16 rows of the per-ray
// vector b + W[:, ind columns] emb(ind) (nof_raybias_kernel; C/D order: rb[4q + i] = row 8q + 4h + i) instead of the LDS bias.
MF_D f32x16 bias_acc(uint32_t bias_off, int h) {            // C/D order: reg 4q + i <- bias[8q + 4h + i]
  const f32x4 b0 = lds_f4(bias_off + (0 + 4 * h) * 4), b1 = lds_f4(bias_off + (8 + 4 * h) * 4);
  const f32x4 b2 = lds_f4(bias_off + (16 + 4 * h) * 4), b3 = lds_f4(bias_off + (24 + 4 * h) * 4);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i] = b0[i]; acc[4 + i] = b1[i]; acc[8 + i] = b2[i]; acc[12 + i] = b3[i]; }
  return acc;
}

// Epilogue step u (0..7) of a finished tile: accumulator pair -> one packed bf16 dword, floored (ReLU / pass-through).
// u < 4: registers 2u, 2u+1 -> out0[u]; u >= 4: registers 8 + 2(u-4), +1 -> out1[u-4].
// The register file is full: each packed dword must exist HERE.  Left alone, hipcc sinks the pure convert / max chain down
// to the outputs' first use (the next layer), keeps every tile's 16 accumulators alive until then and spills ~200
// registers.  The empty asm makes the dword a value that exists at this point.
MF_D void epi_pair(const f32x16& a, int u, unsigned floor, u32x4& out0, u32x4& out1) {
  const int w = u & 3, r = (u < 4 ? 0 : 8) + 2 * w;
  unsigned v = pk_floor_bf16(pack_bf16x2(a[r], a[r + 1]), floor);
  asm volatile("" : "+v"(v));
  (u < 4 ? out0 : out1)[w] = v;
}
// the steps of gap gi of an NG-group tile: the 8 steps spread over gaps 1 .. min(8, NG - 3) (gap 0 carries the panel barrier,
// gap NG - 2 the next tile's bias reads)
MF_D constexpr int epi_lo(int gi, int ng) { const int e = ng - 3 < 8 ? ng - 3 : 8; return gi < 1 ? 0 : (gi - 1 >= e ? 8 : 8 * (gi - 1) / e); }
MF_D constexpr int epi_hi(int gi, int ng) { const int e = ng - 3 < 8 ? ng - 3 : 8; return gi < 1 ? 0 : (gi >= e ? 8 : 8 * gi / e); }

template <int NGE, int KHID, bool EMB_FIRST, bool SPLIT, class Hook, class Piece, class Gap>
MF_D void mma_tile(Carry& carry, const u32x4* hid, const u32x4* xhi, const u32x4* xlo, uint32_t p, uint32_t pn,
                   f32x16& acc, Hook&& hook, Piece&& piece, Gap&& gap) {
  constexpr int NEG = (SPLIT ? 2 : 1) * NGE;            // groups of the embedded block
  constexpr int NG = NEG + KHID;
  static_assert(NG > PD, "panel shorter than the fragment pipeline");
  static_assert(NG >= 4, "panel too short for the DMA pieces");
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = carry.w[i];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int s = gi % (PD + 1);
    const int ge = EMB_FIRST ? gi : gi - KHID;            // index within the embedded groups
    if (ge >= 0 && ge < NEG) {
      acc = MF_MFMA32(r[s], xhi[SPLIT ? ge >> 1 : ge], acc);   // SPLIT: even: Whi * xhi ; odd: Wlo * xhi
    } else {
      acc = MF_MFMA32(r[s], hid[EMB_FIRST ? gi - NEG : gi], acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
    if (nb < NG) r[sp] = lds_u4(p + nb * kGroupBytes);
    if (gi == 0) hook();
    if (gi >= 1 && gi <= 4) piece(gi - 1);     // (the 4th piece: only the 32-group panels of MF_PREC_BF16X3 have one)
    if (nb >= NG) r[sp] = lds_u4(pn + (nb - NG) * kGroupBytes);
    gap(gi, NG);
    __builtin_amdgcn_sched_barrier(0);
#ifndef MF_BF_BREAK_LO      // (-DMF_BF_BREAK_LO: the deliberately broken build the oracle-of-the-arithmetic tests must reject)
    if (SPLIT && ge >= 0 && ge < NEG && !(ge & 1)) {
      acc = MF_MFMA32(r[s], xlo[ge >> 1], acc);           // Whi * xlo
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  }
#pragma unroll
  for (int i = 0; i < PD; ++i) carry.w[i] = r[(NG + i) % (PD + 1)];
}

// Head tile (NoF 3|9-row head): acc = bias + (Whi + Wlo) * hidden, raw fp32 accumulators (rows (r&3)+8(r>>2)+4h).
// TERMS = groups per hidden k-step: 2 = the NoF's (hi, lo) group pairs; 1 = ONE group whose rows carry the terms (the NeRF's
// sigma / rgb heads, NetLayout::head_tiles: row c = bf16(w_c), row 8 + c = bf16(w_c - hi) -- registers c and 4 + c of lane half
// 0 -- half the MFMAs and fragment reads of the group-pair form for heads of <= 4 rows).  ZERO: accumulators start at 0 (the
// caller adds its scalar biases) instead of a 32-row bias vector in LDS.
template <int KHID, int TERMS = 2, bool ZERO = false, class Hook, class Piece>
MF_D f32x16 head_tile(Carry& carry, const u32x4* hid, uint32_t p, uint32_t pn, uint32_t bias_off, int h, Hook&& hook,
                      Piece&& piece) {
  constexpr int NG = TERMS * KHID;
  static_assert(NG > PD && NG >= 4, "head panel too short");
  f32x16 acc;
  if constexpr (ZERO) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  } else {
    const f32x4 b0 = lds_f4(bias_off + (0 + 4 * h) * 4), b1 = lds_f4(bias_off + (8 + 4 * h) * 4);
    const f32x4 b2 = lds_f4(bias_off + (16 + 4 * h) * 4), b3 = lds_f4(bias_off + (24 + 4 * h) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = b0[i]; acc[4 + i] = b1[i]; acc[8 + i] = b2[i]; acc[12 + i] = b3[i]; }
  }
  u32x4 r[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i) r[i] = carry.w[i];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int s = gi % (PD + 1);
    acc = MF_MFMA32(r[s], hid[TERMS == 2 ? gi >> 1 : gi], acc);
    __builtin_amdgcn_sched_barrier(0);
    const int sp = (gi + PD) % (PD + 1), nb = gi + PD;
    if (nb < NG) r[sp] = lds_u4(p + nb * kGroupBytes);
    if (gi == 0) hook();
    if (gi >= 1 && gi <= 3) piece(gi - 1);
    if (nb >= NG) r[sp] = lds_u4(pn + (nb - NG) * kGroupBytes);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < PD; ++i) carry.w[i] = r[(NG + i) % (PD + 1)];
  return acc;
}

// One trunk layer: act <- relu?(W_l [emb ; act] + b_l), NT = KH/2 tiles of 32 features.
// MODE: 1 = embedded input only (layer 0), 2 = hidden only, 3 = both (skip layers, embedded input first).  A template
// parameter, not a switch inside the tile loop: the register file is full here and every control-flow merge inside
// the unrolled tile sequence costs copies.
// RB: the layer's accumulators start from the per-ray bias `rb` (one f32x16 per tile, see mma_tile).
struct RayBias { f32x16 t[4]; };    // per tile of a 128-wide layer: the accumulators' initial value (whole vectors: a
                                    // f32x4[16] read back as f32x16 defeats SROA and lands in scratch)
struct NoRayBias {};
// The same vectors staged in LDS (the render pass: the rows of the tile's rays arrive by LDS-DMA one chain step ahead,
// stage_raybias in mf_render_bf16.hip): `lane_off` = LDS byte offset of this lane's ray's [embedded layer][128] block.
// The accumulators then start with four ds_reads exactly like the static bias, only from a per-lane address.
struct LdsRayBias { uint32_t lane_off; };
template <int KH, int NGE, int MODE, bool SPLIT, class RBT = NoRayBias, int TPP = 1>
MF_D void trunk_layer_m(const Net& net, int layer, bool relu, const u32x4 (&act)[KH], u32x4 (&out)[KH], const u32x4 (&xhi)[NGE],
                        const u32x4 (&xlo)[NGE], Stream& st, Carry& carry, const Lane& id, const Next& nxt, const RBT& rb) {
  constexpr bool RB = __is_same(RBT, RayBias);
  static_assert(!RB || KH == 8, "per-ray bias: 128-wide layers (4 tiles)");
  constexpr int NT = KH / 2;
  constexpr int NP = NT / TPP;                          // panels of the layer (TPP tiles each)
  static_assert(NT % TPP == 0, "tiles per panel");
  const int groups = ((MODE & 1) ? (SPLIT ? 2 : 1) * NGE : 0) + ((MODE & 2) ? KH : 0);      // of one tile
  const int pgroups = TPP * groups;                                                           // of one panel
  const unsigned lo = relu ? 0u : 0x80008000u;      // ReLU / pass-through floor
  uint32_t bias_off = net.res_lds + layer * (16 * KH) * 4;
  if constexpr (__is_same(RBT, LdsRayBias) && (MODE & 1))      // embedded layer number popcount(mask below `layer`)
    bias_off = rb.lane_off + (uint32_t)__builtin_popcount(net.emb_mask & ((1u << layer) - 1u)) * (16 * KH) * 4;
  auto init = [&](int t) __attribute__((always_inline)) -> f32x16 {       // tile t's accumulators before its first MFMA
    if constexpr (RB && (MODE & 1)) return rb.t[t < 4 ? t : 0];
    else return bias_acc(bias_off + 32 * t * 4, id.h);
  };
  f32x16 acc = init(0), pend = {}, nacc = {};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pi = t / TPP, sub = t % TPP;             // panel of this tile, position inside it
    const bool second = sub != 0;                      // (not the tile that carries the panel's barrier and DMA)
    const uint32_t p = st.slot_off(0) + id.lane * 16 + sub * groups * kGroupBytes;
    const uint32_t pn = sub == TPP - 1 ? st.slot_off(1) + id.lane * 16 : p + groups * kGroupBytes;
    // panel two ahead: same layer while pi + 2 < NP, else panel (pi + 2 - NP) of the next layer
    auto hook = [&]() {
      if (second) return;
      st.sync(pi + 2 < NP ? pgroups : (pi == NP - 2 ? nxt.groups : nxt.groups2),
              pi == NP - 2 ? nxt.jump : (pi == NP - 1 ? nxt.jump2 : nullptr), id, true, pi + 2 < NP ? pgroups : -1);
    };
    // this wave's pieces of the panel two ahead: up to four per tile (gaps 1-4), the panel's later tiles carry on where a
    // short first tile stops (layer 0 of the NeRF: eight 4-group tiles in one panel, 3 + 1 pieces)
    const int ppt = groups - 1 < 4 ? groups - 1 : 4;
    auto piece = [&](int k) { if (k < ppt) st.piece(sub * ppt + k, id); };
    const int tp = t > 0 ? t - 1 : 0;                  // the tile whose epilogue is pending
    auto gap = [&](int gi, int ng) __attribute__((always_inline)) {
      if (t > 0) {
#pragma unroll
        for (int u = epi_lo(gi, ng); u < epi_hi(gi, ng); ++u) epi_pair(pend, u, lo, out[2 * tp], out[2 * tp + 1]);
      }
      if (gi == ng - 2 && t + 1 < NT) nacc = init(t + 1);
    };
    mma_tile<(MODE & 1) ? NGE : 0, (MODE & 2) ? KH : 0, true, SPLIT>(carry, act, xhi, xlo, p, pn, acc, hook, piece, gap);
    pend = acc;
    acc = nacc;
    if (sub == TPP - 1) st.advance();
  }
  // the layer's last tile: nobody's gaps to run in (the next layer's first tile reads every input)
#pragma unroll
  for (int u = 0; u < 8; ++u) epi_pair(pend, u, lo, out[2 * (NT - 1)], out[2 * (NT - 1) + 1]);
  __builtin_amdgcn_sched_barrier(0);
}

// in -> out (two register sets: the callers alternate them from layer to layer, so no layer ends in a 64-register copy)
template <int KH, int NGE, bool SPLIT, class RBT = NoRayBias, int TPP0 = 1, int TPPH = 1, int TPPS = 1>
MF_D void trunk_layer(const Net& net, int layer, bool relu, const u32x4 (&act)[KH], u32x4 (&out)[KH], const u32x4 (&xhi)[NGE],
                      const u32x4 (&xlo)[NGE], Stream& st, Carry& carry, const Lane& id, const Next& nxt, const RBT& rb) {
  const int has_emb = (net.emb_mask >> layer) & 1;
  if (layer == 0) trunk_layer_m<KH, NGE, 1, SPLIT, RBT, TPP0>(net, layer, relu, act, out, xhi, xlo, st, carry, id, nxt, rb);
  else if (has_emb) trunk_layer_m<KH, NGE, 3, SPLIT, RBT, TPPS>(net, layer, relu, act, out, xhi, xlo, st, carry, id, nxt, rb);
  else trunk_layer_m<KH, NGE, 2, SPLIT, RBT, TPPH>(net, layer, relu, act, out, xhi, xlo, st, carry, id, nxt, rb);
}

// This lane's rows of the per-ray bias of embedded layer number `el` (0 = layer 0, 1 = the first skip layer, ...):
// rbp = table row of the lane's (ray, network, index value) + 4 h floats; 16 global loads of 16 bytes (L2 hits: a wave's
// 32 samples share one or two rays), issued a whole layer ahead of their use.
MF_D void load_raybias(RayBias& rb, const float* rbp, int el) {
  const float* src = rbp + el * 128;
  typedef float f32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + 32 * t), b = *reinterpret_cast<const f32x4*>(src + 32 * t + 8);
    const f32x4 c = *reinterpret_cast<const f32x4*>(src + 32 * t + 16), d = *reinterpret_cast<const f32x4*>(src + 32 * t + 24);
    const f32x8 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
    rb.t[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
  }
}
MF_D void load_raybias(NoRayBias&, const float*, int) {}
MF_D void load_raybias(LdsRayBias&, const float*, int) {}

// the D trunk layers of a network, in pairs a -> b -> a; returns with the last layer's output in `a`
// RB: `rb` arrives loaded for layer 0; the set of the next embedded layer is fetched at the START of the layer in front of
// it (a whole layer of MFMAs to land) -- or, when that layer consumes the current set itself (adjacent embedded layers),
// right behind it.  One set is live at a time.
// `after_first()` runs once behind layer 0 (the render pass stages the NEXT chain step's per-ray bias rows there: this
// wave is past the step's first panel barrier, and every later barrier of the step publishes them).
template <int KH, int NGE, bool SPLIT, int TPP0 = 1, int TPPH = 1, int TPPS = 1, class NextOf, class RBT, class AfterFirst>
MF_D void trunk(const Net& net, int D, u32x4 (&a)[KH], const u32x4 (&xhi)[NGE], const u32x4 (&xlo)[NGE], Stream& st, Carry& carry,
                const Lane& id, NextOf&& next_of, RBT& rb, const float* rbp, AfterFirst&& after_first) {
  constexpr bool RB = __is_same(RBT, RayBias);
  u32x4 b[KH];
  int l = 0;
  int el = 1;                                               // ordinal of the next embedded layer behind layer 0
  auto one = [&](int layer, const u32x4 (&in)[KH], u32x4 (&out)[KH]) {
    if constexpr (RB) {
      const bool emb_here = (net.emb_mask >> layer) & 1, emb_next = layer + 1 < D && ((net.emb_mask >> (layer + 1)) & 1);
      if (!emb_here && emb_next) load_raybias(rb, rbp, el);
      trunk_layer<KH, NGE, SPLIT, RBT, TPP0, TPPH, TPPS>(net, layer, true, in, out, xhi, xlo, st, carry, id, next_of(layer), rb);
      if (emb_here && emb_next) load_raybias(rb, rbp, el);
      if (emb_next) ++el;
    } else {
      trunk_layer<KH, NGE, SPLIT, RBT, TPP0, TPPH, TPPS>(net, layer, true, in, out, xhi, xlo, st, carry, id, next_of(layer), rb);
    }
  };
  for (; l + 1 < D; l += 2) {
    one(l, a, b);
    if (l == 0) after_first();
    st.tl.stamp(10 + l, id);
    one(l + 1, b, a);
    st.tl.stamp(11 + l, id);
  }
  if (l < D) {
    one(l, a, b);
    if (l == 0) after_first();
#pragma unroll
    for (int t = 0; t < KH; ++t) a[t] = b[t];
  }
}

// ------------------------------------------------------------------ embedding in registers (two lane halves)
// Embedding parameters live in LDS (par_off: freq[16] then weight[16], floats): this kernel keeps its SGPRs for
// addresses.  dst[0..SLOTS) of block (C,F) for lane half h; arg = freq*x rounded to fp32 before sin/cos exactly as
// `func(freq*x)` in embedding.py:45.
// sin / cos of an angle in radians on the transcendental unit: v_sin_f32 / v_cos_f32 take REVOLUTIONS in [-256, 256], so
// the angle is scaled by 1/(2 pi) and reduced with v_fract_f32.  Error: the fp32 rounding of the scaled angle, |rev| x
// 6e-8 revolutions (2e-4 rad at the NeRF's largest argument, 512 x 6 / 2 pi = 490 revolutions; 1e-6 rad where the angle
// stays under a few revolutions) -- an order of magnitude under the 4e-3 of the bf16 operands the values become.  Four
// instructions per pair against ~110 for OCML's sincosf with its exact range reduction: on a pass whose clock drops with its activity
// the VALU work saved is worth more than the cycles.
MF_D void sincos_rev(float rad, float& sn, float& cs) {
  const float f = __builtin_amdgcn_fractf(rad * 0.15915494309189535f);
  sn = __builtin_amdgcn_sinf(f);
  cs = __builtin_amdgcn_cosf(f);
}

// (`h` is made opaque at the top of each evaluation: everything derived from it here -- the half's component index, its table
//  address -- is loop-invariant per lane, and hipcc hoisted those ~15 values out of the whole group loop into registers the MFMA
//  section has no room for: the 17 spilled VGPRs / 72 B of scratch per lane of round 5's fast kernels.  Re-derived per evaluation
//  they are a v_cndmask each, inside a VALU phase.)
MF_D int opaque_lane_half(int h) {
  asm volatile("" : "+v"(h));
  return h;
}
template <int C, int F, bool HW = false>
MF_D void emb_eval_direct(float* dst, const float (&v)[C], uint32_t par_off, int h_in) {
  using B = EmbBlock2<C, F>;
  const int h = opaque_lane_half(h_in);
#pragma unroll
  for (int pi = 0; pi < B::NPI; ++pi) {
    const int p0 = 2 * pi, p1 = 2 * pi + 1;                // the two halves' pairs
    const bool real0 = p0 < B::NPAIR, real1 = p1 < B::NPAIR;
    const int f0 = real0 ? p0 / C : 0, c0 = real0 ? p0 % C : 0;
    const int f1 = real1 ? p1 / C : 0, c1 = real1 ? p1 % C : 0;
    const int q0 = real0 ? 0 : 2 * (p0 - B::NPAIR), q1 = real1 ? 0 : 2 * (p1 - B::NPAIR);   // raw pseudo-pairs
    const float ra0 = (!real0 && p0 < B::NALL && q0 < C) ? v[q0 < C ? q0 : 0] : 0.f;
    const float rb0 = (!real0 && p0 < B::NALL && q0 + 1 < C) ? v[q0 + 1 < C ? q0 + 1 : 0] : 0.f;
    const float ra1 = (!real1 && p1 < B::NALL && q1 < C) ? v[q1 < C ? q1 : 0] : 0.f;
    const float rb1 = (!real1 && p1 < B::NALL && q1 + 1 < C) ? v[q1 + 1 < C ? q1 + 1 : 0] : 0.f;
    if (!real0 && !real1) {                                // (compile time) raw components only
      dst[2 * pi] = h ? ra1 : ra0;
      dst[2 * pi + 1] = h ? rb1 : rb0;
      continue;
    }
    const float x = h ? v[c1] : v[c0];
    const int f = h ? f1 : f0;
    const float fr = lds_f(par_off + 4 * f);
    const float w = lds_f(par_off + 64 + 4 * f);
    const bool real = h ? real1 : real0;
    // skipped when both halves' frequencies are muted (coarse-to-fine start, trainer_moco_flow.py:113-114)
    const bool live = lds_f(par_off + 64 + 4 * f0) != 0.f || (real1 && lds_f(par_off + 64 + 4 * f1) != 0.f);
    float sn = 0.f, cs = 0.f;
    if (__builtin_amdgcn_readfirstlane((int)live)) {
      if (HW) sincos_rev(fr * x, sn, cs);
      else sincosf(fr * x, &sn, &cs);
    }
    dst[2 * pi] = real ? w * sn : (h ? ra1 : ra0);
    dst[2 * pi + 1] = real ? w * cs : (h ? rb1 : rb0);
  }
}

// The same for the logscale tables every configuration of the reference uses (freq_bands = 2^k, embedding.py:19;
// `pow2` is checked on the host): a lane half owns, for each of its C chains (chain j = pi mod C: component
// (2j + h) mod C), every SECOND frequency f = f_start + 2m, so consecutive entries of a chain are angle quadruplings:
//     sin 2a = 2 sin a cos a,  cos 2a = 1 - 2 sin^2 a,   twice.
// An exact sincosf re-seeds each chain every third entry, so no value is more than two quadruplings (four doublings:
// <= 16 x the 1e-7 of the seed, 2e-6) from an exact one -- an order of magnitude under the 2^-16 of the split bf16
// operands the values are rounded to.  6 sincosf instead of 15 per NeRF encoding and lane, 6 instead of 18 per NoF input.
// HW: every pair straight from the transcendental unit (sincos_rev) -- the blocks whose values become plain bf16 operands
// or whose arguments stay small (NeRF xyz / dir / ind, NoF xyz); the NoF's image-index block (arguments up to 2^15, split
// operands) keeps the exact seeds + doubling chains.
template <int C, int F, bool HW = false>
MF_D void emb_eval(float* dst, const float (&v)[C], uint32_t par_off, int h_in, bool pow2) {
  using B = EmbBlock2<C, F>;
  if (HW) { emb_eval_direct<C, F, true>(dst, v, par_off, h_in); return; }
  if (!pow2) { emb_eval_direct<C, F>(dst, v, par_off, h_in); return; }
  const int h = opaque_lane_half(h_in);
  float cs_[C], sn_[C];                                    // the chains' current (cos, sin)
#pragma unroll
  for (int pi = 0; pi < B::NPI; ++pi) {
    const int p0 = 2 * pi, p1 = 2 * pi + 1;
    const bool real0 = p0 < B::NPAIR, real1 = p1 < B::NPAIR;
    const int q0 = real0 ? 0 : 2 * (p0 - B::NPAIR), q1 = real1 ? 0 : 2 * (p1 - B::NPAIR);
    const float ra0 = (!real0 && p0 < B::NALL && q0 < C) ? v[q0 < C ? q0 : 0] : 0.f;
    const float rb0 = (!real0 && p0 < B::NALL && q0 + 1 < C) ? v[q0 + 1 < C ? q0 + 1 : 0] : 0.f;
    const float ra1 = (!real1 && p1 < B::NALL && q1 < C) ? v[q1 < C ? q1 : 0] : 0.f;
    const float rb1 = (!real1 && p1 < B::NALL && q1 + 1 < C) ? v[q1 + 1 < C ? q1 + 1 : 0] : 0.f;
    if (!real0 && !real1) {
      dst[2 * pi] = h ? ra1 : ra0;
      dst[2 * pi + 1] = h ? rb1 : rb0;
      continue;
    }
    const int j = pi % C, m = pi / C;                      // chain, entry within the chain
    const int f0 = real0 ? p0 / C : 0, c0 = real0 ? p0 % C : 0;
    const int f1 = real1 ? p1 / C : 0, c1 = real1 ? p1 % C : 0;
    const int f = h ? f1 : f0;
    if (m % 3 == 0) {                                      // seed: exact
      const float x = h ? v[c1] : v[c0];
      const float fr = lds_f(par_off + 4 * f);
      sincosf(fr * x, &sn_[j], &cs_[j]);
    } else {                                               // two doublings: f -> f + 2
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float t = sn_[j] + sn_[j];
        const float s2 = t * cs_[j];
        cs_[j] = __builtin_fmaf(-t, sn_[j], 1.f);
        sn_[j] = s2;
      }
    }
    const float w = lds_f(par_off + 64 + 4 * f);
    const bool real = h ? real1 : real0;
    dst[2 * pi] = real ? w * sn_[j] : (h ? ra1 : ra0);
    dst[2 * pi + 1] = real ? w * cs_[j] : (h ? rb1 : rb0);
  }
}

// fp32 slots -> split bf16 operands of the k-steps: xhi[ks] = bf16(x), xlo[ks] = bf16(x - hi)
template <int KS>
MF_D void split_operands(const float* emb, int n_slots, u32x4 (&xhi)[KS], u32x4 (&xlo)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e0 = 8 * ks + 2 * w, e1 = e0 + 1;
      const float a = e0 < n_slots ? emb[e0] : 0.f, b = e1 < n_slots ? emb[e1] : 0.f;
      const unsigned hi = pack_bf16x2(a, b);
      xhi[ks][w] = hi;
      xlo[ks][w] = pack_bf16x2(a - bflo(hi), b - bfhi(hi));
    }
  }
}

// fp32 slots -> IEEE-half (hi, lo) pairs of kHalfActScale x: hi = half(s x), lo = half(s x - hi)   (22 significand bits)
template <int KS>
MF_D void split_operands_half(const float* emb, int n_slots, u32x4 (&xhi)[KS], u32x4 (&xlo)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e0 = 8 * ks + 2 * w, e1 = e0 + 1;
      const float a = e0 < n_slots ? emb[e0] * kHalfActScale : 0.f, b = e1 < n_slots ? emb[e1] * kHalfActScale : 0.f;
      const unsigned hi = pack_f16x2(a, b);
      xhi[ks][w] = hi;
      xlo[ks][w] = pack_f16x2(a - hflo(hi), b - hfhi(hi));
    }
  }
}

// fp32 slots -> three-term operands: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)   (24 mantissa bits)
template <int KS>
MF_D void split_operands3(const float* emb, int n_slots, u32x4 (&xhi)[KS], u32x4 (&xmid)[KS], u32x4 (&xlo)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e0 = 8 * ks + 2 * w, e1 = e0 + 1;
      const float a = e0 < n_slots ? emb[e0] : 0.f, b = e1 < n_slots ? emb[e1] : 0.f;
      const unsigned hi = pack_bf16x2(a, b);
      const float ra = a - bflo(hi), rb = b - bfhi(hi);
      const unsigned mid = pack_bf16x2(ra, rb);
      xhi[ks][w] = hi;
      xmid[ks][w] = mid;
      xlo[ks][w] = pack_bf16x2(ra - bflo(mid), rb - bfhi(mid));
    }
  }
}

// fp32 slots -> plain bf16 operands (the NeRF's encodings)
template <int KS>
MF_D void pack_operands(const float* emb, int n_slots, u32x4 (&x)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int e0 = 8 * ks + 2 * w, e1 = e0 + 1;
      x[ks][w] = pack_bf16x2(e0 < n_slots ? emb[e0] : 0.f, e1 < n_slots ? emb[e1] : 0.f);
    }
}

// ------------------------------------------------------------------ the two networks
template <int KH, int EKS, bool SPLIT, int TPP = 1, int T = 2, int TPPH = 0, int TPPS = 0, class ST, class CR>
MF_D void start_program(const Net& n, ST& st, CR& carry, uint32_t ring, uint32_t buf_bytes, const Lane& id) {
  const Next f = first_of<KH, EKS, SPLIT, TPP, T, TPPH, TPPS>(n);        // the program's first two panels (contiguous)
  st.start(f.jump, f.groups, f.groups2, ring, buf_bytes, id);
  carry.load(st.slot_off(0) + id.lane * 16);
}

template <int NW = kWaves>
MF_D void load_resident(const Net& n, const Lane& id) {
  const int groups = (int)(n.res_bytes / kGroupBytes);
  for (int g = id.wave; g < groups; g += NW) blds16(n.packed, id.lane * 16, g * kGroupBytes, n.res_lds + g * kGroupBytes);
}

// extra_encoding (nerf.py:98): W/2 outputs from [final (W) ; extra block], ReLU.  NGX = extra k-steps (0, 1, 2).
template <int NGX, bool SPLIT = false>
MF_D void extra_layer(const Net& net, const u32x4 (&act)[16], const u32x4* ex, const u32x4* exlo, u32x4 (&out)[8],
                      Stream& st, Carry& carry, const Lane& id, const Next& nxt) {
  constexpr int NT = 4;
  const int groups = 16 + (SPLIT ? 2 : 1) * NGX;
  const uint32_t bias_off = net.res_lds + (net.D + 1) * 256 * 4;
  f32x16 acc = bias_acc(bias_off, id.h), pend = {}, nacc = {};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const uint32_t p = st.slot_off(0) + id.lane * 16;
    const uint32_t pn = st.slot_off(1) + id.lane * 16;
    auto hook = [&]() {
      st.sync(t + 2 < NT ? groups : (t == NT - 2 ? nxt.groups : nxt.groups2),
              t == NT - 2 ? nxt.jump : (t == NT - 1 ? nxt.jump2 : nullptr), id, true, t + 2 < NT ? groups : -1);
    };
    auto piece = [&](int k) { st.piece(k, id); };
    const int tp = t > 0 ? t - 1 : 0;
    auto gap = [&](int gi, int ng) __attribute__((always_inline)) {
      if (t > 0) {
#pragma unroll
        for (int u = epi_lo(gi, ng); u < epi_hi(gi, ng); ++u) epi_pair(pend, u, 0u, out[2 * tp], out[2 * tp + 1]);
      }
      if (gi == ng - 2 && t + 1 < NT) nacc = bias_acc(bias_off + 32 * (t + 1) * 4, id.h);
    };
    mma_tile<NGX, 16, false, SPLIT>(carry, act, ex, exlo, p, pn, acc, hook, piece, gap);
    pend = acc;
    acc = nacc;
    st.advance();
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) epi_pair(pend, u, 0u, out[2 * (NT - 1)], out[2 * (NT - 1) + 1]);
  __builtin_amdgcn_sched_barrier(0);
}

// Canonical NeRF (W = 256) on this wave's 32 samples.  xe: bf16 operands of the xyz embedding (4 k-steps).
// `make_extra(ex)` builds the extra block's operands; it is called right before extra_encoding so that those
// registers are not held through the trunk.
// tiles per panel of the fast mode's NeRF: layer 0 (eight tiles of 4 groups), hidden-only layers (16 groups), skip layers (20).
// Re-priced on the pipelined tile loop in round 5 (profiles/r05_ab_nof_tpp_and_fp16_nof.txt, same-box A/Bs): layer 0 as ONE
// panel and two hidden tiles per panel both measure +-1 % (rounds 2-4 measured the same on the serial tile loop).
constexpr int kNerfTpp0 = 1, kNerfTppH = 1, kNerfTppS = 1;
template <class MakeExtra>
MF_D void nerf_eval(const Net& net, const u32x4 (&xe)[kKsNerfXyz],
                    MakeExtra&& make_extra, bool sigma_only, Stream& st,
                    Carry& carry, const Lane& id, const Next& follow, float& sigma, float (&rgb)[3]) {
  u32x4 act[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) act[t][i] = 0;
  const int D = net.D;
  NoRayBias norb;
  // Round 6: the sigma and rgb heads are PANELS of the weight stream (NetLayout::head_tiles) -- 16 + 8 MFMAs and fragment reads per
  // wave where the VALU dot products cost 192 + 192 FMAs, as many unpack operations and 32 + 48 ds_read_b128 of fp32 weight rows
  // (80 KiB of LDS reads per wave and tile: what 80 MFMAs' fragments cost), with both waves of a SIMD in VALU code at the same time.
  // Weights as bf16 (hi, lo) row pairs (16 significand bits; the activations they multiply are bf16), fp32 accumulation.
  // The sigma panel follows the last trunk layer; behind it comes xyz_encoding_final's first panel -- or, sigma_only, `follow`'s.
  const Next sg_next = sigma_only ? Next{16, nullptr, follow.groups, follow.jump} : Next{16, nullptr, 16, nullptr};
  trunk<16, kKsNerfXyz, false, kNerfTpp0, kNerfTppH, kNerfTppS>(net, D, act, xe, xe, st, carry, id, [&](int l) {
    return l == D - 1 ? sg_next : next_trunk_bf<16, kKsNerfXyz, false, 0, kNerfTppH, kNerfTppS>(net, l + 1);
  }, norb, nullptr, [] {});
  // resident block: [bias_trunk (D+1) 256 | bias_extra 128 | sigma_w 256 | sigma_b 4 | rgb_w 384 | rgb_b 4]
  const uint32_t r_sigma_w = net.res_lds + ((D + 1) * 256 + 128) * 4;
  {
    const uint32_t p = st.slot_off(0) + id.lane * 16, pn = st.slot_off(1) + id.lane * 16;
    // panel two ahead of the sigma panel: xyz_encoding_final's second tile (contiguous), or the SECOND panel of `follow`
    auto hook = [&]() {
      if (sigma_only) st.sync(follow.groups2, follow.jump2, id);
      else st.sync(16, nullptr, id, true, 16);
    };
    auto piece = [&](int k) { st.piece(k, id); };
    const f32x16 acc = head_tile<16, 1, true>(carry, act, p, pn, 0u, id.h, hook, piece);
    st.advance();
    sigma = acc[0] + acc[4] + lds_f(r_sigma_w + 256 * 4);      // rows 0 (hi) + 8 (lo): valid in lane half 0 (the half that stores it)
  }
  st.tl.stamp(30, id);
  if (sigma_only) return;
  const int xg = 16 + net.aux;
  const Next ex{xg, nullptr, xg, nullptr};
  u32x4 fin[16];
  trunk_layer_m<16, kKsNerfXyz, 2, false, NoRayBias, kNerfTppH>(net, D, false, act, fin, xe, xe, st, carry, id, ex, norb);   // xyz_encoding_final (no ReLU, hidden input only)
  st.tl.stamp(31, id);
  u32x4 e[8], eo[kKsExtraMax];
  make_extra(eo);
  st.tl.stamp(32, id);
  // behind extra_encoding: the rgb panel (8 groups), then `follow`'s first panel
  const Next rg_next{8, nullptr, follow.groups, follow.jump};
  if (net.aux == 2) extra_layer<2>(net, fin, eo, eo, e, st, carry, id, rg_next);
  else if (net.aux == 1) extra_layer<1>(net, fin, eo, eo, e, st, carry, id, rg_next);
  else extra_layer<0>(net, fin, eo, eo, e, st, carry, id, rg_next);
  st.tl.stamp(33, id);
  {
    const uint32_t p = st.slot_off(0) + id.lane * 16, pn = st.slot_off(1) + id.lane * 16;
    auto hook = [&]() { st.sync(follow.groups2, follow.jump2, id); };
    auto piece = [&](int k) { st.piece(k, id); };
    const f32x16 acc = head_tile<8, 1, true>(carry, e, p, pn, 0u, id.h, hook, piece);
    st.advance();
    const uint32_t r_rgb_b = r_sigma_w + (256 + 4 + 384) * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float o = acc[c] + acc[4 + c] + lds_f(r_rgb_b + 4 * c);      // rows c (hi) + 8 + c (lo), lane half 0
      rgb[c] = 1.f / (1.f + expf(-o));                                  // nn.Sigmoid, nerf.py:57-59
    }
  }
}

// Neural motion flow (W = 128) on this wave's 32 samples; xhi/xlo = split operands of the xyz block; `rb` = the per-ray
// bias (image-index block + layer bias) of layer 0, already in flight, `rbp` = where the later embedded layers' sets are.
// (RBT = RayBias: register sets fetched from the global table, the per-point query; LdsRayBias: staged in LDS, the render pass)
// tiles per panel of the fast mode's NoF (128 wide: four tiles per layer of 6 / 8 / 14 groups): two everywhere since round 3
// (one tile per panel: C3g +1.9 %, C5 +2.4 %).  Round 5, with whole layers as one panel where the ring slot holds them -- (4, 4, 2),
// (4, 2, 2), (2, 4, 2); the stream's two-panel look-ahead handles single-panel layers since then (first_of / next_trunk_np) --:
// C3 / C3g / C5 all within +-1 % of (2, 2, 2), every variant green against the oracle of the arithmetic.
constexpr int kNofTpp0 = 2, kNofTppH = 2, kNofTppS = 2;
template <class RBT, class AfterFirst>
MF_D void nof_eval(const Net& net, const u32x4 (&xhi)[kKsNofXyz], const u32x4 (&xlo)[kKsNofXyz], const float (&xyz)[3],
                   Stream& st, Carry& carry, const Lane& id, const Next& follow, float (&out)[3], RBT& rb,
                   const float* rbp, AfterFirst&& after_first) {
  u32x4 act[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) act[t][i] = 0;
  const int D = net.D;
  // the head panel (8 groups: rows c = bf16(w_c), rows 16 + c = bf16(w_c - hi), NetLayout::head_tiles) follows the last trunk
  // layer contiguously; behind it comes `follow`'s first panel
  const Next hd{8, nullptr, follow.groups, follow.jump};
  trunk<8, kKsNofXyz, true, kNofTpp0, kNofTppH, kNofTppS>(net, D, act, xhi, xlo, st, carry, id,
      [&](int l) { return l == D - 1 ? hd : next_trunk_np<8, kKsNofXyz, true, kNofTppH, kNofTppS>(net, l + 1, D, hd); }, rb, rbp, after_first);
  // head on the matrix pipe (8 MFMAs instead of 9 x 64 dependent FMAs + 144 LDS reads per lane; rounds 2-5: 16, the terms as
  // group pairs); T[0..3] come out in half 0's registers 0-3 (+ their lo terms in 8-11), T[4..7] in half 1's registers 0-3
  // (+ 8-11), T[8] in half 0's register 4 (+ 12)
  f32x16 acc;
  {
    const uint32_t p = st.slot_off(0) + id.lane * 16, pn = st.slot_off(1) + id.lane * 16;
    // panel two ahead of the head panel = the SECOND panel of whatever follows
    auto hook = [&]() { st.sync(follow.groups2, follow.jump2, id); };
    auto piece = [&](int k) { st.piece(k, id); };
    // resident block: [bias_trunk D 128 | head_w n_head 128 | head_b 32]
    acc = head_tile<8, 1>(carry, act, p, pn, net.res_lds + (D + net.aux) * 128 * 4, id.h, hook, piece);
    st.advance();
  }
  float own[5], oth[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) { own[i] = acc[i] + acc[8 + i]; oth[i] = __shfl_xor(own[i], 32, 64); }
  if (net.aux == 9) {
    float T[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) { T[i] = id.h ? oth[i] : own[i]; T[4 + i] = id.h ? own[i] : oth[i]; }
    T[8] = id.h ? oth[4] : own[4];
    quat_transform<true>(T, xyz, out);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (id.h ? oth[c] : own[c]) + xyz[c];
  }
}

// NoF matrix operands from a point: its xyz block (rendering.py:70-72); the image-index block (:73-75) is the per-ray bias.
// HW: sin / cos from the transcendental unit (the fast mode: the angle's fp32 rounding in revolutions, <= 6e-6 rad at
// 16 x, is far under the hidden layers' bf16 rounding); the three-product mode takes exact seeds + doubling chains -- the
// canonical point this network produces feeds sin(512 x), and 6e-6 rad is the size of the split operands' own 2^-17.
template <bool HW = true>
MF_D void nof_embed(u32x4 (&xhi)[kKsNofXyz], u32x4 (&xlo)[kKsNofXyz], const float (&xyz)[3], uint32_t par_xyz, int h, bool pow2_xyz) {
  float emb[B2Xyz5::SLOTS];
  emb_eval<3, 5, HW>(emb, xyz, par_xyz, h, pow2_xyz);
  split_operands<kKsNofXyz>(emb, B2Xyz5::SLOTS, xhi, xlo);
}
// the same as T-term operands (T = 2: xmid untouched)
template <bool HW, int T, bool HF = false>
MF_D void nof_embed_t(u32x4 (&xhi)[kKsNofXyz], u32x4 (&xmid)[kKsNofXyz], u32x4 (&xlo)[kKsNofXyz], const float (&xyz)[3], uint32_t par_xyz,
                      int h, bool pow2_xyz) {
  float emb[B2Xyz5::SLOTS];
  emb_eval<3, 5, HW>(emb, xyz, par_xyz, h, pow2_xyz);
  if constexpr (HF) split_operands_half<kKsNofXyz>(emb, B2Xyz5::SLOTS, xhi, xlo);
  else if constexpr (T == 3) split_operands3<kKsNofXyz>(emb, B2Xyz5::SLOTS, xhi, xmid, xlo);
  else split_operands<kKsNofXyz>(emb, B2Xyz5::SLOTS, xhi, xlo);
}


// ================================================================== MF_PREC_BF16X3 (the contract mode of the bf16 pipe)
// EVERY matrix product of the NeRF as a two-term bf16 split of activations AND weights, x = hi + lo, three products
// hi*hi + hi*lo + lo*hi with fp32 accumulation (16 mantissa bits per operand: the dropped lo*lo term is 2^-16 relative);
// every product of the NoFs as a THREE-term split (hi, mid, lo: 24 mantissa bits, six products per k-step, template
// parameter T below) -- their output point feeds sin(512 x) of the canonical encoding, where the 2^-17 of two terms is 1.5e-4
// of the rendered ray (oracle/bf16_ref.py); the sigma / rgb heads as fp32 dot products on the fp32 accumulators inside the
// epilogue of the layer in front of them, the NoF's image index as the exact fp32 per-ray bias.  Measured through the MoCo
// chains at 4096 rays: 111-114 dB, max-rel <= 3.1e-5 against the fp32 oracle on both weight draws (the fast mode: 38-51 dB) --
// fp32-class results at three / six bf16 matrix instructions per fp32 one (the fp32 pipe costs sixteen).  Splitting only parts
// does not get there: the NoF alone in two terms 47-60 dB, + the NeRF's trunk 63-72 dB (round 3's first bf16x3 was the former).
//
// Registers: the (hi, lo) activations of a 256-wide layer are 128 registers per lane, its output as many: more than the
// 256 of a wave at two waves per SIMD.  The x3 kernels therefore run ONE wave per SIMD (workgroup = 4 waves = 128
// samples per pass over the weight stream) with the whole file: 256 VGPRs + 256 AGPRs, hipcc parks the finished output
// tiles in AGPRs (one v_accvgpr_write / _read per register and layer).
// LDS: a tile's groups stream as one panel up to 32 groups and as two half panels beyond (panel_cap, mf_core.hpp: the
// skip layer has 40, extra_encoding 36), so the ring's three slots stay 32 KiB.
//
// One output tile.  Embedded block (NGE k-steps, always split) in front of (EMB_FIRST) or behind the KHID hidden k-steps;
// HMODE 0: hidden plain (one group, one MFMA per k-step), 2: hidden split (groups hi, lo; hi: two MFMAs, lo: one).
// OUTS: 0 no operand output, 1 hi, 2 (hi, lo).  NHEAD: head[o] += w_o[rows of this tile] . act(acc) in fp32, rows
// HSTRIDE bytes apart.  `two` = the panels two ahead of this tile's first / second panel.

struct Ahead { int g0; const char* j0; int g1; const char* j1; int s0 = -1, s1 = -1; };   // s0 / s1: g0 / g1 when static (StreamT::sync's sg)

// Activation dump of the training forward (mf_render_args.dump_acts): `row` = this lane's sample row + the layer's first
// column + 4 (lane >> 5); a tile's 16 accumulators are rows 8 q + 4 h + i, i.e. four 16-byte stores at row[32 t + 8 q].
// `mask` (round 5; with `masks`, uniform): this lane's ReLU BIT ROW of the layer -- the words the fp32 forward writes
// (relu_mask_word / relu_mask_shift, mf_core.hpp: word 4 (t >> 2) + g, byte t & 3, bit = feature of lane group g) and the x3
// backward chains read (relu_mask_pair): lane half h of this 32 x 32 tiling owns the bytes of groups g = h and g = 2 + h.
// Round 5: the ROW stores go through a buffer descriptor -- `rs` = the wave's first row of the plane (uniform), `voff` = this
// lane's byte offset from it (+ 16 (lane >> 5)), `soff` = the layer's first column in bytes (uniform); a lane without a row
// carries an offset no record reaches and the hardware drops its store.  As `if (on) *row = v` every one of the ~530 row stores of
// the sample-tile program sat in its own s_and_saveexec / s_cbranch_execz / s_or exec bracket: 1 600 scalar instructions and as
// many basic-block ends inside the MFMA schedule.
struct RowDump {
  float* row; bool on; bool wave_on; unsigned* mask = nullptr; bool masks = false;     // wave_on: some lane of the wave stores (uniform)
  float* rs = nullptr; uint32_t voff = 0; int soff = 0;      // (rs: the descriptor's base; the descriptor itself is made at the store -- four scalar moves the compiler keeps)
};
constexpr uint32_t kDumpNoRow = 0x7fffff00u;           // voff of a lane without a row = the descriptors' num_records
// the descriptor + lane offset of a dump plane: `row` = this lane's row index (ascending over the wave: lane 0 holds the first), `on` its predicate
MF_D void dump_rows(RowDump& d, float* plane, long long row, long long stride, bool on, int h) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)row), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)row >> 32));
  const long long row0 = (long long)(((unsigned long long)hi << 32) | lo);
  d.rs = plane + row0 * stride;
  d.voff = on ? (uint32_t)((row - row0) * stride * 4) + 16u * (uint32_t)h : kDumpNoRow;
  d.soff = 0;
}
MF_D bool dump_wave_on(const RowDump& d) { return d.wave_on; }
struct NoDump {};
MF_D bool dump_wave_on(const NoDump&) { return true; }
// Tile t's two mask bytes from its accumulators (register 4 q + i = feature 8 q + 4 h + i: q = 0 / 2 -> low / high nibble of
// group h's byte, q = 1 / 3 -> of group 2 + h's), collected four tiles to a word and stored behind every fourth tile.
MF_D void mask_put(const RowDump& d, const f32x16& acc, int t, int h, unsigned& ma, unsigned& mb) {
  if (!d.masks) return;
  // (x > 0 as a bit costs a compare + select + shift-or per element, 48 VALU per tile: +0.33 ms per launch of the joint step's
  //  dumping forward against -0.7 ms in the two backward chains that read 64 bytes instead of 2 KiB per row)
  auto pos = [](float x) { return relu_bit(x); };                     // (v_med3_i32: mf_core.hpp)
  unsigned a = 0, b = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a |= pos(acc[i]) << i | pos(acc[8 + i]) << (4 + i);
    b |= pos(acc[4 + i]) << i | pos(acc[12 + i]) << (4 + i);
  }
  ma = ((t & 3) == 0 ? 0u : ma) | a << (8 * (t & 3));
  mb = ((t & 3) == 0 ? 0u : mb) | b << (8 * (t & 3));
  if ((t & 3) == 3 && d.on) { d.mask[4 * (t >> 2) + h] = ma; d.mask[4 * (t >> 2) + 2 + h] = mb; }
}
MF_D void mask_put(const NoDump&, const f32x16&, int, int, unsigned&, unsigned&) {}
// HF: the accumulators of a half-operand layer are 2^(kNofHalfSA + kNofHalfSW) x the pre-activations (mf_core.hpp): the dump holds
// the activations themselves
template <bool RELU, bool HF = false>
MF_D void dump_store(const RowDump& d, const f32x16& acc, int t, int q) {
  auto act = [](float x) {
    if (!RELU) return x;
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
  };
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = act(acc[4 * q + i]) * (HF ? kHalfInvAcc : 1.f);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.rs, 0, (int)kDumpNoRow, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(d.voff + (uint32_t)((32 * t + 8 * q) * 4)), d.soff, 0);
}
template <bool RELU, bool HF = false>
MF_D void dump_store(const NoDump&, const f32x16&, int, int) {}

MF_D f32x2 lds_f2(uint32_t byte_off) { return *(const f32x2*)(smem + byte_off); }

// Epilogue of one tile in 16 steps of ~5 instructions (a wave alone on its SIMD hides about five issues behind each
// MFMA, MI355X_MICROARCH.md): step 2u (u = 0..7) turns accumulator pair u into the hi dword (+ its share of the head dot
// products), step 2u + 1 into the lo dword.  Pair u < 4: registers 2u, 2u+1 -> out0[u]; u >= 4: registers 8 + 2(u-4), +1
// -> out1[u-4]  (C/D order: register 4q + i = row 8q + 4h + i of the tile).
template <bool RELU, int OUTS, int NHEAD, int HSTRIDE, bool HF = false>
MF_D void epi_step(const f32x16& acc, int step, int h, u32x4& out0, u32x4& out1, u32x4& lo0, u32x4& lo1, uint32_t headw_off,
                   float (&head)[NHEAD ? NHEAD : 1], u32x4& mid0, u32x4& mid1) {
  static_assert(!HF || (OUTS == 2 && NHEAD == 0), "half operands: (hi, lo) outputs, no fused head");
  // OUTS == 3 (three-term operands, the NoF under MF_PREC_BF16X3): three steps per pair -- hi, mid = bf16(v - hi),
  // lo = bf16(v - hi - mid) -- 24 in all; else two (hi, lo), 16 in all
  constexpr int SPP = OUTS == 3 ? 3 : 2;
  const int u = step / SPP, ph = step % SPP, w = u & 3, r = u < 4 ? 2 * u : 8 + 2 * (u - 4);
  // ReLU as a signed-integer max: every negative float (and -0) is a negative int32 -- one v_max_i32 where fmaxf's IEEE
  // canonicalisation costs two v_max_f32
  auto relu = [](float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); };
  // HF: the accumulators are 2^(SA + SW) x the pre-activation, the next operand 2^SA x the activation: un-scale by 2^-SW (exact)
  const float v0 = (RELU ? relu(acc[r]) : acc[r]) * (HF ? kHalfInvW : 1.f), v1 = (RELU ? relu(acc[r + 1]) : acc[r + 1]) * (HF ? kHalfInvW : 1.f);
  u32x4& hv = u < 4 ? out0 : out1;
  if (ph == 0) {
    if constexpr (NHEAD > 0) {                               // rows 8q + 4h + i: q = r / 4, i = r % 4
      const uint32_t so = headw_off + 16 * h + 32 * (r >> 2) + 4 * (r & 3);
#pragma unroll
      for (int o = 0; o < NHEAD; ++o) {
        const f32x2 wv = lds_f2(so + o * HSTRIDE);
        head[o] = __builtin_fmaf(wv[1], v1, __builtin_fmaf(wv[0], v0, head[o]));
        asm volatile("" : "+v"(head[o]));
      }
    }
    if constexpr (OUTS > 0) {
      unsigned hi = HF ? pack_f16x2(v0, v1) : pack_bf16x2(v0, v1);
      // (the register file is full: each packed dword must exist HERE -- left alone, hipcc sinks the pure convert chain to
      //  the outputs' first use in the next layer and keeps every tile's accumulators alive until then)
      asm volatile("" : "+v"(hi));
      hv[w] = hi;
    }
  } else if constexpr (OUTS == 2) {
    const unsigned hi = hv[w];
    unsigned lo = HF ? pack_f16x2(v0 - hflo(hi), v1 - hfhi(hi)) : pack_bf16x2(v0 - bflo(hi), v1 - bfhi(hi));
    asm volatile("" : "+v"(lo));
    (u < 4 ? lo0 : lo1)[w] = lo;
  } else if constexpr (OUTS == 3) {
    const unsigned hi = hv[w];
    const float r0 = v0 - bflo(hi), r1 = v1 - bfhi(hi);          // exact
    if (ph == 1) {
      unsigned mid = pack_bf16x2(r0, r1);
      asm volatile("" : "+v"(mid));
      (u < 4 ? mid0 : mid1)[w] = mid;
    } else {
      const unsigned mid = (u < 4 ? mid0 : mid1)[w];
      unsigned lo = pack_bf16x2(r0 - bflo(mid), r1 - bfhi(mid));
      asm volatile("" : "+v"(lo));
      (u < 4 ? lo0 : lo1)[w] = lo;
    }
  }
}

// extra MFMAs of group gi of a tile beyond its first one: a k-step of T terms is the groups W_0 .. W_{T-1} (hi, [mid,] lo), and
// group W_t multiplies the operand terms x_0 .. x_{T-1-t} -- every product down to 2^(-8 (T-1)): T (T + 1) / 2 per k-step
template <int T, int NEG, int NHG, int HMODE, bool EMB_FIRST>
MF_D constexpr int x_extras(int gi) {
  const int ge = EMB_FIRST ? gi : gi - NHG, gh = EMB_FIRST ? gi - NEG : gi;
  if (ge >= 0 && ge < NEG) return T - 1 - ge % T;
  if (HMODE == 2 && gh >= 0 && gh < NHG) return T - 1 - gh % T;
  return 0;
}
template <int T, int NEG, int NHG, int HMODE, bool EMB_FIRST>
MF_D constexpr int x_slots(int lo, int hi) {                 // groups of [lo, hi) with at least one extra MFMA
  int n = 0;
  for (int g = lo; g < hi; ++g) n += x_extras<T, NEG, NHG, HMODE, EMB_FIRST>(g) >= 1 ? 1 : 0;
  return n;
}

// The matrix part of one output tile: acc = bias + W_tile [emb ; (hid, hidlo)] in three products per k-step.  `gap(m)`
// runs behind the m-th MFMA (m = 0 .. NM-1): the caller's deferred work (the previous tile's epilogue steps).
// KEEP: see StreamT::sync (the tile's FIRST panel barrier only).  LATE_FREE: no LDS-DMA pieces behind the last two (hi, lo)
// group pairs of the tile's last panel -- the caller's dump stores go there, behind every piece of the panel.
template <int NGE, int KHID, int HMODE, bool EMB_FIRST, int KEEP = 0, bool LATE_FREE = false, int T = 2, bool HF = false, class ST, class Gap>
MF_D void mma_tile_x(ST& st, const Lane& id, CarryX& carry, const u32x4* hid, const u32x4* hidlo, const u32x4* xhi, const u32x4* xlo,
                     uint32_t bias_off, const Ahead& two, f32x16& acc, Gap&& gap, bool keep_ok = true,
                     const u32x4* hidmid = nullptr, const u32x4* xmid = nullptr) {
  static_assert(T == 2 || T == 3, "operand terms");
  constexpr int NEG = T * NGE;
  constexpr int NHG = (HMODE ? T : 1) * KHID;
  constexpr int NG = NEG + NHG;
  constexpr int NSEG = NG > 32 ? 2 : 1;
  constexpr int NG1 = NSEG == 2 ? (NG + 1) / 2 : NG;         // groups of the first panel (panel_cap)
  static_assert(NG > PDX && NG - NG1 != 1 && (NSEG == 1 || NG - NG1 > PDX), "panel shorter than the fragment pipeline");
  static_assert(NG <= 64, "tile longer than two panels");
  const int h = id.h;
  const uint32_t b0 = st.off0 + id.lane * 16, b1 = st.off1 + id.lane * 16, b2 = st.off2 + id.lane * 16;
  auto frag = [&](int g) {                                   // group g of this tile; g >= NG: of the tile behind it
    if (g < NG1) return lds_u4(b0 + g * kGroupBytes);
    if (g < NG) return lds_u4(b1 + (g - NG1) * kGroupBytes);
    return lds_u4((NSEG == 2 ? b2 : b1) + (g - NG) * kGroupBytes);
  };
  // (the bias read here stands in front of the tile's first MFMA; reading it into the freed accumulator set three MFMAs before
  //  the previous tile ends -- what the fast mode's mma_tile does -- was built and measured in round 5: no change, C3x 1.0127
  //  vs 1.0081 ms, 2.131e6 vs 2.135e6 cycles: this wave's previous MFMA and its fragment reads cover the round trip)
  acc = bias_acc(bias_off, h);
  u32x4 r[PDX + 1];
#pragma unroll
  for (int i = 0; i < PDX; ++i) r[i] = carry.w[i];
  int m = 0;                                                 // MFMAs issued (compile time after unrolling)
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int s = gi % (PDX + 1);
    const int ge = EMB_FIRST ? gi : gi - NHG, gh = EMB_FIRST ? gi - NEG : gi;
    const bool emb = ge >= 0 && ge < NEG;
    const bool split = emb || HMODE == 2;
    const int ks = emb ? ge / T : (HMODE ? gh / T : gh);     // k-step of the group within its block
    const int nx = x_extras<T, NEG, NHG, HMODE, EMB_FIRST>(gi);
    if (gi == NG1) st.advance();                             // second panel of the tile
    acc = mfma32<HF>(r[s], (emb ? xhi : hid)[ks], acc);      // W_t x_0
    __builtin_amdgcn_sched_barrier(0);
    const int sp = (gi + PDX) % (PDX + 1), nb = gi + PDX;
    if (nb < NG) r[sp] = frag(nb);
    // the panel's barrier + the DMA of the panel two ahead, behind its first group; that panel's pieces go into the MFMA
    // gaps that follow (several per gap where the panel is short)
    const int base = gi >= NG1 ? NG1 : 0, len = gi >= NG1 ? NG - NG1 : NG1, q = gi - base;
    if (q == 0) {
      if (base) st.sync(two.g1, two.j1, id, true, two.s1);
      else st.template sync<KEEP>(two.g0, two.j0, id, keep_ok, two.s0);
    }
    if (nb >= NG) r[sp] = frag(nb);
    gap(m++);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 1; e <= nx; ++e) {                          // W_t x_e: the lower-order products of this group
      const u32x4* op = T == 3 ? (e == 1 ? (emb ? xmid : hidmid) : (emb ? xlo : hidlo)) : (emb ? xlo : hidlo);
#ifdef MF_X3_BREAK_LO       // (the deliberately broken build the oracle-of-the-arithmetic tests must reject: W_0 x_last of the embedded blocks dropped)
      if (!(emb && e == nx && ge % T == 0))
#endif
      acc = mfma32<HF>(r[s], op[ks], acc);
      __builtin_amdgcn_sched_barrier(0);
      if (e == 1) {
        // the pieces of the panel two ahead go behind these second MFMAs -- gaps that carry no fragment read (an LDS-DMA
        // issued next to a ds_read_b128 costs the wave ~50 cycles, alone ~10) -- spread evenly over the panel
        const int nall = x_slots<T, NEG, NHG, HMODE, EMB_FIRST>(base, base + len);
        const int j = x_slots<T, NEG, NHG, HMODE, EMB_FIRST>(base, gi);
        const bool last_panel = base + len == NG;
        const int nslots = (LATE_FREE && last_panel && nall > 2) ? nall - 2 : nall;
        if (j < nslots) {
#pragma unroll
          for (int k = j * ST::kPieces / nslots; k < (j + 1) * ST::kPieces / nslots; ++k) st.piece(k, id);
        }
      }
      gap(m++);
      __builtin_amdgcn_sched_barrier(0);
    }
    (void)split;
  }
#pragma unroll
  for (int i = 0; i < PDX; ++i) carry.w[i] = r[(NG + i) % (PDX + 1)];
}

// the first two panels of a layer whose tiles are `g` groups long
MF_D Next next_of_groups(int g) { return g > 32 ? Next{(g + 1) / 2, nullptr, g / 2, nullptr} : Next{g, nullptr, g, nullptr}; }
template <int KH, int EKS, int T = 2>
MF_D Next next_x(const Net& n, int layer) { return next_of_groups(tgroups<KH, EKS, true, 1, T>(n, layer)); }

// One layer of NT tiles with split operands: (out, outlo) <- act(W [emb ; (in, inlo)] + b), `nxt` = what follows it.
// The epilogue of tile t runs in the MFMA gaps of tile t + 1 (a wave alone on its SIMD has nobody to cover it); only the
// last tile's stands alone.
// T = 3 (OUTS = 3): three-term operands -- `inmid` / `outmid` / `xmid` are the middle terms (unused with T = 2).
template <int NT, int NGE, int KHID, int HMODE, bool EMB_FIRST, bool RELU, int OUTS, int NHEAD, int HSTRIDE, int T = 2, bool HF = false, class DT, class ST, int KI, int KO>
MF_D void layer_x(ST& st, const Lane& id, CarryX& carry, const u32x4 (&in)[KI], const u32x4 (&inlo)[KI], u32x4 (&out)[KO],
                  u32x4 (&outlo)[KO], const u32x4* xhi, const u32x4* xlo, uint32_t bias_off, const Next& nxt, uint32_t headw_off,
                  float (&head)[NHEAD ? NHEAD : 1], const DT& dump, const u32x4* inmid = nullptr, u32x4* outmid = nullptr,
                  const u32x4* xmid = nullptr) {
  constexpr bool DUMP = __is_same(DT, RowDump);
  static_assert(OUTS != 3 || T == 3, "three-term outputs feed three-term layers");
  constexpr int NG = T * NGE + (HMODE ? T : 1) * KHID;
  constexpr int NSEG = NG > 32 ? 2 : 1;
  constexpr int NG1 = NSEG == 2 ? (NG + 1) / 2 : NG;
  constexpr int PK = T * (T + 1) / 2;                        // products per split k-step
  constexpr int NM = PK * NGE + (HMODE == 2 ? PK * KHID : KHID);      // MFMAs of a tile
  constexpr int kSteps = OUTS == 3 ? 24 : 16;
  static_assert(!DUMP || NM >= 8, "dump stores need four piece-free MFMA gaps");
  f32x16 pend = {};
  unsigned mask_a = 0, mask_b = 0;                           // the layer's ReLU bit words being collected (RowDump::mask)
  // DUMP: the pending tile's four row stores go into the LAST four MFMA gaps of the next tile, behind that tile's last
  // LDS-DMA piece (LATE_FREE), so the panel barrier that follows may leave exactly them in flight (KEEP = 4).  Which
  // barriers: the first of tile t >= 2 (tile t - 1 carried tile t - 2's stores) and the first of tile 0 (the layer in
  // front ended with its last tile's stores; a lower bound where other VM traffic sits in between).  Tile 1's waits for
  // everything: tile 0 carries no stores.
  auto epi = [&](const f32x16& a, int sidx, int tile) __attribute__((always_inline)) {
    u32x4* om = OUTS == 3 ? outmid : &outlo[0];              // (placeholders where a term does not exist)
    epi_step<RELU, OUTS, NHEAD, HSTRIDE, HF>(a, sidx, id.h, out[OUTS ? 2 * tile : 0], out[OUTS ? 2 * tile + 1 : 1], outlo[OUTS >= 2 ? 2 * tile : 0],
                                         outlo[OUTS >= 2 ? 2 * tile + 1 : 1], headw_off + 32 * tile * 4, head,
                                         om[OUTS == 3 ? 2 * tile : 0], om[OUTS == 3 ? 2 * tile + 1 : 1]);
  };
  auto run = [&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    Ahead two;
    if constexpr (NSEG == 1) {        // panel t + 2 of this layer, else panel t + 2 - NT of what follows
      two = Ahead{t + 2 < NT ? NG : (t == NT - 2 ? nxt.groups : nxt.groups2),
                  t == NT - 2 ? nxt.jump : (t == NT - 1 ? nxt.jump2 : nullptr), 0, nullptr, t + 2 < NT ? NG : -1, -1};
    } else {                          // the same half of the next tile, else of the first tile of what follows
      two = t + 1 < NT ? Ahead{NG1, nullptr, NG - NG1, nullptr, NG1, NG - NG1} : Ahead{nxt.groups, nxt.jump, nxt.groups2, nxt.jump2, -1, -1};
    }
    constexpr int tp = t > 0 ? t - 1 : 0;                    // the tile whose epilogue is pending
    auto gap = [&](int m) __attribute__((always_inline)) {
      if (t == 0) return;
#pragma unroll
      for (int sidx = kSteps * m / NM; sidx < kSteps * (m + 1) / NM; ++sidx) epi(pend, sidx, tp);
      if (DUMP && RELU && m == NM - 5) mask_put(dump, pend, tp, id.h, mask_a, mask_b);    // (older than the four row stores: KEEP = 4 still leaves exactly those in flight)
      if (DUMP && m >= NM - 4) dump_store<RELU, HF>(dump, pend, tp, m - (NM - 4));
    };
    f32x16 acc;
    constexpr int KEEP = (DUMP && t != 1) ? 4 : 0;
    mma_tile_x<NGE, KHID, HMODE, EMB_FIRST, KEEP, DUMP, T, HF>(st, id, carry, in, inlo, xhi, xlo, bias_off + 32 * t * 4, two, acc, gap,
                                                          dump_wave_on(dump), inmid, xmid);
    st.advance();
    pend = acc;
  };
  static_assert(NT == 4 || NT == 8, "tiles per layer");
  run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 1>{});
  run(std::integral_constant<int, 2>{}); run(std::integral_constant<int, 3>{});
  if constexpr (NT == 8) {
    run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 5>{});
    run(std::integral_constant<int, 6>{}); run(std::integral_constant<int, 7>{});
  }
#pragma unroll
  for (int sidx = 0; sidx < kSteps; ++sidx) epi(pend, sidx, NT - 1);
  if constexpr (DUMP) {
    if (RELU) mask_put(dump, pend, NT - 1, id.h, mask_a, mask_b);
#pragma unroll
    for (int q = 0; q < 4; ++q) dump_store<RELU, HF>(dump, pend, NT - 1, q);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// One trunk layer: MODE as trunk_layer_m (1 embedded input only, 2 hidden only, 3 both, embedded input first).
template <int KH, int NGE, int MODE, bool RELU, int NHEAD, int T = 2, bool HF = false, class RBT, class ST, class DT = NoDump>
MF_D void trunk_layer_x(const Net& net, int layer, const u32x4 (&in)[KH], const u32x4 (&inlo)[KH], u32x4 (&out)[KH],
                        u32x4 (&outlo)[KH], const u32x4 (&xhi)[NGE], const u32x4 (&xlo)[NGE], ST& st, CarryX& carry,
                        const Lane& id, const Next& nxt, const RBT& rb, uint32_t headw_off, float (&head)[NHEAD ? NHEAD : 1],
                        const DT& dump = DT{}, const u32x4* inmid = nullptr, u32x4* outmid = nullptr, const u32x4* xmid = nullptr) {
  uint32_t bias_off = net.res_lds + layer * (16 * KH) * 4;
  if constexpr (__is_same(RBT, LdsRayBias) && (MODE & 1))
    bias_off = rb.lane_off + (uint32_t)__builtin_popcount(net.emb_mask & ((1u << layer) - 1u)) * (16 * KH) * 4;
  layer_x<KH / 2, (MODE & 1) ? NGE : 0, (MODE & 2) ? KH : 0, 2, true, RELU, T, NHEAD, 0, T, HF>(
      st, id, carry, in, inlo, out, outlo, xhi, xlo, bias_off, nxt, headw_off, head, dump, inmid, outmid, xmid);
}

// NoF head with split activations and weights: T groups per k-step (Whi, [Wmid,] Wlo), T (T + 1) / 2 products.
template <int KHID, int T = 2, bool HF = false, class ST>
MF_D f32x16 head_tile_x3(ST& st, const Lane& id, CarryX& carry, const u32x4* hid, const u32x4* hidlo, uint32_t bias_off, int g2,
                         const char* j2, const u32x4* hidmid = nullptr) {
  constexpr int NG = T * KHID;
  const int h = id.h;
  const uint32_t p = st.off0 + id.lane * 16, pn = st.off1 + id.lane * 16;
  f32x16 acc;
  {
    const f32x4 b0 = lds_f4(bias_off + (0 + 4 * h) * 4), b1 = lds_f4(bias_off + (8 + 4 * h) * 4);
    const f32x4 b2 = lds_f4(bias_off + (16 + 4 * h) * 4), b3 = lds_f4(bias_off + (24 + 4 * h) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = b0[i]; acc[4 + i] = b1[i]; acc[8 + i] = b2[i]; acc[12 + i] = b3[i]; }
  }
  u32x4 r[PDX + 1];
#pragma unroll
  for (int i = 0; i < PDX; ++i) r[i] = carry.w[i];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int s = gi % (PDX + 1);
    acc = mfma32<HF>(r[s], hid[gi / T], acc);
    __builtin_amdgcn_sched_barrier(0);
    const int sp = (gi + PDX) % (PDX + 1), nb = gi + PDX;
    if (nb < NG) r[sp] = lds_u4(p + nb * kGroupBytes);
    if (gi == 0) st.sync(g2, j2, id);
    if (gi >= 1 && gi <= ST::kPieces) st.piece(gi - 1, id);
    if (nb >= NG) r[sp] = lds_u4(pn + (nb - NG) * kGroupBytes);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 1; e <= T - 1 - gi % T; ++e) {
      acc = mfma32<HF>(r[s], (T == 3 && e == 1 ? hidmid : hidlo)[gi / T], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < PDX; ++i) carry.w[i] = r[(NG + i) % (PDX + 1)];
  return acc;
}

// T = kNofTermsX3 terms per operand: with T = 3 every product of the NoF carries 24 mantissa bits -- its output point feeds
// sin(512 x) of the canonical encoding, where the 2^-17 of a two-term split is ~1e-4 of the rendered ray (oracle/bf16_ref.py,
// tools/bf16_explore.py: C3 on the dense draw 1.5e-4 max-rel with two terms, 1.4e-5 with three = the exact-fp32 NoF's)
// DT = RowDump (the training forward under NoF, round 5): the evaluation's row [h_1 .. h_D | T (9 | 3) zero-padded to 16] -- what
// mf_nof_backward3 / mf_weight_grads read (dump.row = row start + 4 (lane >> 5)); no ReLU bit rows (the caller's row has no room
// for them: the backward then reads the activations)
template <int T = 2, bool HF = false, class RBT, class AfterFirst, class ST, class DT = NoDump>
MF_D void nof_eval_x3(const Net& net, const u32x4 (&xhi)[kKsNofXyz], const u32x4 (&xmid)[kKsNofXyz], const u32x4 (&xlo)[kKsNofXyz],
                      const float (&xyz)[3], ST& st, CarryX& carry, const Lane& id, const Next& follow, float (&out)[3], const RBT& rb,
                      AfterFirst&& after_first, const DT& dump = DT{}) {
  u32x4 ah[8], am[T == 3 ? 8 : 1], al[8], bh[8], bm[T == 3 ? 8 : 1], bl[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) { ah[t][i] = 0; al[t][i] = 0; if (T == 3) am[T == 3 ? t : 0][i] = 0; }
  const int D = net.D;
  const Next hd{T * 8, nullptr, follow.groups, follow.jump};
  float nohead[1] = {0.f};
  auto one = [&](int layer, const u32x4 (&ih)[8], const u32x4* im, const u32x4 (&il)[8], u32x4 (&oh)[8], u32x4* om, u32x4 (&ol)[8]) __attribute__((always_inline)) {
    const Next nxt = layer == D - 1 ? hd : next_x<8, kKsNofXyz, T>(net, layer + 1);
    const int has_emb = (net.emb_mask >> layer) & 1;
    auto dump_at = [&](int col, int lyr) {                    // (the NoF's bit rows: 4 words per layer)
      if constexpr (__is_same(DT, RowDump)) return RowDump{dump.row + col, dump.on, dump.wave_on, dump.mask + 4 * lyr, dump.masks, dump.rs, dump.voff, col * 4};
      else return NoDump{};
    };
    const auto nd = dump_at(layer * 128, layer);
    if (layer == 0) trunk_layer_x<8, kKsNofXyz, 1, true, 0, T, HF>(net, layer, ih, il, oh, ol, xhi, xlo, st, carry, id, nxt, rb, 0u, nohead, nd, im, om, xmid);
    else if (has_emb) trunk_layer_x<8, kKsNofXyz, 3, true, 0, T, HF>(net, layer, ih, il, oh, ol, xhi, xlo, st, carry, id, nxt, rb, 0u, nohead, nd, im, om, xmid);
    else trunk_layer_x<8, kKsNofXyz, 2, true, 0, T, HF>(net, layer, ih, il, oh, ol, xhi, xlo, st, carry, id, nxt, rb, 0u, nohead, nd, im, om, xmid);
  };
  int l = 0;
  for (; l + 1 < D; l += 2) {
    one(l, ah, am, al, bh, bm, bl);
    if (l == 0) after_first();
    one(l + 1, bh, bm, bl, ah, am, al);
  }
  if (l < D) {
    one(l, ah, am, al, bh, bm, bl);
    if (l == 0) after_first();
#pragma unroll
    for (int t = 0; t < 8; ++t) { ah[t] = bh[t]; al[t] = bl[t]; if (T == 3) am[T == 3 ? t : 0] = bm[T == 3 ? t : 0]; }
  }
  // the head panel; the panel two ahead of it = the SECOND panel of whatever follows
  f32x16 acc = head_tile_x3<8, T, HF>(st, id, carry, ah, al, net.res_lds + (D + net.aux) * 128 * 4, follow.groups2, follow.jump2, am);
  st.advance();
  float own[5], oth[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) { own[i] = acc[i] * (HF ? kHalfInvAcc : 1.f); oth[i] = __shfl_xor(own[i], 32, 64); }
  float T9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (net.aux == 9) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { T9[i] = id.h ? oth[i] : own[i]; T9[4 + i] = id.h ? own[i] : oth[i]; }
    T9[8] = id.h ? oth[4] : own[4];
    quat_transform(T9, xyz, out);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) { T9[c] = id.h ? oth[c] : own[c]; out[c] = T9[c] + xyz[c]; }
  }
  if constexpr (HF) {
    // the packer's poison slot (mf_pack.hip): rows 27 / 31 of the head tile have zero weights and a bias of 0 -- or NaN when one
    // of this network's weights did not fit the half range; register 15 is that row in either lane half.  + 0 changes nothing.
    const float poison = acc[15];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] += poison;
  }
  if constexpr (__is_same(DT, RowDump)) {
    if (dump.on && id.h == 0) {                               // (h == 0: dump.row is the row's start)
      f32x4* tr = reinterpret_cast<f32x4*>(dump.row + D * 128);
      tr[0] = f32x4{T9[0], T9[1], T9[2], T9[3]};
      tr[1] = f32x4{T9[4], T9[5], T9[6], T9[7]};
      tr[2] = f32x4{T9[8], 0.f, 0.f, 0.f};
      tr[3] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// Canonical NeRF, x3: every layer in three products; the sigma head inside the epilogue of the last trunk layer, the rgb
// head inside extra_encoding's (whose outputs never become operands).  `make_extra(eh, el)` builds the extra block's split
// operands right before extra_encoding.
// DT = RowDump (training forward): every layer's activations [h_0 .. h_{D-1} | xyz_encoding_final | extra_encoding] go to the
// sample's dump row as they leave the accumulators (dump.row = row start + 4 (lane >> 5)).
template <class MakeExtra, class ST, class DT = NoDump>
MF_D void nerf_eval_x3(const Net& net, const u32x4 (&xh)[kKsNerfXyz], const u32x4 (&xl)[kKsNerfXyz], MakeExtra&& make_extra,
                       bool sigma_only, ST& st, CarryX& carry, const Lane& id, const Next& follow, float& sigma,
                       float (&rgb)[3], const DT& dump = DT{}) {
  auto dump_at = [&](int col) {                               // the dump of a layer whose first column is `col` (layer col / 256: 8 mask words each)
    if constexpr (__is_same(DT, RowDump)) return RowDump{dump.row + col, dump.on, dump.wave_on, dump.mask + 8 * (col / 256), dump.masks, dump.rs, dump.voff, col * 4};
    else return NoDump{};
  };
  u32x4 ah[16], al[16], bh[16], bl[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) { ah[t][i] = 0; al[t][i] = 0; }
  const int D = net.D;
  NoRayBias norb;
  float nohead[1] = {0.f};
  // (always_inline: called three times; as a real function its array arguments would live in scratch)
  auto one = [&](int layer, const u32x4 (&ih)[16], const u32x4 (&il)[16], u32x4 (&oh)[16], u32x4 (&ol)[16]) __attribute__((always_inline)) {
    const Next nxt = next_x<16, kKsNerfXyz>(net, layer + 1);
    const int has_emb = (net.emb_mask >> layer) & 1;
    const auto dl = dump_at(layer * 256);
    if (layer == 0) trunk_layer_x<16, kKsNerfXyz, 1, true, 0>(net, layer, ih, il, oh, ol, xh, xl, st, carry, id, nxt, norb, 0u, nohead, dl);
    else if (has_emb) trunk_layer_x<16, kKsNerfXyz, 3, true, 0>(net, layer, ih, il, oh, ol, xh, xl, st, carry, id, nxt, norb, 0u, nohead, dl);
    else trunk_layer_x<16, kKsNerfXyz, 2, true, 0>(net, layer, ih, il, oh, ol, xh, xl, st, carry, id, nxt, norb, 0u, nohead, dl);
  };
  int l = 0;                                                  // layers 0 .. D-2, in pairs a -> b -> a
  for (; l + 1 < D - 1; l += 2) {
    one(l, ah, al, bh, bl);
    st.tl.stamp(10 + l, id);
    one(l + 1, bh, bl, ah, al);
    st.tl.stamp(11 + l, id);
  }
  if (l < D - 1) {
    one(l, ah, al, bh, bl);
#pragma unroll
    for (int t = 0; t < 16; ++t) { ah[t] = bh[t]; al[t] = bl[t]; }
  }
  // resident block: [bias_trunk (D+1) 256 | bias_extra 128 | sigma_w 256 | sigma_b 4 | rgb_w 384 | rgb_b 4]
  const uint32_t r_sigma_w = net.res_lds + ((D + 1) * 256 + 128) * 4;
  float sig[1] = {0.f};
  {                                                           // layer D-1 (a -> b), the sigma head in its epilogue
    const Next nx = sigma_only ? follow : next_x<16, kKsNerfXyz>(net, D);      // xyz_encoding_final: 32 groups
    const auto dl = dump_at((D - 1) * 256);
    if ((net.emb_mask >> (D - 1)) & 1)
      trunk_layer_x<16, kKsNerfXyz, 3, true, 1>(net, D - 1, ah, al, bh, bl, xh, xl, st, carry, id, nx, norb, r_sigma_w, sig, dl);
    else
      trunk_layer_x<16, kKsNerfXyz, 2, true, 1>(net, D - 1, ah, al, bh, bl, xh, xl, st, carry, id, nx, norb, r_sigma_w, sig, dl);
  }
  sigma = sig[0] + __shfl_xor(sig[0], 32, 64) + lds_f(r_sigma_w + 256 * 4);
  st.tl.stamp(30, id);
  if (sigma_only) return;
  const Next ex = next_of_groups(32 + 2 * net.aux);
  trunk_layer_x<16, kKsNerfXyz, 2, false, 0>(net, D, bh, bl, ah, al, xh, xl, st, carry, id, ex, norb, 0u, nohead, dump_at(D * 256));   // xyz_encoding_final (b -> a, no ReLU)
  st.tl.stamp(31, id);
  u32x4 eh[kKsExtraMax], el[kKsExtraMax];
  make_extra(eh, el);
  st.tl.stamp(32, id);
  // extra_encoding (nerf.py:98): W/2 outputs from [final (W) ; extra block], ReLU; rgb (nerf.py:57-59) in its epilogue
  float o[3] = {0.f, 0.f, 0.f};
  const uint32_t bias_extra = net.res_lds + (D + 1) * 256 * 4, r_rgb_w = r_sigma_w + (256 + 4) * 4;
  const auto de = dump_at((D + 1) * 256);
  if (net.aux == 2) layer_x<4, 2, 16, 2, false, true, 0, 3, 512>(st, id, carry, ah, al, bh, bl, eh, el, bias_extra, follow, r_rgb_w, o, de);
  else if (net.aux == 1) layer_x<4, 1, 16, 2, false, true, 0, 3, 512>(st, id, carry, ah, al, bh, bl, eh, el, bias_extra, follow, r_rgb_w, o, de);
  else layer_x<4, 0, 16, 2, false, true, 0, 3, 512>(st, id, carry, ah, al, bh, bl, eh, el, bias_extra, follow, r_rgb_w, o, de);
  st.tl.stamp(33, id);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = o[c] + __shfl_xor(o[c], 32, 64) + lds_f(r_rgb_w + (384 + c) * 4);
    rgb[c] = 1.f / (1.f + expf(-t));                          // nn.Sigmoid, nerf.py:57-59
  }
}

}  // namespace bf
}  // namespace mf
