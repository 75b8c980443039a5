// mf_wgrad.hip -- weight / bias gradients of the NeRF layers:  dW_i = G_i^T X_i,  db_i = sum_s G_i[s]
//
// What it replaces: the dW / db halves of torch's addmm backward for the 12 nn.Linear of
// models/nerf.py:78-102 under loss.backward() (trainer/base.py:188-197).  G_i are column slices of the
// pre-activation gradient buffer written by mf_nerf_backward, X_i column slices of the forward's
// activation dump (or the embedded inputs): both (P, .) row-major with P ~ 1e6 samples, outputs at most
// 256 x 256.  Library GEMMs handle this shape (tiny M,N; K = P; strided operands) at 30-60 % of the
// matrix peak and need ~25 launches; here ALL layers run in ONE persistent launch:
//   * the contraction runs over SAMPLES: v_mfma_f32_16x16x4_f32 with A = G^T (16 out-features x 4
//     samples) and B = X (4 samples x 16 in-features); a workgroup owns the whole (n_out x n_in) output
//     of one item in registers (8 waves x up to 4x8 accumulator tiles) and streams 16-sample stages of
//     (G, X) rows HBM -> LDS with LDS-DMA through a 3-slot ring (one barrier per stage);
//   * the linearised (item, stage) space is cut into #CU equal-cost ranges, one per workgroup, so
//     each workgroup flushes at most a few partial results; partials go to a scratch buffer and a
//     second tiny kernel sums them in a fixed order (deterministic, no atomics);
//   * db falls out of the G tile already in LDS (one column per thread).
#include "mf_host.hpp"
#include "mf_core.hpp"
#include <type_traits>
#include <cstddef>
#include <cstdlib>

namespace mf {

int device_cus();   // mf_forward.hip

constexpr int kWgStage = 16;          // samples per stage
constexpr int kWgMaxItems = MF_WG_MAX_ITEMS;

// Output-block shapes (n_out x n_in), the 8 waves arranged (8 / WAVES_C) x WAVES_C, each wave WR x WC
// 16x16 tiles.  GW = valid G columns per row (the rest of the 16-row minimum is zero).
template <int NOUT_, int NIN_, int WR_, int WC_, int WAVES_C_, int GW_>
struct WgShape {
  static constexpr int NOUT = NOUT_, NIN = NIN_, WR = WR_, WC = WC_, WAVES_C = WAVES_C_, GW = GW_;
  static constexpr int PG = NOUT + 4, PX = NIN + 4;                  // LDS pitches (floats): pitch % 8 == 4
  static constexpr int SLOT_BYTES = kWgStage * (PG + PX) * 4;
  static constexpr int OUT_FLOATS = NOUT * NIN + NOUT;               // partial: dW then db
  static_assert((8 / WAVES_C) * WR * 16 == NOUT && WAVES_C * WC * 16 == NIN, "wave tiling must cover the block");
};
using ShapeA = WgShape<256, 256, 4, 8, 2, 256>;   // hidden x hidden
using ShapeB = WgShape<256, 64, 2, 4, 1, 256>;    // hidden x embedded xyz (63 -> 64)
using ShapeC = WgShape<128, 256, 4, 4, 4, 128>;   // extra_encoding x final
using ShapeD = WgShape<128, 32, 1, 2, 1, 128>;    // extra_encoding x embedded dir / ind (<= 32)
using ShapeE = WgShape<16, 640, 1, 5, 8, 4>;      // heads: [d rgb(3), d sigma] x [h_D | (final: not fetched, dW = 0) | extra]
using ShapeF = WgShape<128, 128, 2, 4, 2, 128>;   // NoF hidden x hidden
using ShapeG = WgShape<128, 80, 1, 5, 1, 128>;    // NoF hidden x embedded input (66 -> 80)
using ShapeH = WgShape<16, 128, 1, 1, 8, 12>;     // NoF head: d T (9 | 3, padded 12) x h_D

// MF_PREC_BF16X3 variants of the large blocks (shape ids 8..): the contraction on v_mfma_f32_32x32x16_bf16 with G and X as
// (hi, lo) bf16 pairs, three products per 16-sample k-step (wg_segment_x3).  8 waves as (8 / WAVES_C) x WAVES_C, each wave
// WR x WC tiles of 32 x 32.
template <int NOUT_, int NIN_, int WR_, int WC_, int WAVES_C_, int KS_ = 1, int SETS_ = 3>
struct WgShapeX {
  static constexpr int NOUT = NOUT_, NIN = NIN_, WR = WR_, WC = WC_, WAVES_C = WAVES_C_, KS = KS_, SETS = SETS_;   // KS: k-steps (of 16 samples) per step
  static constexpr int KSTEP_BYTES = (NOUT + NIN) * 2 * 16;         // one k-step of one of (hi | lo): both sample octets of every feature
  static constexpr int PART_BYTES = KS * KSTEP_BYTES;
  static constexpr int BUF_BYTES = 2 * PART_BYTES;                  // a step's fragments: hi + lo
  static constexpr int OUT_FLOATS = NOUT * NIN + NOUT;
  static_assert((8 / WAVES_C) * WR * 32 == NOUT && WAVES_C * WC * 32 == NIN, "wave tiling must cover the block");
  static_assert((NOUT + NIN) * KS <= kThreads && NOUT % 128 == 0 && NIN % 128 == 0, "one conversion unit per thread, a wave inside one octet");
  static_assert((KS * WR) % 2 == 0 && 4 % (KS * WR - KS * WR / 2) == 0, "the conversion pieces are dealt to the groups of the second half");
};
using ShapeAX = WgShapeX<256, 256, 4, 2, 4>;      // hidden x hidden
using ShapeCX = WgShapeX<128, 256, 2, 2, 4>;      // extra_encoding x final
// The NoF's blocks (128 x 128, 128 x 80, 12 x 128) as ONE 128 x 128 shape with the operands' valid widths in the item (the
// columns beyond them are zero: their lanes load nothing), two k-steps per step so that all 512 threads own a conversion unit
// (with one, half of the waves load nothing and 32 KiB per CU in flight do not cover the HBM's latency: round 4's +3 %).
using ShapeFX = WgShapeX<128, 128, 2, 1, 4, 2>;
using ShapeBX = WgShapeX<256, 128, 4, 1, 4>;      // hidden x embedded xyz (64 of the 128 columns valid; two waves without a unit)
constexpr int kWgShapeX0 = 8;                     // first x3 shape id
constexpr uint32_t kWgX3Dump = 3 * ShapeAX::BUF_BYTES;   // LDS: 4 KiB behind the three fragment buffers, written by waves without a conversion unit
static_assert(ShapeCX::BUF_BYTES <= ShapeAX::BUF_BYTES && ShapeFX::BUF_BYTES <= ShapeAX::BUF_BYTES && ShapeBX::BUF_BYTES <= ShapeAX::BUF_BYTES, "the dump area lies behind the largest shape's buffers");

// cost of one stage of each block shape in CU cycles, MEASURED (tools/bench_wgrad.py: each shape alone at 1.3 M samples,
// launch overhead subtracted).  Only the ratios matter: they decide where the linearised (item, stage) space is cut, and
// a shape that is under-priced by 20 % makes its workgroups -- and the launch -- 20 % late (the first-principles model
// max(MFMA, HBM) + 400 this replaces had C / D 20 % low and E 30 % high: the 13-item NeRF launch ran 12 % slower than
// its items one by one).
MF_HD int wg_stage_cost(int shape) {
  switch (shape) {
    case 0: return 9640;
    case 1: return 3700;
    case 2: return 5990;
    case 3: return 2500;
    case 4: return 3260;      // (the `final` columns not fetched: was 4790)
    case 5: return 3440;
    case 6: return 2690;
    case 7: return 2560;
    case 8: return 4420;      // x3 variants (MF_WGRAD=bf16x3 tools/bench_wgrad.py, round 5; planned apart from the fp32 shapes)
    case 9: return 3260;
    case 10: return 1750;     // 128 x 128 at full width (wg_item_dims prices the narrower items)
    default: return 2870;     // 256 x 128 with 64 valid X columns, priced BESIDE 256 x 256 workgroups (alone: 2 350 -- the chip's clock follows the others' matrix load)
  }
}
MF_HD int wg_out_floats(int shape) {
  switch (shape) {
    case 0: return ShapeA::OUT_FLOATS;
    case 1: return ShapeB::OUT_FLOATS;
    case 2: return ShapeC::OUT_FLOATS;
    case 3: return ShapeD::OUT_FLOATS;
    case 4: return ShapeE::OUT_FLOATS;
    case 5: return ShapeF::OUT_FLOATS;
    case 6: return ShapeG::OUT_FLOATS;
    case 7: return ShapeH::OUT_FLOATS;
    case 8: return ShapeAX::OUT_FLOATS;
    case 9: return ShapeCX::OUT_FLOATS;
    case 10: return ShapeFX::OUT_FLOATS;
    default: return ShapeBX::OUT_FLOATS;
  }
}

struct WgItem {
  const float* G; long long g_stride;
  const float* X; long long x_stride;
  int shape, want_bias;
  int cost;               // cycles per 16-sample stage (wg_stage_cost of the shape; the x3 128 x 128 shape: by the operands' widths)
  int gw, xw;             // x3 shapes: valid columns of G / X (even; = the shape's NOUT / NIN unless the item is narrower)
  int out_rows, out_cols; // layout of dW: (out_rows, out_cols), the top-left corner of the shape's (NOUT, NIN) block
  long long cost0;        // start of this item in the linearised cost space
  long long part_off;     // float offset of its first partial in the scratch buffer
  int slot0, n_slots;     // first workgroup touching it, number of partials
  int dense;              // every workgroup in [slot0, slot0 + n_slots) wrote a partial (false only for tiny P)
  float* dW; float* db;   // final (NOUT, NIN) / (NOUT)
};

struct WgParams {
  WgItem it[kWgMaxItems];
  int n_items, grid;
  long long P, stages, total_cost;
  float* scratch;
  int dbg;                // timing-ablation switches (MF_DEBUG_FLAGS; 0 in production): 1 = no half-stage stagger
};

// stage range [b, e) of item `it` that workgroup w owns (same integer formula on host and device)
MF_HD void wg_range(long long total, int grid, int w, long long cost0, int c, long long stages, long long& b, long long& e) {
  const long long lo = total * w / grid, hi = total * (w + 1) / grid;
  auto cut = [&](long long x) {
    long long q = x <= cost0 ? 0 : (x - cost0 + c - 1) / c;
    return q > stages ? stages : q;
  };
  b = cut(lo);
  e = cut(hi);
}

MF_D void lds_zero(uint32_t byte_off) { *(float*)(smem + byte_off) = 0.f; }

// Running source of one segment: this wave's two rows (wave, wave + 8) of the next stage to fetch.
// Held in registers -- re-reading the item from kernarg memory at every stage put four dependent
// s_load round trips between the barrier and the first MFMAs (measured 12 % of a 256x256 stage).
struct WgSource {
  const char* g;            // G row `wave` of the next stage (wave-uniform)
  const char* x;            // X row `wave` of the next stage
  long long g_stage, x_stage, g_half, x_half;   // byte steps: one stage (16 rows), half a stage (8 rows)
  long long row;            // sample index of that row
};

template <class S>
MF_D void wg_load_stage(WgSource& src, long long P, uint32_t slot, const LaneId& id) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int rr = id.wave + 8 * k;
    const uint32_t dg = slot + rr * S::PG * 4, dx = slot + kWgStage * S::PG * 4 + rr * S::PX * 4;
    if (src.row + 8 * k < P) {
      const char* gsrc = src.g + (k ? src.g_half : 0);
      const char* xsrc = src.x + (k ? src.x_half : 0);
      // (buffer form of the LDS-DMA: wave-uniform row address + lane * 16, see blds16)
      if (id.lane < S::GW / 4) blds16(gsrc, id.lane * 16, 0, dg);
#pragma unroll
      for (int c0 = 0; c0 < S::NIN / 4; c0 += 64) {
        if (S::NIN == 640 && c0 == 64) continue;     // heads block: columns 256..511 (`final`, read by no head) are not fetched
        if (c0 + id.lane < S::NIN / 4) blds16(xsrc, id.lane * 16, c0 * 16, dx + c0 * 16);
      }
    } else {      // past the last sample: the rows contribute nothing
      for (int c = id.lane; c < S::GW; c += 64) lds_zero(dg + c * 4);
      for (int c = id.lane; c < S::NIN; c += 64) lds_zero(dx + c * 4);
    }
  }
  src.g += src.g_stage;
  src.x += src.x_stage;
  src.row += kWgStage;
}

template <class S>
MF_D void wg_segment(const WgItem& it, long long sb, long long se, long long P, float* part, const LaneId& id, int dbg) {
  constexpr int WR = S::WR, WC = S::WC;
  const int tid = threadIdx.x;
  const int wr = id.wave / S::WAVES_C, wc = id.wave % S::WAVES_C;
  const int row0 = wr * WR * 16, col0 = wc * WC * 16;
  __syncthreads();                                   // previous segment's readers are done with the ring
  if (S::GW < S::NOUT) {                             // narrow G: the unused columns of every slot stay zero
    for (int i = tid; i < 3 * kWgStage * S::PG; i += kThreads) {
      const int sl = i / (kWgStage * S::PG), o = i % (kWgStage * S::PG);
      lds_zero(sl * S::SLOT_BYTES + o * 4);
    }
    if (S::NIN == 640) {                             // heads block: its unfetched X columns 256..511 stay zero (dW there = 0)
      for (int i = tid; i < 3 * kWgStage * 256; i += kThreads) {
        const int sl = i / (kWgStage * 256), r = (i / 256) % kWgStage, c = 256 + i % 256;
        lds_zero(sl * S::SLOT_BYTES + kWgStage * S::PG * 4 + (r * S::PX + c) * 4);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  f32x4 acc[WR][WC];
#pragma unroll
  for (int a = 0; a < WR; ++a)
#pragma unroll
    for (int b = 0; b < WC; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // db: column sums of the G tile already in LDS.  Spread over the whole workgroup -- thread t owns the column pair
  // (t mod NOUT/2) of the row group t / (NOUT/2): one ds_read_b64 + two adds per row -- instead of one column x 16 rows
  // on the first NOUT threads (that left half of the waves 16 LDS reads per stage behind the others at every
  // barrier: 6 % of a 256x256 item); the row groups' partial sums meet in LDS when the segment ends.
  // (Only where a stage is matrix-bound: on the lighter shapes -- measured on 256x64 and 128x256 -- the waves that
  //  had no column to sum were hiding the others' reads, and sharing the sums costs 4-16 %.)
  constexpr bool kSpread = S::WR * S::WC >= 32;
  constexpr int BC = kSpread ? S::NOUT / 2 : S::NOUT;               // column pairs (single columns when not spread)
  constexpr int BG = kSpread ? ((kThreads / BC) < kWgStage ? (kThreads / BC) : kWgStage) : 1;   // row groups
  constexpr int BR = kWgStage / BG;                                 // rows per group
  const int bcol = tid % BC, bgrp = tid / BC;
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool want_bias = it.want_bias != 0 && bgrp < BG;
  WgSource src;
  {
    const long long gs = it.g_stride * 4, xs = it.x_stride * 4;
    src.row = sb * kWgStage + id.wave;
    src.g = reinterpret_cast<const char*>(it.G) + src.row * gs;
    src.x = reinterpret_cast<const char*>(it.X) + src.row * xs;
    src.g_stage = gs * kWgStage; src.x_stage = xs * kWgStage;
    src.g_half = gs * 8; src.x_half = xs * 8;
  }
  wg_load_stage<S>(src, P, 0, id);
  if (sb + 1 < se) wg_load_stage<S>(src, P, S::SLOT_BYTES, id);
  uint32_t cur = 0;
  // lane (i = lane & 15, kg = lane >> 4), MFMA step m: sample row 4*kg + m of the stage
  const uint32_t aoff = ((4 * id.g) * S::PG + row0 + id.j) * 4;
  const uint32_t boff = kWgStage * S::PG * 4 + ((4 * id.g) * S::PX + col0 + id.j) * 4;
  // One barrier per stage.  At barrier `st` the data of stage st+1 is already complete (every wave
  // drained its share before arriving at barrier st-1 ... st), so the barrier may sit anywhere inside
  // stage st: waves 4-7 take it at the start, waves 0-3 (their SIMD partners) in the middle, which
  // keeps the two waves of a SIMD half a stage out of phase -- one is always in MFMA-dense code while
  // the other crosses a stage boundary (barrier, DMA issue, first fragment reads).
  const bool late = id.wave < kWaves / 2 && !(MF_TIMING_FLAGS && (dbg & 1));
  for (long long st = sb; st < se; ++st) {
    const uint32_t base = cur * S::SLOT_BYTES;
    auto hook = [&]() {      // (MF_WG_ABL_*: timing-ablation builds only, tools/ab_lib.sh; results are garbage there)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (st + 2 < se) wg_load_stage<S>(src, P, (cur >= 1 ? cur - 1 : 2) * S::SLOT_BYTES, id);
    };
    if (!late || st == sb) hook();      // (a segment's first stage has no earlier barrier to rely on)
    if (want_bias) {
#pragma unroll
      for (int s = 0; s < BR; ++s) {
        if constexpr (kSpread) {
          const float2 g2 = *reinterpret_cast<const float2*>(smem + base + ((bgrp * BR + s) * S::PG + 2 * bcol) * 4);
          bsum0 += g2.x;
          bsum1 += g2.y;
        } else {
          bsum0 += lds_f(base + (s * S::PG + tid) * 4);
        }
      }
    }
    float a[2][WR], b[2][WC];
#pragma unroll
    for (int t = 0; t < WR; ++t) a[0][t] = lds_f(base + aoff + (16 * t) * 4);
#pragma unroll
    for (int t = 0; t < WC; ++t) b[0][t] = lds_f(base + boff + (16 * t) * 4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = m & 1, n = c ^ 1;
      if (m == 2 && late && st != sb) hook();
      if (m + 1 < 4) {
#pragma unroll
        for (int t = 0; t < WR; ++t) a[n][t] = lds_f(base + aoff + ((m + 1) * S::PG + 16 * t) * 4);
#pragma unroll
        for (int t = 0; t < WC; ++t) b[n][t] = lds_f(base + boff + ((m + 1) * S::PX + 16 * t) * 4);
      }
#pragma unroll
      for (int ti = 0; ti < WR; ++ti)
#pragma unroll
        for (int tj = 0; tj < WC; ++tj) acc[ti][tj] = MF_MFMA(a[c][ti], b[c][tj], acc[ti][tj]);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  // partial result: C/D layout row = 4*(lane>>4) + r, col = lane & 15
#pragma unroll
  for (int ti = 0; ti < WR; ++ti)
#pragma unroll
    for (int tj = 0; tj < WC; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        part[(long long)(row0 + 16 * ti + 4 * id.g + r) * S::NIN + col0 + 16 * tj + id.j] = acc[ti][tj][r];
  // the row groups' column sums: through LDS (the ring is idle now), fixed order
  if constexpr (kSpread) {
    __syncthreads();
    if (bgrp < BG) *reinterpret_cast<float2*>(smem + (bgrp * S::NOUT + 2 * bcol) * 4) = make_float2(bsum0, bsum1);
    __syncthreads();
    if (tid < S::NOUT) {
      float b = 0.f;
#pragma unroll
      for (int g = 0; g < BG; ++g) b += lds_f((g * S::NOUT + tid) * 4);
      part[(long long)S::NOUT * S::NIN + tid] = b;
    }
  } else {
    if (tid < S::NOUT) part[(long long)S::NOUT * S::NIN + tid] = bsum0;
  }
}

// ---- MF_PREC_BF16X3: dW = G^T X with G and X as two-term bf16 splits, three products per k-step ----
// The fp32 matrix pipe does 16 samples x 256 x 256 in 9 640 cycles per CU; three bf16 products of (hi, lo) pairs carry 16
// mantissa bits per operand (the dropped lo*lo term is 2^-16 relative, the sum over samples accumulates in fp32 like the
// fp32 MFMA's) at a fifth of the matrix time -- the launch then runs against the HBM reads of its two operands.
// Step = KS k-steps of 16 samples.  Thread t owns one conversion unit per step -- two adjacent features x one sample octet of G
// or of X: eight 8-byte loads straight from HBM into a register set, (hi, lo) split in registers, four 16-byte LDS writes in
// the MFMA operand layout [hi | lo][k-step][operand][octet][feature][8 bf16] (A = G^T: 32 out-features x 16 samples, B = X:
// 16 samples x 32 in-features -- both "one feature, eight consecutive samples" per lane).  No raw staging in LDS, no LDS-DMA.
// db: the G units' column sums (fp32) as a by-product of the conversion.
//
// Round 5 -- the pipeline (profiles/r05_wgrad_x3.txt; before: 3.7 TB/s of operand reads, the stage's three phases in a row):
//   * two register sets: during step s a thread converts the set that holds step s + 2 and re-loads it with step s + 4, two
//     rows at a time behind the arithmetic that frees them (16 loads in flight per thread at every moment);
//   * THREE fragment buffers, ONE barrier per step: step s reads buffer s % 3 and writes the fragments of step s + 2 into the
//     buffer step s - 1 read.  After barrier s every wave has left step s - 1 (its buffer may be overwritten) and has written
//     step s + 1 (complete one whole step ahead) -- so the barrier may sit anywhere in step s before the wave's own fragment
//     writes: waves 4-7 take it at the start, waves 0-3 (their SIMD partners) in the middle, and the two waves of a SIMD run
//     half a step out of phase: one multiplies while the other crosses the barrier, reads its first fragments, writes its
//     own (with one buffer less and the barrier at the start for all, both waves of every SIMD stalled together: the matrix
//     pipe idle for 40 % of a step);
//   * the conversion in four pieces between the MFMA groups of the second half of the step.
typedef float wg_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg_bf2 __attribute__((ext_vector_type(2)));
MF_D unsigned wg_pack(float a, float b) {
  wg_f2 v; v[0] = a; v[1] = b;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_bf2));
}
typedef float f32x16w __attribute__((ext_vector_type(16)));

template <class S>
MF_D void wg_segment_x3(const WgItem& it, long long sb, long long se, long long P, float* part, const LaneId& id) {
  constexpr int WR = S::WR, WC = S::WC, KS = S::KS;
  constexpr int NG = KS * WR;                        // MFMA groups of a step: (k-step, row tile), WC tiles x 3 products each
  const int tid = threadIdx.x;
  const int lane = id.lane, li = lane & 31, lh = lane >> 5;
  const int wr = id.wave / S::WAVES_C, wc = id.wave % S::WAVES_C;
  const int row0 = wr * WR * 32, col0 = wc * WC * 32;
  // conversion role of this thread.  Everything but the feature pair is derived from the WAVE index (a scalar register):
  // operand, sample octet, source pointer and stride are wave-uniform (NOUT / 2 and NIN / 2 are multiples of 64), and the
  // compiler has to SEE that -- derived from threadIdx they were per-lane values to it, every buffer load sat in a
  // readfirstlane waterfall loop, and behind those loops its counter bookkeeping gave up: `s_waitcnt vmcnt(0)` at the head of
  // the stage loop, i.e. no load was ever in flight across a stage.
  const int w64 = id.wave * 64;
  const bool is_g = w64 < S::NOUT * KS, is_x = !is_g && w64 - S::NOUT * KS < S::NIN * KS;
  const bool unit = is_g || is_x;
  const int ub = is_g ? w64 : w64 - S::NOUT * KS, nf = is_g ? S::NOUT : S::NIN;
  const int uh = ub / (nf / 2), up = ub % (nf / 2) + lane;           // sample octet of the step (uniform), feature pair
  const float* src = is_g ? it.G : it.X;
  const long long stride = is_g ? it.g_stride : it.x_stride;
  const int width = is_g ? it.gw : it.xw;                            // valid columns of the operand (even; the rest of the block is zero)
  const uint32_t reg0 = is_g ? 0u : (uint32_t)(2 * S::NOUT * 16);    // byte offset of this operand's region inside a k-step
  // Fragment slots: within a 32-feature tile the EVEN features take slots 0..15, the odd ones 16..31 (feature f of the tile: slot
  // 16 (f & 1) + (f >> 1)).  A unit's two features then lie 256 bytes apart and the 64 lanes of one write instruction cover
  // contiguous 256-byte runs -- with the natural order (slot = feature) they were 32 bytes apart: a two-way bank conflict on every
  // fragment write.  The MFMA rows / columns are the slots; the partial store at the end of the segment maps them back.
  const uint32_t wdst = (uint32_t)((uh >> 1) * S::KSTEP_BYTES) + reg0 + (uint32_t)(((uh & 1) * nf + 32 * (up >> 4) + (up & 15)) * 16);
  // (no branch around the fragment writes: a wave without a unit -- 128 x 256 has two -- converts the zeros its loads return and
  //  writes them to a dump area behind the buffers; under a branch the compiler sinks the pieces' arithmetic into it, i.e. behind
  //  the step's MFMAs, and re-loads into fresh registers it then has to copy -- with a wait for the newest loads)
  const uint32_t wbase = unit ? wdst : kWgX3Dump + (uint32_t)lane * 16u, wlo = unit ? (uint32_t)S::PART_BYTES : 2048u;
  __syncthreads();                                   // previous segment's readers are done with the buffers
  f32x16w acc[WR][WC];
#pragma unroll
  for (int a = 0; a < WR; ++a)
#pragma unroll
    for (int b = 0; b < WC; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bs0 = 0.f, bs1 = 0.f;
  constexpr int NS = S::SETS;                        // register sets: 8 NS loads of 8 bytes in flight per thread
  wg_f2 raw[NS][8];
  // Loads through a buffer descriptor: wave-uniform 64-bit base (this wave's octet of the step) + ONE per-lane 32-bit offset
  // + a scalar row offset -- no 64-bit address per lane and row -- and `num_records` = the bytes up to sample P (0 past the
  // segment's last stage), so rows past the end read 0; a lane whose feature pair lies beyond the operand's width carries an
  // offset no record reaches.
  const int voff = 2 * up < width ? up * 8 : 0x7ffffff0;
  const long long row_bytes = stride * 4;
  typedef unsigned wg_u2 __attribute__((ext_vector_type(2)));
  auto desc = [&](long long step) {                  // step = index of the KS-stage step inside the segment's stage numbering
    const long long st = sb + step * KS + (uh >> 1);
    const long long s0 = st * kWgStage + 8 * (uh & 1);
    const long long left = (st < se && unit) ? (P - s0) * row_bytes : 0;
    const unsigned recs = left <= 0 ? 0u : (left > 0x7fffffe0LL ? 0x7fffffe0u : (unsigned)left);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(src + s0 * stride), 0, (int)recs, 0x00020000);
  };
  auto load1 = [&](wg_f2& r, const __amdgpu_buffer_rsrc_t& rs, int e) {
    r = __builtin_bit_cast(wg_f2, (wg_u2)__builtin_amdgcn_raw_buffer_load_b64(rs, voff, (int)(e * row_bytes), 0));
  };
  auto load = [&](wg_f2 (&r)[8], long long step) {
    const __amdgpu_buffer_rsrc_t rs = desc(step);
#pragma unroll
    for (int e = 0; e < 8; ++e) load1(r[e], rs, e);
  };
  // Conversion in four PIECES (w = sample pair 2w, 2w + 1 of the octet, both features of the unit): ~16 VALU each.
  auto piece = [&](wg_f2 (&r)[8], int w, u32x4 (&hv)[2], u32x4 (&lv)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float a0 = r[2 * w][c], a1 = r[2 * w + 1][c];
      const unsigned ha = wg_pack(a0, a1);
      hv[c][w] = ha;
      lv[c][w] = wg_pack(a0 - __builtin_bit_cast(float, ha << 16), a1 - __builtin_bit_cast(float, ha & 0xffff0000u));
      if (c == 0) bs0 += a0 + a1; else bs1 += a0 + a1;
    }
    // pinned HERE: free-floating arithmetic is placed next to its users (the fragment writes at the end of the step) when the
    // block is linearised, the re-loads of the rows it reads would then be issued before it -- into other registers, copied back
    // at the end of the trip behind a wait for the newest loads
    asm volatile("" : "+v"(hv[0][w]), "+v"(hv[1][w]), "+v"(lv[0][w]), "+v"(lv[1][w]), "+v"(bs0), "+v"(bs1) :: "memory");
  };
  auto put = [&](const u32x4 (&hv)[2], const u32x4 (&lv)[2], uint32_t buf) {
    const uint32_t o = wbase + (unit ? buf : 0u);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<u32x4*>(smem + o + 256 * c) = hv[c];
      *reinterpret_cast<u32x4*>(smem + o + wlo + 256 * c) = lv[c];
    }
  };
  // prologue: steps 0 and 1 into buffers 0 and 1, steps 2 .. 1 + NS on their way (set q holds step 2 + q)
  // (issued in the ORDER the loop keeps them in -- set 0 oldest: the compiler merges the counter state of this block with the
  //  loop's own at the loop head, and where the two orders differ it waits for the later position of every register)
  load(raw[0], 0);
  load(raw[1], 1);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    u32x4 hv[2], lv[2];
#pragma unroll
    for (int w = 0; w < 4; ++w) piece(raw[q], w, hv, lv);
    put(hv, lv, q * S::BUF_BYTES);
    load(raw[q], 2 + q);
  }
#pragma unroll
  for (int q = 2; q < NS; ++q) load(raw[q], 2 + q);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // fragment addresses of this lane: A rows row0 + 32 ti + li, B columns col0 + 32 tj + li, octet lh
  const uint32_t aoff = (uint32_t)((lh * S::NOUT + row0 + li) * 16);
  const uint32_t boff = (uint32_t)(2 * S::NOUT * 16 + (lh * S::NIN + col0 + li) * 16);
  const bool late = id.wave < kWaves / 2;
  uint32_t cur = 0, nxt = 2 * S::BUF_BYTES;          // byte offsets of the buffer this step reads / writes
  // The step loop, two steps per trip in straight-line code (PAR = the register set of the step): with the parity a run-time
  // variable the two sets met in one loop body through branches, and the compiler's load counter -- exact in straight-line
  // code: a piece waits for vmcnt(14), the other fourteen loads stay in flight -- degraded to vmcnt(0) at the merges.
  auto stage = [&](long long step, auto parc) {
    constexpr int PAR = decltype(parc)::value;
    jitter();                                                   // (race screen builds only, -DMF_DBG_JITTER)
    if (!late) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this thread's fragment writes of the step before
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const uint32_t ab = cur + aoff, bb = cur + boff;
    u32x4 bh[KS][WC], bl[KS][WC];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int t = 0; t < WC; ++t) {
        bh[k][t] = *reinterpret_cast<const u32x4*>(smem + bb + k * S::KSTEP_BYTES + t * 512);
        bl[k][t] = *reinterpret_cast<const u32x4*>(smem + bb + S::PART_BYTES + k * S::KSTEP_BYTES + t * 512);
      }
    u32x4 ah = *reinterpret_cast<const u32x4*>(smem + ab), al = *reinterpret_cast<const u32x4*>(smem + ab + S::PART_BYTES);
    const __amdgpu_buffer_rsrc_t rs = desc(step + 2 + NS);
    u32x4 hv[2], lv[2];
    constexpr int PPG = 4 / (NG - NG / 2);                      // pieces behind each group of the step's second half
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = g / WR, ti = g % WR;
      if (g == NG / 2) jitter();
      if (g == NG / 2 && late) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      u32x4 nh = ah, nl = al;
      if (g + 1 < NG) {                                         // the next group's A fragments while this one multiplies
        const int kn = (g + 1) / WR, tn = (g + 1) % WR;
        nh = *reinterpret_cast<const u32x4*>(smem + ab + kn * S::KSTEP_BYTES + tn * 512);
        nl = *reinterpret_cast<const u32x4*>(smem + ab + S::PART_BYTES + kn * S::KSTEP_BYTES + tn * 512);
      }
      // product-major: the three products of one accumulator are a dependent chain, the tiles of the group take turns
#pragma unroll
      for (int tj = 0; tj < WC; ++tj)
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[k][tj]), acc[ti][tj], 0, 0, 0);
#pragma unroll
      for (int tj = 0; tj < WC; ++tj)
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[k][tj]), acc[ti][tj], 0, 0, 0);
#pragma unroll
      for (int tj = 0; tj < WC; ++tj)
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[k][tj]), acc[ti][tj], 0, 0, 0);
      if (g >= NG / 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = (g - NG / 2) * PPG; w < (g - NG / 2 + 1) * PPG; ++w) {
          piece(raw[PAR], w, hv, lv);
          __builtin_amdgcn_sched_barrier(0);                    // (the loads stay BEHIND the arithmetic that frees their registers)
          load1(raw[PAR][2 * w], rs, 2 * w);
          load1(raw[PAR][2 * w + 1], rs, 2 * w + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ah = nh; al = nl;
    }
    put(hv, lv, nxt);
    nxt = cur;                                                  // ring of three: the next step writes into the buffer this one read
    cur = cur == 2 * S::BUF_BYTES ? 0u : cur + S::BUF_BYTES;
  };
  // (a segment whose step count is no multiple of NS runs up to NS - 1 steps past its end: `desc` hands out zeros there, the step adds nothing -- and
  //  the loop stays one basic block)
  const long long steps = (se - sb + KS - 1) / KS;
  for (long long q = 0; q < steps; q += NS) {
    stage(q, std::integral_constant<int, 0>{});
    stage(q + 1, std::integral_constant<int, 1>{});
    if constexpr (NS > 2) stage(q + 2, std::integral_constant<int, 2>{});
  }
  // partial result: C/D layout slot row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), slot column = lane & 31; slot s of a tile is
  // feature 2 s (s < 16) or 2 (s - 16) + 1: row feature = 2 (r & 3) + 16 ((r >> 2) & 1) + 8 (lane >> 5) + (r >> 3)
  // (the lane index is made opaque: hipcc otherwise computes the store addresses in front of the stage loop and carries
  //  them through it next to the 128 accumulators)
  int lo_ = lane;
  asm volatile("" : "+v"(lo_));
  const int lc = lo_ & 31;
  float* pl = part + (long long)(row0 + 8 * (lo_ >> 5)) * S::NIN + col0 + (lc < 16 ? 2 * lc : 2 * (lc - 16) + 1);
#pragma unroll
  for (int ti = 0; ti < WR; ++ti)
#pragma unroll
    for (int tj = 0; tj < WC; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        pl[(32 * ti + 2 * (r & 3) + 16 * ((r >> 2) & 1) + (r >> 3)) * S::NIN + 32 * tj] = acc[ti][tj][r];
  // db: the octets' column sums meet in LDS (the fragment buffers are idle now)
  __syncthreads();
  if (is_g) {
    *reinterpret_cast<float*>(smem + (uh * S::NOUT + 2 * up) * 4) = bs0;
    *reinterpret_cast<float*>(smem + (uh * S::NOUT + 2 * up + 1) * 4) = bs1;
  }
  __syncthreads();
  if (tid < S::NOUT) {
    float b = 0.f;
#pragma unroll
    for (int o = 0; o < 2 * KS; ++o) b += lds_f((o * S::NOUT + tid) * 4);
    part[(long long)S::NOUT * S::NIN + tid] = b;
  }
}

// The item table is read through the kernarg segment pointer (scalar loads of one item at a time): indexing the by-value
// `p.it[i]` with the runtime i made hipcc keep a private copy of all of WgParams in scratch (460 bytes per lane, 114
// VGPR + 136 SGPR spills in round 2's build).
MF_D WgItem wg_item(int i) {
  WgItem it;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) char* kptr;
  const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WgParams, it) + (size_t)i * sizeof(WgItem);
  __builtin_memcpy(&it, (const __attribute__((address_space(4))) void*)ka, sizeof(WgItem));
#endif
  return it;
}

// X3 = false: the fp32 block shapes (ids 0-7); true: the three-product shapes (ids 8-10).  Two kernels, because one
// kernel holding both families spills (hipcc hoists lane-derived values of every body in front of the item loop).
template <bool X3>
__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(WgParams p) {
  const LaneId id0;
  const int w = blockIdx.x;
  for (int i = 0; i < p.n_items; ++i) {
    const WgItem it = wg_item(i);
    long long b, e;
    wg_range(p.total_cost, p.grid, w, it.cost0, it.cost, p.stages, b, e);
    if (b >= e) continue;
    float* part = p.scratch + it.part_off + (long long)(w - it.slot0) * wg_out_floats(it.shape);
    // The lane / wave indices are made opaque per item: everything the shape bodies derive from them (fragment
    // offsets, tile origins, bias columns) is invariant in this loop, and hipcc hoists all of it in front of the loop
    // -- ~80 registers live across the 256x256 body with its 128 accumulators: 114 VGPR spills, 460 bytes of scratch per
    // lane in round 2's build.
    LaneId id = id0;
    asm volatile("" : "+v"(id.lane), "+v"(id.j), "+v"(id.g), "+s"(id.wave));
    if constexpr (!X3) {
      switch (it.shape) {
        case 0: wg_segment<ShapeA>(it, b, e, p.P, part, id, p.dbg); break;
        case 1: wg_segment<ShapeB>(it, b, e, p.P, part, id, p.dbg); break;
        case 2: wg_segment<ShapeC>(it, b, e, p.P, part, id, p.dbg); break;
        case 3: wg_segment<ShapeD>(it, b, e, p.P, part, id, p.dbg); break;
        case 4: wg_segment<ShapeE>(it, b, e, p.P, part, id, p.dbg); break;
        case 5: wg_segment<ShapeF>(it, b, e, p.P, part, id, p.dbg); break;
        case 6: wg_segment<ShapeG>(it, b, e, p.P, part, id, p.dbg); break;
        default: wg_segment<ShapeH>(it, b, e, p.P, part, id, p.dbg); break;
      }
    } else {
      switch (it.shape) {
        case kWgShapeX0: wg_segment_x3<ShapeAX>(it, b, e, p.P, part, id); break;
        case kWgShapeX0 + 1: wg_segment_x3<ShapeCX>(it, b, e, p.P, part, id); break;
        case kWgShapeX0 + 2: wg_segment_x3<ShapeFX>(it, b, e, p.P, part, id); break;
        default: wg_segment_x3<ShapeBX>(it, b, e, p.P, part, id); break;
      }
    }
  }
}

// fixed-order sum of the partials of every item
__global__ void wgrad_reduce_kernel(WgParams p) {
  const int i = blockIdx.y;
  if (i >= p.n_items) return;
  const WgItem it = wg_item(i);
  const int nf = wg_out_floats(it.shape);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nf) return;
  const float* src = p.scratch + it.part_off + e;
  // fixed order, summed in double: up to 256 partials of mixed sign -- a weight gradient is a heavily cancelling sum
  // over 1e5..1e6 samples, and a 256-long fp32 chain cost 4x the reference's own fp32 error on layer-0 gradients
  // (tests/test_gpu_shapes.py::test_stage1_training_shape_vs_oracle)
  double s = 0.0;
  if (it.dense) {
    for (int k = 0; k < it.n_slots; ++k) s += src[(long long)k * nf];
  } else {
    for (int k = 0; k < it.n_slots; ++k) {
      long long b, en;    // workgroups whose range of this item is empty (tiny P) wrote nothing
      wg_range(p.total_cost, p.grid, it.slot0 + k, it.cost0, it.cost, p.stages, b, en);
      if (b < en) s += src[(long long)k * nf];
    }
  }
  int nout, nin;
  switch (it.shape) {
    case 0: nout = ShapeA::NOUT; nin = ShapeA::NIN; break;
    case 1: nout = ShapeB::NOUT; nin = ShapeB::NIN; break;
    case 2: nout = ShapeC::NOUT; nin = ShapeC::NIN; break;
    case 3: nout = ShapeD::NOUT; nin = ShapeD::NIN; break;
    case 4: nout = ShapeE::NOUT; nin = ShapeE::NIN; break;
    case 5: nout = ShapeF::NOUT; nin = ShapeF::NIN; break;
    case 6: nout = ShapeG::NOUT; nin = ShapeG::NIN; break;
    case 7: nout = ShapeH::NOUT; nin = ShapeH::NIN; break;
    case 8: nout = ShapeAX::NOUT; nin = ShapeAX::NIN; break;
    case 9: nout = ShapeCX::NOUT; nin = ShapeCX::NIN; break;
    case 10: nout = ShapeFX::NOUT; nin = ShapeFX::NIN; break;
    default: nout = ShapeBX::NOUT; nin = ShapeBX::NIN; break;
  }
  if (e < nout * nin) {
    const int r = e / nin, c = e % nin;
    if (r < it.out_rows && c < it.out_cols) it.dW[r * it.out_cols + c] = (float)s;
  } else if (it.db && e - nout * nin < it.out_rows) {
    it.db[e - nout * nin] = (float)s;
  }
}

static int shape_of(const mf_wgrad_item& a, int precision) {
  if (precision == MF_PREC_BF16X3) {                 // the large blocks have a three-product variant
    if (a.n_out == 256 && a.n_in == 256) return kWgShapeX0;
    if (a.n_out == 128 && a.n_in == 256) return kWgShapeX0 + 1;
    if ((a.n_out == 128 && (a.n_in == 128 || a.n_in == 80)) || (a.n_out == 12 && a.n_in == 128)) return kWgShapeX0 + 2;   // the NoF's
    if (a.n_out == 128 && a.n_in == 32) return kWgShapeX0 + 2;
    if (a.n_out == 256 && a.n_in == 64) return kWgShapeX0 + 3;
  }
  if (a.n_out == 256 && a.n_in == 256) return 0;
  if (a.n_out == 256 && a.n_in == 64) return 1;
  if (a.n_out == 128 && a.n_in == 256) return 2;
  if (a.n_out == 128 && a.n_in == 32) return 3;
  if (a.n_out == 4 && a.n_in == 640) return 4;
  if (a.n_out == 128 && a.n_in == 128) return 5;
  if (a.n_out == 128 && a.n_in == 80) return 6;
  if (a.n_out == 12 && a.n_in == 128) return 7;
  return -1;
}

// widths, output layout and stage cost of an item: the fp32 shapes and the two large x3 shapes are their block; the x3 128 x 128
// shape takes the NoF's narrower blocks too (output layout = the fp32 shape's of the same block: 128 x 80, 16 x 128)
static void wg_item_dims(const mf_wgrad_item& a, int sh, WgItem& it) {
  int nout = 0, nin = 0;
  switch (sh) {
    case 0: nout = ShapeA::NOUT; nin = ShapeA::NIN; break;
    case 1: nout = ShapeB::NOUT; nin = ShapeB::NIN; break;
    case 2: nout = ShapeC::NOUT; nin = ShapeC::NIN; break;
    case 3: nout = ShapeD::NOUT; nin = ShapeD::NIN; break;
    case 4: nout = ShapeE::NOUT; nin = ShapeE::NIN; break;
    case 5: nout = ShapeF::NOUT; nin = ShapeF::NIN; break;
    case 6: nout = ShapeG::NOUT; nin = ShapeG::NIN; break;
    case 7: nout = ShapeH::NOUT; nin = ShapeH::NIN; break;
    case 8: nout = ShapeAX::NOUT; nin = ShapeAX::NIN; break;
    case 9: nout = ShapeCX::NOUT; nin = ShapeCX::NIN; break;
    case 10: nout = a.n_out == 12 ? ShapeH::NOUT : ShapeFX::NOUT; nin = a.n_in; break;
    default: nout = ShapeBX::NOUT; nin = a.n_in; break;
  }
  it.out_rows = nout; it.out_cols = nin;
  it.gw = sh == kWgShapeX0 + 2 && a.n_out == 12 ? ShapeH::GW : nout;
  it.xw = nin;
  it.cost = wg_stage_cost(sh);
  // the narrower items of the 128 x 128 shape run against their operand reads (measured, tools/bench_wgrad.py: 128 x 80 0.67 x the
  // full block, 12 x 128 0.83 x -- its 48-byte G rows cost whole bursts; 128 x 32 0.87 x (beside 256 x 256 workgroups; alone 0.76 x).  An item priced 40 % low makes its
  // workgroups -- and the launch -- 60 % late: the 13 NeRF items took 11.2 ms instead of 7.0 with 128 x 32 at 812)
  if (sh == kWgShapeX0 + 2) it.cost = it.gw < 128 ? 1450 : it.xw >= 128 ? it.cost : it.xw > 32 ? 1250 : 1530;
}

static int wg_plan(const mf_wgrad_item* items, int n, long long P, int precision, WgParams& p, long long& scratch_floats) {
  if (n < 0 || n > kWgMaxItems) return fail(MF_E_INVALID, "mf_weight_grads: %d items (max %d)", n, kWgMaxItems);
  p = WgParams{};
  p.n_items = n;
  p.P = P;
  p.stages = (P + kWgStage - 1) / kWgStage;
  p.grid = device_cus();
  long long cost = 0;
  for (int i = 0; i < n; ++i) {
    const mf_wgrad_item& a = items[i];
    const int sh = shape_of(a, precision);
    if (!a.G || !a.X || !a.dW) return fail(MF_E_INVALID, "mf_weight_grads: item %d has a null pointer", i);
    if (sh >= kWgShapeX0 && ((a.g_stride & 1) || (a.x_stride & 1)))
      return fail(MF_E_INVALID, "mf_weight_grads: item %d strides must be even", i);
    if ((a.g_stride & 3) || (a.x_stride & 3) || (reinterpret_cast<uintptr_t>(a.G) & 15) || (reinterpret_cast<uintptr_t>(a.X) & 15))
      return fail(MF_E_INVALID, "mf_weight_grads: item %d operands must be 16-byte aligned with strides that are multiples of 4 floats", i);
    WgItem& it = p.it[i];
    it.G = a.G; it.g_stride = a.g_stride; it.X = a.X; it.x_stride = a.x_stride;
    it.shape = sh; it.want_bias = a.db ? 1 : 0; it.dW = a.dW; it.db = a.db;
    wg_item_dims(a, sh, it);
    it.cost0 = cost;
    cost += p.stages * it.cost;
  }
  p.total_cost = cost;
  scratch_floats = 0;
  for (int i = 0; i < n; ++i) {
    WgItem& it = p.it[i];
    int first = -1, last = -1, count = 0;
    for (int w = 0; w < p.grid; ++w) {
      long long b, e;
      wg_range(p.total_cost, p.grid, w, it.cost0, it.cost, p.stages, b, e);
      if (b < e) { if (first < 0) first = w; last = w; ++count; }
    }
    it.dense = first >= 0 && count == last - first + 1;
    it.slot0 = first < 0 ? 0 : first;
    it.n_slots = first < 0 ? 0 : last - first + 1;
    it.part_off = scratch_floats;
    scratch_floats += (long long)it.n_slots * wg_out_floats(it.shape);
  }
  return MF_OK;
}

}  // namespace mf

using namespace mf;

static int wg_precision_ok(int precision) { return precision == MF_PREC_F32 || precision == MF_PREC_BF16X3; }

// The items of a call as (at most) two launches: those whose block has a three-product variant (precision BF16X3) and the
// rest (fp32).  Each subset is planned on its own (own cost space, own partials behind the other's in the scratch buffer).
struct WgSplit {
  mf_wgrad_item items[2][kWgMaxItems];
  int n[2];
  WgParams p[2];
  long long fl[2];
};
static int wg_split_plan(const mf_wgrad_item* items, int n, long long P, int precision, WgSplit& sp) {
  if (n < 0 || n > kWgMaxItems) return fail(MF_E_INVALID, "mf_weight_grads: %d items (max %d)", n, kWgMaxItems);
  sp.n[0] = sp.n[1] = 0;
  for (int i = 0; i < n; ++i) {
    const int sh = shape_of(items[i], precision);
    if (sh < 0) return fail(MF_E_UNSUPPORTED, "mf_weight_grads: item %d has unsupported block %d x %d", i, items[i].n_out, items[i].n_in);
    const int k = sh >= kWgShapeX0 ? 1 : 0;
    sp.items[k][sp.n[k]++] = items[i];
  }
  for (int k = 0; k < 2; ++k) {
    sp.fl[k] = 0;
    const int rc = wg_plan(sp.items[k], sp.n[k], P, k ? MF_PREC_BF16X3 : MF_PREC_F32, sp.p[k], sp.fl[k]);
    if (rc != MF_OK) return rc;
  }
  return MF_OK;
}

extern "C" int64_t mf_weight_grads_scratch_bytes_p(int32_t precision, const mf_wgrad_item* items, int32_t n_items, int64_t P) {
  static thread_local WgSplit sp;
  if (!items || P < 0 || !wg_precision_ok(precision) || wg_split_plan(items, n_items, P, precision, sp) != MF_OK) return -1;
  return (sp.fl[0] + sp.fl[1]) * 4 + 16;
}
extern "C" int64_t mf_weight_grads_scratch_bytes(const mf_wgrad_item* items, int32_t n_items, int64_t P) {
  return mf_weight_grads_scratch_bytes_p(MF_PREC_F32, items, n_items, P);
}

extern "C" int32_t mf_weight_grads_p(int32_t precision, const mf_wgrad_item* items, int32_t n_items, int64_t P, void* scratch, void* stream) {
  if (!items || P < 0) return fail(MF_E_INVALID, "mf_weight_grads: null argument");
  if (!wg_precision_ok(precision)) return fail(MF_E_INVALID, "mf_weight_grads: precision %d (MF_PREC_F32 | MF_PREC_BF16X3)", precision);
  if (n_items == 0) return MF_OK;
  static thread_local WgSplit sp;
  const int rc = wg_split_plan(items, n_items, P, precision, sp);
  if (rc != MF_OK) return rc;
  if (sp.fl[0] + sp.fl[1] > 0 && !scratch) return fail(MF_E_INVALID, "mf_weight_grads: scratch buffer missing");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int lds = ShapeA::SLOT_BYTES;
  if (ShapeB::SLOT_BYTES > lds) lds = ShapeB::SLOT_BYTES;
  if (ShapeC::SLOT_BYTES > lds) lds = ShapeC::SLOT_BYTES;
  if (ShapeD::SLOT_BYTES > lds) lds = ShapeD::SLOT_BYTES;
  if (ShapeE::SLOT_BYTES > lds) lds = ShapeE::SLOT_BYTES;
  if (ShapeF::SLOT_BYTES > lds) lds = ShapeF::SLOT_BYTES;
  if (ShapeG::SLOT_BYTES > lds) lds = ShapeG::SLOT_BYTES;
  if (ShapeH::SLOT_BYTES > lds) lds = ShapeH::SLOT_BYTES;
  lds *= 3;
  const int lds_k[2] = {lds, (int)kWgX3Dump + 4096};
  for (int k = 0; k < 2; ++k) {
    if (sp.n[k] == 0) continue;
    WgParams& p = sp.p[k];
    p.scratch = static_cast<float*>(scratch) + (k ? sp.fl[0] : 0);
    if (const char* e = getenv("MF_DEBUG_FLAGS")) p.dbg = atoi(e);
    if (P > 0) {
      void (*kern)(WgParams) = k ? wgrad_kernel<true> : wgrad_kernel<false>;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_k[k]) != hipSuccess)
        return fail(MF_E_LAUNCH, "mf_weight_grads: cannot reserve %d bytes of LDS", lds_k[k]);
      hipLaunchKernelGGL(kern, dim3(p.grid), dim3(kThreads), lds_k[k], st, p);
    }
    int maxf = 0;
    for (int i = 0; i < sp.n[k]; ++i) if (wg_out_floats(p.it[i].shape) > maxf) maxf = wg_out_floats(p.it[i].shape);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((maxf + 255) / 256, sp.n[k]), dim3(256), 0, st, p);
  }
  return check_launch("mf_weight_grads");
}

extern "C" int32_t mf_weight_grads(const mf_wgrad_item* items, int32_t n_items, int64_t P, void* scratch, void* stream) {
  return mf_weight_grads_p(MF_PREC_F32, items, n_items, P, scratch, stream);
}
