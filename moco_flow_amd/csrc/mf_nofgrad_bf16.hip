// mf_nofgrad_bf16.hip -- the backward of one NoF evaluation (mf_nofgrad.hip: mf_nof_backward) in three bf16 products.
//
// Same mathematics -- per point, given dL/d out (3) and the forward's dump [h_1 .. h_D | T]
//     (d T, d x) = backward of the head's transform (nof.py:75-82; flow head: d T = d x = d out);
//     d z_{D-1} = (W_head^T d T) [h_D > 0];  d z_{l-1} = (W_l[:, hidden]^T d z_l) [h_l > 0],  l = D-1 .. 1;
//     d emb[:64] = W_0[:, :64]^T d z_0 (+ W_skip[:, :64]^T d z_skip);  d pts = d x + (sin / cos chain rule of the xyz block)
// (the autograd graph of models/rendering.py:49-83 + models/nof.py:69-82 under loss.backward(), trainer/base.py:188-197), every
// pre-activation gradient stored in the dump's layout for mf_weight_grads -- on the MF_PREC_BF16X3 tile loop of mf_bf16.hpp,
// as mf_backward_bf16.hip does for the canonical NeRF: the D - 1 hidden contractions (K = 128) and the one or two 64-row
// embedded-input layers as (hi, lo) bf16 pairs of the gradients AND the transposed weights, three products per k-step, fp32
// accumulation; 4 waves (one per SIMD), 128 points per pass over the transposed weight stream (136 KiB at D = 4).  The ReLU
// masks come from the dump exactly as in the fp32 kernel, so no unit changes side: the result differs from the fp32 chain's by
// the 2^-16 of the split operands.  The fp32 kernel spends its time on the exact-fp32 matrix pipe (sixteen times the bf16
// pipe's cost per product; ten launches of ~0.3 ms per joint training step at 52 % MFMA busy, profiles/r04_train_joint_*); what
// stays here is the VALU work around the chain -- the transform's forward-mode partials, the 9 x 128 head product, the
// sin / cos of the chain rule.
#include "mf_bf16.hpp"
#include "mf_host.hpp"
#include "mf_layout.hpp"
#include "mf_nofbwd.hpp"

namespace mf {

int device_cus();   // mf_forward.hip

namespace bf {

constexpr int kNW = 128;                       // the NoF's width
// packed buffer: [resident: zeros 32 | head weight n_head x 128 (natural order), 1 KiB-aligned]
//                [panels: backward layers 0 .. D-2 = trunk layers D-1 .. 1 transposed, hidden columns only (K = 128: 4 tiles x 16
//                 groups), then W_0[:, :64]^T and (one skip layer) W_skip[:, :64]^T: 2 tiles x 16 groups each]
// group (hi | lo of k-step ks): lane (i = lane & 31, h = lane >> 5) holds Wt[32 P + i][16 ks + hid_perm2(h, e)], e = 0..7.
constexpr int kN3Zero = 0, kN3Head = 32;
inline int n3_res_bytes(int n_head) { return (int)round_up((int64_t)(32 + n_head * kNW) * 4, kGroupBytes); }
inline long long n3_groups_total(int D, int n_emb) { return (long long)(D - 1) * 4 * 16 + (long long)n_emb * 2 * 16; }

struct N3PackJob {
  const float* W[MF_MAX_LAYERS + 2];   // forward weight feeding backward layer i
  int ld[MF_MAX_LAYERS + 2];           // its row length
  int col0[MF_MAX_LAYERS + 2];         // first column read
  int tiles[MF_MAX_LAYERS + 2];        // 32-row tiles of the backward layer's output
  long long g0[MF_MAX_LAYERS + 3];     // first group of layer i
  int n_layers;
  const float* head_w; int n_head;
  float* res; int res_floats; unsigned* panels;
  long long total_groups;
};

__device__ inline unsigned short n3_rne(float x) {
  const unsigned u = __float_as_uint(x);
  const unsigned rnd = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(((u & 0x7f800000u) == 0x7f800000u ? u : rnd) >> 16);
}

__global__ void pack_nof_bwd3_kernel(N3PackJob job) {
  const long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx < job.res_floats) {
    const int o = (int)gidx - kN3Head;
    job.res[gidx] = (o >= 0 && o < job.n_head * kNW) ? job.head_w[o] : 0.f;
  }
  const long long grp = gidx >> 6;
  if (grp >= job.total_groups) return;
  const int lane = (int)(gidx & 63), i = lane & 31, h = lane >> 5;
  int li = 0;
  while (li + 1 < job.n_layers && grp >= job.g0[li + 1]) ++li;
  const long long local = grp - job.g0[li];
  const int P = (int)(local / 16), gi = (int)(local % 16), ks = gi >> 1, lo = gi & 1;
  const int n = 32 * P + i;                               // output feature of the backward layer = forward input column
  unsigned short h8[8];
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * ks + hid_perm2(h, e);              // forward output row
    const float w = job.W[li][(long long)k * job.ld[li] + job.col0[li] + n];
    const unsigned short hi = n3_rne(w);
    h8[e] = lo ? n3_rne(w - __uint_as_float((unsigned)hi << 16)) : hi;
  }
  unsigned* dst = job.panels + gidx * 4;
  for (int w = 0; w < 4; ++w) dst[w] = (unsigned)h8[2 * w] | ((unsigned)h8[2 * w + 1] << 16);
}

struct Nof3Params {
  Net net;                 // packed, res_lds, res_bytes, D; aux = head rows (3 | 9)
  float exyz[32];          // xyz embedding: freq[16], weight[16]
  int skip;                // the skip layer (-1 = none)
  int pow2;                // the xyz embedding's frequencies are exactly 2^k (the logscale default, embedding.py:19)
  long long P, stride;
  const float* pts; const float* acts; const float* g_out;
  float* gpre; float* g_pts;
  uint32_t par_off, ring_off, buf_bytes;
};

// value of accumulator register r, masked by the forward activation (m: the four float4 of this lane's activations of the tile;
// BITS: m[0][0] carries the two mask bytes of this lane half for the tile -- lane groups g = h and g = 2 + h of the forward's panel
// t, relu_mask_pair in mf_core.hpp: row 8 q + 4 h + i of the tile sits at bit 8 (q & 1) + 4 (q >> 1) + i)
template <bool BITS>
MF_D float n3_val(const f32x16& acc, const f32x4 (&m)[4], int r) {
  if (BITS) return ((__builtin_bit_cast(unsigned, m[0][0]) >> (((r >> 2) & 1) * 8 + ((r >> 2) >> 1) * 4 + (r & 3))) & 1u) ? acc[r] : 0.f;
  return m[r >> 2][r & 3] > 0.f ? acc[r] : 0.f;
}

// One backward layer of the chain: (out, outlo) <- split(mask * (Wt (in, inlo))), the fp32 values to grow[32 t + ...].
// mrow / grow: this lane's dump row / gradient row of the layer + 4 (lane >> 5).  OUT: the result is an operand again.
// BITS: `mrow` points at the layer's 4 mask words of this lane's sample instead (one 4-byte load per tile).
template <bool OUT, bool BITS, class ST>
MF_D void nof3_layer(ST& st, const Lane& id, CarryX& carry, const u32x4 (&in)[8], const u32x4 (&inlo)[8], u32x4 (&out)[8],
                     u32x4 (&outlo)[8], uint32_t zero_off, const Next& nxt, const float* mrow, float* grow) {
  constexpr int NT = 4, KHID = 8, NG = 2 * KHID, NM = 3 * KHID, kSteps = 16;
  f32x16 pend = {};
  f32x4 pm[4] = {}, hm[4] = {};
  // The mask is applied ONCE per element -- in the hi step of its pair, written back into the pending accumulators -- and the lo
  // step and the row store read the masked value (round 5; before, each of the three re-did the bit test: 144 instead of 48
  // VALU per tile in a kernel that runs against the VALU).  Every hi step (sidx <= 14) lies in front of the first store gap.
  auto step = [&](f32x16& acc, const f32x4 (&m)[4], int sidx, int t) __attribute__((always_inline)) {
    const int u = sidx >> 1, w = u & 3, r = u < 4 ? 2 * u : 8 + 2 * (u - 4);
    if (!OUT) return;
    if (!(sidx & 1)) { acc[r] = n3_val<BITS>(acc, m, r); acc[r + 1] = n3_val<BITS>(acc, m, r + 1); }
    const float v0 = acc[r], v1 = acc[r + 1];
    u32x4& hv = u < 4 ? out[2 * t] : out[2 * t + 1];
    if (!(sidx & 1)) {
      unsigned hi = pack_bf16x2(v0, v1);
      asm volatile("" : "+v"(hi));
      hv[w] = hi;
    } else {
      const unsigned hi = hv[w];
      unsigned lo = pack_bf16x2(v0 - bflo(hi), v1 - bfhi(hi));
      asm volatile("" : "+v"(lo));
      (u < 4 ? outlo[2 * t] : outlo[2 * t + 1])[w] = lo;
    }
  };
  auto store = [&](const f32x16& acc, const f32x4 (&m)[4], int t, int q) __attribute__((always_inline)) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = OUT ? acc[4 * q + i] : n3_val<BITS>(acc, m, 4 * q + i);      // (OUT: masked by the hi steps)
    *reinterpret_cast<f32x4*>(grow + 32 * t + 8 * q) = v;
  };
  auto run = [&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const Ahead two{t + 2 < NT ? NG : (t == NT - 2 ? nxt.groups : nxt.groups2),
                    t == NT - 2 ? nxt.jump : (t == NT - 1 ? nxt.jump2 : nullptr), 0, nullptr, t + 2 < NT ? NG : -1, -1};
    if constexpr (BITS) {                                      // this tile's two mask bytes: in flight across its MFMAs
      hm[0][0] = __builtin_bit_cast(float, relu_mask_pair(reinterpret_cast<const unsigned*>(mrow), t, id.h));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) hm[q] = *reinterpret_cast<const f32x4*>(mrow + 32 * t + 8 * q);
    }
    constexpr int tp = t > 0 ? t - 1 : 0;
    auto gap = [&](int m) __attribute__((always_inline)) {
      if (t == 0) return;
#pragma unroll
      for (int sidx = kSteps * m / NM; sidx < kSteps * (m + 1) / NM; ++sidx) step(pend, pm, sidx, tp);
      if (m >= NM - 4) store(pend, pm, tp, m - (NM - 4));
    };
    f32x16 acc;
    // VM operations younger than the previous panel's last piece at this tile's barrier: the four row stores that closed the
    // previous tile (tile 0: the layer in front; none behind tile 0) + this tile's four mask loads
    constexpr int KEEP = (t == 1 ? 0 : 4) + (BITS ? 2 : 4);
    mma_tile_x<0, KHID, 2, true, KEEP, true>(st, id, carry, in, inlo, in, inlo, zero_off, two, acc, gap);
    st.advance();
    pend = acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) pm[q] = hm[q];
  };
  run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 1>{});
  run(std::integral_constant<int, 2>{}); run(std::integral_constant<int, 3>{});
#pragma unroll
  for (int sidx = 0; sidx < kSteps; ++sidx) step(pend, pm, sidx, NT - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) store(pend, pm, NT - 1, q);
  __builtin_amdgcn_sched_barrier(0);
}

// A 64-row layer behind the chain (the embedded-input gradient): res[t] = Wt_tile (in, inlo), t = 0, 1; nothing stored.
template <class ST>
MF_D void nof3_emb(ST& st, const Lane& id, CarryX& carry, const u32x4 (&in)[8], const u32x4 (&inlo)[8], uint32_t zero_off,
                   const Next& nxt, f32x16 (&res)[2]) {
  const Ahead t0{nxt.groups, nxt.jump, 0, nullptr}, t1{nxt.groups2, nxt.jump2, 0, nullptr};
  auto nogap = [](int) {};
  mma_tile_x<0, 8, 2, true>(st, id, carry, in, inlo, in, inlo, zero_off, t0, res[0], nogap);
  st.advance();
  mma_tile_x<0, 8, 2, true>(st, id, carry, in, inlo, in, inlo, zero_off, t1, res[1], nogap);
  st.advance();
}

// eight fp32 values (two float4 of a row: features 16 ks + 4 h + (0..3) and 16 ks + 8 + 4 h + (0..3)) -> (hi, lo) operand dwords
MF_D void n3_split8(const f32x4& v0, const f32x4& v1, u32x4& hi4, u32x4& lo4) {
  const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned hi = pack_bf16x2(v[2 * w], v[2 * w + 1]);
    hi4[w] = hi;
    lo4[w] = pack_bf16x2(v[2 * w] - bflo(hi), v[2 * w + 1] - bfhi(hi));
  }
}

// BITS: the dump rows carry the layers' ReLU bit rows behind T (4 words per layer at float offset D W + 16: written by the
// forward when the row has room, nof_eval's `masks`) -- the chain then reads 16 bytes instead of 512 per layer and point.
template <bool BITS>
__global__ __launch_bounds__(256, 1) void nof_backward_kernel_x3(const Nof3Params p) {
  constexpr int NW = 4, TILE = NW * kWaveSamples;
  const Lane id;
  load_resident<NW>(p.net, id);
  if (threadIdx.x < 32) {      // the xyz embedding's (freq, weight) table: kernarg -> LDS (a runtime index into the by-value struct would put it in scratch)
    typedef const __attribute__((address_space(4))) char* kptr;
    const kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Nof3Params, exyz);
    *(float*)(smem + p.par_off + threadIdx.x * 4) = ((const __attribute__((address_space(4))) float*)ka)[threadIdx.x];
  }
  StreamT<NW> st;
  st.tl.start(nullptr, id);
  CarryX carry;
  const int D = p.net.D, NH = p.net.aux;
  const char* first = p.net.packed + p.net.res_bytes;
  st.start(first, 16, 16, p.ring_off, p.buf_bytes, id);          // (its wait + barrier also publish the resident block / table)
  carry.load(st.slot_off(0) + id.lane * 16);
  const uint32_t zero_off = p.net.res_lds + kN3Zero * 4, headw = p.net.res_lds + kN3Head * 4;
  const Next n16{16, nullptr, 16, nullptr}, nfirst{16, first, 16, nullptr};
  const long long ntiles = (p.P + TILE - 1) / TILE;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long s = tile * TILE + id.wave * kWaveSamples + id.j;
    const bool valid = s < p.P;
    const long long ss = valid ? s : p.P - 1;
    const float x[3] = {p.pts[ss * 3 + 0], p.pts[ss * 3 + 1], p.pts[ss * 3 + 2]};
    const float go[3] = {p.g_out[ss * 3 + 0], p.g_out[ss * 3 + 1], p.g_out[ss * 3 + 2]};
    // "activation rows": BITS: the sample's mask words (4 per layer); else the dump rows + 4 h
    const float* arow = BITS ? p.acts + ss * p.stride + (long long)D * kNW + 16 : p.acts + ss * p.stride + 4 * id.h;
    constexpr int LW = BITS ? 4 : kNW;                           // row elements per layer
    float* grow = p.gpre + s * p.stride + 4 * id.h;            // rows up to round_up(P, 128) exist
    float T[9], dT[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dx[3];
    {
      const float4* tr = reinterpret_cast<const float4*>(p.acts + ss * p.stride + (long long)D * kNW);
      const float4 t0 = tr[0], t1 = tr[1], t2 = tr[2];
      T[0] = t0.x; T[1] = t0.y; T[2] = t0.z; T[3] = t0.w; T[4] = t1.x; T[5] = t1.y; T[6] = t1.z; T[7] = t1.w; T[8] = t2.x;
    }
    if (NH == 9) {
      quat_transform_backward(T, x, go, dT, dx);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { dT[c] = go[c]; dx[c] = go[c]; }
    }
    if (id.h == 0) {
      float4* tg = reinterpret_cast<float4*>(p.gpre + s * p.stride + (long long)D * kNW);
      tg[0] = make_float4(dT[0], dT[1], dT[2], dT[3]);
      tg[1] = make_float4(dT[4], dT[5], dT[6], dT[7]);
      tg[2] = make_float4(dT[8], 0.f, 0.f, 0.f);
      tg[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // d z_{D-1} = (W_head^T d T) [h_D > 0] as the (hi, lo) operands of 8 k-steps: slot e of step ks = feature 16 ks + hid_perm2(h, e)
    u32x4 ah[8], al[8], bh[8], bl[8];
    {
      const float* hrow = arow + (long long)(D - 1) * LW;
      float* ghrow = grow + (long long)(D - 1) * kNW;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        f32x4 g2[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {                        // features 16 ks + 8 c2 + 4 h + (0..3)
          const int f = 16 * ks + 8 * c2;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 9; ++c) {
            if (c < NH) {
              const f32x4 w = lds_f4(headw + (c * kNW + f + 4 * id.h) * 4);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(w[r], dT[c], acc[r]);
            }
          }
          f32x4 h4;
          if constexpr (BITS) {                                  // outputs (f & 31) + 4 h + r of the forward's panel f / 32
            const int fp = (f & 31) + 4 * id.h;
            const unsigned wv = reinterpret_cast<const unsigned*>(hrow)[relu_mask_word(f >> 5, fp)] >> relu_mask_shift(f >> 5, fp);
#pragma unroll
            for (int r = 0; r < 4; ++r) h4[r] = ((wv >> r) & 1u) ? 1.f : 0.f;
          } else {
            h4 = *reinterpret_cast<const f32x4*>(hrow + f);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) g2[c2][r] = h4[r] > 0.f ? acc[r] : 0.f;
          *reinterpret_cast<f32x4*>(ghrow + f) = g2[c2];
        }
        n3_split8(g2[0], g2[1], ah[ks], al[ks]);
      }
    }
    // chain: d z_{l-1} = (W_l[:, hidden]^T d z_l) [h_l > 0],  l = D-1 .. 1   (a -> b, copied back)
    for (int l = D - 1; l >= 1; --l) {
      nof3_layer<true, BITS>(st, id, carry, ah, al, bh, bl, zero_off, n16, arow + (long long)(l - 1) * LW, grow + (long long)(l - 1) * kNW);
#pragma unroll
      for (int t = 0; t < 8; ++t) { ah[t] = bh[t]; al[t] = bl[t]; }
    }
    // embedded-input gradient, columns 0..63:  W_0[:, :64]^T d z_0 (+ W_skip[:, :64]^T d z_skip)
    f32x16 ge[2];
    nof3_emb(st, id, carry, ah, al, zero_off, p.skip >= 0 ? n16 : nfirst, ge);
    if (p.skip >= 0) {
      // d z_skip was stored by this very lane several layers ago (the panel barriers' vmcnt waits retired the stores)
      const float* zrow = grow + (long long)p.skip * kNW;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        n3_split8(*reinterpret_cast<const f32x4*>(zrow + 16 * ks), *reinterpret_cast<const f32x4*>(zrow + 16 * ks + 8), ah[ks], al[ks]);
      f32x16 g2[2];
      nof3_emb(st, id, carry, ah, al, zero_off, nfirst, g2);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ge[t][r] += g2[t][r];
    }
    // sin / cos chain rule (embedding.py:42-46): register r of tile t is column f = 32 t + 8 (r / 4) + 4 h + r % 4 of
    // [x | sin f0 x | cos f0 x | ...]; the xyz block is columns 0..32.  The 15 (frequency, component) pairs: with the
    // reference's 2^k table (embedding.py:19; checked on the host) an exact sincosf per component at f_0 and angle doublings
    // from there (<= 16 x the seed's 1e-7: far under a gradient's 1e-4), else 15 exact ones -- the fp32 kernel calls
    // sincosf per column (12 per lane, ~110 instructions each).
    float sn[5][3], cs[5][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sincosf(lds_f(p.par_off) * x[c], &sn[0][c], &cs[0][c]);
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        if (p.pow2) {
          const float t2 = sn[k - 1][c] + sn[k - 1][c];
          sn[k][c] = t2 * cs[k - 1][c];
          cs[k][c] = __builtin_fmaf(-t2, sn[k - 1][c], 1.f);
        } else {
          sincosf(lds_f(p.par_off + 4 * k) * x[c], &sn[k][c], &cs[k][c]);
        }
      }
    }
    float ex[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (t == 1 && r > 0) continue;                         // columns >= 33: the image-index block (no point gradient)
        const float gv = ge[t][r];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                       // (the column depends on the lane half: both resolved at compile time)
          const int f = 32 * t + 8 * (r >> 2) + 4 * hh + (r & 3);
          if (f >= 33) continue;
          float contrib;
          int comp;
          if (f < 3) {
            contrib = gv;
            comp = f;
          } else {
            const int k = (f - 3) / 6, rem = (f - 3) % 6;
            comp = rem % 3;
            const float fr = lds_f(p.par_off + 4 * k), w = lds_f(p.par_off + 64 + 4 * k);
            contrib = w * fr * (rem >= 3 ? -sn[k][comp] : cs[k][comp]) * gv;
          }
          if (id.h == hh) ex[comp] += contrib;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[c] += __shfl_xor(ex[c], 32, 64);
    if (valid && id.h == 0 && p.g_pts) {
      p.g_pts[s * 3 + 0] = dx[0] + ex[0];
      p.g_pts[s * 3 + 1] = dx[1] + ex[1];
      p.g_pts[s * 3 + 2] = dx[2] + ex[2];
    }
  }
  wait_vm0();
}

}  // namespace bf
}  // namespace mf

using namespace mf;

// the shapes of mf_nof_backward (W = 128, 33 + 33 input columns, at most one skip layer); skip = -1: none
static bool nof_bwd3_shape(const mf_nof_desc* d, int& skip) {
  NetLayout F;
  if (!d || !nof_layout(*d, F, 0) || d->D < 2) return false;
  // ONE condition for the size query, the packer and the launch: the transposed embedded-input tiles are built for the
  // full 33 + 33 input block (64 packed columns); narrower blocks take the fp32 kernel (autograd.nof_backward_hip falls back)
  if (d->in_channels_xyz + d->extra_feat_dim < 64) return false;
  skip = -1;
  for (int l = 1; l < d->D; ++l)
    if ((d->skip_mask >> l) & 1u) {
      if (skip >= 0) return false;
      skip = l;
    }
  return true;
}

extern "C" int64_t mf_nof_bwd3_packed_bytes(const mf_nof_desc* d) {
  int skip;
  if (!nof_bwd3_shape(d, skip)) { fail(MF_E_UNSUPPORTED, "mf_nof_bwd3_packed_bytes: unsupported NoF configuration"); return 0; }
  return bf::n3_res_bytes(d->use_quat ? 9 : 3) + bf::n3_groups_total(d->D, skip >= 0 ? 2 : 1) * kGroupBytes;
}

extern "C" int32_t mf_nof_pack_bwd3(const mf_nof_desc* d, void* packed, void* stream) {
  int skip;
  if (!d || !packed) return fail(MF_E_INVALID, "mf_nof_pack_bwd3: null argument");
  if (!nof_bwd3_shape(d, skip)) return fail(MF_E_UNSUPPORTED, "mf_nof_pack_bwd3: unsupported NoF configuration (W=%d D=%d)", d->W, d->D);
  bf::N3PackJob job{};
  const int cin = d->in_channels_xyz + d->extra_feat_dim, n_head = d->use_quat ? 9 : 3;
  long long g0 = 0;
  for (int i = 0; i + 1 < d->D; ++i) {                       // backward layer i = trunk layer l = D-1-i, hidden columns
    const int l = d->D - 1 - i;
    job.W[i] = d->trunk_w[l];
    job.ld[i] = bf::kNW + (l == skip ? cin : 0);
    job.col0[i] = l == skip ? cin : 0;
    job.tiles[i] = 4; job.g0[i] = g0; g0 += 4 * 16;
  }
  job.n_layers = d->D - 1;
  for (int e = 0; e < (skip >= 0 ? 2 : 1); ++e) {              // embedded columns 0..63 of layer 0 / the skip layer
    const int i = job.n_layers++, l = e == 0 ? 0 : skip;
    job.W[i] = d->trunk_w[l];
    job.ld[i] = l == 0 ? cin : bf::kNW + cin;
    job.col0[i] = 0; job.tiles[i] = 2; job.g0[i] = g0; g0 += 2 * 16;
  }
  job.g0[job.n_layers] = g0;
  for (int i = 0; i < job.n_layers; ++i)
    if (!job.W[i]) return fail(MF_E_INVALID, "mf_nof_pack_bwd3: missing weight pointer (backward layer %d)", i);
  if (!d->head_w) return fail(MF_E_INVALID, "mf_nof_pack_bwd3: missing head weight");
  if (cin < 64) return fail(MF_E_UNSUPPORTED, "mf_nof_pack_bwd3: needs >= 64 embedded input columns");
  job.head_w = d->head_w; job.n_head = n_head;
  job.res = static_cast<float*>(packed);
  job.res_floats = bf::n3_res_bytes(n_head) / 4;
  job.panels = reinterpret_cast<unsigned*>(static_cast<char*>(packed) + bf::n3_res_bytes(n_head));
  job.total_groups = bf::n3_groups_total(d->D, skip >= 0 ? 2 : 1);
  if (g0 != job.total_groups) return fail(MF_E_INVALID, "mf_nof_pack_bwd3: layout mismatch");
  const long long slots = job.total_groups * 64 > job.res_floats ? job.total_groups * 64 : job.res_floats;
  hipLaunchKernelGGL(bf::pack_nof_bwd3_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), job);
  return check_launch("mf_nof_pack_bwd3");
}

extern "C" int32_t mf_nof_backward3(const mf_nof_desc* d, const void* packed_bwd3, const mf_embedding* emb_xyz, int64_t P,
                                    const float* pts, const float* acts, int64_t stride, const float* g_out,
                                    float* gpre, float* g_pts, void* stream) {
  int skip;
  if (!d || !packed_bwd3 || !emb_xyz || (P > 0 && (!pts || !acts || !g_out || !gpre)))
    return fail(MF_E_INVALID, "mf_nof_backward3: null argument");
  if (!nof_bwd3_shape(d, skip)) return fail(MF_E_UNSUPPORTED, "mf_nof_backward3: unsupported NoF configuration");
  if (emb_xyz->in_channels != 3 || emb_xyz->n_freqs > 5)
    return fail(MF_E_UNSUPPORTED, "mf_nof_backward3: xyz embedding must have 3 channels and <= 5 frequencies");
  if (stride < (int64_t)d->D * bf::kNW + 16 || (stride & 3) || (reinterpret_cast<uintptr_t>(acts) & 15) || (reinterpret_cast<uintptr_t>(gpre) & 15))
    return fail(MF_E_INVALID, "mf_nof_backward3: dump rows must be 16-byte aligned with a stride >= D W + 16 that is a multiple of 4 floats");
  if (P == 0) return MF_OK;
  bf::Nof3Params p{};
  const int n_head = d->use_quat ? 9 : 3;
  p.net.packed = static_cast<const char*>(packed_bwd3);
  p.net.res_lds = 0; p.net.res_bytes = (uint32_t)bf::n3_res_bytes(n_head); p.net.D = d->D; p.net.emb_mask = 0; p.net.aux = n_head;
  for (int k = 0; k < 16; ++k) {
    p.exyz[k] = k < emb_xyz->n_freqs ? emb_xyz->freq[k] : 0.f;
    p.exyz[16 + k] = k < emb_xyz->n_freqs ? emb_xyz->weight[k] : 0.f;
  }
  p.pow2 = 1;
  for (int k = 0; k < emb_xyz->n_freqs; ++k)
    if (emb_xyz->freq[k] != (float)(1 << k)) p.pow2 = 0;
  p.skip = skip; p.P = P; p.stride = stride;
  p.pts = pts; p.acts = acts; p.g_out = g_out; p.gpre = gpre; p.g_pts = g_pts;
  uint32_t lds = p.net.res_bytes;
  p.par_off = lds; lds += 128;
  lds = (lds + 1023u) & ~1023u;
  p.ring_off = lds; p.buf_bytes = 16 * kGroupBytes; lds += 3 * p.buf_bytes;
  const long long ntiles = (P + 127) / 128;
  // 207 VGPRs + 48 AGPRs and 53 KiB of LDS per workgroup: TWO workgroups fit a CU (two waves per SIMD, each other's cover
  // during the VALU phases around the chain), so the persistent grid is two per CU
  const long long slots = 2LL * device_cus();
  const int grid = (int)(ntiles < slots ? ntiles : slots);
  // the forward wrote the layers' ReLU bit rows when the row has room for them (nof_eval's `masks`)
  const bool bits = stride >= (int64_t)d->D * bf::kNW + 16 + 4 * d->D;
  void (*kern)(const bf::Nof3Params) = bits ? bf::nof_backward_kernel_x3<true> : bf::nof_backward_kernel_x3<false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nof_backward3: cannot reserve %u bytes of LDS", lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nof_backward3");
}
