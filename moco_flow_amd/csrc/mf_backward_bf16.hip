// mf_backward_bf16.hip -- the input-gradient chain of the canonical NeRF (mf_backward.hip) in three bf16 products.
//
// Same mathematics as mf_nerf_backward -- per sample, given dL/d[rgb, sigma] and the forward's activation dump,
//     d_o = d_rgb rgb (1 - rgb);  d_e = (W_rgb^T d_o) [e > 0];  d_g = W_e[:, :W]^T d_e;
//     d_z_{D-1} = (W_f^T d_g + w_sigma d_sigma) [h_{D-1} > 0];  d_z_{l-1} = (W_l[:, hidden]^T d_z_l) [h_{l-1} > 0],  l = D-1 .. 1
// with every pre-activation gradient stored in the dump's layout for mf_weight_grads -- on the MF_PREC_BF16X3 core of
// mf_bf16.hpp: the D + 1 W-wide contractions as (hi, lo) bf16 pairs of the gradients AND the transposed weights, three
// products per k-step, fp32 accumulation; 4 waves (one per SIMD), 128 samples per pass over the transposed weight
// stream.  The ReLU masks come from the dump exactly as in the fp32 chain, so no unit changes side: the result differs from
// the fp32 chain's by the 2^-16 of the split operands (measured ~1e-5 max-rel on d z), not by mask flips.
// A tile's epilogue (sigma term, mask, (hi, lo) split of the next layer's operand) runs in the MFMA gaps of the next tile;
// its four 16-byte row stores sit behind that tile's last LDS-DMA piece so that the panel barrier leaves exactly them (and
// the next tile's four mask loads) in flight (StreamT::sync<KEEP>).
// The gradient of the embedded input (g_emb, ABI v9: the joint stage's NoF training) = W_0[:, :63]^T d_z_0 (+ one skip
// layer: W_skip[:, :63]^T d_z_skip) follows as one or two 64-row layers behind the chain, d_z_skip re-read from the rows
// this lane stored several layers earlier.
#include "mf_bf16.hpp"
#include "mf_host.hpp"
#include "mf_layout.hpp"

namespace mf {

int device_cus();   // mf_forward.hip

namespace bf {

// packed buffer: [resident: zeros 32 | rgb.0.weight 3 x 128 (natural order) | sigma.weight 256, 1 KiB-aligned]
//                [panels: backward layer 0 = extra_encoding[:, :W]^T (K = 128: 8 tiles x 16 groups),
//                 layers 1 .. D = xyz_encoding_final^T, trunk layers D-1 .. 1 transposed (K = 256: 8 tiles x 32 groups)]
// group (hi | lo of k-step ks): lane (i = lane & 31, h = lane >> 5) holds Wt[32 P + i][16 ks + hid_perm2(h, e)], e = 0..7.
constexpr int kB3Zero = 0, kB3Rgb = 32, kB3Sig = 32 + 384, kB3ResFloats = 32 + 384 + 256;
constexpr int kB3ResBytes = ((kB3ResFloats * 4 + kGroupBytes - 1) / kGroupBytes) * kGroupBytes;
//                 [then, for g_emb: W_0[:, :63]^T and (one skip layer) W_skip[:, :63]^T: 2 tiles x 32 groups each, rows >= 63 zero]
inline long long bwd3_groups_total(int D, int n_emb) { return 8LL * 16 + (long long)D * 8 * 32 + (long long)n_emb * 2 * 32; }
inline int bwd3_skip_layer(const mf_nerf_desc& d) {      // the single skip layer, 0 = none, -1 = several
  int s = 0;
  for (int l = 1; l < d.D; ++l)
    if ((d.skip_mask >> l) & 1u) { if (s) return -1; s = l; }
  return s;
}

struct Bwd3PackJob {
  const float* W[MF_MAX_LAYERS + 4];   // forward weight feeding backward layer i
  int ld[MF_MAX_LAYERS + 4];           // its row length
  int col0[MF_MAX_LAYERS + 4];         // first column read
  int ncols[MF_MAX_LAYERS + 4];        // output rows present (rows beyond are zero): the embedded-input layers
  int gpt[MF_MAX_LAYERS + 4];          // groups per tile
  long long g0[MF_MAX_LAYERS + 5];     // first group of layer i
  int n_layers;
  const float* sigma_w; const float* rgb_w;
  float* res; unsigned* panels;
  long long total_groups;
};

__device__ inline unsigned short b3_rne(float x) {
  const unsigned u = __float_as_uint(x);
  const unsigned rnd = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(((u & 0x7f800000u) == 0x7f800000u ? u : rnd) >> 16);
}

__global__ void pack_bwd3_kernel(Bwd3PackJob job) {
  const long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx < kB3ResBytes / 4) {
    const int o = (int)gidx;
    float v = 0.f;
    if (o >= kB3Rgb && o < kB3Rgb + 384) v = job.rgb_w[o - kB3Rgb];
    else if (o >= kB3Sig && o < kB3Sig + 256) v = job.sigma_w[o - kB3Sig];
    job.res[o] = v;
  }
  const long long grp = gidx >> 6;
  if (grp >= job.total_groups) return;
  const int lane = (int)(gidx & 63), i = lane & 31, h = lane >> 5;
  int li = 0;
  while (li + 1 < job.n_layers && grp >= job.g0[li + 1]) ++li;
  const long long local = grp - job.g0[li];
  const int gpt = job.gpt[li];
  const int P = (int)(local / gpt), gi = (int)(local % gpt), ks = gi >> 1, lo = gi & 1;
  const int n = 32 * P + i;                               // output feature of the backward layer = forward input column
  unsigned short h8[8];
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * ks + hid_perm2(h, e);              // forward output row
    const float w = n < job.ncols[li] ? job.W[li][(long long)k * job.ld[li] + job.col0[li] + n] : 0.f;
    const unsigned short hi = b3_rne(w);
    h8[e] = lo ? b3_rne(w - __uint_as_float((unsigned)hi << 16)) : hi;
  }
  unsigned* dst = job.panels + gidx * 4;
  for (int w = 0; w < 4; ++w) dst[w] = (unsigned)h8[2 * w] | ((unsigned)h8[2 * w + 1] << 16);
}

struct Bwd3Params {
  Net net;                 // packed, res_lds, res_bytes, D
  long long P, stride;
  const float* g_out; const float* acts; const float* rgbsigma;
  float* gpre; float* ghead;
  float* g_emb;            // (P,64) dL/d embedded input (natural column order, column 63 = 0), or null
  int skip;                // the skip layer (0 = none)
  const unsigned* mask; long long mask_stride;   // the forward's ReLU bit-mask rows (BITS instantiation), else acts is read
  uint32_t ring_off, buf_bytes;
};

// value of accumulator register r of tile t: + the sigma term, masked by the forward activation
// (BITS: m[0][0] carries the two mask bytes of this lane half for the tile -- lane groups g = h (low byte) and g = 2 + h of the
//  forward's panel t, relu_mask_word / relu_mask_shift in mf_core.hpp: row 8 q + 4 h + i of the tile sits at bit
//  8 (q & 1) + 4 (q >> 1) + i)
template <bool MASK, bool SIG, bool BITS>
MF_D float b3_val(const f32x16& acc, const f32x4 (&m)[4], int r, uint32_t sigw_off, int t, int h, float dsig) {
  float v = acc[r];
  if (SIG) v = __builtin_fmaf(lds_f(sigw_off + (32 * t + 8 * (r >> 2) + 4 * h + (r & 3)) * 4), dsig, v);
  if (MASK) {
    if (BITS) v = ((__builtin_bit_cast(unsigned, m[0][0]) >> (((r >> 2) & 1) * 8 + ((r >> 2) >> 1) * 4 + (r & 3))) & 1u) ? v : 0.f;
    else v = m[r >> 2][r & 3] > 0.f ? v : 0.f;
  }
  return v;
}

// One backward layer: (out, outlo) <- split(mask * (Wt (in, inlo) [+ w_sigma d_sigma])), the fp32 values to grow[32 t + ...].
// KHID = k-steps of the input (8 | 16).  mrow / grow: this lane's dump row / gradient row of the layer + 4 (lane >> 5).
// BITS: `mrow` points at the layer's 8 mask words of this lane's sample instead (one 4-byte load per tile).
template <int KHID, bool MASK, bool SIG, bool OUT, bool BITS, class ST>
MF_D void bwd_layer_x(ST& st, const Lane& id, CarryX& carry, const u32x4 (&in)[16], const u32x4 (&inlo)[16], u32x4 (&out)[16],
                      u32x4 (&outlo)[16], uint32_t zero_off, const Next& nxt, const float* mrow, float* grow, uint32_t sigw_off,
                      float dsig) {
  constexpr int NT = 8, NG = 2 * KHID, NM = 3 * KHID, kSteps = 16;
  f32x16 pend = {};
  f32x4 pm[4] = {}, hm[4] = {};
  // The sigma term and the mask are applied ONCE per element -- in the hi step of its pair, written back into the pending
  // accumulators -- and the lo step and the row store read the finished value (round 5; before, each of the three re-did them).
  // Every hi step (sidx <= 14) lies in front of the first store gap.
  auto step = [&](f32x16& acc, const f32x4 (&m)[4], int sidx, int t) __attribute__((always_inline)) {
    const int u = sidx >> 1, w = u & 3, r = u < 4 ? 2 * u : 8 + 2 * (u - 4);
    if (!OUT) return;
    if (!(sidx & 1)) {
      acc[r] = b3_val<MASK, SIG, BITS>(acc, m, r, sigw_off, t, id.h, dsig);
      acc[r + 1] = b3_val<MASK, SIG, BITS>(acc, m, r + 1, sigw_off, t, id.h, dsig);
    }
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
    for (int i = 0; i < 4; ++i) v[i] = OUT ? acc[4 * q + i] : b3_val<MASK, SIG, BITS>(acc, m, 4 * q + i, sigw_off, t, id.h, dsig);   // (OUT: finished by the hi steps)
    *reinterpret_cast<f32x4*>(grow + 32 * t + 8 * q) = v;
  };
  auto run = [&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const Ahead two{t + 2 < NT ? NG : (t == NT - 2 ? nxt.groups : nxt.groups2),
                    t == NT - 2 ? nxt.jump : (t == NT - 1 ? nxt.jump2 : nullptr), 0, nullptr, t + 2 < NT ? NG : -1, -1};
    if constexpr (MASK && BITS) {                             // this tile's two mask bytes: in flight across its MFMAs
      hm[0][0] = __builtin_bit_cast(float, relu_mask_pair(reinterpret_cast<const unsigned*>(mrow), t, id.h));
    } else if constexpr (MASK) {
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
    // VM operations younger than the previous panel's last piece at this tile's first barrier: the four row stores that
    // closed the previous tile (tile 0: the layer in front; none behind tile 0) + this tile's four mask loads
    constexpr int KEEP = (t == 1 ? 0 : 4) + (MASK ? (BITS ? 2 : 4) : 0);
    mma_tile_x<0, KHID, 2, true, KEEP, true>(st, id, carry, in, inlo, in, inlo, zero_off, two, acc, gap);
    st.advance();
    pend = acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) pm[q] = hm[q];
  };
  run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 1>{});
  run(std::integral_constant<int, 2>{}); run(std::integral_constant<int, 3>{});
  run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 5>{});
  run(std::integral_constant<int, 6>{}); run(std::integral_constant<int, 7>{});
#pragma unroll
  for (int sidx = 0; sidx < kSteps; ++sidx) step(pend, pm, sidx, NT - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) store(pend, pm, NT - 1, q);
  __builtin_amdgcn_sched_barrier(0);
}

// A 64-row layer behind the chain (the embedded-input gradient): res[t] = Wt_tile (in, inlo), t = 0, 1; nothing stored.
template <class ST>
MF_D void bwd_emb_x(ST& st, const Lane& id, CarryX& carry, const u32x4 (&in)[16], const u32x4 (&inlo)[16], uint32_t zero_off,
                    const Next& nxt, f32x16 (&res)[2]) {
  const Ahead t0{nxt.groups, nxt.jump, 0, nullptr}, t1{nxt.groups2, nxt.jump2, 0, nullptr};
  auto nogap = [](int) {};
  mma_tile_x<0, 16, 2, true>(st, id, carry, in, inlo, in, inlo, zero_off, t0, res[0], nogap);
  st.advance();
  mma_tile_x<0, 16, 2, true>(st, id, carry, in, inlo, in, inlo, zero_off, t1, res[1], nogap);
  st.advance();
}

template <bool BITS>
__global__ __launch_bounds__(256, 1) void nerf_backward_kernel_x3(const Bwd3Params p) {
  constexpr int NW = 4, TILE = NW * kWaveSamples;
  const Lane id;
  load_resident<NW>(p.net, id);
  StreamT<NW> st;
  st.tl.start(nullptr, id);
  CarryX carry;
  const int D = p.net.D;
  const char* first = p.net.packed + p.net.res_bytes;
  st.start(first, 16, 16, p.ring_off, p.buf_bytes, id);
  carry.load(st.slot_off(0) + id.lane * 16);
  const uint32_t zero_off = p.net.res_lds + kB3Zero * 4, rgbw = p.net.res_lds + kB3Rgb * 4, sigw = p.net.res_lds + kB3Sig * 4;
  const Next n32{32, nullptr, 32, nullptr}, nfirst{16, first, 16, nullptr};
  const long long ntiles = (p.P + TILE - 1) / TILE;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long s = tile * TILE + id.wave * kWaveSamples + id.j;
    const bool valid = s < p.P;
    const long long ss = valid ? s : p.P - 1;
    const float4 go = *reinterpret_cast<const float4*>(p.g_out + ss * 4);
    const float4 rs = *reinterpret_cast<const float4*>(p.rgbsigma + ss * 4);
    const float d0 = go.x * rs.x * (1.f - rs.x), d1 = go.y * rs.y * (1.f - rs.y), d2 = go.z * rs.z * (1.f - rs.z);
    if (valid && id.h == 0) *reinterpret_cast<float4*>(p.ghead + s * 4) = make_float4(d0, d1, d2, go.w);
    // BITS: "activation rows" are the sample's mask words (8 per layer); else the dump rows + 4 h
    const float* arow = BITS ? reinterpret_cast<const float*>(p.mask + ss * p.mask_stride) : p.acts + ss * p.stride + 4 * id.h;
    constexpr int LW = BITS ? 8 : 256;                           // row elements per layer
    float* grow = p.gpre + s * p.stride + 4 * id.h;             // rows up to round_up(P, 128) exist
    // d_e = (W_rgb^T d_o) [e > 0] as the (hi, lo) operands of 8 k-steps: slot e of step ks = feature 16 ks + hid_perm2(h, e)
    u32x4 ah[16], al[16], bh[16], bl[16];
    {
      const float* erow = arow + (long long)(D + 1) * LW;
      float* gerow = grow + (long long)(D + 1) * 256;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 2; ++c) {                            // features 16 ks + 8 c + 4 h + (0..3)
          const int f = 16 * ks + 8 * c;
          const f32x4 w0 = lds_f4(rgbw + (0 * 128 + f + 4 * id.h) * 4), w1 = lds_f4(rgbw + (1 * 128 + f + 4 * id.h) * 4);
          const f32x4 w2 = lds_f4(rgbw + (2 * 128 + f + 4 * id.h) * 4);
          f32x4 e4;
          if constexpr (BITS) {                                  // outputs (f & 31) + 4 h + r of the forward's panel f / 32
            const int fp = (f & 31) + 4 * id.h;
            const unsigned wv = reinterpret_cast<const unsigned*>(erow)[relu_mask_word(f >> 5, fp)] >> relu_mask_shift(f >> 5, fp);
#pragma unroll
            for (int r = 0; r < 4; ++r) e4[r] = ((wv >> r) & 1u) ? 1.f : 0.f;
          } else {
            e4 = *reinterpret_cast<const f32x4*>(erow + f);
          }
          f32x4 g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = __builtin_fmaf(w2[r], d2, __builtin_fmaf(w1[r], d1, w0[r] * d0));
            g[r] = e4[r] > 0.f ? x : 0.f;
            v[4 * c + r] = g[r];
          }
          *reinterpret_cast<f32x4*>(gerow + f) = g;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned hi = pack_bf16x2(v[2 * w], v[2 * w + 1]);
          ah[ks][w] = hi;
          al[ks][w] = pack_bf16x2(v[2 * w] - bflo(hi), v[2 * w + 1] - bfhi(hi));
        }
      }
#pragma unroll
      for (int ks = 8; ks < 16; ++ks) { ah[ks] = u32x4{0u, 0u, 0u, 0u}; al[ks] = u32x4{0u, 0u, 0u, 0u}; }
    }
    // layer 0: d_g = W_e[:, :W]^T d_e (a -> b; xyz_encoding_final has no activation)
    bwd_layer_x<8, false, false, true, BITS>(st, id, carry, ah, al, bh, bl, zero_off, n32, nullptr, grow + (long long)D * 256, 0u, 0.f);
    // layer 1: d_z_{D-1} = (W_f^T d_g + w_sigma d_sigma) [h_{D-1} > 0] (b -> a)
    bwd_layer_x<16, true, true, true, BITS>(st, id, carry, bh, bl, ah, al, zero_off, D >= 2 ? n32 : nfirst, arow + (long long)(D - 1) * LW,
                                      grow + (long long)(D - 1) * 256, sigw, go.w);
    // layers 2 .. D: d_z_{l-1} = (W_l^T d_z_l) [h_{l-1} > 0], l = D-1 .. 1 (a -> b, copied back); the last one only stores
    for (int i = 2; i < D; ++i) {
      const int l = D + 1 - i;
      bwd_layer_x<16, true, false, true, BITS>(st, id, carry, ah, al, bh, bl, zero_off, n32, arow + (long long)(l - 1) * LW,
                                         grow + (long long)(l - 1) * 256, 0u, 0.f);
#pragma unroll
      for (int t = 0; t < 16; ++t) { ah[t] = bh[t]; al[t] = bl[t]; }
    }
    if (!p.g_emb) {
      bwd_layer_x<16, true, false, false, BITS>(st, id, carry, ah, al, bh, bl, zero_off, nfirst, arow, grow, 0u, 0.f);
    } else {
      // d emb = W_0[:, :63]^T d_z_0 (+ W_skip[:, :63]^T d_z_skip): two 32-row tiles each, K = W
      bwd_layer_x<16, true, false, true, BITS>(st, id, carry, ah, al, bh, bl, zero_off, n32, arow, grow, 0u, 0.f);     // d_z_0 as an operand too
      f32x16 ge[2];
      bwd_emb_x(st, id, carry, bh, bl, zero_off, p.skip > 0 ? n32 : nfirst, ge);
      if (p.skip > 0) {
        // d_z_skip was stored by this very lane several layers ago (the panel barriers' vmcnt waits retired the stores)
        const float* zrow = grow + (long long)p.skip * 256;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(zrow + 16 * ks), v1 = *reinterpret_cast<const f32x4*>(zrow + 16 * ks + 8);
          const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const unsigned hi = pack_bf16x2(v[2 * w], v[2 * w + 1]);
            ah[ks][w] = hi;
            al[ks][w] = pack_bf16x2(v[2 * w] - bflo(hi), v[2 * w + 1] - bfhi(hi));
          }
        }
        f32x16 g2[2];
        bwd_emb_x(st, id, carry, ah, al, zero_off, nfirst, g2);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) ge[t][r] += g2[t][r];
      }
      if (valid) {
        float* er = p.g_emb + s * 64 + 4 * id.h;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ge[t][4 * q + i];
            *reinterpret_cast<f32x4*>(er + 32 * t + 8 * q) = v;
          }
      }
    }
  }
  wait_vm0();
}

}  // namespace bf
}  // namespace mf

using namespace mf;

static bool bwd3_supported(const mf_nerf_desc* d) {
  NetLayout F;
  return d && nerf_layout(*d, F, MF_PREC_BF16X3) && F.W == 256 && d->D >= 2;
}
static int bwd3_n_emb(const mf_nerf_desc* d) {           // embedded-input layers behind the chain (0 = g_emb unsupported)
  const int sk = bf::bwd3_skip_layer(*d);
  return sk < 0 ? 0 : (sk > 0 ? 2 : 1);
}

extern "C" int64_t mf_nerf_bwd3_packed_bytes(const mf_nerf_desc* d) {
  if (!bwd3_supported(d)) { fail(MF_E_UNSUPPORTED, "mf_nerf_bwd3_packed_bytes: unsupported NeRF configuration"); return 0; }
  return bf::kB3ResBytes + bf::bwd3_groups_total(d->D, bwd3_n_emb(d)) * kGroupBytes;
}

extern "C" int32_t mf_nerf_pack_bwd3(const mf_nerf_desc* d, void* packed, void* stream) {
  if (!d || !packed) return fail(MF_E_INVALID, "mf_nerf_pack_bwd3: null argument");
  if (!bwd3_supported(d)) return fail(MF_E_UNSUPPORTED, "mf_nerf_pack_bwd3: unsupported NeRF configuration (W=%d D=%d)", d->W, d->D);
  bf::Bwd3PackJob job{};
  const int ext = d->extra_feat_type == MF_EXTRA_NONE ? 0 : d->extra_feat_dim;
  long long g0 = 0;
  job.W[0] = d->extra_w; job.ld[0] = 256 + ext; job.col0[0] = 0; job.ncols[0] = 256; job.gpt[0] = 16; job.g0[0] = g0; g0 += 8 * 16;
  job.W[1] = d->final_w; job.ld[1] = 256; job.col0[1] = 0; job.ncols[1] = 256; job.gpt[1] = 32; job.g0[1] = g0; g0 += 8 * 32;
  for (int i = 2; i <= d->D; ++i) {
    const int l = d->D + 1 - i;
    const bool emb = ((1u | d->skip_mask) >> l) & 1u;
    job.W[i] = d->trunk_w[l];
    job.ld[i] = (emb ? d->in_channels_xyz : 0) + 256;
    job.col0[i] = emb ? d->in_channels_xyz : 0;
    job.ncols[i] = 256; job.gpt[i] = 32; job.g0[i] = g0; g0 += 8 * 32;
  }
  job.n_layers = d->D + 1;
  const int n_emb = bwd3_n_emb(d), sk = bf::bwd3_skip_layer(*d);
  for (int e = 0; e < n_emb; ++e) {
    const int i = job.n_layers++, l = e == 0 ? 0 : sk;
    job.W[i] = d->trunk_w[l];
    job.ld[i] = (l == 0 ? 0 : 256) + d->in_channels_xyz;
    job.col0[i] = 0; job.ncols[i] = d->in_channels_xyz; job.gpt[i] = 32; job.g0[i] = g0; g0 += 2 * 32;
  }
  job.g0[job.n_layers] = g0;
  for (int i = 0; i < job.n_layers; ++i)
    if (!job.W[i]) return fail(MF_E_INVALID, "mf_nerf_pack_bwd3: missing weight pointer (backward layer %d)", i);
  if (!d->sigma_w || !d->rgb_w) return fail(MF_E_INVALID, "mf_nerf_pack_bwd3: missing sigma / rgb weight");
  job.sigma_w = d->sigma_w; job.rgb_w = d->rgb_w;
  job.res = static_cast<float*>(packed);
  job.panels = reinterpret_cast<unsigned*>(static_cast<char*>(packed) + bf::kB3ResBytes);
  job.total_groups = bf::bwd3_groups_total(d->D, n_emb);
  if (g0 != job.total_groups) return fail(MF_E_INVALID, "mf_nerf_pack_bwd3: layout mismatch");
  const long long slots = job.total_groups * 64;
  hipLaunchKernelGGL(bf::pack_bwd3_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), job);
  return check_launch("mf_nerf_pack_bwd3");
}

extern "C" int32_t mf_nerf_backward3(const mf_nerf_desc* d, const void* packed_bwd3, int64_t P, const float* g_out,
                                     const float* acts, int64_t stride, const float* rgbsigma, float* gpre, float* ghead,
                                     float* g_emb, const uint32_t* mask, int64_t mask_stride, void* stream) {
  if (!d || !packed_bwd3 || (P > 0 && (!g_out || (!acts && !mask) || !rgbsigma || !gpre || !ghead)))
    return fail(MF_E_INVALID, "mf_nerf_backward3: null argument");
  if (!bwd3_supported(d)) return fail(MF_E_UNSUPPORTED, "mf_nerf_backward3: unsupported NeRF configuration (W=%d D=%d)", d->W, d->D);
  if (mask && mask_stride < (int64_t)(d->D + 2) * 8) return fail(MF_E_INVALID, "mf_nerf_backward3: mask_stride %lld < 8 (D + 2)", (long long)mask_stride);
  if (stride < (int64_t)(d->D + 1) * 256 + 128 || (stride & 3) || (reinterpret_cast<uintptr_t>(acts) & 15) || (reinterpret_cast<uintptr_t>(gpre) & 15))
    return fail(MF_E_INVALID, "mf_nerf_backward3: dump rows must be 16-byte aligned, stride >= (D + 1) W + W / 2 and a multiple of 4 floats");
  if (g_emb && bwd3_n_emb(d) == 0)
    return fail(MF_E_UNSUPPORTED, "mf_nerf_backward3: the embedded-input gradient is built for at most one skip layer");
  if (P == 0) return MF_OK;
  bf::Bwd3Params p{};
  p.g_emb = g_emb; p.skip = bf::bwd3_skip_layer(*d) > 0 ? bf::bwd3_skip_layer(*d) : 0;
  p.mask = mask; p.mask_stride = mask_stride;
  p.net.packed = static_cast<const char*>(packed_bwd3);
  p.net.res_lds = 0; p.net.res_bytes = bf::kB3ResBytes; p.net.D = d->D; p.net.emb_mask = 0; p.net.aux = 0;
  p.P = P; p.stride = stride; p.g_out = g_out; p.acts = acts; p.rgbsigma = rgbsigma; p.gpre = gpre; p.ghead = ghead;
  uint32_t lds = bf::kB3ResBytes;
  p.ring_off = lds; p.buf_bytes = 32 * kGroupBytes; lds += 3 * p.buf_bytes;
  const long long ntiles = (P + 127) / 128;
  const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
  void (*kern)(const bf::Bwd3Params) = mask ? bf::nerf_backward_kernel_x3<true> : bf::nerf_backward_kernel_x3<false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(MF_E_LAUNCH, "mf_nerf_backward3: cannot reserve %u bytes of LDS", lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return check_launch("mf_nerf_backward3");
}
