// mf_bwd.hpp -- one layer of an input-gradient chain on the fused MLP core (shared by the NeRF and NoF
// backward kernels):  out = W^T-panels * [sig ; in]  (+ ReLU mask of the forward activation the output
// is the gradient of), optionally stored to this lane's row of the gradient buffer.
#pragma once
#include "mf_core.hpp"

namespace mf {

constexpr int kBwdSigSteps = 4;   // k-steps of the optional leading block (NeRF: d_sigma against sigma.weight)

// MODE: 2 = hidden input only, 3 = [sig ; hidden].  NKI = k-tiles of the input (16 features each),
// NPO = output panels (32 features each).  MASK: multiply by [mask_row > 0]; STORE: write the result to
// store_row (both rows in the activation dump's natural feature order, offset to this layer's slot).
template <int MODE, int NKI, int NPO, bool MASK, bool STORE>
MF_D void bwd_layer(const f32x4 (&in)[NKI], const float (&sig)[kBwdSigSteps], f32x4 (&out)[2 * NPO], int groups,
                    uint32_t zero_bias, Stream& st, CarryT<kPD>& carry, const LaneId& id,
                    const NextLayer& nxt, const float* mask_row, float* store_row) {
#pragma unroll
  for (int t = 0; t < NPO; ++t) {
    const uint32_t p = st.slot_off(0) + id.lane * 16;
    const uint32_t pn = st.slot_off(1) + id.lane * 16;
    f32x4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = {1.f, 1.f, 1.f, 1.f};
    auto hook = [&](int ph) {
      st.template sync_and_dma<true>(t + 2 < NPO ? groups : nxt.groups, t == NPO - 2 ? nxt.jump : nullptr, id, ph);
      if (ph == 1) return;
      if constexpr (MASK) {     // behind the barrier: in flight for the rest of the panel
        m0 = *reinterpret_cast<const f32x4*>(mask_row + 32 * t + 4 * id.g);
        m1 = *reinterpret_cast<const f32x4*>(mask_row + 32 * t + 16 + 4 * id.g);
      }
    };
    const bool late = id.wave < kWaves / 2;
    f32x4 E, O;
    out_pair<MODE, NKI, kBwdSigSteps>(carry, in, sig, p, pn, zero_bias, id.g, late, hook, -__builtin_inff(), E, O);
    if constexpr (MASK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        E[i] = m0[i] > 0.f ? E[i] : 0.f;
        O[i] = m1[i] > 0.f ? O[i] : 0.f;
      }
    }
    if constexpr (STORE) {
      *reinterpret_cast<f32x4*>(store_row + 32 * t + 4 * id.g) = E;
      *reinterpret_cast<f32x4*>(store_row + 32 * t + 16 + 4 * id.g) = O;
    }
    st.keep2 = (STORE && !(st.dbg & 4)) ? 2 : 0;   // the two youngest VM operations are this panel's row stores
    out[2 * t] = E;
    out[2 * t + 1] = O;
    st.advance();
  }
}

}  // namespace mf
