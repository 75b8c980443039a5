/*
 * mocoflow_hip.h -- C ABI of the MI355X (gfx950) volume-rendering path of MoCo-Flow.
 *
 * The reference (wyysf-98/MoCo_Flow) is pure Python/PyTorch and has NO native
 * boundary of its own; this ABI is the build-defined drop-in surface described in
 * SURVEY.md §8(b).  Every entry point below names the reference interface it
 * replaces (file:line relative to the reference repository).
 *
 * Conventions
 *   - extern "C", POD only: device pointers, 32/64-bit sizes, a hipStream_t passed
 *     as void*.  No torch types, no exceptions, no allocation, no ownership
 *     transfer: the caller allocates every output and workspace buffer.
 *   - All pointers are DEVICE pointers (fp32 / int32 as documented) unless marked
 *     "host".  Descriptor structs are read on the host during the call.
 *   - Return value: 0 = ok; <0 = error (MF_E_*), text via mf_last_error()
 *     (thread-local).  Work is enqueued on `stream`; nothing synchronises.
 *   - The library is stateless; "packed weights" are plain caller-owned device
 *     buffers produced by mf_*_pack from the PyTorch parameter tensors.
 */
#ifndef MOCOFLOW_HIP_H
#define MOCOFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: activation dump of the training forward; 4: mf_nerf_backward, mf_weight_grads; 5: NoF backward
 * (mf_nof_points_dump, mf_nof_backward); 6: mf_composite_backward, mf_image_compose;
 * 7: mf_nof_forward_dump; 8: mf_loss_partials, new packed layout of MF_PREC_BF16 (32x32x16 fragments);
 * 9: mf_valid_rays_mask, mf_nerf_backward_x (embedded-input gradient in the chain launch), mf_embedding_backward;
 * 10: mf_nerf_forward_dump, mf_render_args.dump_nof_* (+ mf_nof_emb_slot_features), mf_smpl_lbs,
 *     mf_smpl_frame_transforms, mf_apply_vertex_transforms
 * 12: MF_PREC_BF16 NoF takes its image-index block as a per-ray fp32 bias: mf_render_args.workspace(+_bytes),
 *     mf_render_workspace_bytes, mf_render_prepare, mf_loss_partials_backward, perturb arguments of mf_z_vals, mf_points_sigma_workspace_bytes, workspace arguments of mf_points_sigma_p
 * 13: MF_PREC_BF16X3 is the full three-product mode (own packed layout); mf_weight_grads_p, mf_weight_grads_scratch_bytes_p,
 *     mf_nerf_backward3 (+ mf_nerf_bwd3_packed_bytes, mf_nerf_pack_bwd3), MF_PREC_BF16X3 in mf_points_sigma_p and with the NeRF dump
 * 14: mf_embedding_forward_rows
 * 15: MF_PREC_BF16X3 packs the NoF in three-term operands (six products per k-step); mf_nof_backward3 (+ mf_nof_bwd3_packed_bytes,
 *     mf_nof_pack_bwd3); ReLU bit rows: four panels per word, written for NoF evaluations (behind T) and under NoF; mf_sample_pdf_eps
 * 16: MF_PREC_BF16X3 packs the NoF as IEEE-half (hi, lo) pairs at 2^5 x (three products per k-step on the f16 matrix
 *     instruction, biases at 2^10 x); MF_PREC_BF16X3 training forward under NoF (mf_render_args.dump_nof_acts / dump_nof_out without
 *     dump_nof_emb) + mf_nof_embed_rows */
#define MF_ABI_VERSION 16

enum {
  MF_OK = 0,
  MF_E_INVALID = -1,     /* bad argument / unsupported configuration */
  MF_E_LAUNCH = -2,      /* HIP launch or runtime failure            */
  MF_E_UNSUPPORTED = -3  /* valid for the reference, not built here  */
};

enum { MF_MAX_FREQS = 16, MF_MAX_LAYERS = 16 };

/* Arithmetic of the W-wide ("hidden") GEMMs of the fused pass.  F32: exact-fp32 MFMA everywhere
 * (the reference's arithmetic; BASELINE configs C1-C2).  BF16: hidden-layer weights and
 * activations rounded to bf16 (RNE), fp32 accumulate (v_mfma_f32_32x32x16_bf16, 32 samples per wave); the
 * embedded-input k-ranges and the NoF head use a two-term bf16 split (16 mantissa bits); biases, the NeRF
 * heads and the composite stay fp32 (BASELINE configs C3-C5).
 * BF16X3 (ABI v12; NoF in three-term bf16 operands in v15, in IEEE-half pairs since v16): the fp32 CONTRACT on the bf16 pipe (1e-4 max-rel on every per-ray
 * output, measured <= 4.6e-5 on the reference-generated golden vectors and <= 3.1e-5 through the MoCo chains at 4096 rays on two
 * weight draws).  The NeRF's matrix products as a two-term bf16 split of activations AND weights (hi + lo, 16 mantissa bits
 * each), three products hi*hi + hi*lo + lo*hi; the NoFs' -- whose output point feeds sin(512 x) -- as IEEE-half (hi, lo) pairs of
 * 2^5 x the value (22 significand bits), the same three products on v_mfma_f32_32x32x16_f16; fp32 accumulation; sigma / rgb heads as fp32 dot products on the fp32 accumulators; the
 * NoF's image index an exact fp32 per-ray bias.  Three matrix instructions per product where F32 costs sixteen:
 * 0.33-0.4 of F32's time.  Own packed layout (every k-step a (hi, lo) group pair: bf16 in the NeRF, halves in the NoF).  Render passes
 * (with the activation dumps of the training forward, v16: under NoF too) and mf_points_sigma_p (scalar
 * image index or no NoF). */
enum { MF_PREC_F32 = 0, MF_PREC_BF16 = 1, MF_PREC_BF16X3 = 2 };

/* ---- Embedding: models/embedding.py:4-47 ------------------------------------
 * out = [x, w0*sin(f0 x), w0*cos(f0 x), w1*sin(f1 x), ...]; freq[] are the module's
 * freq_bands (2^k or linear), weight[] its per-frequency weights (set_weights /
 * trainer_moco_flow.py:289,301).  n_freqs == 0 is legal (identity). */
typedef struct mf_embedding {
  int32_t in_channels;             /* 3 (xyz, dir) or 1 (ind)             */
  int32_t n_freqs;                 /* 0..MF_MAX_FREQS                     */
  float freq[MF_MAX_FREQS];
  float weight[MF_MAX_FREQS];
} mf_embedding;

/* ---- NeRF: models/nerf.py:5-102 ---------------------------------------------
 * Parameter pointers follow the reference's state_dict:
 *   trunk_w[i] / trunk_b[i]  = xyz_encoding_{i+1}.0.{weight,bias}
 *   final_w / final_b        = xyz_encoding_final.{weight,bias}
 *   extra_w / extra_b        = extra_encoding.0.{weight,bias}   (W/2, W+extra_feat_dim)
 *   sigma_w / sigma_b        = sigma.{weight,bias}              (1, W)
 *   rgb_w / rgb_b            = rgb.0.{weight,bias}              (3, W/2)
 * All row-major (out, in) fp32 as nn.Linear stores them. */
enum { MF_EXTRA_NONE = 0, MF_EXTRA_IND = 1, MF_EXTRA_DIR = 2 };

typedef struct mf_nerf_desc {
  int32_t D;                       /* trunk layers, 2..MF_MAX_LAYERS-1       */
  int32_t W;                       /* hidden width: 256 (bf16 modes, backward); the fp32 forward also takes 128 */
  int32_t in_channels_xyz;         /* 63 (= 3*(2*10+1)); narrower embeddings are zero-padded to it by the caller */
  uint32_t skip_mask;              /* bit i set <=> i in skips (nerf.py:84-86) */
  int32_t extra_feat_type;         /* MF_EXTRA_*                             */
  int32_t extra_feat_dim;          /* columns of extra_encoding beyond W     */
  const float* trunk_w[MF_MAX_LAYERS];
  const float* trunk_b[MF_MAX_LAYERS];
  const float* final_w; const float* final_b;
  const float* extra_w; const float* extra_b;
  const float* sigma_w; const float* sigma_b;
  const float* rgb_w;   const float* rgb_b;
} mf_nerf_desc;

/* ---- NoF: models/nof.py:6-85 --------------------------------------------------
 *   trunk_w[i] / trunk_b[i] = nof_encoding_{i+1}.0.{weight,bias}
 *   head_w / head_b         = nof_encoding_final.{weight,bias}  (9|3, W) */
typedef struct mf_nof_desc {
  int32_t D;                       /* trunk layers                           */
  int32_t W;                       /* hidden width: 128                      */
  int32_t in_channels_xyz;         /* 33                                     */
  int32_t extra_feat_dim;          /* 33 (ind embedding width)               */
  uint32_t skip_mask;
  int32_t use_quat;                /* nof.py:75-82                           */
  const float* trunk_w[MF_MAX_LAYERS];
  const float* trunk_b[MF_MAX_LAYERS];
  const float* head_w; const float* head_b;
} mf_nof_desc;

int32_t mf_version(void);
const char* mf_last_error(void);

/* Size in bytes of the packed-weights buffer for a network (0 on unsupported). */
int64_t mf_nerf_packed_bytes(const mf_nerf_desc* d);
int64_t mf_nof_packed_bytes(const mf_nof_desc* d);
/* Re-order the nn.Linear tensors into the kernels' MFMA fragment stream.
 * Must be re-run whenever the parameters change (optimizer.step). */
int32_t mf_nerf_pack(const mf_nerf_desc* d, void* packed, void* stream);
int32_t mf_nof_pack(const mf_nof_desc* d, void* packed, void* stream);
/* The same with an explicit MF_PREC_* (the plain forms are MF_PREC_F32).  A packed buffer can only
 * be used by a render pass of the same precision. */
int64_t mf_nerf_packed_bytes_p(const mf_nerf_desc* d, int32_t precision);
int64_t mf_nof_packed_bytes_p(const mf_nof_desc* d, int32_t precision);
int32_t mf_nerf_pack_p(const mf_nerf_desc* d, int32_t precision, void* packed, void* stream);
int32_t mf_nof_pack_p(const mf_nof_desc* d, int32_t precision, void* packed, void* stream);

/* Embedding.forward, models/embedding.py:30-47:  x (B, in_channels) -> out (B, C*(2F+1)). */
int32_t mf_embedding_forward(const mf_embedding* e, const float* x, int64_t B, float* out, void* stream);
/* The same for B OUTPUT rows of `out_stride` >= C*(2F+1) floats (the columns past the embedding are written 0;
 * out_stride <= 0: the embedding's width), output row b embedding input row b / repeat (x has ceil(B / repeat)
 * rows): the embedded inputs as the weight-gradient launches read them (64 / 32-column operands of mf_weight_grads;
 * the per-ray `ind` / `dir` embedding of rendering.py:133-146 repeated for the ray's samples) in ONE launch
 * instead of embedding + repeat_interleave + zero pad. */
int32_t mf_embedding_forward_rows(const mf_embedding* e, const float* x, int64_t B, int32_t repeat, float* out,
                                  int64_t out_stride, void* stream);

/* NeRF.forward, models/nerf.py:61-102: inputs (B, in_channels_xyz [+ extra_feat_dim]) row
 * stride `in_stride` floats -> out (B,4) = [rgb, sigma], or (B,1) sigma when sigma_only. */
int32_t mf_nerf_forward(const mf_nerf_desc* d, const void* packed, const float* inputs,
                        int64_t in_stride, int64_t B, int32_t sigma_only, float* out, void* stream);

/* The same call when gradients are wanted (trainer_moco_flow.py:146-157 `forwarf_nerf`, :337-362: NeRF called on
 * embedded points with requires_grad inputs): the full forward (out (B,4)) that also writes the per-sample layer
 * outputs [h_0 .. h_{D-1} | xyz_encoding_final | extra_encoding] to dump_acts (B, dump_stride >= (D+1) W + W/2),
 * the layout mf_nerf_backward / mf_weight_grads consume -- the module-level counterpart of
 * mf_render_args.dump_acts.  (ABI v10) */
int32_t mf_nerf_forward_dump(const mf_nerf_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                             int64_t B, float* out, float* dump_acts, int64_t dump_stride, void* stream);

/* NoF.forward, models/nof.py:55-85: inputs (B, in_channels_xyz+extra_feat_dim), xyz (B,3) -> (B,3). */
int32_t mf_nof_forward(const mf_nof_desc* d, const void* packed, const float* inputs,
                       int64_t in_stride, const float* xyz, int64_t B, float* out, void* stream);

/* ---- backward of NeRF.forward over the training forward's dump (ABI v4) -------------------
 * Replaces the autograd graph torch records for models/nerf.py:78-102 under loss.backward()
 * (trainer/base.py:188-197).  The input-gradient chain (nine W-wide contractions per sample) runs
 * on the same register-resident MFMA core as the forward, on the transposed weights:
 *   mf_nerf_bwd_packed_bytes / mf_nerf_pack_bwd : transposed fragment stream of a NeRF (fp32, W=256);
 *     re-run whenever the parameters change.
 *   mf_nerf_backward : g_out (P,4) = dL/d[rgb (after the sigmoid), sigma], acts (P,stride) and
 *     rgbsigma (P,4) = mf_render_args.dump_acts / dump_rgbsigma of the forward ->
 *       gpre  (round_up(P,128), stride): pre-activation gradients in the dump's layout
 *             [d z_0 .. d z_{D-1} | d xyz_encoding_final | d extra_encoding]  (rows >= P are scratch)
 *       ghead (P,4): [d rgb pre-sigmoid (3), d sigma]
 *     The weight gradients are then plain GEMMs on (acts, gpre):  dW_l = gpre_l^T in_l,  db_l = sum gpre_l. */
int64_t mf_nerf_bwd_packed_bytes(const mf_nerf_desc* d);
int32_t mf_nerf_pack_bwd(const mf_nerf_desc* d, void* packed, void* stream);
int32_t mf_nerf_backward(const mf_nerf_desc* d, const void* packed_bwd, int64_t P, const float* g_out,
                         const float* acts, int64_t stride, const float* rgbsigma, float* gpre,
                         float* ghead, void* stream);

/* The same launch, additionally producing the gradient of the EMBEDDED INPUT (ABI v9): g_emb (P,64), natural column
 * order of the 63-wide xyz embedding (column 63 = 0), = xyz_encoding_1[:, :63]^T d_z_0 + (one skip layer)
 * xyz_encoding_{skip+1}[:, :63]^T d_z_skip, as two more panels of the transposed stream behind the chain; NULL =
 * mf_nerf_backward.  mf_nerf_pack_bwd packs those panels; more than one skip layer: MF_E_UNSUPPORTED. */
int32_t mf_nerf_backward_x(const mf_nerf_desc* d, const void* packed_bwd, int64_t P, const float* g_out,
                           const float* acts, int64_t stride, const float* rgbsigma, float* gpre,
                           float* ghead, float* g_emb, void* stream);
/* The chain of mf_nerf_backward in three bf16 products (ABI v13; csrc/mf_backward_bf16.hip): gradients and transposed
 * weights as (hi, lo) bf16 pairs, fp32 accumulation, ReLU masks from the dump as in the fp32 chain (no unit changes side:
 * the result differs by the 2^-16 of the split operands).  Own packed stream (mf_nerf_bwd3_packed_bytes /
 * mf_nerf_pack_bwd3); same arguments and outputs as mf_nerf_backward_x (g_emb may be NULL).  `mask` (nullable): the
 * forward's mf_render_args.dump_mask rows -- the chain then reads no activation at all (`acts` may be NULL).  W = 256, D >= 2. */
int64_t mf_nerf_bwd3_packed_bytes(const mf_nerf_desc* d);
int32_t mf_nerf_pack_bwd3(const mf_nerf_desc* d, void* packed, void* stream);
int32_t mf_nerf_backward3(const mf_nerf_desc* d, const void* packed_bwd3, int64_t P, const float* g_out,
                          const float* acts, int64_t stride, const float* rgbsigma, float* gpre,
                          float* ghead, float* g_emb, const uint32_t* mask, int64_t mask_stride, void* stream);
/* Backward of Embedding.forward (models/embedding.py:42-46) through the embedded values themselves:
 * g_x[c] = g_emb[c] + sum_k f_k (emb[cos_kc] g_emb[sin_kc] - emb[sin_kc] g_emb[cos_kc]);  g_emb (P, >= C(2F+1))
 * with row stride g_stride, emb = the forward's output rows (stride e_stride), g_x (P, C). */
int32_t mf_embedding_backward(const mf_embedding* e, const float* g_emb, int64_t g_stride, const float* emb,
                              int64_t e_stride, int64_t P, float* g_x, void* stream);

/* Backward of the alpha-composite of nerf_inference (models/rendering.py:157-192) on per-sample planes:
 * g_rgb (N,3) / g_depth (N) / g_opacity (N) (any may be NULL) = dL/d of the pass's outputs ->
 * g_rgbsigma (N*S,4) = dL/d[rgb (after the sigmoid), raw sigma] per sample, ready for mf_nerf_backward.
 * rgbsigma = mf_render_args.dump_rgbsigma of the forward; z_vals / noise / activation / background as in
 * the forward.  Nothing flows to z_vals (rendering.py:323).  S <= 2048. */
int32_t mf_composite_backward(const float* rays, int64_t ray_stride, int64_t n_rays, int32_t S,
                              const float* z_vals, const float* rgbsigma, const float* noise,
                              int32_t activation, const float* background, const float* g_rgb,
                              const float* g_depth, const float* g_opacity, float* g_rgbsigma, void* stream);

/* Weight / bias gradients of a set of linear layers over P samples, ONE persistent launch:
 *     dW_i = G_i[:P]^T X_i[:P]   (n_out x n_in),      db_i = sum_s G_i[s]   (n_out)
 * G_i / X_i: fp32 row-major device matrices (column slices of the mf_nerf_backward gradient buffer, of
 * the forward's activation dump, or of the embedded inputs), 16-byte aligned, strides multiples of 4
 * floats.  Supported blocks (n_out x n_in): NeRF 256x256, 256x64, 128x256, 128x32 and 4x640 (the two heads
 * at once: G = ghead (P,4), X = dump columns [h_D | final | extra]; no head reads `final`: its 256 columns
 * are not fetched and columns 256..511 of this block's dW are returned as 0); NoF 128x128, 128x80 (embedded input,
 * 66 -> 80) and 12x128 (head, G = d T padded to 12 columns).  dW is written as a dense
 * (max(n_out,16), n_in) matrix, db (optional, may be NULL) as max(n_out,16) floats.  Deterministic
 * (fixed-order partial sums through `scratch`, no atomics).  Replaces the dW/db halves of torch's
 * addmm backward for models/nerf.py:78-102. */
#define MF_WG_MAX_ITEMS 32
typedef struct mf_wgrad_item {
  const float* G; int64_t g_stride; int32_t n_out;
  const float* X; int64_t x_stride; int32_t n_in;
  float* dW; float* db;
} mf_wgrad_item;
int64_t mf_weight_grads_scratch_bytes(const mf_wgrad_item* items, int32_t n_items, int64_t P);
int32_t mf_weight_grads(const mf_wgrad_item* items, int32_t n_items, int64_t P, void* scratch, void* stream);
/* The same with the arithmetic of the contraction chosen (ABI v13).  MF_PREC_F32: exact-fp32 MFMA (the two calls above).
 * MF_PREC_BF16X3: every block but the heads' 4x640 (round 5; before: 256x256 and 128x256 only) contracts G and X as two-term
 * bf16 splits, three bf16 products per 16-sample step with fp32 accumulation (16 mantissa bits per operand; a fifth of the
 * fp32 pipe's matrix time, the launch then runs against the HBM reads of its operands) -- 256x64 on a 256x128 block, the NoF's
 * 128x128 / 128x80 / 12x128 and 128x32 on a 128x128 block, columns beyond the operands' widths zero; dW keeps the (rows, n_in)
 * layout of the fp32 call, the operand strides must be even.  The heads block stays fp32.  The scratch size depends on the
 * precision. */
int64_t mf_weight_grads_scratch_bytes_p(int32_t precision, const mf_wgrad_item* items, int32_t n_items, int64_t P);
int32_t mf_weight_grads_p(int32_t precision, const mf_wgrad_item* items, int32_t n_items, int64_t P, void* scratch, void* stream);

/* ---- backward of one NoF evaluation on points (rendering.py:49-83 + nof.py:69-82; ABI v5) ---------
 * mf_nof_points_dump: pts (P,3), per-ray image index ind[ray * ind_stride] with ray = sample / S ->
 *   out (P,3) and the dump the backward reads: acts (P,stride) = [h_1 .. h_D | T padded to 16],
 *   emb (P,80) = the embedded input [xyz 33 | ind 33 | 0] in the reference's column order.
 * mf_nof_bwd_packed_bytes / mf_nof_pack_bwd: transposed fragment stream (W=128, at most one skip layer).
 * mf_nof_backward: g_out (P,3) = dL/d out -> g_pts (P,3, may be NULL) = dL/d pts and
 *   gpre (round_up(P,128), stride) = [d z_0 .. d z_{D-1} | d T padded to 16] for mf_weight_grads. */
int32_t mf_nof_points_dump(const mf_nof_desc* d, const void* packed, const mf_embedding* emb_xyz,
                           const mf_embedding* emb_ind, const float* pts, const float* ind,
                           int64_t ind_stride, int32_t S, int64_t P, float* out, float* acts,
                           int64_t stride, float* emb, void* stream);
/* NoF.forward (models/nof.py:55-85) on pre-embedded inputs, storing the same activation dump: the forward
 * half of the module-level training call (trainer_nof.py:85-112, trainer_moco_flow.py:159-187). */
int32_t mf_nof_forward_dump(const mf_nof_desc* d, const void* packed, const float* inputs, int64_t in_stride,
                            const float* xyz, int64_t B, float* out, float* acts, int64_t stride, void* stream);
int64_t mf_nof_bwd_packed_bytes(const mf_nof_desc* d);
int32_t mf_nof_pack_bwd(const mf_nof_desc* d, void* packed, void* stream);
int32_t mf_nof_backward(const mf_nof_desc* d, const void* packed_bwd, const mf_embedding* emb_xyz, int64_t P,
                        const float* pts, const float* acts, int64_t stride, const float* g_out,
                        float* gpre, float* g_pts, void* stream);
/* mf_nof_backward in three bf16 products (ABI v15; csrc/mf_nofgrad_bf16.hip): the hidden contractions and the embedded-input
 * layers on gradients and transposed weights as (hi, lo) bf16 pairs with fp32 accumulation, ReLU masks from the dump as in
 * the fp32 kernel (no unit changes side), the transform's backward / head product / sin-cos chain rule in fp32.  Own packed
 * stream (mf_nof_bwd3_packed_bytes / mf_nof_pack_bwd3); same arguments and outputs as mf_nof_backward; dump rows 16-byte
 * aligned.  Replaces the same autograd range: models/rendering.py:49-83 + models/nof.py:69-82 under loss.backward()
 * (trainer/base.py:188-197). */
int64_t mf_nof_bwd3_packed_bytes(const mf_nof_desc* d);
int32_t mf_nof_pack_bwd3(const mf_nof_desc* d, void* packed, void* stream);
int32_t mf_nof_backward3(const mf_nof_desc* d, const void* packed_bwd3, const mf_embedding* emb_xyz, int64_t P,
                         const float* pts, const float* acts, int64_t stride, const float* g_out,
                         float* gpre, float* g_pts, void* stream);

/* Fused point query: xyz (B,3) -> [backward NoF at image index ind] -> positional encoding -> NeRF
 * trunk -> raw sigma (B,), one launch.  Replaces the per-chunk module sequence forward_nof /
 * nerf_embedding_xyz / zero-pad / NeRF(sigma_only=True) of trainer_moco_flow.py:146-187 and the
 * lattice loop of visualize_mesh (trainer_moco_flow.py:500-526, trainer_nerf.py:215-245).
 * nof == NULL: canonical-space query.  ind: per-point indices (B,) or NULL -> ind_scalar for all.
 * canon (B,3), optional: the point after the backward flow. */
int32_t mf_points_sigma(const mf_nerf_desc* nerf, const void* nerf_packed, const mf_embedding* emb_xyz,
                        const mf_nof_desc* nof, const void* nof_packed, const mf_embedding* nof_emb_xyz,
                        const mf_embedding* nof_emb_ind, const float* xyz, const float* ind,
                        float ind_scalar, int64_t B, float* sigma, float* canon, void* stream);
/* The same query with the arithmetic of `precision` (MF_PREC_F32 | MF_PREC_BF16 | MF_PREC_BF16X3; the packed buffers must
 * have been packed for it): bf16 runs the hidden GEMMs of both networks on the bf16 matrix pipe like mf_render_pass does
 * (inference only; the 512^3 mesh-extraction lattice of test.py in ~0.15 s instead of ~0.9 s); bf16x3 (ABI v13) the
 * three-product kernels (fp32-class, ~0.3 s), with `ind` == NULL only (scalar image index, or no NoF): a per-point index
 * tensor returns MF_E_UNSUPPORTED.  (ABI v11) */
int32_t mf_points_sigma_p(int32_t precision, const mf_nerf_desc* nerf, const void* nerf_packed, const mf_embedding* emb_xyz,
                          const mf_nof_desc* nof, const void* nof_packed, const mf_embedding* nof_emb_xyz,
                          const mf_embedding* nof_emb_ind, const float* xyz, const float* ind,
                          float ind_scalar, int64_t B, float* sigma, float* canon, void* workspace,
                          int64_t workspace_bytes, void* stream);
/* Bytes of `workspace` the bf16 query needs when a NoF is given (ABI v12): the per-point (ind array) or single
 * (ind_scalar) fp32 bias  b_l + W_l[:, 33:66] emb(ind)  of the NoF layers that consume the embedded input -- in bf16 mode
 * the image index does not go through the matrix pipe (see mf_render_args.workspace).  0 for MF_PREC_F32 / no NoF. */
int64_t mf_points_sigma_workspace_bytes(int32_t precision, const mf_nof_desc* nof, int32_t per_point_ind, int64_t B);

/* ---- one rendering pass: nof_inference* + nerf_inference of models/rendering.py:49-192 as
 * called from render_rays (rendering.py:262-314 coarse, 329-373 fine) -------------------- */
enum {
  MF_ACT_RELU = 0, MF_ACT_SOFTPLUS = 1        /* rendering.py:169-172 */
};
enum {
  MF_F_SIGMA_ONLY   = 1 << 0,  /* weights_only=True: no rgb/depth (rendering.py:290-294)   */
  MF_F_CHAIN_LOCAL  = 1 << 1,  /* fw(bw(x,i),i) consensus (rendering.py:275-277)            */
  MF_F_CHAIN_GLOBAL = 1 << 2   /* fw_i(bw_j(fw_j(bw_i(x)))) (rendering.py:279-282)          */
};

typedef struct mf_render_args {
  /* rays (N, ray_stride>=9|10): o(3) d(3) near far img_ind [chained_img_ind]
   * (rendering.py:237-242) */
  const float* rays; int64_t ray_stride; int64_t n_rays;
  const float* background;          /* (N,3) or NULL (rendering.py:189-190)              */
  int32_t n_samples;                /* S of this pass                                     */
  /* depths: either z_vals (N,S) explicit (perturbed / fine pass), or z_steps (S,)
   * = torch.linspace(0,1,S) with use_disp choosing rendering.py:247 vs :249 */
  const float* z_vals; const float* z_steps; int32_t use_disp;
  const float* noise;               /* (N,S) sigma noise already scaled by noise_std, or NULL */
  int32_t activation;               /* MF_ACT_*                                           */
  int32_t flags;                    /* MF_F_*                                             */
  /* canonical NeRF */
  const mf_nerf_desc* nerf; const void* nerf_packed;
  mf_embedding emb_xyz;             /* nerf_embeddings[0]                                 */
  mf_embedding emb_extra;           /* nerf_embeddings[1] (ind) or [2] (dir); ignored for none */
  /* neural motion flow (NULL => plain NeRF) */
  const mf_nof_desc* nof_bw; const void* nof_bw_packed;
  const mf_nof_desc* nof_fw; const void* nof_fw_packed;
  mf_embedding nof_emb_xyz, nof_emb_ind;
  /* outputs (any may be NULL) */
  float* rgb;                       /* (N,3) */
  float* depth;                     /* (N,)  */
  float* opacity;                   /* (N,)  weights.sum(1)                               */
  float* weights;                   /* (N,S) */
  float* alphas;                    /* (N,S) */
  float* disp_local;                /* (N,S) mean_c |xyz - recon|         (rendering.py:310-311) */
  float* disp_global;               /* (N,S) mean_c |xyz - chained_recon| (rendering.py:313-314) */
  int32_t precision;                /* MF_PREC_*: must match how every *_packed buffer was packed      */
  /* training forward (MF_PREC_F32 only; all optional): what the backward needs, written once by the
   * fused kernel instead of being recomputed.  dump_acts (N*S, dump_stride): per sample the NeRF's
   * post-activation layer outputs in natural feature order [h_0 | ... | h_{D-1} | xyz_encoding_final |
   * extra_encoding] (dump_stride >= (D+1)*W + W/2 floats); dump_rgbsigma (N*S, 4): per-sample rgb (after
   * the sigmoid) and raw sigma; dump_xyz (N*S, 3): the point fed to the NeRF (canonical under NoF). */
  float* dump_acts; int64_t dump_stride;
  float* dump_rgbsigma;
  float* dump_xyz;
  /* The same for the NoF evaluations of the chain program (rendering.py:270-282), one plane per step k in
   * [bw(x,i), fw(canon,i), fw(canon,j), bw(.,j), fw(.,i)] (as many as the flags run), each N*S rows:
   * dump_nof_acts (steps, N*S, dump_nof_stride): [h_1 | ... | h_D | T (9 | 3) zero-padded to 16]
   * (dump_nof_stride >= D*W + 16), dump_nof_emb (steps, N*S, 80): the embedded input [xyz 33 | ind 33] in the
   * kernel's register-slot order (column c holds reference column mf_nof_emb_slot_features()[c], -1 = zero pad: the
   * weight gradient taken against it is un-permuted once per launch, on 128 x 80 numbers),
   * dump_nof_out (steps, N*S, 3): the step's output points.  This is what mf_nof_points_dump writes for one
   * evaluation (there in natural column order); with it the backward needs no forward re-evaluation of the chains.
   * All three or none.  dump_nof_plane[k] = plane (0 .. steps-1, a permutation) that step k writes: lets the caller
   * keep the evaluations of one network adjacent (bw: steps 0, 3; fw: steps 1, 2, 4), so that its weight gradients
   * are one contraction over all of them. */
  float* dump_nof_acts; int64_t dump_nof_stride;
  float* dump_nof_emb;
  float* dump_nof_out;
  int32_t dump_nof_plane[5];
  /* MF_PREC_BF16 with NoF (ABI v12): caller-allocated device scratch of mf_render_workspace_bytes(a) bytes holding,
   * per ray and (network, image index) combination of the chain program, the fp32 vectors  b_l + W_l[:, 33:66] emb(ind)
   * of the NoF layers that consume the embedded input (models/rendering.py:73-75, models/nof.py:69-73: the index block
   * is constant along a ray).  mf_render_prepare fills it (one small launch); mf_render_pass reads it as those layers'
   * initial accumulator values.  Ignored (may be NULL) otherwise. */
  void* workspace; int64_t workspace_bytes;
  /* ReLU bit mask of the NeRF's dumped activations (ABI v13; with dump_acts; ABI v15: passes with NoF too; NULL = none): dump_mask (N*S, dump_mask_stride
   * >= 8 (D + 2) words): layer l owns words 8 l .. 8 l + 7; output 32 t + f (f < 32) of the layer sits in word
   * 8 l + 4 (t / 4) + (f % 16) / 4 at bit 8 (t % 4) + 4 (f / 16) + f % 4 (ABI v15: the forward kernels' lane order, one 4-byte store per lane
   * and four 32-row panels; v13 / v14 stored a byte per panel) -- for the trunk layers
   * l = 0 .. D-1 and extra_encoding (l = D + 1, 4 words; l = D, xyz_encoding_final, is not written: no activation) -- what
   * mf_nerf_backward3 needs of the activations: 32 bytes instead of 1 KiB per layer and sample. */
  uint32_t* dump_mask; int64_t dump_mask_stride;
} mf_render_args;

/* mf_render_prepare(a) must have run on the same stream with the same rays / NoFs / nof_emb_ind / chain flags whenever
 * mf_render_workspace_bytes(a) > 0; the coarse and the fine pass of one render_rays call share one prepared workspace
 * (same rays, same NoFs).  A no-op (MF_OK) when no workspace is needed. */
int32_t mf_render_prepare(const mf_render_args* a, void* stream);
int32_t mf_render_pass(const mf_render_args* a, void* stream);
int64_t mf_render_workspace_bytes(const mf_render_args* a);

/* Column map of mf_render_args.dump_nof_emb: features80[c] = column of the NoF's embedded input ([xyz 33 | ind 33])
 * stored at dump column c, or -1 (zero).  Host-side, no stream. */
int32_t mf_nof_emb_slot_features(int32_t* features80);

/* The embedded input of a NoF evaluation as rows, NATURAL column order (ABI v16): out (P, 80) = [emb_xyz(pts) zero-padded to 33 |
 * ind_emb[r / S] (ind_width <= 33 columns: the index embedding of the point's ray, embedded ONCE per ray by the caller,
 * mf_embedding_forward) | 0], row r belongs to ray r / S.  Reference: models/rendering.py:70-75, models/embedding.py:42-47.
 * The X operand of the NoF's 128 x 80 weight-gradient blocks for training forwards that do not write
 * mf_render_args.dump_nof_emb themselves (MF_PREC_BF16X3).  xyz embedding: 3 channels, <= 5 frequencies. */
int32_t mf_nof_embed_rows(const mf_embedding* emb_xyz, const float* pts, const float* ind_emb, int32_t ind_width, int32_t S,
                          int64_t P, float* out, void* stream);

/* ---- hierarchical resampling: sample_pdf (rendering.py:5-46) [+ cat + sort, :321-326] ------
 * General form.  Per ray: n_bins bin positions -- either explicit `bins` (N, n_bins), the
 * reference's first argument, or the mid-points of z_coarse (N, n_bins+1) (rendering.py:321) --
 * and n_bins-1 weights starting at weights + ray*w_stride.  u: uniform draws (N, M) with row
 * stride u_stride, or ONE shared row with u_stride = 0 (the deterministic linspace(0,1,M)).
 * cdf_in (N, n_bins), optional: use this cdf instead of normalising/summing the weights.
 * Outputs (each optional): z_new_out (N,M) the drawn samples; inds_out (N,M) int32 the
 * searchsorted(right=True) indices; z_sorted_out (N, n_bins+1+M) = sort(cat(z_coarse, samples))
 * (needs z_coarse). */
int32_t mf_sample_pdf(const float* bins, const float* z_coarse, const float* weights, int64_t w_stride,
                      int64_t n_rays, int32_t n_bins, int32_t M, const float* u, int64_t u_stride,
                      const float* cdf_in, float* z_new_out, int32_t* inds_out, float* z_sorted_out,
                      void* stream);
/* The same with sample_pdf's `eps` argument (rendering.py:5, :20, :41-42; ABI v15): mf_sample_pdf is eps = 1e-5, the value of
 * every call the reference makes. */
int32_t mf_sample_pdf_eps(const float* bins, const float* z_coarse, const float* weights, int64_t w_stride,
                          int64_t n_rays, int32_t n_bins, int32_t M, const float* u, int64_t u_stride,
                          const float* cdf_in, float* z_new_out, int32_t* inds_out, float* z_sorted_out,
                          float eps, void* stream);
/* render_rays' use of it: z_coarse (N,S), weights (N,S) of the coarse pass (the kernel takes
 * weights[:,1:-1]); u (N,M) draws; writes the sorted union z_out (N,S+M). */
int32_t mf_sample_pdf_merge(const float* z_coarse, const float* weights, int64_t n_rays,
                            int32_t S, int32_t M, const float* u, float* z_out,
                            int32_t* inds_out, float* z_new_out, void* stream);

/* ---- consensus compaction, rendering.py:306-314: mask = alphas >= 0.01 (all-true if none);
 * out[k] = vals[p_k] for masked positions in row-major (ray, sample) order.
 * count (device int64[1]) receives the number of outputs; scratch >= mf_compact_scratch_bytes. */
int64_t mf_compact_scratch_bytes(int64_t n_rays);
int32_t mf_compact_mask(const float* alphas, const float* vals_a, const float* vals_b,
                        int64_t n_rays, int32_t S, float* out_a, float* out_b,
                        int64_t* count, void* scratch, void* stream);

/* ---- fused loss partials (second half): the additive pieces of MSELoss over both passes
 * (models/losses.py:4-14) and of the consensus means over mask = alphas >= 0.01, all-true if none
 * (models/rendering.py:306-314, trainer/trainer_moco_flow.py:317-328), from the arrays a render pass already wrote:
 * no mask compaction, no data-dependent length, no host sync.  out12 (device, 12 doubles) = one (sum, count) pair
 * per term the reference averages separately:
 *   [0:4]  sum (rgb_coarse - target)^2, 3N | sum (rgb_fine - target)^2, 3N
 *   [4:8]  sum_masked disp_local_coarse, n_masked | ... fine        (disp_* = mf_render_args.disp_local/global planes)
 *   [8:12] sum_masked disp_global_coarse, n_masked | ... fine
 * A NULL array (or a NULL `fine`) leaves its pair at (0, 0).  loss terms = sum / count; the multi-GPU caller
 * all-reduces the 96 bytes first.  Deterministic (fixed-order partial sums through `scratch`). */
typedef struct mf_loss_pass {
  const float* rgb;          /* (N,3) rendered colour of the pass, or NULL            */
  const float* alphas;       /* (N,S) of the pass: the consensus mask source, or NULL */
  const float* disp_local;   /* (N,S) or NULL                                         */
  const float* disp_global;  /* (N,S) or NULL                                         */
  int32_t n_samples;         /* S of the pass                                         */
} mf_loss_pass;
int64_t mf_loss_partials_scratch_bytes(void);
/* means6 (device, 6 floats, optional; ABI v12): sum / count of each pair in fp32 -- what torch.mean of the corresponding
 * vector returns (nan for an empty one) -- so that a mean-only caller launches nothing else. */
int32_t mf_loss_partials(const mf_loss_pass* coarse, const mf_loss_pass* fine, const float* target, int64_t n_rays,
                         double* out12, float* means6, void* scratch, void* stream);

/* Backward of mf_loss_partials (ABI v12): the training-mode loss epilogue.  g12 (device, 12 doubles) = dL/d out12 (only the
 * six sums carry a gradient; out12 = the forward's result, its counts tell whether the consensus mask fell back to
 * all-true).  Writes the gradient seeds the backward nodes of the pass consume, each buffer whole:
 *   g_rgb (N,3)              = g12[2q] * 2 (rgb - target)                               models/losses.py:4-14
 *   g_recon_local  (N*S,3)   = g12[4+2q] * mask * (-sign(x - recon_local)) / 3           models/rendering.py:310-311
 *   g_recon_global (N*S,3)   = g12[8+2q] * mask * (-sign(x - recon_global)) / 3          models/rendering.py:313-314
 * x = o + d z (rays columns 0-5, z_vals (N,S)); recon_* = the output points of the chain's last forward-flow evaluation
 * (mf_render_args.dump_nof_out planes of steps 1 / 4).  Any output pointer may be NULL. */
typedef struct mf_loss_grad_pass {
  const float* rgb; float* g_rgb;
  const float* alphas; int32_t n_samples;
  const float* rays; int64_t ray_stride; const float* z_vals;
  const float* recon_local; float* g_recon_local;
  const float* recon_global; float* g_recon_global;
} mf_loss_grad_pass;
int32_t mf_loss_partials_backward(const mf_loss_grad_pass* coarse, const mf_loss_grad_pass* fine, const float* target,
                                  int64_t n_rays, const double* out12, const double* g12, void* stream);

/* The sample depths of a pass as render_rays materialises them when something outside the fused kernel needs them
 * (the resample, stratified jitter, the backward): z_out (N, S) = near*(1-t) + far*t, or 1/(1/near*(1-t) + 1/far*t)
 * with use_disp (models/rendering.py:245-251; near / far = columns 6 / 7 of the ray rows, t = z_steps (S) = linspace(0,1,S));
 * every product and sum separately rounded, bit-identical to the torch expression and to the fused pass's own z.
 * perturb_rand (N, S) uniform draws, or NULL (ABI v12): the stratified jitter of rendering.py:253-260 in the same launch,
 * z = lower + (upper - lower) * (perturb * rand) with lower / upper the neighbouring mid-points (the ray's ends at the
 * ends), again bit-identical to the torch expression. */
int32_t mf_z_vals(const float* rays, int64_t ray_stride, int64_t n_rays, const float* z_steps, int32_t n_samples,
                  int32_t use_disp, const float* perturb_rand, float perturb, float* z_out, void* stream);

/* ---- producers either side of the path -----------------------------
 * Camera.make_rays (utils/camera.py:134-148 with gen_ray_directions :29-50 and gen_rays :52-81):
 * rays_out (H*W, 9) = [o(3), unit d(3), near, far, idx], pixel order row-major; the direction is
 * ((i-cx)/focal, -(j-cy)/focal, -1) rotated by c2w[:, :3] (host pointer to the 3x4 row-major matrix, or
 * NULL for camera coordinates).  near/far/idx are the scalars make_rays broadcasts. */
int32_t mf_make_rays(int32_t H, int32_t W, float focal, float cx, float cy, const float* c2w_host,
                     float nearv, float farv, float idx, float* rays_out, void* stream);
/* Camera.get_valid_rays_mask (utils/camera.py:119-132): mask_out (H*W bytes, device, row-major) = 1 inside the filled
 * convex hull of the projected AABB corners pts_xy (host, n_pts x (x = column, y = row) int32: the output of
 * calculate_2d_projections, camera.py:83-103), else 0.  Hull on the host, fill on the device, one thread per pixel.
 * Rule = cv2.fillConvexPoly's scan-line fill (line_type 8) with exact intersections: row y in [ymin, ymax] sets
 * pixels round_half_up(XL(y)) .. round_half_up(XR(y)).  PARITY UNPINNED vs cv2 itself (absent from the image). */
int32_t mf_valid_rays_mask(int32_t H, int32_t W, const int32_t* pts_xy_host, int32_t n_pts, uint8_t* mask_out,
                           void* stream);
/* Foreground scatter-back of the full-image drivers (MoCoFlowTrainer.render trainer_moco_flow.py:249-266,
 * NeRFTrainer.render trainer_nerf.py:128-140), on the device: for every pixel b of the (B) image
 *   not rendered (rays_msk[b] == 0)            -> img = background[b], depth = 10
 *   rendered, opacity[rank[b]] > 0             -> img = rgb[rank[b]],  depth = depth[rank[b]]
 *   rendered, opacity == 0                     -> img = background[b], depth = 8
 * rays_msk (B) bytes or NULL (all rendered, rank may be NULL); rank (B) int64 = position of pixel b among
 * the rendered rays. */
int32_t mf_image_compose(const uint8_t* rays_msk, const int64_t* rank, int64_t B, const float* opacity,
                         const float* rgb, const float* depth, const float* background, float* img,
                         float* depth_out, void* stream);

/* knn_cuda.KNN(k=1, transpose_mode=True) of the vendored wheel (datasets/moco_flow_dataset.py:35,120):
 * ref (V,3), query (Q,3) -> dist (Q,) Euclidean distance to, and ind (Q,) int64 0-based index of, the
 * nearest reference point; first minimum on ties. */
int32_t mf_knn1(const float* ref, int64_t V, const float* query, int64_t Q, float* dist, int64_t* ind,
                void* stream);

/* ---- SMPL linear blend skinning + per-vertex frame transforms (ABI v10) ------------
 * The model arrays of utils/smpl/smpl_model.py:60-82 as device pointers (fp32, row-major):
 *   v_template (V,3), shapedirs (V*3,10) [= shapedirs[:, :, :10]], posedirs (V*3,207), j_regressor (24,V) dense,
 *   weights (V,24); parent[i] (i = 1..23) = index of joint i's parent (an earlier joint), parent[0] ignored. */
typedef struct mf_smpl_model {
  int32_t n_verts;
  const float* v_template;
  const float* shapedirs;
  const float* posedirs;
  const float* j_regressor;
  const float* weights;
  int32_t parent[24];
} mf_smpl_model;

/* SMPL.forward (smpl_model.py:96-139) and SMPL.get_vertex_transformation (:141-186) in one call:
 * pose (B,72) axis-angle, or (B,24,3,3) rotation matrices when pose_is_rotmat; betas (B,10) ->
 * verts (B,V,3) and / or T (B,V,4,4) (either may be NULL).  scratch: mf_smpl_scratch_bytes(V,B) bytes. */
int64_t mf_smpl_scratch_bytes(int64_t n_verts, int64_t B);
int32_t mf_smpl_lbs(const mf_smpl_model* m, const float* pose, int32_t pose_is_rotmat, const float* betas, int64_t B,
                    float* verts, float* T, void* scratch, void* stream);

/* datasets/moco_flow_dataset.py:96-99: trans (V,4,4) = T_tgt @ inverse(T_src), per vertex. */
int32_t mf_smpl_frame_transforms(const float* T_src, const float* T_tgt, int64_t V, float* trans, void* stream);

/* datasets/moco_flow_dataset.py:127-129: cano (Q,3) = (trans[ind] @ [query, 1])[:3]; ind (Q,) int64 into the V
 * transforms (mf_knn1's output). */
int32_t mf_apply_vertex_transforms(const float* trans, const int64_t* ind, int64_t V, const float* query, int64_t Q,
                                   float* cano, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOCOFLOW_HIP_H */
