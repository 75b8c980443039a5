"""ctypes binding of the C-ABI library (include/mocoflow_hip.h).

The product path has NO fallback: if ``libmocoflow_hip.so`` is missing or a call
fails, a RuntimeError is raised (build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C moco_flow_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

MF_MAX_FREQS = 16
MF_MAX_LAYERS = 16
MF_EXTRA_NONE, MF_EXTRA_IND, MF_EXTRA_DIR = 0, 1, 2
MF_ACT_RELU, MF_ACT_SOFTPLUS = 0, 1
MF_F_SIGMA_ONLY, MF_F_CHAIN_LOCAL, MF_F_CHAIN_GLOBAL = 1, 2, 4
MF_PREC_F32, MF_PREC_BF16, MF_PREC_BF16X3 = 0, 1, 2
PRECISIONS = {"f32": MF_PREC_F32, "bf16": MF_PREC_BF16, "bf16x3": MF_PREC_BF16X3}
MF_ABI_VERSION = 16

LIB_PATH = os.environ.get("MOCOFLOW_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmocoflow_hip.so")   # (override: A/B builds)

_fp = C.c_void_p   # device pointers travel as integers


class mf_embedding(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("n_freqs", C.c_int32),
                ("freq", C.c_float * MF_MAX_FREQS), ("weight", C.c_float * MF_MAX_FREQS)]


class mf_smpl_model(C.Structure):
    _fields_ = [("n_verts", C.c_int32), ("v_template", _fp), ("shapedirs", _fp), ("posedirs", _fp),
                ("j_regressor", _fp), ("weights", _fp), ("parent", C.c_int32 * 24)]


class mf_nerf_desc(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("in_channels_xyz", C.c_int32),
                ("skip_mask", C.c_uint32), ("extra_feat_type", C.c_int32), ("extra_feat_dim", C.c_int32),
                ("trunk_w", _fp * MF_MAX_LAYERS), ("trunk_b", _fp * MF_MAX_LAYERS),
                ("final_w", _fp), ("final_b", _fp), ("extra_w", _fp), ("extra_b", _fp),
                ("sigma_w", _fp), ("sigma_b", _fp), ("rgb_w", _fp), ("rgb_b", _fp)]


class mf_nof_desc(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("in_channels_xyz", C.c_int32),
                ("extra_feat_dim", C.c_int32), ("skip_mask", C.c_uint32), ("use_quat", C.c_int32),
                ("trunk_w", _fp * MF_MAX_LAYERS), ("trunk_b", _fp * MF_MAX_LAYERS),
                ("head_w", _fp), ("head_b", _fp)]


class mf_wgrad_item(C.Structure):
    _fields_ = [("G", _fp), ("g_stride", C.c_int64), ("n_out", C.c_int32),
                ("X", _fp), ("x_stride", C.c_int64), ("n_in", C.c_int32),
                ("dW", _fp), ("db", _fp)]


MF_WG_MAX_ITEMS = 32


class mf_loss_pass(C.Structure):
    _fields_ = [("rgb", _fp), ("alphas", _fp), ("disp_local", _fp), ("disp_global", _fp), ("n_samples", C.c_int32)]


class mf_loss_grad_pass(C.Structure):
    _fields_ = [("rgb", _fp), ("g_rgb", _fp), ("alphas", _fp), ("n_samples", C.c_int32),
                ("rays", _fp), ("ray_stride", C.c_int64), ("z_vals", _fp),
                ("recon_local", _fp), ("g_recon_local", _fp), ("recon_global", _fp), ("g_recon_global", _fp)]


class mf_render_args(C.Structure):
    _fields_ = [("rays", _fp), ("ray_stride", C.c_int64), ("n_rays", C.c_int64),
                ("background", _fp), ("n_samples", C.c_int32),
                ("z_vals", _fp), ("z_steps", _fp), ("use_disp", C.c_int32),
                ("noise", _fp), ("activation", C.c_int32), ("flags", C.c_int32),
                ("nerf", C.POINTER(mf_nerf_desc)), ("nerf_packed", _fp),
                ("emb_xyz", mf_embedding), ("emb_extra", mf_embedding),
                ("nof_bw", C.POINTER(mf_nof_desc)), ("nof_bw_packed", _fp),
                ("nof_fw", C.POINTER(mf_nof_desc)), ("nof_fw_packed", _fp),
                ("nof_emb_xyz", mf_embedding), ("nof_emb_ind", mf_embedding),
                ("rgb", _fp), ("depth", _fp), ("opacity", _fp), ("weights", _fp), ("alphas", _fp),
                ("disp_local", _fp), ("disp_global", _fp), ("precision", C.c_int32),
                ("dump_acts", _fp), ("dump_stride", C.c_int64), ("dump_rgbsigma", _fp), ("dump_xyz", _fp),
                ("dump_nof_acts", _fp), ("dump_nof_stride", C.c_int64), ("dump_nof_emb", _fp), ("dump_nof_out", _fp),
                ("dump_nof_plane", C.c_int32 * 5),
                ("workspace", _fp), ("workspace_bytes", C.c_int64),
                ("dump_mask", _fp), ("dump_mask_stride", C.c_int64)]


# every symbol include/mocoflow_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "mf_version": (C.c_int32, []),
    "mf_last_error": (C.c_char_p, []),
    "mf_nerf_packed_bytes": (C.c_int64, [C.POINTER(mf_nerf_desc)]),
    "mf_nof_packed_bytes": (C.c_int64, [C.POINTER(mf_nof_desc)]),
    "mf_nerf_pack": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, _fp]),
    "mf_nof_pack": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, _fp]),
    "mf_nerf_packed_bytes_p": (C.c_int64, [C.POINTER(mf_nerf_desc), C.c_int32]),
    "mf_nof_packed_bytes_p": (C.c_int64, [C.POINTER(mf_nof_desc), C.c_int32]),
    "mf_nerf_pack_p": (C.c_int32, [C.POINTER(mf_nerf_desc), C.c_int32, _fp, _fp]),
    "mf_nof_pack_p": (C.c_int32, [C.POINTER(mf_nof_desc), C.c_int32, _fp, _fp]),
    "mf_embedding_forward": (C.c_int32, [C.POINTER(mf_embedding), _fp, C.c_int64, _fp, _fp]),
    "mf_embedding_forward_rows": (C.c_int32, [C.POINTER(mf_embedding), _fp, C.c_int64, C.c_int32, _fp, C.c_int64, _fp]),
    "mf_nerf_forward": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, _fp, C.c_int64, C.c_int64, C.c_int32, _fp, _fp]),
    "mf_nerf_forward_dump": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, _fp, C.c_int64, C.c_int64, _fp, _fp, C.c_int64, _fp]),
    "mf_nof_forward": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, _fp, C.c_int64, _fp, C.c_int64, _fp, _fp]),
    "mf_nerf_bwd_packed_bytes": (C.c_int64, [C.POINTER(mf_nerf_desc)]),
    "mf_nerf_pack_bwd": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, _fp]),
    "mf_nof_points_dump": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, C.POINTER(mf_embedding), C.POINTER(mf_embedding), _fp, _fp,
                                       C.c_int64, C.c_int32, C.c_int64, _fp, _fp, C.c_int64, _fp, _fp]),
    "mf_nof_forward_dump": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, _fp, C.c_int64, _fp, C.c_int64, _fp, _fp, C.c_int64, _fp]),
    "mf_nof_bwd_packed_bytes": (C.c_int64, [C.POINTER(mf_nof_desc)]),
    "mf_nof_pack_bwd": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, _fp]),
    "mf_nof_backward": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, C.POINTER(mf_embedding), C.c_int64, _fp, _fp, C.c_int64,
                                    _fp, _fp, _fp, _fp]),
    "mf_nof_backward3": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, C.POINTER(mf_embedding), C.c_int64, _fp, _fp, C.c_int64,
                                    _fp, _fp, _fp, _fp]),
    "mf_nof_bwd3_packed_bytes": (C.c_int64, [C.POINTER(mf_nof_desc)]),
    "mf_nof_pack_bwd3": (C.c_int32, [C.POINTER(mf_nof_desc), _fp, _fp]),
    "mf_composite_backward": (C.c_int32, [_fp, C.c_int64, C.c_int64, C.c_int32, _fp, _fp, _fp, C.c_int32, _fp, _fp, _fp, _fp,
                                          _fp, _fp]),
    "mf_image_compose": (C.c_int32, [_fp, _fp, C.c_int64, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "mf_weight_grads_scratch_bytes": (C.c_int64, [C.POINTER(mf_wgrad_item), C.c_int32, C.c_int64]),
    "mf_weight_grads": (C.c_int32, [C.POINTER(mf_wgrad_item), C.c_int32, C.c_int64, _fp, _fp]),
    "mf_weight_grads_scratch_bytes_p": (C.c_int64, [C.c_int32, C.POINTER(mf_wgrad_item), C.c_int32, C.c_int64]),
    "mf_weight_grads_p": (C.c_int32, [C.c_int32, C.POINTER(mf_wgrad_item), C.c_int32, C.c_int64, _fp, _fp]),
    "mf_nerf_bwd3_packed_bytes": (C.c_int64, [C.POINTER(mf_nerf_desc)]),
    "mf_nerf_pack_bwd3": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, _fp]),
    "mf_nerf_backward3": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, C.c_int64, _fp, _fp, C.c_int64, _fp, _fp, _fp, _fp, _fp, C.c_int64, _fp]),
    "mf_nerf_backward": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, C.c_int64, _fp, _fp, C.c_int64, _fp, _fp, _fp, _fp]),
    "mf_render_pass": (C.c_int32, [C.POINTER(mf_render_args), _fp]),
    "mf_render_workspace_bytes": (C.c_int64, [C.POINTER(mf_render_args)]),
    "mf_render_prepare": (C.c_int32, [C.POINTER(mf_render_args), _fp]),
    "mf_points_sigma_workspace_bytes": (C.c_int64, [C.c_int32, C.POINTER(mf_nof_desc), C.c_int32, C.c_int64]),
    "mf_points_sigma": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, C.POINTER(mf_embedding), C.POINTER(mf_nof_desc), _fp,
                                    C.POINTER(mf_embedding), C.POINTER(mf_embedding), _fp, _fp, C.c_float, C.c_int64,
                                    _fp, _fp, _fp]),
    "mf_points_sigma_p": (C.c_int32, [C.c_int32, C.POINTER(mf_nerf_desc), _fp, C.POINTER(mf_embedding), C.POINTER(mf_nof_desc), _fp,
                                      C.POINTER(mf_embedding), C.POINTER(mf_embedding), _fp, _fp, C.c_float, C.c_int64,
                                      _fp, _fp, _fp, C.c_int64, _fp]),
    "mf_sample_pdf_merge": (C.c_int32, [_fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp]),
    "mf_sample_pdf": (C.c_int32, [_fp, _fp, _fp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_int64,
                                  _fp, _fp, _fp, _fp, _fp]),
    "mf_sample_pdf_eps": (C.c_int32, [_fp, _fp, _fp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_int64,
                                      _fp, _fp, _fp, _fp, C.c_float, _fp]),
    "mf_z_vals": (C.c_int32, [_fp, C.c_int64, C.c_int64, _fp, C.c_int32, C.c_int32, _fp, C.c_float, _fp, _fp]),
    "mf_make_rays": (C.c_int32, [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float),
                                 C.c_float, C.c_float, C.c_float, _fp, _fp]),
    "mf_knn1": (C.c_int32, [_fp, C.c_int64, _fp, C.c_int64, _fp, _fp, _fp]),
    "mf_nof_emb_slot_features": (C.c_int32, [C.POINTER(C.c_int32)]),
    "mf_nof_embed_rows": (C.c_int32, [C.POINTER(mf_embedding), _fp, _fp, C.c_int32, C.c_int32, C.c_int64, _fp, _fp]),
    "mf_smpl_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "mf_smpl_lbs": (C.c_int32, [C.POINTER(mf_smpl_model), _fp, C.c_int32, _fp, C.c_int64, _fp, _fp, _fp, _fp]),
    "mf_smpl_frame_transforms": (C.c_int32, [_fp, _fp, C.c_int64, _fp, _fp]),
    "mf_apply_vertex_transforms": (C.c_int32, [_fp, _fp, C.c_int64, _fp, C.c_int64, _fp, _fp]),
    "mf_nerf_backward_x": (C.c_int32, [C.POINTER(mf_nerf_desc), _fp, C.c_int64, _fp, _fp, C.c_int64, _fp, _fp, _fp, _fp, _fp]),
    "mf_embedding_backward": (C.c_int32, [C.POINTER(mf_embedding), _fp, C.c_int64, _fp, C.c_int64, C.c_int64, _fp, _fp]),
    "mf_valid_rays_mask": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _fp, _fp]),
    "mf_loss_partials_scratch_bytes": (C.c_int64, []),
    "mf_loss_partials": (C.c_int32, [C.POINTER(mf_loss_pass), C.POINTER(mf_loss_pass), _fp, C.c_int64, _fp, _fp, _fp, _fp]),
    "mf_loss_partials_backward": (C.c_int32, [C.POINTER(mf_loss_grad_pass), C.POINTER(mf_loss_grad_pass), _fp, C.c_int64, _fp, _fp, _fp]),
    "mf_compact_scratch_bytes": (C.c_int64, [C.c_int64]),
    "mf_compact_mask": (C.c_int32, [_fp, _fp, _fp, C.c_int64, C.c_int32, _fp, _fp, _fp, _fp, _fp]),
}

_lock = threading.Lock()
_lib = None


def lib():
    """The loaded C-ABI library; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"moco_flow_amd: HIP library not built ({LIB_PATH} missing). "
                    "Run `make -C moco_flow_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(handle, name)   # AttributeError if the export is missing
                fn.restype = res
                fn.argtypes = args
            if handle.mf_version() != MF_ABI_VERSION:
                raise RuntimeError("moco_flow_amd: ABI version mismatch")
            _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().mf_last_error().decode(errors="replace")
        if rc == -3:
            raise NotImplementedError(f"moco_flow_amd HIP path: {msg}")
        raise RuntimeError(f"moco_flow_amd HIP call failed ({what}, rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (or NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"moco_flow_amd.{what}: tensor is on '{t.device}'. This package is the MI355X (HIP) path and has "
            "no CPU implementation; move the module and its inputs to a 'cuda' device.")
