"""SMPL linear blend skinning and the NoF-supervision transforms: stands in for
utils/smpl/smpl_model.py (``SMPL.forward`` :96-139, ``get_vertex_transformation`` :141-186) and for the
per-vertex part of ``get_frame_correspondence`` (datasets/moco_flow_dataset.py:87-142, datasets/nof_dataset.py).

The licensed SMPL pickle is not part of either repository: ``SMPL(model=...)`` takes the model arrays (a dict, an
``.npz`` or the original pickle's path) -- `gender=` resolves the reference's file name next to ``data_dir``."""
import os

import numpy as np
import torch
from torch import nn

from . import _lib as L


def _load_model(model):
    if isinstance(model, dict):
        return model
    if str(model).endswith(".npz"):
        return dict(np.load(model))
    import pickle
    with open(model, "rb") as f:                      # smpl_model.py:63-64 (needs chumpy for the original file)
        return pickle.load(f, encoding="iso-8859-1")


class SMPL(nn.Module):
    """Same buffers, attributes and call contract as the reference's class; the arithmetic is mf_smpl_lbs."""

    def __init__(self, gender="neutral", model=None, data_dir=None):
        super().__init__()
        self.gender = gender
        if model is None:
            base = data_dir or os.path.join(os.path.dirname(__file__), "data")
            model = os.path.join(base, "basicmodel_%s_lbs_10_207_0_v1.1.0.pkl" % gender)        # smpl_model.py:61
            if not os.path.exists(model):
                raise FileNotFoundError(f"SMPL model file {model} not found (licensed asset: pass model=<dict | .npz | .pkl>)")
        m = _load_model(model)
        f32 = lambda a: torch.as_tensor(np.array(a), dtype=torch.float32)
        self.vert_num = int(np.array(m["v_template"]).shape[0])
        jr = m["J_regressor"]
        jr = jr.toarray() if hasattr(jr, "toarray") else np.array(jr)                        # scipy sparse in the pickle, :68-77
        self.register_buffer("J_regressor", f32(jr))
        self.register_buffer("weights", f32(m["weights"]))
        self.register_buffer("posedirs", f32(m["posedirs"]))
        self.register_buffer("v_template", f32(m["v_template"]))
        self.register_buffer("shapedirs", f32(m["shapedirs"]))
        if "f" in m:
            self.register_buffer("faces", torch.as_tensor(np.array(m["f"]).astype(np.int64)))
        if "parent" in m:
            parent = [int(p) for p in np.array(m["parent"]).tolist()]
            if len(parent) == 24:        # some exports keep the root's (meaningless) entry in front: joints 1..23 have parents
                parent = parent[1:]
            if len(parent) != 23:
                raise ValueError(f"SMPL model: 'parent' must list the parents of joints 1..23 (23 entries), got {len(parent)}")
        else:                                                                                # :83-85
            kt = np.array(m["kintree_table"]).astype(np.int64)
            id_to_col = {int(kt[1, i]): i for i in range(kt.shape[1])}
            parent = [id_to_col[int(kt[0, it])] for it in range(1, kt.shape[1])]
        self.register_buffer("parent", torch.as_tensor(parent, dtype=torch.int64))
        self.pose_shape, self.beta_shape, self.translation_shape = [24, 3], [10], [3]
        self._packed = None

    def _model(self, dev):
        """(descriptor, tensors kept alive): contiguous device copies in the layout the kernels read."""
        key = (str(dev), self.shapedirs.data_ptr(), self.posedirs.data_ptr())
        if self._packed is None or self._packed[0] != key:
            V = self.vert_num
            keep = dict(vt=self.v_template.to(dev).contiguous(),
                        sd=self.shapedirs[:, :, :10].to(dev).reshape(V * 3, 10).contiguous(),       # :99
                        pd=self.posedirs.to(dev).reshape(V * 3, 207).contiguous(),                  # :119
                        jr=self.J_regressor.to(dev).contiguous(), w=self.weights.to(dev).contiguous())
            d = L.mf_smpl_model()
            d.n_verts = V
            d.v_template, d.shapedirs, d.posedirs = keep["vt"].data_ptr(), keep["sd"].data_ptr(), keep["pd"].data_ptr()
            d.j_regressor, d.weights = keep["jr"].data_ptr(), keep["w"].data_ptr()
            d.parent[0] = 0
            for i, p in enumerate(self.parent.tolist()):
                d.parent[i + 1] = int(p)
            self._packed = (key, d, keep)
        return self._packed[1]

    def _lbs(self, pose, beta, want_verts, want_T):
        L.require_gpu(pose, "SMPL")
        dev = pose.device
        B = pose.shape[0]
        if pose.dim() == 4:
            is_rot, p = 1, pose.detach().float().reshape(B, 24, 9).contiguous()                 # :110-111
        elif pose.dim() == 2:
            is_rot, p = 0, pose.detach().float().reshape(B, 72).contiguous()                    # :113-116
        else:
            raise RuntimeError(f"SMPL: pose must be (B,72) or (B,24,3,3), got {tuple(pose.shape)}")
        b = beta.detach().float().reshape(B, beta.shape[-1] if beta.dim() > 1 else 10)[:, :10].contiguous()
        if b.shape[1] != 10:
            raise RuntimeError(f"SMPL: betas must be (B,10), got {tuple(beta.shape)}")
        V = self.vert_num
        desc = self._model(dev)
        verts = torch.empty((B, V, 3), device=dev, dtype=torch.float32) if want_verts else None
        T = torch.empty((B, V, 4, 4), device=dev, dtype=torch.float32) if want_T else None
        lib = L.lib()
        scratch = torch.empty((max(int(lib.mf_smpl_scratch_bytes(V, B)), 4),), device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            L.check(lib.mf_smpl_lbs(desc, L.ptr(p), is_rot, L.ptr(b), B, L.ptr(verts), L.ptr(T), L.ptr(scratch),
                                    L.current_stream(dev)), "mf_smpl_lbs")
        return verts, T

    def forward(self, pose, beta):
        """smpl_model.py:96-139: posed vertices (B,V,3)."""
        return self._lbs(pose, beta, True, False)[0]

    def get_vertex_transformation(self, pose, beta):
        """smpl_model.py:141-186: blended per-vertex transforms (B,V,4,4)."""
        return self._lbs(pose, beta, False, True)[1]

    def get_smpl_joints(self, vertices):
        """smpl_model.py:188-197."""
        return torch.einsum('bik,ji->bjk', [vertices, self.J_regressor.to(vertices.device)])


def frame_transforms(T_src, T_tgt):
    """datasets/moco_flow_dataset.py:96-99: per-vertex source pose -> t-pose -> target pose, (V,4,4)."""
    L.require_gpu(T_src, "frame_transforms")
    a, b = T_src.detach().float().contiguous(), T_tgt.detach().float().contiguous()
    if a.shape != b.shape or a.dim() != 3 or a.shape[1:] != (4, 4):
        raise RuntimeError(f"frame_transforms expects two (V,4,4) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        L.check(L.lib().mf_smpl_frame_transforms(L.ptr(a), L.ptr(b), a.shape[0], L.ptr(out), L.current_stream(a.device)),
                "mf_smpl_frame_transforms")
    return out


def apply_vertex_transforms(trans, ind, query):
    """datasets/moco_flow_dataset.py:127-129: query (Q,3), ind (Q,) | (Q,1) int64 -> canonical points (Q,3)."""
    L.require_gpu(query, "apply_vertex_transforms")
    t = trans.detach().float().contiguous()
    q = query.detach().float().contiguous()
    i = ind.detach().reshape(-1).to(torch.int64).contiguous()
    if i.shape[0] != q.shape[0]:
        raise RuntimeError(f"apply_vertex_transforms: {i.shape[0]} indices for {q.shape[0]} points")
    out = torch.empty_like(q)
    with torch.cuda.device(q.device):
        L.check(L.lib().mf_apply_vertex_transforms(L.ptr(t), L.ptr(i), t.shape[0], L.ptr(q), q.shape[0], L.ptr(out),
                                                   L.current_stream(q.device)), "mf_apply_vertex_transforms")
    return out


def frame_correspondence(smpl, src_pose, src_betas, tgt_pose, tgt_betas, query_xyzs, thickness=0.2, knn=None):
    """The device part of ``get_frame_correspondence`` (moco_flow_dataset.py:87-142) for given query points (the
    reference samples them with trimesh + randn, :101-110): nearest source-pose vertex -> that vertex's
    source -> target transform -> (inside_xyzs, outside_xyzs), rows [query | canonical], split at dist < thickness."""
    from .knn import KNN
    trans = frame_transforms(smpl.get_vertex_transformation(src_pose, src_betas)[0],
                             smpl.get_vertex_transformation(tgt_pose, tgt_betas)[0])
    src_verts = smpl.forward(src_pose, src_betas)[0]
    knn = knn or KNN(k=1, transpose_mode=True)
    dist, ind = knn(src_verts.unsqueeze(0), query_xyzs.unsqueeze(0))                         # :120-121
    dist, ind = dist[0], ind[0]
    cano = apply_vertex_transforms(trans, ind, query_xyzs)
    inside = dist.flatten() < thickness                                                       # :123-125
    both = torch.cat([query_xyzs.view(-1, 3), cano.view(-1, 3)], dim=-1)
    return both[inside], both[~inside]
