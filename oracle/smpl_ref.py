"""TEST INFRASTRUCTURE -- CPU restatement of the reference's SMPL linear blend skinning and of the per-vertex
transforms the datasets use for NoF supervision.  Only tests/, smoke() and bench.py's
cpu_baseline may import this; the product (moco_flow_amd/) never does.

Follows /root/reference/utils/smpl/smpl_model.py and datasets/moco_flow_dataset.py; pinned by
tests/golden/u_smpl.npz, which tests/golden/gen_golden.py produced by running the REFERENCE's own `SMPL.forward` /
`get_vertex_transformation` / correspondence lines on the synthetic assets of moco_flow_amd.synth.smpl_model (the
licensed SMPL pickle is absent; the array shapes and the arithmetic are the reference's)."""
import torch


def quat2mat(quat):
    """smpl_model.py:17-37: (B,4) (w,x,y,z) -> (B,3,3), after normalisation."""
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def rodrigues(theta):
    """smpl_model.py:40-55: axis-angle (B,3) -> (B,3,3) through the half-angle quaternion; the norm is taken of
    theta + 1e-8, the division uses theta itself."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1).unsqueeze(-1)
    normalized = theta / angle
    angle = angle * 0.5
    return quat2mat(torch.cat([torch.cos(angle), torch.sin(angle) * normalized], dim=1))


class SMPL:
    """smpl_model.py:58-186 on explicit model arrays (dict of tensors: v_template, shapedirs, posedirs, J_regressor,
    weights, parent)."""

    def __init__(self, model):
        t = lambda a: torch.as_tensor(a)
        self.v_template = t(model["v_template"]).float()
        self.shapedirs = t(model["shapedirs"]).float()
        self.posedirs = t(model["posedirs"]).float()
        self.J_regressor = t(model["J_regressor"]).float()
        self.weights = t(model["weights"]).float()
        self.parent = [int(p) for p in t(model["parent"]).tolist()]

    def _skeleton(self, pose, beta):
        """smpl_model.py:97-134 / 142-181 (the two methods share these lines): -> (v_posed (B,V,3), G (B,24,4,4))."""
        B = pose.shape[0]
        v_shaped = torch.matmul(self.shapedirs[:, :, :10].reshape(-1, 10)[None].expand(B, -1, -1),
                                beta[:, :, None]).view(B, -1, 3) + self.v_template[None]           # :100-103
        J = torch.stack([self.J_regressor @ v_shaped[i] for i in range(B)], dim=0)                # :105-108
        if pose.dim() == 4:
            R = pose                                                                              # :110-111
        else:
            R = rodrigues(pose.reshape(-1, 3)).view(B, 24, 3, 3)                                  # :113-116
        lrotmin = (R[:, 1:] - torch.eye(3)[None, None]).reshape(B, -1)                            # :117-119
        v_posed = v_shaped + torch.matmul(self.posedirs.reshape(-1, 207)[None].expand(B, -1, -1),
                                          lrotmin[:, :, None]).view(B, -1, 3)                     # :120-121
        J_ = J.clone()
        J_[:, 1:] = J[:, 1:] - J[:, self.parent]                                                  # :122-123
        G_ = torch.cat([R, J_[:, :, :, None]], dim=-1)
        pad = torch.tensor([0., 0., 0., 1.]).view(1, 1, 1, 4).expand(B, 24, -1, -1)
        G_ = torch.cat([G_, pad], dim=2)                                                          # :124-126
        G = [G_[:, 0].clone()]
        for i in range(1, 24):
            G.append(G[self.parent[i - 1]] @ G_[:, i])                                            # :127-129
        G = torch.stack(G, dim=1)
        rest = torch.cat([J, torch.zeros(B, 24, 1)], dim=2).view(B, 24, 4, 1)
        rest = torch.cat([torch.zeros(B, 24, 4, 3), rest], dim=-1)
        return v_posed, G - G @ rest                                                              # :131-135

    def get_vertex_transformation(self, pose, beta):
        """smpl_model.py:141-186 -> T (B,V,4,4)."""
        B = pose.shape[0]
        _, G = self._skeleton(pose, beta)
        return torch.matmul(self.weights, G.permute(1, 0, 2, 3).contiguous().view(24, -1)).view(-1, B, 4, 4).transpose(0, 1)

    def forward(self, pose, beta):
        """smpl_model.py:96-139 -> vertices (B,V,3)."""
        B = pose.shape[0]
        v_posed, G = self._skeleton(pose, beta)
        T = torch.matmul(self.weights, G.permute(1, 0, 2, 3).contiguous().view(24, -1)).view(-1, B, 4, 4).transpose(0, 1)
        h = torch.cat([v_posed, torch.ones_like(v_posed)[:, :, [0]]], dim=-1)
        return torch.matmul(T, h[:, :, :, None])[:, :, :3, 0]

    __call__ = forward


def frame_transforms(T_src, T_tgt):
    """moco_flow_dataset.py:96-99: source pose -> t-pose -> target pose, per vertex: T_tgt @ inverse(T_src)."""
    return T_tgt @ torch.inverse(T_src)


def apply_vertex_transforms(trans, ind, query):
    """moco_flow_dataset.py:127-129: cano = (trans[ind] @ [query, 1])[:3]; ind (Q,) int64 nearest-vertex indices."""
    h = torch.cat([query, torch.ones((query.shape[0], 1))], dim=-1)
    return (trans[ind] @ h.unsqueeze(-1))[:, :3, 0]


def split_inside_outside(query, cano, dist, thickness):
    """moco_flow_dataset.py:122-132: rows [query | cano] with dist < thickness / the rest, original order."""
    inside = dist.flatten() < thickness
    both = torch.cat([query, cano], dim=-1)
    return both[inside], both[~inside]
