"""k = 1 nearest reference point: stands in for knn_cuda.KNN(k=1,
transpose_mode=True), the vendored CUDA wheel the datasets use for NoF supervision
(datasets/moco_flow_dataset.py:35,120; datasets/nof_dataset.py:27,79)."""
import torch
from torch import nn

from . import _lib as L


class KNN(nn.Module):
    """forward(ref (B,V,3), query (B,Q,3)) -> (dist (B,Q,1) float32, ind (B,Q,1) int64), like the wheel's
    class with transpose_mode=True. Only k = 1 (the only value the reference uses) is built."""

    def __init__(self, k, transpose_mode=False):
        super().__init__()
        if k != 1 or not transpose_mode:
            raise NotImplementedError("moco_flow_amd.knn.KNN: only k=1, transpose_mode=True is built")
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        L.require_gpu(ref, "KNN.forward")
        with torch.no_grad():
            D, I = [], []
            for bi in range(ref.size(0)):
                r = ref[bi].float().contiguous()
                q = query[bi].float().contiguous()
                d = torch.empty((q.shape[0], 1), device=q.device, dtype=torch.float32)
                i = torch.empty((q.shape[0], 1), device=q.device, dtype=torch.int64)
                with torch.cuda.device(q.device):
                    L.check(L.lib().mf_knn1(L.ptr(r), r.shape[0], L.ptr(q), q.shape[0], L.ptr(d), L.ptr(i),
                                            L.current_stream(q.device)), "mf_knn1")
                D.append(d)
                I.append(i)
            return torch.stack(D, dim=0), torch.stack(I, dim=0)
