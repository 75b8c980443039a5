"""Race screen of wg_segment_x3 (three fragment buffers, staggered barrier): run with a library built by
`tools/ab_lib.sh build mf_wgrad.hip "jit=-DMF_DBG_JITTER"` (MOCOFLOW_HIP_LIB=build/ab/lib_jit.so): every block shape, ragged sample
counts, twelve repeats each -- bit-identical results under random per-wave stalls in front of both barrier sites."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import moco_flow_amd as M
from moco_flow_amd import autograd as A
dev = torch.device("cuda")
torch.manual_seed(1)
worst = 0.0
for P in (37, 5000, 70001, 300000):
    W, stride = 256, 9 * 256 + 128
    acts, gpre = torch.randn(P, stride, device=dev), torch.randn(P, stride, device=dev)
    emb, ext, ghead = torch.randn(P, 64, device=dev), torch.randn(P, 32, device=dev), torch.randn(P, 4, device=dev)
    ns = 4 * 128 + 16
    nacts, ngpre, emb80 = torch.randn(P, ns, device=dev), torch.randn(P, ns, device=dev), torch.randn(P, 80, device=dev)
    sl = lambda t, l, w=W: t[:, l * W:l * W + w]
    nsl = lambda t, l, w=128: t[:, l * 128:l * 128 + w]
    jobs = [(sl(gpre, 1), sl(acts, 0), 256, 256, True), (sl(gpre, 5), sl(acts, 4), 256, 256, False), (sl(gpre, 9, 128), sl(acts, 8), 128, 256, True),
            (sl(gpre, 0), emb, 256, 64, True), (sl(gpre, 9, 128), ext, 128, 32, False),
            (nsl(ngpre, 1), nsl(nacts, 0), 128, 128, True), (nsl(ngpre, 0), emb80, 128, 80, True), (ngpre[:, 512:524], nsl(nacts, 3), 12, 128, True)]
    ref = A.weight_grads(jobs, P, dev)
    for rep in range(12):
        res = A.weight_grads(jobs, P, dev)
        for (a, ab), (b, bb) in zip(ref, res):
            assert torch.equal(a, b), (P, rep)
            if ab is not None: assert torch.equal(ab, bb), (P, rep)
    for (G, X, no, ni, b), (dW, db) in zip(jobs, ref):
        want = G.double().t() @ X.double()
        worst = max(worst, float((dW[:no].double() - want).norm() / want.norm()))
print("jitter screen: 12 repeats x 4 sample counts bit-identical; worst l2-rel vs float64", worst)
