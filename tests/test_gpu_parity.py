"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP path, reached through the
C ABI, against (a) the golden vectors produced by the reference itself and (b) the CPU oracle
on the same seeded inputs. Tolerance: 1e-4 max-rel for fp32 values (north_star), bit-exact for
index bookkeeping (searchsorted indices from a given cdf, mask-compaction order, ray order)."""
import numpy as np
import os

import pytest
import torch

from cases import RENDER_CASES
from helpers import OracleOps, build_case, case_inputs, load_golden, pad_to, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def M():
    import moco_flow_amd
    assert torch.cuda.is_available()
    moco_flow_amd._lib.lib()          # fail loudly if the HIP library is missing
    return moco_flow_amd


@pytest.fixture(scope="module")
def R():
    from oracle import cpu_ref
    return cpu_ref


def test_embedding_vs_golden(M):
    g = load_golden("u_embedding")
    x3, x1 = torch.from_numpy(g["in_x3"]).cuda(), torch.from_numpy(g["in_x1"]).cuda()
    with torch.no_grad():
        for nf in (0, 2, 4, 5, 10, 16):
            assert relerr(M.Embedding(3, nf)(x3), g[f"out_x3_f{nf}"]) <= 2e-6
            assert relerr(M.Embedding(1, nf)(x1), g[f"out_x1_f{nf}"]) <= 2e-6
        e = M.Embedding(3, 10)
        e.set_weights(0)
        assert relerr(e(x3), g["out_x3_f10_w0"]) <= 2e-6
        e.weights = list(g["in_ramp"])
        assert relerr(e(x3), g["out_x3_f10_ramp"]) <= 2e-6
        assert relerr(M.Embedding(3, 6, logscale=False)(x3), g["out_x3_f6_linear"]) <= 2e-6


def test_networks_vs_golden(M):
    from moco_flow_amd import synth
    g = load_golden("u_networks")
    with torch.no_grad():
        for extra, dim in (("dir", 27), ("ind", 5), ("none", 0)):
            sd = synth.nerf_state(11, extra_feat_type=extra, extra_feat_dim=dim, regime="dense", tag="unit")
            m = M.NeRF(8, 256, 63, [4], extra, dim)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            m = m.cuda()
            inp = torch.from_numpy(g[f"in_nerf_{extra}"]).cuda()
            out = m(inp)
            assert out.shape == (inp.shape[0], 4)
            assert relerr(out, g[f"out_nerf_{extra}_full"]) <= 2e-5, extra
            sg = m(inp[:, :63].contiguous(), sigma_only=True)
            assert sg.shape == (inp.shape[0], 1)
            assert relerr(sg, g[f"out_nerf_{extra}_sigma"]) <= 2e-5
            # non-contiguous rows (a column slice) go through the row stride
            sg2 = m(inp[:, :63], sigma_only=True) if dim else sg
            assert torch.equal(sg, sg2)
        for quat in (True, False):
            m = M.NoF(4, 128, 33, [2], "ind", 33, quat)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nof_state(13, use_quat=quat, tag="unit").items()})
            m = m.cuda()
            out = m(torch.from_numpy(g["in_nof_inputs"]).cuda(), torch.from_numpy(g["in_nof_xyz"]).cuda())
            assert relerr(out, g[f"out_nof_{'quat' if quat else 'flow'}"]) <= 2e-5, quat


def _flipped_rays(z_mine, z_ref):
    """Rays whose sorted fine depths differ: the u = 1.0 hazard. The deterministic resample's last
    draw (rendering.py:27,33) falls in the last or the last-but-one bin depending on whether
    cdf[-1] rounded to <= 1 or > 1; that depends on the association order of torch.sum
    (rendering.py:21), which is backend/ISA specific even inside the reference.
    It moves ONE fine sample of the ray by up to a bin; all other samples agree to rounding."""
    zm = torch.as_tensor(z_mine).cpu().double()
    zr = torch.as_tensor(z_ref).double()
    per_ray = (zm - zr).abs().amax(1)
    return per_ray > 1e-5


def _check_result(res, want, c, flipped=None, fine_tol=TOL):
    """Values within 1e-4 max-rel (north_star). Documented fp hazards handled explicitly:
    * alpha >= 0.01 (rendering.py:306) can flip for an alpha within an ulp of 0.01, so a consensus
      vector may differ in LENGTH by a couple of entries;
    * rays in ``flipped`` (see _flipped_rays) are excluded from the tight comparison of the
      per-ray fine outputs and checked loosely; the compacted per-sample fine consensus vectors may
      have a few outliers per flipped ray."""
    assert sorted(res) == sorted(want)
    n_flip = int(flipped.sum()) if flipped is not None else 0
    for k, v in want.items():
        got = res[k]
        assert got.dtype == torch.float32 and got.is_cuda
        v = np.asarray(v)
        if k.startswith("nof_"):
            n_g, n_w = got.shape[0], v.shape[0]
            assert abs(n_g - n_w) <= max(2, int(0.002 * n_w)), (k, n_g, n_w)
            if n_g == n_w:
                d = (got.cpu().double() - torch.from_numpy(v).double()).abs() / max(float(np.abs(v).max()), 1e-30)
                n_out = int((d > (fine_tol if k.endswith("_fine") else TOL)).sum())
                assert n_out <= (16 * n_flip if k.endswith("_fine") else 0), (k, n_out, float(d.max()))
            else:
                assert abs(float(got.mean()) - float(v.mean())) <= 1e-3 * abs(float(v.mean()))
        else:
            assert tuple(got.shape) == v.shape, k
            if k.endswith("_fine") and n_flip:
                keep = ~flipped
                assert relerr(got.cpu()[keep], v[keep.numpy()]) <= fine_tol, (k, relerr(got.cpu()[keep], v[keep.numpy()]))
                assert relerr(got, v) <= 0.1, (k, "flipped rays", relerr(got, v))
            else:
                tol = fine_tol if k.endswith("_fine") else TOL
                assert relerr(got, v) <= tol, (k, relerr(got, v))


@pytest.mark.parametrize("name", sorted(RENDER_CASES))
def test_render_rays_vs_golden(M, name):
    c = RENDER_CASES[name]
    g = load_golden(name)
    seed = int(g["meta_seed"])
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    rays = torch.from_numpy(g["in_rays"]).cuda()
    bg = torch.from_numpy(g["in_background"]).cuda() if c.get("bg", True) else None
    cap = {}
    with torch.no_grad():
        res = M.render_rays(rays, bg, embs, nerfs, _capture=cap, **kw)
        res2 = M.render_rays(rays, bg, embs, nerfs, **kw)
    for k in res:                                            # the hook must not change the result
        assert torch.equal(res[k], res2[k]), k
    want = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    flipped = None
    if c["M"] > 0 and rays.shape[0] > 0:
        flipped = _flipped_rays(cap["z_fine"], g["mid_z_fine"])
        print(f"{name}: {int(flipped.sum())}/{rays.shape[0]} rays hit the u=1.0 resampling hazard")
        assert relerr(cap["weights_coarse"], g["mid_weights_coarse"]) <= TOL
    # fine outputs vs the FIXED golden vectors inherit the conditioning of the resample (a low-weight
    # bin divides an O(ulp) cdf difference by ~1e-5, rendering.py:41-45): 3e-4 here; the fine pass
    # itself is pinned to 1e-4 on identical depths in test_render_rays_vs_oracle_larger.
    _check_result(res, want, c, flipped, fine_tol=3e-4)
    if rays.shape[0] > 0 and "alphas_coarse" in cap and cap["alphas_coarse"] is not None and "mid_alphas_coarse" in g:
        assert relerr(cap["alphas_coarse"], g["mid_alphas_coarse"]) <= TOL
        assert relerr(cap["weights_coarse"], g["mid_weights_coarse"]) <= TOL


@pytest.mark.parametrize("name", ["r_nerf_dir_dense", "r_moco_global", "r_nerf_dir_fine_train", "r_moco_global_fine"])
def test_render_rays_vs_oracle_larger(M, R, name):
    """Same seeded inputs through the oracle (CPU) and the HIP path at a size where tiles, groups
    and persistent workgroups all wrap (600 rays, not a multiple of anything). With a fine pass the
    comparison is decomposed so that it is exact despite the u = 1.0 hazard:
      (1) resample parity: the HIP resample fed the ORACLE's coarse depths/weights reproduces the
          oracle's fine depths except for the hazard's one sample on flagged rays;
      (2) fine-pass parity: the oracle re-run on the HIP path's own fine depths must agree on every
          ray to 1e-4."""
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 600 if c.get("nof", "none") == "none" else 200
    rays, bg = case_inputs(c, seed, n=n)
    embs_o, nerfs_o, kw_o = build_case(R, c, seed)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    cap_o, cap = {}, {}
    with torch.no_grad():
        want = R.render_rays(rays, bg, embs_o, nerfs_o, _capture=cap_o, **kw_o)
        res = M.render_rays(rays.cuda(), bg.cuda() if bg is not None else None, embs, nerfs, _capture=cap, **kw)
    flipped = None
    if c["M"] > 0:
        with torch.no_grad():
            z_p, inds_p, z_new_p = M.resample_merge(cap_o["z_coarse"].cuda(), cap_o["weights_coarse"].cuda(),
                                                    c["M"], det=True, return_aux=True)
            full = R.sample_pdf_full(0.5 * (cap_o["z_coarse"][:, :-1] + cap_o["z_coarse"][:, 1:]),
                                     cap_o["weights_coarse"][:, 1:-1], c["M"], det=True)
        same = inds_p.cpu().long() == full["inds"]
        assert bool(same[:, :-1].all()), "only the u = 1.0 column may differ"       # index bookkeeping
        n_flip_cols = int((~same[:, -1]).sum())
        # low-weight bins divide an O(ulp) cdf difference by a denominator as small as 1e-5
        # (rendering.py:41-45): the drawn depths agree to ~1e-4 of the range, not to rounding
        assert relerr(z_new_p.cpu()[same], full["samples"][same]) <= 3e-4
        flipped = _flipped_rays(cap["z_fine"], cap_o["z_fine"])
        print(f"{name}: last-column index differs on {n_flip_cols}/{n} rays; {int(flipped.sum())} rays moved a sample")
        with torch.no_grad():                                                       # (2)
            want2 = R.render_rays(rays, bg, embs_o, nerfs_o, _z_fine_override=cap["z_fine"].cpu(), **kw_o)
        _check_result({k: v for k, v in res.items() if "fine" in k},
                      {k: v.numpy() for k, v in want2.items() if "fine" in k}, c, None)
    _check_result(res, {k: v.numpy() for k, v in want.items()}, c, flipped, fine_tol=3e-4)


def test_z_vals_bit_identical_to_the_torch_expression(M):
    """mf_z_vals (rendering.py:245-251): the (N, S) sample depths render_rays materialises for the resample / jitter /
    backward must be the very floats of `near * (1 - t) + far * t` (and of the disparity form): they decide
    searchsorted indices downstream.  torch.equal against the reference's own expression evaluated by torch on the GPU."""
    import moco_flow_amd._lib as L
    torch.manual_seed(3)
    for N, S in ((0, 64), (1, 1), (37, 40), (600, 64), (1024, 192)):
        rays = torch.randn(N, 9, device="cuda")
        rays[:, 6] = 0.5 + 3.0 * torch.rand(N, device="cuda")
        rays[:, 7] = rays[:, 6] + 0.1 + 5.0 * torch.rand(N, device="cuda")
        t = torch.linspace(0, 1, S, device="cuda")
        near, far = rays[:, 6:7], rays[:, 7:8]
        for use_disp in (0, 1):
            want = (near * (1 - t) + far * t) if not use_disp else 1 / (1 / near * (1 - t) + 1 / far * t)
            got = torch.empty(N, S, device="cuda")
            L.check(L.lib().mf_z_vals(L.ptr(rays), rays.stride(0), N, L.ptr(t), S, use_disp, None, 0.0, L.ptr(got),
                                      L.current_stream(rays.device)), "mf_z_vals")
            assert torch.equal(got, want.expand(N, S)), (N, S, use_disp, float((got - want).abs().max()) if N else 0)
            # the stratified jitter of rendering.py:253-260 in the same launch: the reference's own expression, bit for bit
            for perturb in (1.0, 0.37):
                z_vals = want.expand(N, S).contiguous()
                pr = torch.rand(N, S, device="cuda")
                z_mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
                upper = torch.cat([z_mid, z_vals[:, -1:]], -1)
                lower = torch.cat([z_vals[:, :1], z_mid], -1)
                want_p = lower + (upper - lower) * (perturb * pr)
                L.check(L.lib().mf_z_vals(L.ptr(rays), rays.stride(0), N, L.ptr(t), S, use_disp, L.ptr(pr), perturb, L.ptr(got),
                                          L.current_stream(rays.device)), "mf_z_vals")
                assert torch.equal(got, want_p), (N, S, use_disp, perturb, float((got - want_p).abs().max()) if N else 0)


def test_sample_pdf_indices_bit_exact(M):
    """Index parity at unit level: the search fed the golden cdf and u must
    return identical indices; samples within 1e-6."""
    g = load_golden("u_sample_pdf")
    bins = torch.from_numpy(g["in_bins"]).cuda()
    w = torch.from_numpy(g["in_weights"]).cuda()
    cdf = torch.from_numpy(g["mid_cdf"]).cuda()
    N, nb = bins.shape
    M_ = 128
    L = M._lib
    for tag, u, ustride in (("det", torch.from_numpy(np.ascontiguousarray(g["mid_u_det"][0])).cuda(), 0),
                            ("rand", torch.from_numpy(g["in_u_rand"]).cuda(), M_)):
        inds = torch.empty((N, M_), dtype=torch.int32, device="cuda")
        out = torch.empty((N, M_), dtype=torch.float32, device="cuda")
        L.check(L.lib().mf_sample_pdf(bins.data_ptr(), None, w.data_ptr(), nb - 1, N, nb, M_, u.data_ptr(), ustride,
                                      cdf.data_ptr(), out.data_ptr(), inds.data_ptr(), None,
                                      L.current_stream(bins.device)))
        torch.cuda.synchronize()
        assert torch.equal(inds.cpu().long(), torch.from_numpy(g[f"mid_inds_{tag}"])), tag
        assert relerr(out, g[f"out_samples_{tag}"]) <= 1e-6
    # own cdf (left-to-right normaliser): everything but the u = 1.0 column agrees to rounding
    with torch.no_grad():
        s = M.sample_pdf(bins, w, M_, det=True)
    assert relerr(s[:, :-1], g["out_samples_det"][:, :-1]) <= 3e-4
    assert s.shape == (N, M_)
    # sample_pdf's `eps` argument (rendering.py:5; every call of the reference leaves it at 1e-5 -- rounds 1-3 raised on any
    # other value): against the oracle on the same draws, all columns but u = 1.0
    from oracle import cpu_ref as Ro
    for eps in (1e-3, 1e-7):
        with torch.no_grad():
            s2 = M.sample_pdf(bins, w, M_, det=True, eps=eps)
        want = Ro.sample_pdf(bins.cpu(), w.cpu(), M_, det=True, eps=eps)
        assert relerr(s2[:, :-1], want[:, :-1]) <= 3e-4, eps
    assert relerr(M.sample_pdf(bins, w, M_, det=True, eps=1e-3)[:, :-1], s[:, :-1]) > 1e-4        # (the argument matters)


@pytest.mark.parametrize("S,Mi", [(8, 5), (64, 64), (64, 128), (64, 192), (40, 100), (128, 256), (200, 312), (192, 400), (256, 768),
                                  (300, 800)])
def test_resample_merge_is_torch_sort_of_the_union(M, S, Mi):
    """rendering.py:326 `torch.sort(torch.cat([z_vals, z_vals_], -1), -1)`: the merged depths of the one-launch resample are
    bit for bit torch.sort of [coarse depths, the launch's own new samples], for every width class of the in-register
    bitonic network (T = S + M <= 128 / 256 / 512 / 1024, the exactly full networks T = 128 / 256 / 512 / 1024 included) and the
    rank-sort path behind it (T > 1024), with sorted
    (linspace) and unsorted (random) draws, duplicates included (zero-weight bins collapse samples onto bin edges)."""
    N = 8192 if (S, Mi) == (64, 128) else 77                   # (BASELINE config C5's full ray count at its own widths)
    g = torch.Generator(device="cuda").manual_seed(S * 1000 + Mi)
    z = torch.sort(2.0 + 4.0 * torch.rand(N, S, device="cuda", generator=g), -1)[0]
    w = torch.rand(N, S, device="cuda", generator=g)
    w[:, S // 3: S // 2] = 0.0
    w[::5] = 0.0                                                                 # eps-only pdf: uniform
    for u in (None, torch.rand(N, Mi, device="cuda", generator=g)):
        with torch.no_grad():
            z_out, _, z_new = M.resample_merge(z, w, Mi, det=True, u=u, return_aux=True)
        want = torch.sort(torch.cat([z, z_new], -1), -1)[0]
        assert z_out.shape == (N, S + Mi)
        assert torch.equal(z_out, want), (S, Mi, u is None, int((z_out != want).sum()))


def test_compaction_order_bit_exact(M):
    torch.manual_seed(0)
    N, S = 37, 64
    alphas = torch.rand(N, S, device="cuda") * 0.03
    va, vb = torch.randn(N, S, device="cuda"), torch.randn(N, S, device="cuda")
    from moco_flow_amd.rendering import _compact
    oa, ob = _compact(alphas, va, vb)
    mask = alphas.ge(0.01)
    assert torch.equal(oa, va[mask]) and torch.equal(ob, vb[mask])
    oa, _ = _compact(torch.zeros(N, S, device="cuda"), va, None)       # all-true fallback
    assert torch.equal(oa, va.reshape(-1))


def test_errors_and_edge_cases(M):
    with torch.no_grad():
        embs, nerfs, kw = build_case(M, RENDER_CASES["r_nerf_dir_dense"], 5, device="cuda")
        rays = torch.zeros(0, 9, device="cuda")
        res = M.render_rays(rays, torch.zeros(0, 3, device="cuda"), embs, nerfs, **kw)
        assert res["rgb_coarse"].shape == (0, 3) and res["depth_coarse"].shape == (0,)
        with pytest.raises(ValueError):
            M.render_rays(torch.zeros(4, 9, device="cuda"), None, embs, nerfs, **{**kw, "nerf_activate_type": "tanh"})
    with pytest.raises(RuntimeError):
        M.NeRF(8, 256, 63, [4], "dir", 27)(torch.zeros(2, 90))          # CPU tensor: no fallback


def test_trainer_glue_vs_golden(M):
    """The trainers call the networks directly (trainer_moco_flow.py:146-187): embed -> NeRF(sigma_only)
    -> softplus alpha, and embed(xyz) ++ embed(ind) -> NoF. Same module calls here, HIP underneath."""
    from moco_flow_amd import synth
    g = load_golden("u_trainer_glue")
    xyz = torch.from_numpy(g["in_xyz"]).cuda()
    B = xyz.shape[0]
    with torch.no_grad():
        nerf = M.NeRF(8, 256, 63, [4], "ind", 5)
        nerf.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(
            32, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag="glue").items()})
        nerf = nerf.cuda()
        xe = torch.zeros((B, nerf.in_channels_xyz), device="cuda")
        e = M.Embedding(3, 10)(xyz)
        xe[:, :e.shape[1]] = e
        sig = nerf(xe, sigma_only=True)
        alphas = 1 - torch.exp(-float(g["in_delta"]) * torch.nn.Softplus()(sig))
        assert relerr(alphas, g["out_alphas"]) <= TOL
        nof = M.NoF(4, 128, 33, [2], "ind", 33, True)
        nof.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nof_state(33, use_quat=True, tag="glue").items()})
        nof = nof.cuda()
        ind = torch.from_numpy(g["in_ind"]).cuda()
        xe2 = torch.zeros((B, nof.in_channels_xyz), device="cuda")
        e2 = M.Embedding(3, 5)(xyz)
        xe2[:, :e2.shape[1]] = e2
        ie = torch.zeros((B, nof.extra_feat_dim), device="cuda")
        indf = ind.unsqueeze(dim=0).repeat((B, 1)).float() * 2 / int(g["in_num_frames"]) - 1.0
        ie_ = M.Embedding(1, 16)(indf)
        ie[:, :ie_.shape[1]] = ie_
        out = nof(torch.cat([xe2, ie], -1), xyz, ind)
        assert relerr(out, g["out_nof_xyz"]) <= TOL


def test_full_size_properties(M):
    """BASELINE sizes (4096 x 64 and the C5-shaped 192-sample fine pass): size-independent properties.
    * permutation equivariance / shard consistency: rendering a shuffled or split batch gives the same
      per-ray results bit-for-bit (rays are independent units; this is what the N-GPU split relies on);
    * opacity in [0,1], weights >= 0 and summing to opacity, depth within [near*op, far*op];
    * sorted fine depths; idempotent re-run (bit-identical: the kernels are deterministic)."""
    from moco_flow_amd import synth
    from moco_flow_amd.dist import shard_bounds
    n = 4096
    rays_np, bg_np = synth.rays(0, n)
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    embs, nerfs, kw = build_case(M, dict(RENDER_CASES["r_nerf_dir_fine_train"]), 7, device="cuda")
    cap = {}
    with torch.no_grad():
        a = M.render_rays(rays, bg, embs, nerfs, _capture=cap, **kw)
        b = M.render_rays(rays, bg, embs, nerfs, **kw)
        perm = torch.randperm(n, device="cuda")
        c = M.render_rays(rays[perm], bg[perm], embs, nerfs, **kw)
        parts = [M.render_rays(rays[lo:hi], bg[lo:hi], embs, nerfs, **kw)
                 for lo, hi in (shard_bounds(n, r, 8) for r in range(8))]
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(a[k][perm], c[k]), k
        assert torch.equal(a[k], torch.cat([p[k] for p in parts], 0)), k
    for tag in ("coarse", "fine"):
        op, w, z = a[f"opacity_{tag}"], cap[f"weights_{tag}"], cap[f"z_{tag}"]
        assert float(op.min()) >= 0 and float(op.max()) <= 1 + 1e-5
        assert float(w.min()) >= 0
        assert relerr(w.sum(1), op) <= 1e-5
        assert bool((z[:, 1:] >= z[:, :-1]).all())
        d = a[f"depth_{tag}"]
        assert bool((d >= 2.0 * op - 1e-4).all()) and bool((d <= 6.0 * op + 1e-4).all())
    assert cap["z_fine"].shape == (n, 192)


def _oracle_grads(R, c, seed, rays, bg, loss_fn, dtype=torch.float32):
    """Oracle forward + autograd of `loss_fn` at `dtype` (float64: the arithmetic-noise-free truth of the same function)."""
    embs_o, nerfs_o, kw_o = build_case(R, c, seed)
    nets = list(nerfs_o) + (list(kw_o["nof_models"]) if kw_o["nof_models"] else [])
    for m in nets:
        for k in m.p:
            m.p[k] = m.p[k].to(dtype).clone().requires_grad_(True)
    for e in list(embs_o) + list(kw_o["nof_embeddings"] or []):
        if e is not None:
            e.freq_bands = e.freq_bands.to(dtype)
    torch.set_default_dtype(dtype)              # the oracle allocates its pads / ones with the default dtype
    try:
        res = R.render_rays(rays.to(dtype), bg.to(dtype) if bg is not None else None, embs_o, nerfs_o, **kw_o)
        loss = loss_fn(res)
    finally:
        torch.set_default_dtype(torch.float32)
    flat = [(i, k) for i, m in enumerate(nets) for k in m.p]
    grads = torch.autograd.grad(loss, [nets[i].p[k] for i, k in flat], allow_unused=True)
    return res, {f"{i}.{k}": g for (i, k), g in zip(flat, grads)}


# End-to-end gradient bars.  The truth is the ORACLE IN FLOAT64 (the same function without arithmetic noise); the yardstick
# per parameter tensor is the reference arithmetic's own distance to it, noise = l2-rel(fp32 oracle autograd, float64 oracle
# autograd): a HIP tensor must be within max(1e-4, 3 x noise) of the truth in l2-rel, with the max-rel distance to the fp32
# oracle bounded the same way (guard against a wrong element).  NeRF-only passes: HIP and the fp32 oracle evaluate the same
# ReLU masks and agree to 1e-6 .. 8e-6 although both sit up to 7e-3 from the float64 truth on the first layers (mask flips) --
# there the fixed 1e-4 against the fp32 oracle stays as the tighter bar.  The MoCo cases carry sin(512 x) of a canonical point through dense NoFs: there the fp32 and
# float64 oracles differ by percents on some tensors (r3 measured 30-120 % max-rel on r_moco_global's NeRF / backward-NoF
# tensors, 0.6 % at default init), and a fixed bar either hides a regression on the well-conditioned tensors or fails on the
# others -- rounds 1-3 used 5e-2 / 3e-3 against the fp32 oracle; per-tensor noise floors replace them.  What pins each backward
# KERNEL at 1e-4 is test_*_backward_vs_oracle* (same function at the same points, masks included).
GRAD_CASES = ["r_nerf_dir_dense", "r_nerf_ind_dense", "r_moco_global", "r_moco_global_default"]


def _check_grads_vs_float64(nets, want32, want64, skip=lambda k: False, label="", fp32_bar=None):
    """fp32_bar: fixed max-rel bar against the fp32 oracle (NeRF-only passes: HIP and the fp32 oracle evaluate the same ReLU
    masks and agree to ~1e-6 where both sit 1e-3 from the float64 truth -- there the fixed 1e-4 is the tighter test).
    The yardstick of a tensor is the larger of its own noise and the MEDIAN noise of its network's tensors: one fp32-vs-float64
    pair is a single draw of a random distance, and on some tensor it comes out several times under the typical one
    (r_moco_global_default, xyz_encoding_8.weight: 1.6e-5 where the network's median is 3e-4 and HIP sits at 2.1e-4)."""
    checked, worst, floor = 0, (0.0, ""), (0.0, "")
    for i, m in enumerate(nets):
        live = [k for k, _ in m.named_parameters() if not skip(k) and want64[f"{i}.{k}"] is not None
                and float(want64[f"{i}.{k}"].abs().max()) > 0.0]
        net_l2 = float(np.median([_l2rel(want32[f"{i}.{k}"], want64[f"{i}.{k}"]) for k in live])) if live else 0.0
        net_mr = float(np.median([relerr(want32[f"{i}.{k}"], want64[f"{i}.{k}"]) for k in live])) if live else 0.0
        for k, p in m.named_parameters():
            key = f"{i}.{k}"
            if skip(k):
                assert p.grad is None, key
                continue
            w64, w32 = want64[key], want32[key]
            if w64 is None or float(w64.abs().max()) == 0.0:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, key
                continue
            assert p.grad is not None, key
            noise_l2, noise_mr = max(_l2rel(w32, w64), net_l2), max(relerr(w32, w64), net_mr)
            e_l2, e_mr32 = _l2rel(p.grad, w64), relerr(p.grad, w32)
            worst, floor = max(worst, (e_l2, key)), max(floor, (noise_l2, key))
            # (NeRF-only passes, fp32_bar given: the multiplier is capped -- ADVICE r4 -- 3 x a 7.6e-3 first-layer floor would let
            #  a 2e-2 error through; the MoCo cases' floors reach 100 % on some tensors and cannot be capped)
            # (HIP shares the fp32 oracle's ReLU masks there and so sits AT the fp32 oracle's own distance from the truth: the cap
            #  cannot go under 1.5 x that distance; the fixed fp32_bar below is what catches a mis-scaled gradient)
            assert e_l2 <= max(TOL, 3 * noise_l2 if fp32_bar is None else min(3 * noise_l2, max(1.5 * noise_l2, 5e-3))), (key, e_l2, noise_l2)
            # max-rel guard against a wrong element.  Its floor is 2e-3, not 1e-4: ONE ReLU unit whose pre-activation rounds to the
            # other side of zero between two fp32 evaluation orders moves a weight-gradient row of the layers in front of it by
            # that sample's whole contribution -- a discrete event (measured 1.4e-4 .. 3.8e-4 of max|dW| at 6144 samples, layers
            # 1-3 only, the layers behind the unit agreeing to 2e-7, with the fp32 backward kernels too: a one-off script, since removed) that
            # the fp32-vs-float64 pair of the same batch need not contain
            assert e_mr32 <= (fp32_bar if fp32_bar is not None else max(2e-3, 3 * noise_mr)), (key, e_mr32, noise_mr)
            checked += 1
    print(f"{label}: end-to-end gradients vs the float64 oracle, worst l2-rel {worst[0]:.2e} at {worst[1]} (the fp32 oracle's own worst "
          f"{floor[0]:.2e} at {floor[1]}; {checked} tensors, each within max(1e-4, 3 x its noise floor))")
    return checked


@pytest.mark.parametrize("name", ["r_nerf_dir_dense", "r_nerf_ind_dense"])
def test_train_forward_bf16x3(M, R, name):
    """rendering.set_train_forward_precision("bf16x3") (opt-in; NeRF-only passes): the training forward on the three-product
    kernels, writing the same activation dump; dX chain in fp32 on that dump.  Forward values hold the fp32 contract (1e-4
    max-rel, measured ~1e-5).  Gradients do NOT: a ReLU network's gradient is discontinuous where a pre-activation crosses
    zero, and the forward's 1e-5 moves ~100x more units across than fp32 rounding does -- the same effect separates the
    oracle's own fp32 and fp64 gradients, here at 2e-3 max-rel / 3e-4 l2-rel per parameter tensor.
    That is why the mode is opt-in and "f32" the default; the bars document the measured level."""
    from moco_flow_amd import rendering
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 48
    rays, bg = case_inputs(c, seed, n=n)
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))

    def loss_fn(res, gt=gt):
        return ((res["rgb_coarse"] - gt.to(res["rgb_coarse"].device)) ** 2).mean() + 0.1 * res["depth_coarse"].mean()

    want_res, want = _oracle_grads(R, c, seed, rays, bg, loss_fn)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    try:
        rendering.set_train_forward_precision("bf16x3")
        res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, **kw)
        loss_fn(res).backward()
    finally:
        rendering.set_train_forward_precision("f32")
    for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
        assert relerr(res[k], want_res[k]) <= TOL, (k, relerr(res[k], want_res[k]))
    worst_mr, worst_l2 = 0.0, 0.0
    for k, p in nerfs[0].named_parameters():
        w = want[f"0.{k}"]
        if w is None:
            continue
        mr, l2 = relerr(p.grad, w), _l2rel(p.grad, w)
        worst_mr, worst_l2 = max(worst_mr, mr), max(worst_l2, l2)
    print(f"{name}: bf16x3 training forward: gradients vs oracle autograd worst max-rel {worst_mr:.2e}, l2-rel {worst_l2:.2e}")
    assert worst_mr <= 1e-2 and worst_l2 <= 2e-3
    # the same gradients against the FLOAT64 oracle (the truth of the function), with the fp32 oracle's own distance to it as the
    # yardstick (GRAD_CASES above): the three-product forward's mask flips are of the size of the reference arithmetic's own --
    # every tensor within max(1e-4, 3 x noise) -- which is the case for making this mode a default some day, and why it is safe for SGD
    _, want64 = _oracle_grads(R, c, seed, rays, bg, loss_fn, torch.float64)
    assert _check_grads_vs_float64([nerfs[0]], want, want64, label=name + " (bf16x3 training forward)") >= 20
    # bit-identical between runs at a size that keeps every workgroup busy for several tiles (the dump stores stay in flight
    # across the panel barriers: a piece of the weight stream that had not landed would show up here)
    rays2, bg2 = case_inputs(c, seed, n=1500)
    grads = []
    try:
        rendering.set_train_forward_precision("bf16x3")
        for _ in range(3):
            for p in nerfs[0].parameters():
                p.grad = None
            r2 = M.render_rays(rays2.cuda(), bg2.cuda(), embs, nerfs, **kw)
            (r2["rgb_coarse"].square().mean() + r2["depth_coarse"].mean()).backward()
            grads.append([p.grad.clone() for p in nerfs[0].parameters()] + [r2["rgb_coarse"].detach().clone()])
    finally:
        rendering.set_train_forward_precision("f32")
    for g in grads[1:]:
        assert all(torch.equal(a, b) for a, b in zip(grads[0], g))
    # ragged ends: sample counts that leave whole waves of the last 128-sample tile without a sample (such a wave issues no
    # dump stores, so its panel barriers must wait for everything: StreamT::sync's keep_ok)
    c40 = dict(RENDER_CASES["r_nerf_dir_S40"])
    embs40, nerfs40, kw40 = build_case(M, c40, seed, device="cuda")
    for n_r in (1, 5, 7):
        r40, b40 = case_inputs(c40, seed, n=n_r)
        _, want40 = None, None
        embs_o, nerfs_o, kw_o = build_case(R, c40, seed)
        with torch.no_grad():
            want40 = R.render_rays(r40, b40, embs_o, nerfs_o, **kw_o)
        outs = []
        try:
            rendering.set_train_forward_precision("bf16x3")
            for _ in range(3):
                outs.append(M.render_rays(r40.cuda(), b40.cuda(), embs40, nerfs40, **kw40))
        finally:
            rendering.set_train_forward_precision("f32")
        assert outs[0]["rgb_coarse"].requires_grad
        for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
            assert relerr(outs[0][k], want40[k]) <= TOL, (n_r, k, relerr(outs[0][k], want40[k]))
            assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k])


@pytest.mark.parametrize("name", ["r_moco_global", "r_moco_global_default", "r_moco_local"])
def test_train_forward_bf16x3_moco(M, R, name):
    """Round 5: rendering.set_train_forward_precision("bf16x3") for passes WITH NoF -- render_kernel_bf16<true, true, true>: the
    MoCo chain (NoFs in IEEE-half pairs) + canonical NeRF in three products, dumping the NeRF's activations and, per chain
    step, the NoF rows [h_1 .. h_D | T] + output points; the embedded-input plane comes from mf_nof_embed_rows (natural column
    order).  Forward values hold the fp32 contract (1e-4 max-rel on rgb / depth / opacity; consensus means 1e-4); gradients
    are judged like the fp32 default's (test_gradients_vs_oracle): float64 oracle as the truth, per tensor within
    max(1e-4, 3 x the fp32 oracle's own distance to it) -- the three-product forward moves ReLU units across zero as fp32 rounding
    does, only more of them (opt-in for that reason).  Same values and gradients between runs, bit for bit."""
    from moco_flow_amd import rendering
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 48
    rays, bg = case_inputs(c, seed, n=n)
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))

    def loss_fn(res, gt=gt):
        loss = ((res["rgb_coarse"] - gt.to(res["rgb_coarse"])) ** 2).mean() + 0.1 * res["depth_coarse"].mean()
        for k in ("nof_local_disp_coarse", "nof_global_disp_coarse"):
            if k in res:
                loss = loss + 0.2 * res[k].mean()
        return loss

    want_res, want32 = _oracle_grads(R, c, seed, rays, bg, loss_fn)
    _, want64 = _oracle_grads(R, c, seed, rays, bg, loss_fn, torch.float64)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    nets = list(nerfs) + list(kw["nof_models"])
    runs = []
    try:
        rendering.set_train_forward_precision("bf16x3")
        for _ in range(2):
            for m in nets:
                m.zero_grad(set_to_none=True)
            res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, **kw)
            assert res["rgb_coarse"].requires_grad
            loss_fn(res).backward()
            runs.append([res["rgb_coarse"].detach().clone()] + [q.grad.clone() for m in nets for q in m.parameters() if q.grad is not None])
    finally:
        rendering.set_train_forward_precision("f32")
    assert len(runs[0]) == len(runs[1]) and all(torch.equal(a, b) for a, b in zip(*runs))
    for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
        assert relerr(res[k], want_res[k]) <= TOL, (k, relerr(res[k], want_res[k]))
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse"):
        if k in want_res:
            a, b = float(res[k].mean()), float(want_res[k].mean())
            assert abs(a - b) <= 1e-4 * abs(b), (k, a, b)
    checked = _check_grads_vs_float64(nets, want32, want64, label=name + " (bf16x3 training forward, MoCo)")
    assert checked >= 20
    # ragged ends: 7 rays x 64 samples = 3.5 tiles of 128 -- whole waves of the last tile without a sample issue no dump stores
    # (StreamT::sync's keep_ok), the NoF planes are not whole 128-row blocks (per-node weight gradients instead of the sinks)
    rays7, bg7 = case_inputs(c, seed, n=7)
    with torch.no_grad():
        embs_o, nerfs_o, kw_o = build_case(R, c, seed)
        want7 = R.render_rays(rays7, bg7, embs_o, nerfs_o, **kw_o)
    outs = []
    try:
        rendering.set_train_forward_precision("bf16x3")
        for _ in range(2):
            for m in nets:
                m.zero_grad(set_to_none=True)
            r7 = M.render_rays(rays7.cuda(), bg7.cuda(), embs, nerfs, **kw)
            loss_fn(r7, gt[:7]).backward()
            outs.append([r7["rgb_coarse"].detach().clone()] + [q.grad.clone() for m in nets for q in m.parameters() if q.grad is not None])
    finally:
        rendering.set_train_forward_precision("f32")
    assert all(torch.equal(a, b) for a, b in zip(*outs)) and all(bool(torch.isfinite(t).all()) for t in outs[0])
    for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
        assert relerr(r7[k], want7[k]) <= TOL, (k, relerr(r7[k], want7[k]))


@pytest.mark.parametrize("name", GRAD_CASES)
def test_gradients_vs_oracle(M, R, name, wgrad):
    """Training contract (moco_flow_amd/autograd.py): forward values from the HIP kernels, gradients from
    the HIP backward nodes (composite, NeRF dX chain + weight gradients, NoF evaluations); against the CPU oracle's autograd
    for the reference's loss shape (MSE on rgb + consensus means, trainer_moco_flow.py:317-328), float64 oracle as the truth
    and the fp32 oracle's own distance to it as the per-tensor yardstick (see above)."""
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 48
    rays, bg = case_inputs(c, seed, n=n)
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))

    def loss_fn(res, gt=gt):
        loss = ((res["rgb_coarse"] - gt.to(res["rgb_coarse"])) ** 2).mean() + 0.1 * res["depth_coarse"].mean()
        for k in ("nof_local_disp_coarse", "nof_global_disp_coarse"):
            if k in res:
                loss = loss + 0.2 * res[k].mean()
        return loss

    _, want32 = _oracle_grads(R, c, seed, rays, bg, loss_fn)
    _, want64 = _oracle_grads(R, c, seed, rays, bg, loss_fn, torch.float64)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    nets = list(nerfs) + (list(kw["nof_models"]) if kw["nof_models"] else [])
    # frozen sub-module (trainer_moco_flow.py:391-404): no grad must reach it
    for p in nerfs[0].rgb.parameters():
        p.requires_grad_(False)
    res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, **kw)
    assert res["rgb_coarse"].requires_grad
    loss_fn(res).backward()
    checked = _check_grads_vs_float64(nets, want32, want64, skip=lambda k: k.startswith("rgb."), label=name,
                                      fp32_bar=TOL if c.get("nof", "none") == "none" else None)
    assert checked >= 10


def test_consensus_mean_node_equals_the_torch_reductions(M):
    """torch.mean of a training pass's consensus vector is ONE autograd node (autograd.ConsensusMean: mf_loss_partials forward on
    the distances the fused pass wrote, mf_loss_partials_backward with the seed g / count).  Against the same step with
    rendering.FUSED_CONSENSUS_MEAN off -- torch ops on |x - recon|, the mask and the masked sums, rounds 2-3: the means agree to fp32
    rounding, every parameter gradient to 1e-5 l2-rel (both seed -g mask sign(x - recon) / (3 count)); the trainer's in-place
    accumulation of the coarse and fine means (trainer_moco_flow.py:318-321) works on the node's output, and the other
    reductions (.sum(), the vector itself) still come from the per-sample planes."""
    from moco_flow_amd import rendering
    c = dict(RENDER_CASES["r_moco_global"])
    c["M"] = 16                                                # (a fine pass: two vectors per chain)
    seed = int(load_golden("r_moco_global")["meta_seed"])
    n = 40
    rays, bg = case_inputs(c, seed, n=n)
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(1)).cuda()

    def step(fused):
        embs, nerfs, kw = build_case(M, c, seed, device="cuda")
        nets = list(nerfs) + list(kw["nof_models"])
        rendering.FUSED_CONSENSUS_MEAN = fused
        try:
            res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, **kw)
            local = torch.mean(res["nof_local_disp_coarse"])
            local += torch.mean(res["nof_local_disp_fine"])
            glob = torch.mean(res["nof_global_disp_coarse"])
            glob += torch.mean(res["nof_global_disp_fine"])
            assert local.requires_grad and local.dtype == torch.float32 and local.dim() == 0
            extra = res["nof_local_disp_fine"].sum() / res["nof_local_disp_fine"].shape[0]      # the plane path, both settings
            loss = ((res["rgb_coarse"] - gt) ** 2).mean() + ((res["rgb_fine"] - gt) ** 2).mean() + 0.3 * local + 0.2 * glob
            (loss + 0.1 * extra).backward()
        finally:
            rendering.FUSED_CONSENSUS_MEAN = True
        return float(local.detach()), float(glob.detach()), float(extra.detach()), {f"{i}.{k}": p.grad for i, m in enumerate(nets) for k, p in m.named_parameters()}

    l1, g1, e1, grads1 = step(True)
    l0, g0, e0, grads0 = step(False)
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and abs(g1 - g0) <= 2e-6 * abs(g0), (l1, l0, g1, g0)
    assert abs(e1 - e0) <= 2e-6 * abs(e0)
    worst = 0.0
    for k, g in grads0.items():
        assert (g is None) == (grads1[k] is None), k
        if g is not None and float(g.abs().max()) > 0:
            worst = max(worst, _l2rel(grads1[k], g))
    print(f"ConsensusMean node vs torch reductions: means {l1:.6g} / {l0:.6g}, worst parameter-gradient l2-rel {worst:.2e}")
    assert worst <= 1e-5


@pytest.mark.parametrize("name", ["r_nerf_dir_dense", "r_moco_global_default"])
def test_gradients_when_the_loss_skips_outputs(M, R, name):
    """A loss that does not touch rgb (depth + opacity only): the composite node gets NO seed for rgb (autograd hands None,
    the node does not materialise zeros -- mf_composite_backward takes a null pointer) and the gradients still match the
    oracle's autograd (float64 truth, per-tensor noise floor: GRAD_CASES above); an output-free NoF evaluation chain leaves
    its parameters without gradient."""
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    rays, bg = case_inputs(c, seed, n=40)

    def loss_fn(res):
        return 0.3 * res["depth_coarse"].mean() + 0.2 * res["opacity_coarse"].square().mean()

    _, want32 = _oracle_grads(R, c, seed, rays, bg, loss_fn)
    _, want64 = _oracle_grads(R, c, seed, rays, bg, loss_fn, torch.float64)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    nets = list(nerfs) + (list(kw["nof_models"]) if kw["nof_models"] else [])
    loss_fn(M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, **kw)).backward()
    assert _check_grads_vs_float64(nets, want32, want64, label=name + " (no rgb term)",
                                   fp32_bar=TOL if c.get("nof", "none") == "none" else None) >= 8


@pytest.mark.parametrize("name", ["r_nerf_dir_fine_test", "r_moco_global_fine_test"])
def test_test_time_pass_with_gradients(M, R, name):
    """render_rays(test_time=True, N_importance > 0) under grad (rendering.py:290-294: the coarse pass returns opacity_coarse
    only, from a sigma-only evaluation; the reference renders such passes under no_grad, but its signature allows gradients
    and rounds 1-3 raised here).  Same keys as the reference; values 1e-4; gradients of a loss on opacity_coarse + the fine
    outputs against the oracle's autograd (float64 truth, per-tensor noise floors), the coarse network's rgb branch left
    without gradient exactly as torch autograd leaves it."""
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 40
    rays, bg = case_inputs(c, seed, n=n)
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    nets = list(nerfs) + (list(kw["nof_models"]) if kw["nof_models"] else [])
    cap = {}
    res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, _capture=cap, **kw)
    assert list(res) == ["opacity_coarse", "rgb_fine", "depth_fine", "opacity_fine"] and res["opacity_coarse"].requires_grad
    z_fine = cap["z_fine"].cpu()

    def loss_fn(r):
        return ((r["rgb_fine"] - gt.to(r["rgb_fine"])) ** 2).mean() + 0.3 * r["opacity_coarse"].square().mean() + 0.1 * r["depth_fine"].mean()

    loss_fn(res).backward()

    def oracle(dtype):
        embs_o, nerfs_o, kw_o = build_case(R, c, seed)
        nets_o = list(nerfs_o) + (list(kw_o["nof_models"]) if kw_o["nof_models"] else [])
        for m in nets_o:
            for k in m.p:
                m.p[k] = m.p[k].to(dtype).clone().requires_grad_(True)
        for e in list(embs_o) + list(kw_o["nof_embeddings"] or []):
            if e is not None:
                e.freq_bands = e.freq_bands.to(dtype)
        torch.set_default_dtype(dtype)
        try:
            r = R.render_rays(rays.to(dtype), bg.to(dtype), embs_o, nerfs_o, _z_fine_override=z_fine.to(dtype), **kw_o)
            loss = loss_fn(r)
        finally:
            torch.set_default_dtype(torch.float32)
        flat = [(i, k) for i, m in enumerate(nets_o) for k in m.p]
        g = torch.autograd.grad(loss, [nets_o[i].p[k] for i, k in flat], allow_unused=True)
        return r, {f"{i}.{k}": v for (i, k), v in zip(flat, g)}

    want, g32 = oracle(torch.float32)
    _, g64 = oracle(torch.float64)
    for k, v in want.items():
        assert relerr(res[k].detach(), v.detach()) <= TOL, (k, relerr(res[k].detach(), v.detach()))
    for k in ("xyz_encoding_final.weight", "extra_encoding.0.weight", "rgb.0.weight"):      # the coarse NeRF's rgb branch: unused
        assert g32[f"0.{k}"] is None and dict(nerfs[0].named_parameters())[k].grad is None, k
    assert _check_grads_vs_float64(nets, g32, g64, label=name + " (test_time)") >= 30


def test_module_gradients(M, R):
    """NoF / Embedding called directly with grad (trainer_nof.py:111, trainer_moco_flow.py:153,185) against the oracle's
    CPU autograd; gradients w.r.t. the points of a module-level NoF call are not built and say so."""
    O = OracleOps(R)
    torch.manual_seed(0)
    nof = M.NoF(4, 128, 33, [2], "ind", 33, True).cuda()
    inp = torch.randn(40, 66, device="cuda")
    xyz = torch.randn(40, 3, device="cuda")
    out = nof(inp, xyz)
    out.square().sum().backward()
    ref = O.nof_forward(nof, inp.cpu(), xyz.cpu())
    assert relerr(out, ref) <= 1e-5
    ref.square().sum().backward()
    want = O.grads(nof)
    for n, q in nof.named_parameters():
        assert relerr(q.grad, want[n]) <= 1e-4, n
    with pytest.raises(NotImplementedError):
        nof(inp, xyz.clone().requires_grad_(True))
    x = torch.randn(16, 3, device="cuda", requires_grad=True)
    e = M.Embedding(3, 4)
    e(x).sum().backward()
    xc = x.detach().cpu().requires_grad_(True)
    want_x = torch.autograd.grad(O.embed(e, xc).sum(), xc)[0]
    assert relerr(x.grad, want_x) <= 1e-5


def _psnr(a, b):
    """models/metrics.py:4-13 with unit peak: -10 log10(mean((a-b)^2))."""
    mse = float(((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()) ** 2).mean())
    return -10 * np.log10(mse) if mse > 0 else 200.0


def _l2rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# bf16 bars: a few dB / a factor ~2 under what the kernels measure (printed by the tests; profiles/README.md), so
# that a regression of the bf16 arithmetic fails instead of hiding under a generic "> 38 dB".
#            case                     PSNR-equiv rgb (dB)   l2-rel rgb   l2-rel depth/opacity
# measured r2 (dB / l2 rgb / worst l2 of depth, opacity): dense 61.5 / 1.2e-3 / 7.5e-4; fine_train 57.5 / 1.7e-3 /
# 4.1e-3; default 95.0 / 3.6e-5 / 1.7e-5; moco_global 46.1 / 9.2e-3; moco_global_fine 50.4 / 4.6e-3 / 5.7e-4
# measured with the NeRF's encodings as plain bf16 operands (later in r2): dense 61.8 / 1.2e-3 / 7.1e-4; fine_train
# 57.4 / 1.7e-3 / 5.0e-3; default 90.4 / 6.0e-5; moco_global 46.5 / 8.8e-3 / 1.4e-2; moco_global_fine 50.8 / 4.4e-3
BF16_BARS = {"r_nerf_dir_dense":      (58.0,                3e-3,        2e-3),
             "r_nerf_dir_fine_train": (55.0,                4e-3,        8e-3),
             "r_nerf_dir_default":    (88.0,                1e-4,        1e-4),
             "r_moco_global":         (44.0,                1.4e-2,      2e-2),
             "r_moco_global_fine":    (48.0,                8e-3,        2e-3)}


# the same fixtures in bf16x3: PSNR-equiv / l2-rel / max-rel bars (1e-4 max-rel = the fp32 contract, except where the
# fixture's own fp32 conditioning is worse: see the per-case note)
X3_BARS = {"r_nerf_dir_dense":      (100.0, 3e-5, 3e-5, 1e-4),
           "r_nerf_dir_fine_train": (100.0, 3e-5, 3e-5, 1e-4),
           "r_nerf_dir_default":    (110.0, 1e-5, 1e-5, 1e-4),
           "r_moco_global":         (95.0,  5e-5, 5e-5, 1e-4),
           "r_moco_global_fine":    (95.0,  5e-5, 5e-5, 1e-4)}


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("name", sorted(BF16_BARS))
def test_bf16_hidden_gemms(M, R, name, precision):
    """BASELINE configs C3-C5: bf16 hidden GEMMs (fp32 accumulate; the NoF's embedded-input k-ranges and head as a
    16-bit two-term bf16 split, the NeRF's encodings as plain bf16 operands -- measured 90.4 dB at default init (95.0
    with the split), 61.8 dB dense (unchanged); heads and composite in fp32).  Not the 1e-4 contract -- north_star allows a PSNR-equivalent
    error for bf16 -- but the bars sit just under the measured values (BF16_BARS), per case,
    with l2-rel bounds on every per-ray output.  With a fine pass the oracle is re-run on the HIP path's own
    fine depths, so the resample's conditioning does not enter."""
    from moco_flow_amd import rendering
    c = dict(RENDER_CASES[name])
    g = load_golden(name)
    seed = int(g["meta_seed"])
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    rays = torch.from_numpy(g["in_rays"]).cuda()
    bg = torch.from_numpy(g["in_background"]).cuda() if c.get("bg", True) else None
    cap = {}
    try:
        rendering.set_precision(precision)
        with torch.no_grad():
            res = M.render_rays(rays, bg, embs, nerfs, _capture=cap, **kw)
    finally:
        rendering.set_precision("f32")
    want = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("out_")}
    if c["M"] > 0:
        embs_o, nerfs_o, kw_o = build_case(R, c, seed)
        with torch.no_grad():
            w2 = R.render_rays(rays.cpu(), bg.cpu() if bg is not None else None, embs_o, nerfs_o,
                               _z_fine_override=cap["z_fine"].cpu(), **kw_o)
        want.update({k: v for k, v in w2.items() if "fine" in k})
    bar_db, bar_rgb, bar_other = BF16_BARS[name] if precision == "bf16" else X3_BARS[name][:3]
    for k, v in want.items():
        if k.startswith("nof_"):
            continue
        ps, l2 = _psnr(res[k], v), _l2rel(res[k], v)
        print(f"{name} {k}: {precision} PSNR-equiv {ps:.1f} dB, l2-rel {l2:.2e}, max-rel {relerr(res[k], v):.2e}")
        if precision == "bf16x3":
            assert relerr(res[k], v) <= X3_BARS[name][3], (k, relerr(res[k], v))
        if k.startswith("rgb"):
            assert ps >= bar_db, (k, ps)
            assert l2 <= bar_rgb, (k, l2)
        else:
            assert l2 <= bar_other, (k, l2)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("name", sorted(n for n in RENDER_CASES if n not in BF16_BARS))
def test_bf16_every_case_shape(M, R, name, precision):
    """Structural screen of the bf16 kernels over EVERY fixture shape (S = 40 / 128, ragged ray groups, softplus,
    disparity sampling, no background, none / ind / dir extra blocks, muted and absent frequencies, flow head, bw-only
    and local chains, the test-time sigma-only coarse pass, N = 0): same keys and shapes as the reference, per-ray
    outputs within the generic bf16 band (rgb >= 36 dB, l2-rel <= 6e-2) -- a wrong sample-to-lane map or head is
    orders of magnitude outside it.  The tight per-case bars are test_bf16_hidden_gemms'.
    bf16x3 (three bf16 products per matrix product, fp32 accumulation and heads): held to the fp32 contract, 1e-4 max-rel
    on every per-ray output."""
    from moco_flow_amd import rendering
    c = dict(RENDER_CASES[name])
    g = load_golden(name)
    seed = int(g["meta_seed"])
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    rays = torch.from_numpy(g["in_rays"]).cuda()
    bg = torch.from_numpy(g["in_background"]).cuda() if c.get("bg", True) else None
    cap = {}
    try:
        rendering.set_precision(precision)
        with torch.no_grad():
            res = M.render_rays(rays, bg, embs, nerfs, _capture=cap, **kw)
    finally:
        rendering.set_precision("f32")
    want = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("out_")}
    assert list(res.keys()) == list(want.keys())
    cap_o = {}
    if c["M"] > 0 and c["n"] > 0:
        embs_o, nerfs_o, kw_o = build_case(R, c, seed)
        with torch.no_grad():
            w2 = R.render_rays(rays.cpu(), bg.cpu() if bg is not None else None, embs_o, nerfs_o,
                               _z_fine_override=cap["z_fine"].cpu(), _capture=cap_o, **kw_o)
        want.update({k: v for k, v in w2.items() if "fine" in k})
        # per-sample opacities of the fine pass, all but the last sample: its delta is 1e10 (rendering.py:158-160), i.e.
        # its alpha is a step function of sigma's SIGN, which a bf16 error of 0.3 % of the sigma range flips on the rays
        # whose far end sits near sigma = 0 -- and with it the ray's opacity by the whole remaining transmittance
        a_hip, a_ref = cap["alphas_fine"][:, :-1], cap_o["alphas_fine"][:, :-1]
        # (under NoF the bf16 chain moves the canonical points by ~1e-3, across which these fixtures' dense-regime
        #  densities change at unit scale: a structural bound only)
        assert _l2rel(a_hip, a_ref) <= (0.3 if c.get("nof", "none") != "none" else 6e-2), _l2rel(a_hip, a_ref)
    # (the two test-time fixtures draw fine networks with a third of space at sigma > 0: there that flip moves the
    #  per-ray outputs by 21-28 dB in bf16 -- also in a torch emulation of the rounding --, so only the per-sample
    #  check above applies to their fine pass)
    flip_prone = name.endswith("_fine_test")
    for k, v in want.items():
        if k.startswith("nof_"):
            assert res[k].dim() == 1           # data-dependent length: the mask is taken on bf16 alphas
            continue
        assert res[k].shape == v.shape, k
        if v.numel() == 0 or (flip_prone and k.endswith("_fine")):
            continue
        ps, l2 = _psnr(res[k], v), _l2rel(res[k], v)
        print(f"{name} [{precision}] {k}: PSNR-equiv {ps:.1f} dB, l2-rel {l2:.2e}, max-rel {relerr(res[k], v):.2e}")
        if precision == "bf16x3":
            # every matrix product as three bf16 products of (hi, lo) pairs: the fp32 contract itself (north_star: 1e-4
            # max-rel; measured <= 4.6e-5 over all fixtures, fp32 kernels <= 2e-5)
            assert relerr(res[k], v) <= TOL, (k, relerr(res[k], v))
            continue
        if k.startswith("rgb"):
            assert ps >= 36.0, (k, ps)
        assert l2 <= 6e-2, (k, l2)


def test_bf16_linear_frequency_table(M, R):
    """Embeddings whose frequencies are not 2^k (logscale=False, embedding.py:21) take the bf16 kernels' direct
    sin / cos path instead of the angle-doubling chains: checked against the oracle like the default tables."""
    from moco_flow_amd import rendering
    c = dict(RENDER_CASES["r_nerf_dir_dense"])
    seed, n = int(load_golden("r_nerf_dir_dense")["meta_seed"]), 300
    rays, bg = case_inputs(c, seed, n=n)
    outs = []
    for backend, dev in ((M, "cuda"), (R, "cpu")):
        embs, nerfs, kw = build_case(backend, c, seed, device=dev)
        embs[0], embs[2] = backend.Embedding(3, 10, False), backend.Embedding(3, 4, False)
        try:
            if backend is M:
                rendering.set_precision("bf16")
            with torch.no_grad():
                outs.append(backend.render_rays(rays.to(dev), bg.to(dev), embs, nerfs, **kw))
        finally:
            rendering.set_precision("f32")
    got, want = outs
    ps = _psnr(got["rgb_coarse"], want["rgb_coarse"])
    print(f"bf16, linear frequency tables: rgb PSNR-equiv {ps:.1f} dB")
    assert ps >= 45.0
    for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
        assert _l2rel(got[k], want[k]) <= 2e-2, (k, _l2rel(got[k], want[k]))


BENCH_TAGS = dict(coarse="nerf", fine="nerf_fine")      # the weight draw bench.py times (tags of synth.*_state)


def _full_size_case(M, R, name, n, precision, tags=None):
    """HIP path at a BASELINE size vs the oracle on the same seeded synthetic batch (bench.py's generator).
    ``tags``: which random weight draw (None = the golden cases' draw, BENCH_TAGS = bench.py's)."""
    from moco_flow_amd import rendering, synth
    c = dict(RENDER_CASES[name])
    seed = 0
    rays_np, bg_np = synth.rays(seed, n, chained=(c.get("nof") == "global"))
    rays, bg = torch.from_numpy(rays_np), torch.from_numpy(bg_np)
    embs_o, nerfs_o, kw_o = build_case(R, c, seed, tags=tags)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda", tags=tags)
    cap = {}
    try:
        rendering.set_precision(precision)
        with torch.no_grad():
            res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, _capture=cap, **kw)
    finally:
        rendering.set_precision("f32")
    with torch.no_grad():
        extra = dict(_z_fine_override=cap["z_fine"].cpu()) if c["M"] > 0 else {}
        want = R.render_rays(rays, bg, embs_o, nerfs_o, **extra, **kw_o)
    return c, res, want


@pytest.mark.parametrize("name", ["r_moco_local", "r_moco_global"])
def test_c3_full_size_fp32_vs_oracle(M, R, name):
    """BASELINE config C3's shape (4096 rays x 64 samples, bw NoF -> NeRF(ind) -> fw NoF, local and local+global
    chains) in fp32 against the oracle: 1e-4 max-rel on every per-ray output and on the consensus vectors."""
    c, res, want = _full_size_case(M, R, name, 4096, "f32")
    _check_result(res, {k: v.numpy() for k, v in want.items()}, c, None)
    for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
        print(f"C3 {name} fp32 {k}: max-rel {relerr(res[k], want[k]):.2e}")


# bf16 error of the MoCo chain depends strongly on the weight draw (the canonical point feeds sin(512 x)): two
# seeded draws, each with bars just under its measured values (r2: bench draw 51.8 dB, l2 4.7e-3 / 5.6e-3 / 2.2e-3;
# golden-case draw 38.1 dB, l2 3.0e-2 / 6.0e-2 / 3.1e-2).
C3_BF16_BARS = {"bench": (BENCH_TAGS, 49.0, 8e-3, 1.2e-2, 6e-3), "case": (None, 36.0, 4.5e-2, 9e-2, 5e-2)}


@pytest.mark.parametrize("draw", sorted(C3_BF16_BARS))
@pytest.mark.parametrize("name", ["r_moco_local", "r_moco_global"])
def test_c3_full_size_bf16_vs_oracle(M, R, name, draw):
    """BASELINE config C3 proper (4096 x 64, MoCo chain, bf16 hidden GEMMs) against the fp32 oracle: PSNR-equiv
    and l2-rel bars just under the measured values of each weight draw; the consensus distances
    (mean |x - fw(bw(x))|) agree to 2 % in the mean."""
    tags, bar_db, bar_rgb, bar_depth, bar_op = C3_BF16_BARS[draw]
    c, res, want = _full_size_case(M, R, name, 4096, "bf16", tags=tags)
    ps = _psnr(res["rgb_coarse"], want["rgb_coarse"])
    l2 = {k: _l2rel(res[k], want[k]) for k in ("rgb_coarse", "depth_coarse", "opacity_coarse")}
    print(f"C3 {name} bf16 [{draw} draw]: PSNR-equiv {ps:.1f} dB; l2-rel rgb {l2['rgb_coarse']:.2e} "
          f"depth {l2['depth_coarse']:.2e} opacity {l2['opacity_coarse']:.2e}")
    assert ps >= bar_db
    assert l2["rgb_coarse"] <= bar_rgb and l2["depth_coarse"] <= bar_depth and l2["opacity_coarse"] <= bar_op
    for k in want:
        if k.startswith("nof_"):
            assert abs(float(res[k].mean()) - float(want[k].mean())) <= 2e-2 * abs(float(want[k].mean())), k
            assert abs(res[k].shape[0] - want[k].shape[0]) <= 0.02 * want[k].shape[0], k


# bf16x3 (MF_PREC_BF16X3): the fp32 CONTRACT on the bf16 matrix pipe -- the NeRF's products as three bf16 products of (hi, lo)
# operand pairs, the NoF's (its output point feeds sin(512 x)) as three products of IEEE-half (hi, lo) pairs at 2^5 x their value
# (22 significand bits on v_mfma_f32_32x32x16_f16; rounds 3-4: six products of bf16 (hi, mid, lo) triples), fp32
# accumulation, heads on the fp32 accumulators, exact per-ray image-index bias, exact seeds + doubling chains for the encodings.
# Measured at 4096 rays (round 4): bench draw 114.1 dB, max-rel 1.0e-5 / 1.4e-5 / 1.0e-5 (rgb / depth / opacity; the fp32
# kernels: 123 dB); golden-case draw 111.3 dB, 2.1e-5 / 2.3e-5 / 3.1e-5 (fp32 kernels 118.7 dB).  History: round 3's kernel (two-term
# NoF, its xyz encoding from the transcendental unit) measured 104.4 dB on the bench draw but 88.6 dB / 7e-3 max-rel on the
# golden-case draw -- oracle/bf16_ref.py located both causes: the unit's 6e-6 rad argument error flipped the sign of a
# far-sample sigma (exact seeds: 99.1 dB, 1.4e-4 .. 2.0e-4), and the two-term split's 2^-17 is 1.5e-4 of the rendered ray (three
# terms: the numbers above = an exact-fp32 NoF's).  The fast mode on the same batches: 51.4 / 38.1 dB.
C3_X3_DRAWS = {"bench": BENCH_TAGS, "case": None}


@pytest.mark.parametrize("draw", sorted(C3_X3_DRAWS))
@pytest.mark.parametrize("name", ["r_moco_local", "r_moco_global"])
def test_c3_full_size_bf16x3_vs_oracle(M, R, name, draw):
    """BASELINE config C3 in the contract mode of the bf16 pipe (set_precision("bf16x3")) against the fp32 oracle: north_star's
    1e-4 max-rel on EVERY per-ray output, on both weight draws, local and local + global chains; >= 50 dB better than the fast
    bf16 mode on the same batch; the consensus means to 1e-4."""
    c, res, want = _full_size_case(M, R, name, 4096, "bf16x3", tags=C3_X3_DRAWS[draw])
    _, fast, _ = _full_size_case(M, R, name, 4096, "bf16", tags=C3_X3_DRAWS[draw])
    ps, ps_fast = _psnr(res["rgb_coarse"], want["rgb_coarse"]), _psnr(fast["rgb_coarse"], want["rgb_coarse"])
    mr = {k: relerr(res[k], want[k]) for k in ("rgb_coarse", "depth_coarse", "opacity_coarse")}
    print(f"C3 {name} bf16x3 [{draw} draw]: PSNR-equiv {ps:.1f} dB (fast bf16: {ps_fast:.1f}); max-rel rgb {mr['rgb_coarse']:.2e} "
          f"depth {mr['depth_coarse']:.2e} opacity {mr['opacity_coarse']:.2e}")
    assert ps >= 106.0 and ps >= ps_fast + 50.0
    for k, e in mr.items():
        assert e <= TOL, (k, e)
    for k in want:
        if k.startswith("nof_"):
            assert abs(float(res[k].mean()) - float(want[k].mean())) <= 1e-4 * abs(float(want[k].mean())), k
            assert abs(res[k].shape[0] - want[k].shape[0]) <= 0.002 * want[k].shape[0] + 2, k


def test_c5_shard_shape_bf16x3_vs_oracle(M, R):
    """BASELINE config C5's shard (1024 rays x (64 + 128), two NeRFs, local + global chains) in bf16x3, both passes against the
    oracle on identical samples: 1e-4 max-rel on every per-ray output of BOTH passes.  Measured (round 4): coarse 111.2 dB,
    max-rel <= 3.1e-5; fine 116.2 dB, <= 8.1e-6 (round 3: coarse 82.9 dB / 8e-3 -- the coarse NeRF of this fixture is the dense
    golden-case draw, see C3_X3_DRAWS; the fast mode 39.9 / 52.7 dB)."""
    c, res, want = _full_size_case(M, R, "r_moco_global_fine", 1024, "bf16x3")
    for k in ("rgb_coarse", "rgb_fine", "depth_coarse", "depth_fine", "opacity_coarse", "opacity_fine"):
        print(f"C5 shard bf16x3 {k}: PSNR-equiv {_psnr(res[k], want[k]):.1f} dB, l2-rel {_l2rel(res[k], want[k]):.2e}, "
              f"max-rel {relerr(res[k], want[k]):.2e}")
        assert relerr(res[k], want[k]) <= TOL, k
    assert _psnr(res["rgb_fine"], want["rgb_fine"]) >= 106.0 and _psnr(res["rgb_coarse"], want["rgb_coarse"]) >= 106.0


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_c5_shard_shape_vs_oracle(M, R, precision):
    """BASELINE config C5's single-GPU shard: 1024 rays x (64 coarse + 128 fine), two NeRFs, MoCo local+global
    chains in both passes.  The oracle is re-run on the HIP path's own fine depths (rendering.py:323 detaches
    them), so both passes are compared on identical samples: fp32 1e-4 max-rel; bf16 by PSNR-equiv / l2-rel
    (bars under the measured values; the fine pass is the worst case of the bf16 mode: 192 thin intervals
    behind two bf16 NoF evaluations)."""
    c, res, want = _full_size_case(M, R, "r_moco_global_fine", 1024, precision)
    assert res["rgb_fine"].shape == (1024, 3)
    if precision == "f32":
        _check_result({k: v for k, v in res.items() if "fine" in k},
                      {k: v.numpy() for k, v in want.items() if "fine" in k}, c, None)
        for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
            assert relerr(res[k], want[k]) <= TOL, (k, relerr(res[k], want[k]))
        return
    for k in ("rgb_coarse", "rgb_fine", "depth_coarse", "depth_fine", "opacity_coarse", "opacity_fine"):
        print(f"C5 shard bf16 {k}: PSNR-equiv {_psnr(res[k], want[k]):.1f} dB, l2-rel {_l2rel(res[k], want[k]):.2e}")
    for k in ("rgb_coarse", "rgb_fine"):
        assert _psnr(res[k], want[k]) >= C5_BF16_BARS[0] and _l2rel(res[k], want[k]) <= C5_BF16_BARS[1], k
    for k in ("depth_coarse", "depth_fine", "opacity_coarse", "opacity_fine"):
        assert _l2rel(res[k], want[k]) <= C5_BF16_BARS[2], k


# measured r2 (golden-case draw): rgb_coarse 40.2 dB / 2.4e-2, rgb_fine 53.0 dB / 3.2e-3, depth_coarse 3.9e-2,
# opacity_coarse 2.1e-2, depth_fine 9.0e-4
C5_BF16_BARS = (38.0, 3.2e-2, 5.5e-2)     # PSNR rgb, l2 rgb, l2 depth / opacity


def test_c4_shards_moco_bf16_bit_identical(M):
    """BASELINE config C4: the MoCo bf16 pass split over 8 contiguous ray shards (what the 8 ranks render)
    reproduces the single-launch result row for row, bit for bit -- per-ray outputs and the compacted consensus
    vectors (row-major order = global ray order)."""
    from moco_flow_amd import rendering, synth
    from moco_flow_amd.dist import shard_bounds
    n = 4096
    rays_np, bg_np = synth.rays(0, n, chained=True)
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    embs, nerfs, kw = build_case(M, dict(RENDER_CASES["r_moco_global"]), 0, device="cuda")
    try:
        rendering.set_precision("bf16")
        with torch.no_grad():
            a = M.render_rays(rays, bg, embs, nerfs, **kw)
            parts = [M.render_rays(rays[lo:hi], bg[lo:hi], embs, nerfs, **kw)
                     for lo, hi in (shard_bounds(n, r, 8) for r in range(8))]
    finally:
        rendering.set_precision("f32")
    assert "nof_global_disp_coarse" in a
    for k in a:
        assert torch.equal(a[k], torch.cat([p[k] for p in parts], 0)), k


@pytest.mark.parametrize("precision", ["f32", "bf16", "bf16x3"])
@pytest.mark.parametrize("name", ["r_nerf_dir_dense", "r_moco_local", "r_moco_global", "r_moco_global_fine"])
def test_repeat_runs_bit_identical(M, name, precision):
    """Race / hazard screen of the fused render kernels: the same call, repeated, returns the same bits in every
    plane.  (The weight stream is LDS-DMA under one barrier per panel with two wave roles; a bf16 build whose
    embedding used packed-fp32 VALU ops differed between runs in ~6 % of the rays -- csrc/Makefile.)"""
    from moco_flow_amd import rendering, synth
    c = dict(RENDER_CASES[name])
    n = 4096 if c["M"] == 0 else 1024
    rays_np, bg_np = synth.rays(0, n, chained=(c.get("nof") == "global"))
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    embs, nerfs, kw = build_case(M, c, 0, device="cuda")
    strict = rendering.STRICT_RNG
    try:
        rendering.STRICT_RNG = False
        rendering.set_precision(precision)
        with torch.no_grad():
            ref = M.render_rays(rays, bg, embs, nerfs, **kw)
            for rep in range(10):
                out = M.render_rays(rays, bg, embs, nerfs, **kw)
                for k in ref:
                    assert ref[k].shape == out[k].shape, (k, rep)
                    assert torch.equal(ref[k], out[k]), (k, rep, int((ref[k] != out[k]).sum()))
    finally:
        rendering.set_precision("f32")
        rendering.STRICT_RNG = strict


@pytest.mark.parametrize("name,n", [("r_nerf_dir_dense", 4096), ("r_moco_local", 4096), ("r_moco_global", 1000), ("r_moco_global_fine", 777)])
def test_two_block_fast_kernels_bit_identical_to_the_default(M, name, n):
    """Round 6: the fast bf16 mode's opt-in two-column-block kernels (csrc/mf_bf16_2b.hpp, MF_BF16_BLOCKS=2: 4 waves x 2 x 32
    samples, every LDS weight fragment feeding two MFMAs) evaluate the SAME arithmetic in the same order as the default
    8-wave kernels: every output plane bit-identical, ragged tails (n not a multiple of the 256-sample tile) included, and
    repeatable.  (They ship opt-in: measured at parity, profiles/r06_two_blocks.txt.)"""
    from moco_flow_amd import rendering, synth
    c = dict(RENDER_CASES[name])
    rays_np, bg_np = synth.rays(3, n, chained=(c.get("nof") == "global"))
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    embs, nerfs, kw = build_case(M, c, 0, device="cuda")
    strict, lazy, prev = rendering.STRICT_RNG, rendering.LAZY_CONSENSUS, os.environ.get("MF_BF16_BLOCKS")
    try:
        rendering.STRICT_RNG, rendering.LAZY_CONSENSUS = False, False
        rendering.set_precision("bf16")
        with torch.no_grad():
            os.environ["MF_BF16_BLOCKS"] = "1"
            ref = M.render_rays(rays, bg, embs, nerfs, **kw)
            os.environ["MF_BF16_BLOCKS"] = "2"
            for rep in range(3):
                out = M.render_rays(rays, bg, embs, nerfs, **kw)
                assert set(out) == set(ref)
                for k in ref:
                    assert ref[k].shape == out[k].shape, (k, rep)
                    assert torch.equal(ref[k], out[k]), (k, rep, int((ref[k] != out[k]).sum()))
    finally:
        if prev is None:
            os.environ.pop("MF_BF16_BLOCKS", None)
        else:
            os.environ["MF_BF16_BLOCKS"] = prev
        rendering.set_precision("f32")
        rendering.STRICT_RNG, rendering.LAZY_CONSENSUS = strict, lazy


def test_bf16x3_nof_weight_beyond_the_half_range_fails_loudly(M):
    """under bf16x3 the NoF's weights travel as IEEE-half (hi, lo) pairs of 32 w; |w| >= 2047 does not fit.  The packer
    used to saturate such a weight at +-65504 -- finite, WRONG results with no diagnostic.  Now it goes into the stream as a
    NaN: every ray whose chain touches it renders NaN (the same loud failure as an activation beyond the range), while the
    fp32 mode evaluates the same network normally."""
    from moco_flow_amd import rendering, synth
    c = dict(RENDER_CASES["r_moco_local"])
    rays_np, bg_np = synth.rays(0, 256)
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    embs, nerfs, kw = build_case(M, c, 0, device="cuda")
    bw = kw["nof_models"][0]
    with torch.no_grad():
        bw.nof_encoding_2[0].weight[5, 7] = 3000.0
    strict = rendering.STRICT_RNG
    try:
        rendering.STRICT_RNG = False
        with torch.no_grad():
            rendering.set_precision("f32")
            ok = M.render_rays(rays, bg, embs, nerfs, **kw)
            assert bool(torch.isfinite(ok["rgb_coarse"]).all())
            rendering.set_precision("bf16x3")
            bad = M.render_rays(rays, bg, embs, nerfs, **kw)
            assert bool(torch.isnan(bad["rgb_coarse"]).all())
    finally:
        rendering.set_precision("f32")
        rendering.STRICT_RNG = strict


@pytest.mark.parametrize("name", ["r_nerf_dir_fine_train", "r_moco_global_fine"])
def test_stochastic_branches_with_injected_draws(M, R, name):
    """perturb > 0 (stratified jitter, rendering.py:253-260), noise_std > 0 (:166) and the stochastic
    resample (u ~ rand, :30): the random tensors are drawn once here and injected into both the
    oracle and the HIP path, which then must agree to 1e-4 -- with random u there is no u = 1.0 hazard,
    so the fine pass is compared directly."""
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n, S, Mi = 96, c["S"], c["M"]
    rays, bg = case_inputs(c, seed, n=n)
    gen = torch.Generator().manual_seed(5)
    rng = dict(perturb_rand=torch.rand(n, S, generator=gen), noise_coarse=0.3 * torch.randn(n, S, generator=gen),
               noise_fine=0.3 * torch.randn(n, S + Mi, generator=gen), u=torch.rand(n, Mi, generator=gen))
    embs_o, nerfs_o, kw_o = build_case(R, c, seed)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    kw_o.update(perturb=1.0, noise_std=0.3)
    kw.update(perturb=1.0, noise_std=0.3)
    cap_o, cap = {}, {}
    with torch.no_grad():
        want = R.render_rays(rays, bg, embs_o, nerfs_o, _rng=rng, _capture=cap_o, **kw_o)
        res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, _rng={k: v.cuda() for k, v in rng.items()},
                            _capture=cap, **kw)
    assert relerr(cap["z_coarse"], cap_o["z_coarse"]) <= 1e-6
    # resample conditioning (a low-weight bin divides an O(ulp) cdf difference by ~1e-5): the drawn
    # depths agree to ~1e-3 of the range; the fine pass itself is then pinned on identical depths
    assert relerr(cap["z_fine"], cap_o["z_fine"]) <= 1e-3
    _check_result(res, {k: v.numpy() for k, v in want.items()}, c, None, fine_tol=5e-3)
    with torch.no_grad():
        want2 = R.render_rays(rays, bg, embs_o, nerfs_o, _rng=rng, _z_fine_override=cap["z_fine"].cpu(), **kw_o)
    _check_result({k: v for k, v in res.items() if "fine" in k},
                  {k: v.numpy() for k, v in want2.items() if "fine" in k}, c, None)


def test_fused_point_query(M, R):
    """mf_points_sigma == the trainer's five-call sequence (forward_nof -> embed -> pad -> NeRF sigma_only),
    checked against the oracle's restatement of trainer_moco_flow.py:146-187."""
    from moco_flow_amd import synth
    torch.manual_seed(1)
    B = 1000                                     # not a multiple of the 128-point tile
    xyz = (torch.rand(B, 3) * 3 - 1.5)
    sd_n = synth.nerf_state(41, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag="pts")
    sd_f = synth.nof_state(42, use_quat=True, tag="pts", head_scale=0.25)
    nerf = M.NeRF(8, 256, 63, [4], "ind", 5)
    nerf.load_state_dict({k: torch.from_numpy(v) for k, v in sd_n.items()})
    nof = M.NoF(4, 128, 33, [2], "ind", 33, True)
    nof.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    nerf, nof = nerf.cuda(), nof.cuda()
    ex, nx, ni = M.Embedding(3, 10), M.Embedding(3, 5), M.Embedding(1, 16)
    frame, num_frames = torch.tensor([17]), 300
    ind = float(frame.item()) * 2 / num_frames - 1.0
    with torch.no_grad():
        sig, canon = M.query_sigma(xyz.cuda(), nerf, ex, bw_nof=nof, nof_embeddings=[nx, ni], ind=ind,
                                   return_canonical=True)
        sig0 = M.query_sigma(xyz.cuda(), nerf, ex)
        onerf = R.NeRF(8, 256, 63, [4], "ind", 5, state=sd_n)
        onof = R.NoF(4, 128, 33, [2], "ind", 33, True, state=sd_f)
        ocanon = R.forward_nof_points(xyz, frame, num_frames, R.Embedding(3, 5), R.Embedding(1, 16), onof)
        osig = onerf(R.Embedding(3, 10)(ocanon), sigma_only=True)
        osig0 = onerf(R.Embedding(3, 10)(xyz), sigma_only=True)
    assert sig.shape == (B, 1) and canon.shape == (B, 3)
    assert relerr(canon, ocanon) <= TOL
    assert relerr(sig, osig) <= TOL
    assert relerr(sig0, osig0) <= TOL


def test_fused_point_query_bf16(M, R):
    """query_sigma(precision="bf16") (mf_points_sigma_p): the lattice / SMPL-point query on the bf16 core.  Not the 1e-4
    contract: l2-rel bars a factor ~2 over what the kernels measure against the fp32 oracle (raw sigma of the canonical query
    7.4e-3, through the backward NoF 2.0e-2 on these dense random weights, canonical point 9.4e-5), every launch shape (B = 1, ragged, > one
    tile per CU), per-point and scalar indices agreeing bit for bit, repeat runs bit-identical."""
    from moco_flow_amd import synth
    torch.manual_seed(1)
    sd_n = synth.nerf_state(41, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag="pts")
    sd_f = synth.nof_state(42, use_quat=True, tag="pts", head_scale=0.25)
    nerf = M.NeRF(8, 256, 63, [4], "ind", 5)
    nerf.load_state_dict({k: torch.from_numpy(v) for k, v in sd_n.items()})
    nof = M.NoF(4, 128, 33, [2], "ind", 33, True)
    nof.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    nerf, nof = nerf.cuda(), nof.cuda()
    ex, nx, ni = M.Embedding(3, 10), M.Embedding(3, 5), M.Embedding(1, 16)
    frame, num_frames = torch.tensor([17]), 300
    ind = float(frame.item()) * 2 / num_frames - 1.0
    onerf = R.NeRF(8, 256, 63, [4], "ind", 5, state=sd_n)
    onof = R.NoF(4, 128, 33, [2], "ind", 33, True, state=sd_f)
    for B in (1, 1000, 70000):
        xyz = (torch.rand(B, 3) * 3 - 1.5)
        with torch.no_grad():
            sig, canon = M.query_sigma(xyz.cuda(), nerf, ex, bw_nof=nof, nof_embeddings=[nx, ni], ind=ind,
                                       return_canonical=True, precision="bf16")
            sig_t = M.query_sigma(xyz.cuda(), nerf, ex, bw_nof=nof, nof_embeddings=[nx, ni],
                                  ind=torch.full((B,), ind), precision="bf16")
            sig0 = M.query_sigma(xyz.cuda(), nerf, ex, precision="bf16")
            sig0_again = M.query_sigma(xyz.cuda(), nerf, ex, precision="bf16")
            k = min(B, 2000)
            ocanon = R.forward_nof_points(xyz[:k], frame, num_frames, R.Embedding(3, 5), R.Embedding(1, 16), onof)
            osig = onerf(R.Embedding(3, 10)(ocanon), sigma_only=True)
            osig0 = onerf(R.Embedding(3, 10)(xyz[:k]), sigma_only=True)
        assert sig.shape == (B, 1) and canon.shape == (B, 3) and sig0.shape == (B, 1)
        assert torch.equal(sig, sig_t) and torch.equal(sig0, sig0_again)
        if B > 1:
            print(f"bf16 point query B={B}: l2-rel canonical {_l2rel(sig0[:k], osig0):.2e}, through bw NoF {_l2rel(sig[:k], osig):.2e}, "
                  f"canonical point {_l2rel(canon[:k], ocanon):.2e}")
            assert _l2rel(sig0[:k], osig0) <= 1.5e-2
            assert _l2rel(canon[:k], ocanon) <= 3e-4
            assert _l2rel(sig[:k], osig) <= 4e-2


def test_fused_point_query_bf16x3(M, R):
    """query_sigma(precision="bf16x3"): the lattice query on the three-product kernels (scalar image index or canonical
    space): raw sigma in canonical space, behind the NoF, and the canonical point to 1e-4 max-rel against the oracle (the fp32
    contract), at every launch shape; a per-point index tensor runs the exact-fp32 kernel; repeat runs bit-identical."""
    from moco_flow_amd import synth
    torch.manual_seed(1)
    sd_n = synth.nerf_state(41, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag="pts")
    sd_f = synth.nof_state(42, use_quat=True, tag="pts", head_scale=0.25)
    nerf = M.NeRF(8, 256, 63, [4], "ind", 5)
    nerf.load_state_dict({k: torch.from_numpy(v) for k, v in sd_n.items()})
    nof = M.NoF(4, 128, 33, [2], "ind", 33, True)
    nof.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    nerf, nof = nerf.cuda(), nof.cuda()
    ex, nx, ni = M.Embedding(3, 10), M.Embedding(3, 5), M.Embedding(1, 16)
    frame, num_frames = torch.tensor([17]), 300
    ind = float(frame.item()) * 2 / num_frames - 1.0
    onerf = R.NeRF(8, 256, 63, [4], "ind", 5, state=sd_n)
    onof = R.NoF(4, 128, 33, [2], "ind", 33, True, state=sd_f)
    for B in (1, 1000, 70000):
        xyz = (torch.rand(B, 3) * 3 - 1.5)
        with torch.no_grad():
            sig, canon = M.query_sigma(xyz.cuda(), nerf, ex, bw_nof=nof, nof_embeddings=[nx, ni], ind=ind,
                                       return_canonical=True, precision="bf16x3")
            sig_again = M.query_sigma(xyz.cuda(), nerf, ex, bw_nof=nof, nof_embeddings=[nx, ni], ind=ind, precision="bf16x3")
            sig_t = M.query_sigma(xyz.cuda(), nerf, ex, bw_nof=nof, nof_embeddings=[nx, ni],
                                  ind=torch.full((B,), ind), precision="bf16x3")
            sig0 = M.query_sigma(xyz.cuda(), nerf, ex, precision="bf16x3")
            k = min(B, 2000)
            ocanon = R.forward_nof_points(xyz[:k], frame, num_frames, R.Embedding(3, 5), R.Embedding(1, 16), onof)
            osig = onerf(R.Embedding(3, 10)(ocanon), sigma_only=True)
            osig0 = onerf(R.Embedding(3, 10)(xyz[:k]), sigma_only=True)
        assert sig.shape == (B, 1) and canon.shape == (B, 3) and sig0.shape == (B, 1)
        assert torch.equal(sig, sig_again)
        print(f"bf16x3 point query B={B}: max-rel canonical {relerr(sig0[:k], osig0):.2e}, through bw NoF {relerr(sig[:k], osig):.2e}, "
              f"canonical point {relerr(canon[:k], ocanon):.2e}; per-point index (fp32 kernel) {relerr(sig_t[:k], osig):.2e}")
        # (raw sigma behind the NoF on these dense random weights: the fp32 kernel measures 2.6e-5; round 3's two-term NoF
        #  1.3e-4 (bar 3e-4), the three-term NoF of round 4 holds the contract)
        assert relerr(sig0[:k], osig0) <= TOL and relerr(canon[:k], ocanon) <= TOL and relerr(sig[:k], osig) <= TOL
        assert relerr(sig_t[:k], osig) <= TOL


def test_make_rays_vs_golden(M):
    from moco_flow_amd import camera
    g = load_golden("u_camera")
    H, W = [int(v) for v in g["in_HW"]]
    K, c2w = g["in_K"], g["in_c2w"]
    near, far = camera.near_far_from_aabb(g["in_aabb_verts"], c2w)
    rays = camera.make_rays(H, W, K[0][0], (K[0][2], K[1][2]), c2w, near, far, float(g["in_idx"]))
    assert rays.shape == (H * W, 9) and rays.is_cuda
    assert relerr(rays, g["out_rays"]) <= 1e-6
    assert torch.equal(rays[:, 6:9].cpu(), torch.from_numpy(g["out_rays"])[:, 6:9])     # near / far / idx columns
    cam = camera.make_rays(H, W, K[0][0], (K[0][2], K[1][2]), None, 0.0, 1.0, 0.0)
    assert relerr(cam[:, 3:6], g["out_dirs_cam"]) <= 1e-6
    assert float(cam[:, :3].abs().max()) == 0.0
    # the generated rays feed the fused pass directly (no host round trip)
    embs, nerfs, kw = build_case(M, RENDER_CASES["r_nerf_dir_dense"], 5, device="cuda")
    with torch.no_grad():
        out = M.render_rays(rays, None, embs, nerfs, **kw)
    assert out["rgb_coarse"].shape == (H * W, 3)


def test_knn1_vs_oracle(M, R):
    from moco_flow_amd.knn import KNN
    torch.manual_seed(3)
    V, Q = 6890, 20000                      # SMPL vertex count, 2 x N_sampled of the joint stage
    ref = torch.randn(V, 3) * 0.5
    ref[100] = ref[37]                      # an exact duplicate: the first index must win
    query = torch.cat([torch.rand(Q // 2, 3) * 3 - 1.5, ref[torch.randint(V, (Q // 2,))] + 0.05 * torch.randn(Q // 2, 3)])
    query[5] = ref[100]
    d, i = KNN(k=1, transpose_mode=True)(ref[None].cuda(), query[None].cuda())
    assert d.shape == (1, Q, 1) and i.shape == (1, Q, 1) and i.dtype == torch.int64
    od, oi = R.knn1(ref, query)
    same = i[0].cpu() == oi
    # fp32 distances may tie-break differently from the fp64 oracle on near-equidistant pairs only
    assert float(same.float().mean()) > 0.9995
    assert int(i[0, 5, 0]) == 37
    bad = ~same.view(-1)
    if bool(bad.any()):
        alt = (query[bad] - ref[i[0].cpu().view(-1)[bad]]).norm(dim=1)
        assert relerr(alt, od.view(-1)[bad]) <= 1e-5
    assert relerr(d[0], od) <= 1e-5
    with pytest.raises(NotImplementedError):
        KNN(k=2, transpose_mode=True)


def test_training_loop_converges(M):
    """The drop-in trains: a few Adam steps of the stage-1 objective (MSELoss on rgb_coarse + rgb_fine,
    trainer_nerf.py:149-169) on a fixed synthetic target must reduce the loss, with parameters re-packed
    for the HIP kernels after every optimizer step."""
    from moco_flow_amd import synth
    torch.manual_seed(0)
    c = dict(RENDER_CASES["r_nerf_dir_fine_train"])
    c["M"] = 32
    embs, nerfs, kw = build_case(M, c, 11, device="cuda")
    rays, bg = case_inputs(c, 11, n=256)
    rays, bg = rays.cuda(), bg.cuda()
    target = torch.rand(256, 3, device="cuda") * 0.5 + 0.25
    params = [p for m in nerfs for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=5e-4)
    crit = M.get_loss(dict(type="MSE"))
    losses = []
    for it in range(12):
        res = M.render_rays(rays, bg, embs, nerfs, **kw)
        loss = crit(res, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.7 * losses[0], losses
    with torch.no_grad():                       # the re-packed weights are what inference now uses
        res = M.render_rays(rays, bg, embs, nerfs, **kw)
    assert float(crit(res, target)) < losses[0]


def _oracle_param_grads(model, out, gout, extra_inputs=()):
    """d <out, gout> / d (every parameter of the oracle model, extra inputs) by CPU autograd."""
    names = list(model.p)
    wrt = [model.p[k] for k in names] + list(extra_inputs)
    got = torch.autograd.grad(out, wrt, gout, allow_unused=True)
    return dict(zip(names, got[:len(names)])), got[len(names):]


def _with_grad(model):
    for k in model.p:
        model.p[k] = model.p[k].clone().requires_grad_(True)
    return model


def test_nerf_backward_vs_oracle_on_dumped_points(M, R):
    """the HIP backward of the NeRF (mf_nerf_backward + mf_weight_grads over the kernel's own
    activation dump) against the ORACLE's CPU autograd -- not the product's torch restatement -- evaluated on the
    points the kernel dumped (so the two sides differentiate the same function at the same place): every
    parameter gradient and the input-point gradient to 1e-4 max-rel."""
    from moco_flow_amd import autograd as A, rendering
    torch.manual_seed(0)
    c = dict(RENDER_CASES["r_nerf_ind_dense"])
    seed, n_rays, S = 21, 40, 64
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    embs_o, nerfs_o, _ = build_case(R, c, seed)
    nerf, onerf = nerfs[0], _with_grad(nerfs_o[0])
    rays, bg = case_inputs(c, seed, n=n_rays)
    rays_g = rays.cuda()
    t = torch.linspace(0, 1, S, device="cuda")
    z = (rays_g[:, 6:7] * (1 - t) + rays_g[:, 7:8] * t).contiguous()
    with torch.no_grad():
        p = rendering._render_pass(rays_g, bg.cuda(), z, None, False, None, 0, nerf, embs, None, None, False, False,
                                   False, True, dump=True)
    xin = p["xyz_in"].clone().requires_grad_(True)
    ind = rays_g[:, 8:9]
    with torch.no_grad():
        emb_in = pad_to(embs[0](p["xyz_in"]), 63)
        extra_in = pad_to(torch.repeat_interleave(embs[1](ind.contiguous()), S, dim=0), 5)
    gout = torch.randn(n_rays * S, 4, device="cuda")
    out = A.NerfSamples.apply(nerf, p["acts"], p["rgbsig"], emb_in, extra_in, embs[0], xin, False, *nerf.parameters())
    out.backward(gout)
    # oracle, CPU, on the kernel's dumped points
    x_o = p["xyz_in"].cpu().clone().requires_grad_(True)
    e_ind = torch.repeat_interleave(embs_o[1](rays[:, 8:9]), S, dim=0)
    inp = torch.cat([R._embed_padded(embs_o[0], x_o, 63), e_ind], -1)
    ref = onerf(inp)
    assert relerr(out, ref) <= 1e-5                                  # same function, same place
    want, (want_x,) = _oracle_param_grads(onerf, ref, gout.cpu(), (x_o,))
    worst = 0.0
    for name, q in nerf.named_parameters():
        e = relerr(q.grad, want[name])
        worst = max(worst, e)
        assert e <= 1e-4, (name, e)
    ex = relerr(xin.grad, want_x)
    print(f"NeRF backward vs oracle autograd on dumped points: worst parameter max-rel {worst:.2e}, d/d point {ex:.2e}")
    assert ex <= 1e-4


@pytest.mark.parametrize("extra,extra_dim,sigma_only,B", [("ind", 5, True, 1000), ("dir", 27, False, 700),
                                                           ("ind", 5, False, 129), ("none", 0, True, 1)])
def test_nerf_module_backward_vs_oracle(M, R, extra, extra_dim, sigma_only, B):
    """``NeRF(inputs, sigma_only)`` called directly with gradients (trainer_moco_flow.py:146-157, 337-362): the HIP
    node (autograd.NerfModule: mf_nerf_forward_dump + mf_nerf_backward_x + mf_weight_grads) against the ORACLE's CPU
    autograd of the same module call on the same embedded inputs: value 1e-5, every parameter gradient and the
    gradient of the inputs 1e-4; parameters off the sigma path get no gradient, as under torch."""
    from moco_flow_amd import synth
    torch.manual_seed(6)
    sd = {k: torch.from_numpy(v) for k, v in synth.nerf_state(11, extra_feat_type=extra, extra_feat_dim=extra_dim,
                                                              regime="dense").items()}
    nerf = M.NeRF(8, 256, 63, [4], extra, extra_dim)
    nerf.load_state_dict(sd)
    nerf = nerf.cuda()
    onerf = R.NeRF(8, 256, 63, [4], extra, extra_dim)
    onerf.load_state_dict(sd)
    onerf = _with_grad(onerf)
    exyz = M.Embedding(3, 10)
    width = 63 + (0 if sigma_only else extra_dim)
    with torch.no_grad():
        x0 = torch.cat([exyz(torch.randn(B, 3, device="cuda") * 0.7), torch.randn(B, extra_dim, device="cuda")], -1)[:, :width]
    x = x0.clone().requires_grad_(True)
    out = nerf(x, sigma_only=sigma_only)
    assert "NerfModule" in type(out.grad_fn).__name__
    gout = torch.randn_like(out)
    out.backward(gout)
    x_o = x0.cpu().clone().requires_grad_(True)
    ref = onerf(x_o, sigma_only=sigma_only)
    assert relerr(out, ref) <= 1e-5
    want, (want_x,) = _oracle_param_grads(onerf, ref, gout.cpu(), (x_o,))
    off_path = ("xyz_encoding_final", "extra_encoding", "rgb")
    for name, q in nerf.named_parameters():
        if sigma_only and name.startswith(off_path):
            assert q.grad is None and want[name] is None, name
            continue
        assert relerr(q.grad, want[name]) <= 1e-4, (name, relerr(q.grad, want[name]))
    assert relerr(x.grad, want_x) <= 1e-4, relerr(x.grad, want_x)
    # only the parameters require grad (the usual module-level training call)
    nerf.zero_grad(set_to_none=True)
    out2 = nerf(x0, sigma_only=sigma_only)
    assert torch.equal(out2, out.detach())
    out2.backward(gout)
    for name, q in nerf.named_parameters():
        if not (sigma_only and name.startswith(off_path)):
            assert relerr(q.grad, want[name]) <= 1e-4, name


def test_embedding_module_backward_vs_oracle(M, R):
    """Embedding(x) with x requiring grad: mf_embedding_forward / mf_embedding_backward vs the oracle's autograd,
    incl. muted frequencies (coarse-to-fine weights, trainer_moco_flow.py:113-114) and an empty batch."""
    torch.manual_seed(7)
    for cin, nf, w in ((3, 10, None), (3, 5, [1.0, 1.0, 0.4, 0.0, 0.0]), (1, 16, None)):
        e, eo = M.Embedding(cin, nf), R.Embedding(cin, nf)
        if w is not None:
            e.weights, eo.weights = list(w), list(w)
        x = (torch.randn(257, cin, device="cuda") * 0.5).requires_grad_(True)
        out = e(x)
        g = torch.randn_like(out)
        out.backward(g)
        xo = x.detach().cpu().requires_grad_(True)
        ref = eo(xo)
        ref.backward(g.cpu())
        assert relerr(out, ref) <= 2e-6
        assert relerr(x.grad, xo.grad) <= 1e-5, (cin, nf, relerr(x.grad, xo.grad))
    x = torch.zeros(0, 3, device="cuda", requires_grad=True)
    M.Embedding(3, 4)(x).sum().backward()
    assert x.grad.shape == (0, 3)


def test_embedding_rows_is_embed_repeat_pad(M):
    """Embedding.rows (mf_embedding_forward_rows): every input row embedded for `repeat` output rows at a wider, zero-padded
    row stride == Embedding.forward + repeat_interleave + pad, bit for bit (the operands the weight-gradient launches read)."""
    torch.manual_seed(3)
    for cin, nf, rep, width in ((3, 10, 1, 64), (1, 16, 7, 40), (3, 4, 5, 32), (3, 10, 3, None)):
        e = M.Embedding(cin, nf)
        x = torch.randn(131, cin + 2, device="cuda")[:, 1:1 + cin]          # a strided view, like rays[:, 8:9]
        got = e.rows(x, rep, width)
        want = torch.repeat_interleave(e(x.contiguous()), rep, dim=0)
        w = want.shape[1] if width is None else width
        assert got.shape == (131 * rep, w)
        assert torch.equal(got[:, :want.shape[1]], want) and not got[:, want.shape[1]:].any()
    assert M.Embedding(3, 4).rows(torch.zeros(0, 3, device="cuda"), 4, 32).shape == (0, 32)


@pytest.mark.parametrize("quat", [True, False])
def test_nof_backward_vs_oracle(M, R, quat, wgrad):
    """One NoF evaluation on points (rendering.py:49-83 + nof.py:69-82): HIP forward-with-dump / backward node
    (autograd.NofPoints) against the oracle's CPU autograd of nof_inference on the same points: output, every
    parameter gradient and d/d point to 1e-4 (quaternion head: kornia restated, like the forward)."""
    from moco_flow_amd import autograd as A
    torch.manual_seed(1)
    c = dict(RENDER_CASES["r_moco_global" if quat else "r_moco_global_flowhead"])
    seed, N, S = 5, 24, 40
    _, _, kw = build_case(M, c, seed, device="cuda")
    _, _, kw_o = build_case(R, c, seed)
    nof, onof = kw["nof_models"][1], _with_grad(kw_o["nof_models"][1])
    pts = (torch.randn(N, S, 3) * 0.7)
    ind = torch.rand(N, 1) * 2 - 1
    pts_g = pts.cuda().requires_grad_(True)
    out = A.nof_points(pts_g, ind.cuda(), kw["nof_embeddings"], nof)
    gout = torch.randn(N, S, 3)
    out.backward(gout.cuda())
    pts_o = pts.clone().requires_grad_(True)
    ref = R.nof_inference(pts_o, ind, kw_o["nof_embeddings"], onof)
    assert relerr(out, ref) <= 1e-5
    want, (want_x,) = _oracle_param_grads(onof, ref, gout, (pts_o,))
    worst = 0.0
    for name, q in nof.named_parameters():
        e = relerr(q.grad, want[name])
        worst = max(worst, e)
        assert e <= 1e-4, (name, e)
    ex = relerr(pts_g.grad, want_x)
    print(f"NoF(quat={quat}) backward vs oracle autograd: worst parameter max-rel {worst:.2e}, d/d point {ex:.2e}")
    assert ex <= 1e-4


@pytest.mark.parametrize("act,use_noise,use_bg", [("relu", False, True), ("softplus", True, True), ("relu", True, False)])
def test_composite_backward_vs_oracle(M, R, act, use_noise, use_bg):
    """mf_composite_backward (rendering.py:157-192 differentiated by hand) against the oracle's CPU autograd of
    its own composite() on the same per-sample planes: d/d(rgb, sigma) per sample to 1e-4."""
    from moco_flow_amd import autograd as A
    torch.manual_seed(2)
    N, S = 33, 64
    c = dict(RENDER_CASES["r_nerf_dir_dense"])
    rays, bg = case_inputs(c, 3, n=N)
    t = torch.linspace(0, 1, S)
    z = (rays[:, 6:7] * (1 - t) + rays[:, 7:8] * t).contiguous()
    rgbsig = torch.cat([torch.rand(N * S, 3), torch.randn(N * S, 1) * 2.0], -1)
    noise = torch.randn(N, S) * 0.5 if use_noise else None
    background = bg if use_bg else None
    x = rgbsig.cuda().requires_grad_(True)
    with torch.no_grad():
        ref0 = R.composite(rgbsig[:, 3].view(N, S), rgbsig[:, :3].view(N, S, 3), z, rays[:, 3:6],
                           noise if use_noise else torch.zeros(N, S), act, background)
    vals = [v.cuda() for v in (ref0[0], ref0[1], ref0[2].sum(1))]
    rgb, depth, opac = A.CompositeSamples.apply(x, rays.cuda(), z.cuda(), noise.cuda() if use_noise else None, act,
                                                background.cuda() if use_bg else None, *vals)
    g = [torch.randn(N, 3), torch.randn(N), torch.randn(N)]
    torch.autograd.backward([rgb, depth, opac], [t_.cuda() for t_ in g])
    xo = rgbsig.clone().requires_grad_(True)
    r_rgb, r_depth, r_w, _ = R.composite(xo[:, 3].view(N, S), xo[:, :3].view(N, S, 3), z, rays[:, 3:6],
                                         noise if use_noise else torch.zeros(N, S), act, background)
    (want,) = torch.autograd.grad([r_rgb, r_depth, r_w.sum(1)], [xo], g)
    e = relerr(x.grad, want)
    print(f"composite backward ({act}, noise={use_noise}, bg={use_bg}) vs oracle autograd: max-rel {e:.2e}")
    assert e <= 1e-4


@pytest.mark.parametrize("n_rays", [40, 37, 1])
def test_explicit_nerf_backward_unit(M, R, n_rays, wgrad):
    """autograd.NerfSamples (fused HIP dX chain mf_nerf_backward_x + mf_weight_grads over the kernel's activation dump)
    against the oracle's CPU autograd of the same network on the SAME points: parameter and input-point gradients to
    1e-4, frozen sub-module honoured, ragged sample counts (not a multiple of the 128-sample tile)."""
    from moco_flow_amd import autograd as A, rendering
    O = OracleOps(R)
    torch.manual_seed(0)
    c = dict(RENDER_CASES["r_nerf_ind_dense"])
    embs, nerfs, kw = build_case(M, c, 21, device="cuda")
    nerf = nerfs[0]
    rays, bg = case_inputs(c, 21, n=n_rays)
    rays, bg = rays.cuda(), bg.cuda()
    S = 64
    z = (rays[:, 6:7] * (1 - torch.linspace(0, 1, S, device="cuda")) + rays[:, 7:8] * torch.linspace(0, 1, S, device="cuda")).contiguous()
    with torch.no_grad():
        p = rendering._render_pass(rays, bg, z, None, False, None, 0, nerf, embs, None, None, False, False, False,
                                   True, dump=True)
        emb_in = pad_to(embs[0](p["xyz_in"]), 63)
        extra_in = pad_to(torch.repeat_interleave(embs[1](rays[:, 8:9].contiguous()), S, dim=0), 5)
    xin = p["xyz_in"].clone().requires_grad_(True)
    for q in nerf.rgb.parameters():
        q.requires_grad_(False)
    gout = torch.randn(n_rays * S, 4, device="cuda")
    out = A.NerfSamples.apply(nerf, p["acts"], p["rgbsig"], emb_in, extra_in, embs[0], xin, False, *nerf.parameters())
    out.backward(gout)
    x2 = p["xyz_in"].cpu().clone().requires_grad_(True)
    e_ind = O.pad_to(torch.repeat_interleave(O.embed(embs[1], rays[:, 8:9].cpu()), S, dim=0), 5)
    ref = O.nerf_forward(nerf, torch.cat([O.pad_to(O.embed(embs[0], x2), 63), e_ind], -1))
    assert relerr(out, ref) <= 1e-5                                  # the dump IS the forward
    ref.backward(gout.cpu())
    want = O.grads(nerf)
    for n, q in nerf.named_parameters():
        if n.startswith("rgb."):
            assert q.grad is None
        else:
            assert relerr(q.grad, want[n]) <= 1e-4, (n, relerr(q.grad, want[n]))
    assert relerr(xin.grad, x2.grad) <= 1e-4


@pytest.mark.parametrize("quat,n_rays,S", [(True, 40, 64), (False, 40, 64), (True, 3, 37), (True, 1, 1)])
def test_nof_points_backward_unit(M, R, quat, n_rays, S, wgrad):
    """autograd.NofPoints (mf_nof_points_dump + mf_nof_backward + mf_weight_grads) against the oracle's CPU autograd of
    the same evaluation (nof_inference: rendering.py:49-83 + nof.py:69-82) on the same points: output to 1e-5,
    parameter and point gradients to 1e-4; ragged sample counts."""
    from moco_flow_amd import autograd as A, synth
    O = OracleOps(R)
    torch.manual_seed(1)
    nof = M.NoF(4, 128, 33, [2], "ind", 33, quat)
    nof.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nof_state(5, use_quat=quat, tag="bw", head_scale=0.25).items()})
    nof = nof.cuda()
    embs = [M.Embedding(3, 5), M.Embedding(1, 16)]
    embs[0].weights = [1.0, 1.0, 0.5, 0.25, 0.0]            # coarse-to-fine ramp, trainer_moco_flow.py:270-305
    rays = torch.zeros(n_rays, 10, device="cuda")
    rays[:, 8] = torch.linspace(-0.9, 0.8, n_rays, device="cuda")
    ind = rays[:, 8:9]
    pts = (torch.randn(n_rays, S, 3, device="cuda") * 0.7).requires_grad_(True)
    gout = torch.randn(n_rays, S, 3, device="cuda")
    out = A.nof_points(pts, ind, embs, nof)
    assert "NofPoints" in type(out.grad_fn.next_functions[0][0]).__name__          # the HIP node, not the torch recompute
    out.backward(gout)
    p2 = pts.detach().cpu().clone().requires_grad_(True)
    ref = O.nof_points(p2, ind, embs, nof)
    assert relerr(out, ref) <= 1e-5
    ref.backward(gout.cpu())
    want = O.grads(nof)
    for n, q in nof.named_parameters():
        assert relerr(q.grad, want[n]) <= 1e-4, (n, relerr(q.grad, want[n]))
    assert relerr(pts.grad, p2.grad) <= 1e-4, relerr(pts.grad, p2.grad)


def test_render_image_vs_golden(M):
    """image.render_image (device-side chunk driver + mf_image_compose) against the fixture
    made from the reference's render_rays and the scatter-back of trainer_moco_flow.py:249-266: values to
    1e-4, the foreground / background / not-rendered classification of every pixel exactly, for chunked
    and single-pass rendering, with and without a mask."""
    import functools
    from test_oracle_golden import _image_case
    from moco_flow_amd import image
    g, embs, nerfs = _image_case(M, "cuda")
    rays, bg = torch.from_numpy(g["in_rays"]), torch.from_numpy(g["in_background"])       # host tensors, as the trainers hold them
    fwd = functools.partial(M.render_rays, nerf_embeddings=embs, nerf_models=nerfs, N_samples=int(g["in_S"]),
                            N_importance=int(g["in_M"]), perturb=0, noise_std=0)
    render = lambda r, b: fwd(r, b)
    with torch.no_grad():
        for n_rand in (int(g["in_N_rand"]), 100000):
            res = image.render_image(rays, bg, render, n_rand, g["in_rays_msk"])
            want_d = torch.from_numpy(g["out_depth_fine"])
            got_d = res["depth_fine"].cpu()
            for code in (8.0, 10.0):                                   # rendered-but-empty / not rendered pixels
                assert torch.equal(got_d == code, want_d == code)
            assert relerr(res["rgb_fine"], g["out_rgb_fine"]) <= 1e-4
            assert relerr(got_d, want_d) <= 1e-4
            assert relerr(res["opacity_fine"], g["out_opacity_fine"]) <= 1e-4
            assert relerr(res["rgb_coarse"], g["out_rgb_coarse"]) <= 1e-4
        full = image.render_image(rays.cuda(), bg.cuda(), render, 128, None)
        assert full["rgb_fine"].shape == (rays.shape[0], 3) and full["depth_fine"].shape == (rays.shape[0],)
        msk = torch.from_numpy(g["in_rays_msk"])
        fg = torch.from_numpy(g["out_opacity_fine"]) > 0
        sel = torch.nonzero(msk).squeeze(1)[fg]
        assert relerr(full["rgb_fine"].cpu()[sel], torch.from_numpy(g["out_rgb_fine"])[sel]) <= 1e-4


@pytest.mark.parametrize("act,use_noise,use_bg,N,S", [("relu", False, True, 33, 64), ("softplus", True, True, 9, 200),
                                                     ("relu", True, False, 5, 2), ("softplus", False, False, 3, 700)])
def test_composite_backward_unit(M, R, act, use_noise, use_bg, N, S):
    """autograd.CompositeSamples (mf_composite_backward) against the oracle's CPU autograd of the composite
    (cpu_ref.composite: rendering.py:157-192) on the same planes: dL/d[rgb, sigma] per sample to 1e-4,
    including the 1e10 last interval, multi-chunk rays (S > 64) and S = 2 (S = 1 is degenerate in the
    reference itself: `ones_like(deltas[:, :1])` of an empty tensor drops the only interval, rendering.py:158-160)."""
    from moco_flow_amd import autograd as A
    O = OracleOps(R)
    torch.manual_seed(3)
    dev = "cuda"
    rays = torch.randn(N, 9, device=dev)
    rays[:, 3:6] = torch.nn.functional.normalize(rays[:, 3:6], dim=1) * 1.3
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, dim=1).values.contiguous()
    rgbsig = torch.rand(N * S, 4, device=dev)
    rgbsig[:, 3] = torch.randn(N * S, device=dev) * 2.0
    noise = torch.randn(N, S, device=dev) * 0.3 if use_noise else None
    bg = torch.rand(N, 3, device=dev) if use_bg else None
    g = [torch.randn(N, 3, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)]
    a = rgbsig.cpu().clone().requires_grad_(True)
    comp = O.composite(a, z, rays[:, 3:6], noise, act, bg)
    torch.autograd.backward([comp["rgb"], comp["depth"], comp["opacity"]], [t.cpu() for t in g])
    b = rgbsig.clone().requires_grad_(True)
    outs = A.CompositeSamples.apply(b, rays, z, noise, act, bg, comp["rgb"].detach().cuda(), comp["depth"].detach().cuda(),
                                    comp["opacity"].detach().cuda())
    torch.autograd.backward(list(outs), g)
    assert relerr(b.grad[:, :3], a.grad[:, :3]) <= 1e-5
    assert relerr(b.grad[:, 3], a.grad[:, 3]) <= 1e-4, relerr(b.grad[:, 3], a.grad[:, 3])


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
@pytest.mark.parametrize("D,skips,extra,extra_dim", [(6, [2], "none", 0), (4, [], "dir", 27), (10, [3], "ind", 5), (8, [7], "dir", 27)])
def test_bf16_modes_other_network_shapes(M, R, D, skips, extra, extra_dim, precision):
    """The bf16 kernels' panel programs follow (D, skip position, extra block) at run time: depths 4-10, no skip / skip at
    the last layer, every extra input type, against the ORACLE's render of the same network on the same rays -- bf16x3 to
    the fp32 contract (1e-4 max-rel; measured <= 3e-6), the fast mode inside a generic bf16 band (rgb >= 30 dB)."""
    from moco_flow_amd import rendering, synth
    O = OracleOps(R)
    torch.manual_seed(11)
    nerf = M.NeRF(D, 256, 63, skips, extra, extra_dim).cuda()
    with torch.no_grad():
        nerf.sigma.weight.mul_(8.0)
    embs = [M.Embedding(3, 10), M.Embedding(1, 2) if extra == "ind" else None, M.Embedding(3, 4) if extra == "dir" else None]
    embs_o = [R.Embedding(3, 10), R.Embedding(1, 2) if extra == "ind" else None, R.Embedding(3, 4) if extra == "dir" else None]
    r, b = synth.rays(3, 333)
    rays, bg = torch.from_numpy(r), torch.from_numpy(b)
    kw = dict(N_samples=48, N_importance=0, use_disp=False, perturb=0, noise_std=0, nerf_activate_type="relu", test_time=False)
    try:
        rendering.set_precision(precision)
        with torch.no_grad():
            res = M.render_rays(rays.cuda(), bg.cuda(), embs, [nerf], **kw)
    finally:
        rendering.set_precision("f32")
    with torch.no_grad():
        want = R.render_rays(rays, bg, embs_o, [O.twin(nerf)], **kw)
    for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
        e = relerr(res[k], want[k])
        print(f"D={D} skips={skips} extra={extra} [{precision}] {k}: max-rel {e:.2e}, PSNR-equiv {_psnr(res[k], want[k]):.1f} dB")
        if precision == "bf16x3":
            assert e <= TOL, (k, e)
        else:
            # (default-initialised networks with an 8x sigma head: thin densities -- rgb measured 34.7-65 dB in the fast
            #  mode, while depth / opacity of nearly empty rays are all error: structural check on rgb only)
            if k.startswith("rgb"):
                assert _psnr(res[k], want[k]) >= 30.0, (k, _psnr(res[k], want[k]))


@pytest.mark.parametrize("D,skips,extra,extra_dim", [(6, [2], "none", 0), (4, [], "dir", 27), (10, [3], "ind", 5), (8, [7], "dir", 27)])
def test_nerf_backward_other_shapes(M, R, D, skips, extra, extra_dim):
    """The fused backward (mf_nerf_backward_x + mf_weight_grads) on network shapes other than the configs'
    8x256 / skip 4: depth, skip position (none / last layer) and extra input type; against the oracle's CPU autograd
    of the same network on the same points, 1e-4."""
    from moco_flow_amd import autograd as A, rendering
    O = OracleOps(R)
    torch.manual_seed(11)
    nerf = M.NeRF(D, 256, 63, skips, extra, extra_dim).cuda()
    with torch.no_grad():
        nerf.sigma.weight.mul_(8.0)
    embs = [M.Embedding(3, 10), M.Embedding(1, 2) if extra == "ind" else None, M.Embedding(3, 4) if extra == "dir" else None]
    n_rays, S = 21, 32
    from moco_flow_amd import synth
    r, b = synth.rays(3, n_rays)
    rays, bg = torch.from_numpy(r).cuda(), torch.from_numpy(b).cuda()
    z = (rays[:, 6:7] * (1 - torch.linspace(0, 1, S, device="cuda")) + rays[:, 7:8] * torch.linspace(0, 1, S, device="cuda")).contiguous()
    with torch.no_grad():
        p = rendering._render_pass(rays, bg, z, None, False, None, 0, nerf, embs, None, None, False, False, False, True, dump=True)
    xin = p["xyz_in"].clone().requires_grad_(True)
    with torch.no_grad():
        emb_in = pad_to(embs[0](p["xyz_in"]), 63)
        extra_in, extra_o = None, None
        if extra == "ind":
            extra_in = pad_to(torch.repeat_interleave(embs[1](rays[:, 8:9].contiguous()), S, dim=0), extra_dim)
            extra_o = O.pad_to(torch.repeat_interleave(O.embed(embs[1], rays[:, 8:9].cpu()), S, dim=0), extra_dim)
        elif extra == "dir":
            extra_in = pad_to(torch.repeat_interleave(embs[2](rays[:, 3:6].contiguous()), S, dim=0), extra_dim)
            extra_o = O.pad_to(torch.repeat_interleave(O.embed(embs[2], rays[:, 3:6].cpu()), S, dim=0), extra_dim)
    gout = torch.randn(n_rays * S, 4, device="cuda")
    out = A.NerfSamples.apply(nerf, p["acts"], p["rgbsig"], emb_in, extra_in, embs[0], xin, False, *nerf.parameters())
    out.backward(gout)
    x2 = p["xyz_in"].cpu().clone().requires_grad_(True)
    full = O.pad_to(O.embed(embs[0], x2), 63)
    ref = O.nerf_forward(nerf, full if extra_o is None else torch.cat([full, extra_o], -1))
    assert relerr(out, ref) <= 1e-5
    ref.backward(gout.cpu())
    want = O.grads(nerf)
    for n, q in nerf.named_parameters():
        assert relerr(q.grad, want[n]) <= 1e-4, (n, relerr(q.grad, want[n]))
    assert relerr(xin.grad, x2.grad) <= 1e-4


@pytest.mark.parametrize("D,skips,quat", [(3, [1], True), (5, [], False), (6, [5], True), (2, [], True)])
def test_nof_backward_other_shapes(M, R, D, skips, quat, wgrad):
    """NofPoints on depths / skip positions other than the configs' 4x128 / skip 2 (default nn.Linear init), against the
    oracle's CPU autograd."""
    from moco_flow_amd import autograd as A
    O = OracleOps(R)
    torch.manual_seed(5)
    nof = M.NoF(D, 128, 33, skips, "ind", 33, quat).cuda()
    embs = [M.Embedding(3, 5), M.Embedding(1, 16)]
    n_rays, S = 7, 50
    rays = torch.zeros(n_rays, 10, device="cuda")
    rays[:, 8] = torch.linspace(-0.7, 0.9, n_rays, device="cuda")
    pts = (torch.randn(n_rays, S, 3, device="cuda") * 0.6).requires_grad_(True)
    gout = torch.randn(n_rays, S, 3, device="cuda")
    assert A.nof_hip_supported(nof, embs)
    out = A.nof_points(pts, rays[:, 8:9], embs, nof)
    out.backward(gout)
    p2 = pts.detach().cpu().clone().requires_grad_(True)
    ref = O.nof_points(p2, rays[:, 8:9], embs, nof)
    assert relerr(out, ref) <= 1e-5
    ref.backward(gout.cpu())
    want = O.grads(nof)
    for n, q in nof.named_parameters():
        assert relerr(q.grad, want[n]) <= 1e-4, (n, relerr(q.grad, want[n]))
    assert relerr(pts.grad, p2.grad) <= 1e-4


@pytest.mark.parametrize("name", ["r_nerf_dir_fine_train", "r_moco_global_fine"])
def test_training_gradients_are_reproducible(M, name):
    """The HIP backward has no atomics (fixed-order partial sums in mf_weight_grads, scans in the
    composite): two identical training steps give bit-identical gradients, and scaling the loss by 4
    scales every gradient by exactly 4 (linearity of the backward, a size-independent property)."""
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 700
    rays, bg = case_inputs(c, seed, n=n)
    rays, bg = rays.cuda(), bg.cuda()
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    nets = list(nerfs) + (list(kw["nof_models"]) if kw["nof_models"] else [])
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(1)).cuda()

    def grads(scale):
        for m in nets:
            m.zero_grad(set_to_none=True)
        res = M.render_rays(rays, bg, embs, nerfs, **kw)
        loss = ((res["rgb_coarse"] - gt) ** 2).mean() + ((res["rgb_fine"] - gt) ** 2).mean()
        for k in res:
            if k.startswith("nof_"):
                loss = loss + 0.25 * res[k].mean()
        (loss * scale).backward()
        return [p.grad.clone() for m in nets for p in m.parameters()]

    a, b, c4 = grads(1.0), grads(1.0), grads(4.0)
    assert len(a) >= 24
    for x, y, z in zip(a, b, c4):
        assert torch.equal(x, y)
        assert torch.equal(4.0 * x, z)


@pytest.mark.parametrize("quat,B", [(True, 1000), (False, 333)])
def test_nof_module_training_call(M, R, quat, B, wgrad):
    """NoF(inputs, xyz) on data points with only the parameters requiring grad (trainer_nof.py:85-125, the
    stage-2 step; trainer_moco_flow.py:159-187): autograd.NofModule (mf_nof_forward_dump + mf_nof_backward
    + mf_weight_grads) against the oracle's CPU autograd of the same call, values 1e-5, gradients 1e-4."""
    from moco_flow_amd import synth
    O = OracleOps(R)
    torch.manual_seed(2)
    nof = M.NoF(4, 128, 33, [2], "ind", 33, quat)
    nof.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nof_state(9, use_quat=quat, tag="fw", head_scale=0.25).items()})
    nof = nof.cuda()
    exyz, eind = M.Embedding(3, 5), M.Embedding(1, 16)
    xyz = torch.randn(B, 3, device="cuda") * 0.6
    ind = torch.full((B, 1), 0.31, device="cuda")
    inputs = torch.cat([exyz(xyz), eind(ind)], -1)                  # (B, 66), as forward() of trainer_nof.py builds it
    target = torch.randn(B, 3, device="cuda")
    out = nof(inputs, xyz)
    assert "NofModule" in type(out.grad_fn).__name__
    torch.nn.functional.mse_loss(out, target).backward()
    ref = O.nof_forward(nof, inputs.detach().cpu(), xyz.cpu())
    assert relerr(out, ref) <= 1e-5
    torch.nn.functional.mse_loss(ref, target.cpu()).backward()
    want = O.grads(nof)
    for n, q in nof.named_parameters():
        assert relerr(q.grad, want[n]) <= 1e-4, (n, relerr(q.grad, want[n]))


def test_aux_point_losses_train(M, R):
    """The joint stage's point losses (trainer_moco_flow.py:330-363): outside points -> bw NoF (module call,
    HIP NofModule) -> xyz embedding -> NeRF(sigma_only) -> alphas -> mask loss.  Gradients reach the NoF through
    the NeRF's input; compared with the same chain through the oracle on the CPU."""
    from moco_flow_amd import synth
    O = OracleOps(R)
    torch.manual_seed(4)
    B = 1000
    load = lambda m, sd: (m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}), m.cuda())[1]
    nof = load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(3, use_quat=True, tag="bw", head_scale=0.25))
    nerf = load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(3, extra_feat_type="ind", extra_feat_dim=5, regime="dense"))
    exyz, eind, nxyz = M.Embedding(3, 5), M.Embedding(1, 16), M.Embedding(3, 10)
    pts = torch.randn(B, 3, device="cuda") * 0.5
    ind = torch.full((B, 1), 0.1, device="cuda")

    def chain(nof_call, emb_call, nerf_call, pts, ind):
        inp = torch.cat([emb_call(exyz, pts), emb_call(eind, ind)], -1)
        canon = nof_call(inp, pts)
        sig = nerf_call(emb_call(nxyz, canon))
        alphas = 1 - torch.exp(-(1.0 / 128) * torch.nn.functional.softplus(sig))
        return canon, (alphas ** 2).mean()

    canon, loss = chain(lambda i, x: nof(i, x), lambda e, x: e(x), lambda z: nerf(z, sigma_only=True), pts, ind)
    loss.backward()
    canon_t, loss_t = chain(lambda i, x: O.nof_forward(nof, i, x), lambda e, x: O.embed(e, x),
                            lambda z: O.nerf_forward(nerf, z, sigma_only=True), pts.cpu(), ind.cpu())
    loss_t.backward()
    assert relerr(canon, canon_t) <= 1e-5 and relerr(loss, loss_t) <= 1e-4
    want_nof, want_nerf = O.grads(nof), O.grads(nerf)
    # The NoF's gradient runs through sin(512 x) of the canonical point into a dense-regime NeRF: ill-conditioned in the
    # reference itself (see GRAD_CASES above: its own fp32 and fp64 autograd differ by tens of percent on such chains),
    # so two fp32 evaluation orders agree to ~1e-2 here (measured 7e-3 on nof_encoding_1.0.weight); the NeRF's own
    # parameters are well conditioned.  Each node is pinned at 1e-4 on identical inputs by test_*_backward_vs_oracle*.
    for n, q in nof.named_parameters():
        assert relerr(q.grad, want_nof[n]) <= 3e-2, (n, relerr(q.grad, want_nof[n]))
    for n, q in nerf.named_parameters():
        if want_nerf[n] is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
        else:
            assert relerr(q.grad, want_nerf[n]) <= 5e-3, (n, relerr(q.grad, want_nerf[n]))


@pytest.mark.parametrize("name", ["r_moco_global_fine", "r_moco_local", "r_nerf_dir_fine_train", "r_nerf_dir_dense"])
def test_fused_loss_partials_vs_trainer_formulas(M, R, name):
    """render_rays(..., _loss_target=gt): the 12 partials of mf_loss_partials (no compaction, no host sync) give the
    reference's loss terms -- MSELoss over both passes (models/losses.py:4-14) and mean(coarse) + mean(fine) of the
    consensus vectors (trainer_moco_flow.py:317-328) -- as computed op for op from the ORACLE's result dict; and the
    default call (with the vectors) agrees with the fast path."""
    from moco_flow_amd import dist as D, losses
    c = dict(RENDER_CASES[name])
    seed = int(load_golden(name)["meta_seed"])
    n = 96
    rays, bg = case_inputs(c, seed, n=n)
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(3))
    embs_o, nerfs_o, kw_o = build_case(R, c, seed)
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    cap = {}
    with torch.no_grad():
        fast = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, _loss_target=gt.cuda(), _capture=cap, **kw)
        full = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, **kw)
        extra = dict(_z_fine_override=cap["z_fine"].cpu()) if c["M"] > 0 else {}
        want = R.render_rays(rays, bg, embs_o, nerfs_o, **extra, **kw_o)
    assert not any(k.startswith("nof_") for k in fast) and fast["loss_partials"].shape == (12,)
    mse = torch.nn.MSELoss(reduction="mean")
    ref = {"img_loss": float(mse(want["rgb_coarse"], gt) + (mse(want["rgb_fine"], gt) if "rgb_fine" in want else 0.0))}
    for key in ("nof_local", "nof_global"):
        if f"{key}_disp_coarse" in want:
            ref[key] = float(torch.mean(want[f"{key}_disp_coarse"]) +
                             (torch.mean(want[f"{key}_disp_fine"]) if f"{key}_disp_fine" in want else 0.0))
    got = {k: float(v) for k, v in losses.from_partials(fast["loss_partials"]).items()}
    slow = D.reduce_loss(D.loss_partials(full, gt.cuda()))
    for k, v in ref.items():
        assert got[k] == pytest.approx(v, rel=2e-4), (k, got[k], v)            # a mask flip at alpha ~ 0.01 moves a mean by 1/n
        assert got[k] == pytest.approx(slow[k], rel=1e-6), k                     # same kernels, two routes
    for k in ("nof_local", "nof_global"):
        if k not in ref:
            assert got[k] == 0.0


def test_fused_loss_partials_train_without_sync(M):
    """Training through the fast path: loss from losses.from_partials(res["loss_partials"]) back-propagates the same
    gradients as the reference-shaped loss on the full result dict (MSE + 0.1 * consensus means)."""
    from moco_flow_amd import losses
    c = dict(RENDER_CASES["r_moco_global_fine"])
    seed = int(load_golden("r_moco_global_fine")["meta_seed"])
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    rays, bg = case_inputs(c, seed, n=24)
    rays, bg = rays.cuda(), bg.cuda()
    gt = torch.rand(24, 3, device="cuda")
    nets = list(nerfs) + list(kw["nof_models"])

    def grads(fast):
        for m in nets:
            m.zero_grad(set_to_none=True)
        if fast:
            res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)
            t = losses.from_partials(res["loss_partials"])
            loss = t["img_loss"] + 0.1 * (t["nof_local"] + t["nof_global"])
        else:
            res = M.render_rays(rays, bg, embs, nerfs, **kw)
            loss = M.get_loss(dict(type="MSE"))(res, gt)
            for key in ("nof_local_disp", "nof_global_disp"):
                loss = loss + 0.1 * (res[key + "_coarse"].mean() + res[key + "_fine"].mean())
        loss.backward()
        return float(loss), [p.grad.clone() for m in nets for p in m.parameters()]

    la, ga = grads(False)
    lb, gb = grads(True)
    assert lb == pytest.approx(la, rel=1e-5)
    for x, y in zip(ga, gb):
        assert relerr(y, x) <= 1e-4


def test_valid_rays_mask_bit_exact(M, R):
    """mf_valid_rays_mask (Camera.get_valid_rays_mask, utils/camera.py:119-132) against the brute-force oracle:
    bit-exact masks (index bookkeeping) for the fixture's camera, 40 random poses / boxes at a full image size, and
    degenerate hulls (point, segment, off-screen, partially clipped).  The projection is pinned to the reference's
    calculate_2d_projections output; hull + fill: parity unpinned vs cv2 (absent), rule stated in the header."""
    from moco_flow_amd import camera
    import moco_flow_amd._lib as L
    import ctypes as C
    g = load_golden("u_camera")
    H, W = [int(v) for v in g["in_HW"]]
    assert np.array_equal(camera.project_aabb(g["in_aabb_verts"], g["in_c2w"], g["in_K"]), g["out_projected_pixels"])
    got = camera.valid_rays_mask(g["in_aabb_verts"], g["in_c2w"], g["in_K"], (H, W))
    want = R.valid_rays_mask(g["out_projected_pixels"], H, W)
    assert got.dtype == torch.bool and got.shape == (H * W,) and np.array_equal(got.cpu().numpy(), want)
    rng = np.random.default_rng(0)
    H, W = 135, 240
    K = np.array([[180.0, 0, 120.0], [0, 180.0, 67.5], [0, 0, 1]])
    n_nonempty = 0
    for it in range(40):
        th, ph = rng.uniform(-0.9, 0.9), rng.uniform(-0.4, 0.4)
        Rz = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(ph), -np.sin(ph)], [0, np.sin(ph), np.cos(ph)]])
        c2w = np.eye(4)
        c2w[:3, :3] = Rz @ Rx
        c2w[:3, 3] = rng.uniform(-0.5, 0.5, 3) + np.array([0, 0, 2.8])
        lo = rng.uniform(-0.9, -0.2, 3)
        hi = rng.uniform(0.2, 0.9, 3)
        verts = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
        pix = camera.project_aabb(verts, c2w, K)
        got = camera.valid_rays_mask(verts, c2w, K, (H, W)).cpu().numpy()
        want = R.valid_rays_mask(pix, H, W)
        assert np.array_equal(got, want), it
        n_nonempty += int(want.any())
    assert n_nonempty >= 30

    def raw(points, H, W):
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.int32).reshape(-1, 2))
        out = torch.empty(H * W, dtype=torch.uint8, device="cuda")
        arr = (C.c_int32 * pts.size)(*pts.reshape(-1).tolist())
        L.check(L.lib().mf_valid_rays_mask(H, W, arr, pts.shape[0], out.data_ptr(), L.current_stream(out.device)), "mask")
        return out.cpu().numpy().astype(bool)

    for pts, H, W in ([[(1, 1)] * 8, 3, 3], [[(0, 0), (3, 3)], 4, 4], [[(-9, -9), (-5, -9), (-7, -3)], 4, 4],
                      [[(0, 0), (3, 0), (0, 2)], 3, 4], [[(-4, 2), (9, -3), (5, 12), (-2, 7)], 8, 6]):
        assert np.array_equal(raw(pts, H, W), R.valid_rays_mask(np.array(pts), H, W)), pts


@pytest.mark.gpu
def test_smpl_lbs_vs_reference_golden(M):
    """SMPL.forward / get_vertex_transformation / the correspondence transforms (utils/smpl/smpl_model.py:96-186,
    datasets/moco_flow_dataset.py:96-99,127-129) on the HIP kernels against the REFERENCE's outputs on the synthetic
    assets (tests/golden/u_smpl.npz): axis-angle poses incl. exactly-zero rotations, rotation-matrix poses, B = 3."""
    from moco_flow_amd import smpl as S, synth
    g = load_golden("u_smpl")
    m = S.SMPL(model=synth.smpl_model(int(g["meta_seed"]), int(g["meta_V"]))).cuda()
    pose, betas = torch.from_numpy(g["in_pose"]).cuda(), torch.from_numpy(g["in_betas"]).cuda()
    assert relerr(m(pose, betas), g["out_verts"]) <= 1e-5
    T = m.get_vertex_transformation(pose, betas)
    assert relerr(T, g["out_T"]) <= 1e-5
    assert relerr(m(torch.from_numpy(g["out_R"]).cuda(), betas), g["out_verts_from_R"]) <= 1e-5
    trans = S.frame_transforms(T[0], T[1])
    assert relerr(trans, g["out_trans"]) <= 1e-4
    ind = torch.from_numpy(g["out_ind"]).cuda()
    query = torch.from_numpy(g["in_query"]).cuda()
    assert relerr(S.apply_vertex_transforms(trans, ind, query), g["out_cano"]) <= 1e-4
    # the nearest-vertex search of the same pipeline reproduces the fixture's indices
    from moco_flow_amd.knn import KNN
    d, i = KNN(k=1, transpose_mode=True)(m(pose[:1], betas[:1]), query[None])
    assert torch.equal(i[0, :, 0].cpu(), torch.from_numpy(g["out_ind"]))
    assert relerr(d[0], g["out_dist"]) <= 1e-5


@pytest.mark.gpu
def test_smpl_full_size_vs_oracle(M):
    """The standard model size (6890 vertices, the shape the datasets run: B = 1 per frame) against oracle/smpl_ref.py,
    end to end through frame_correspondence; plus the empty cases."""
    from moco_flow_amd import smpl as S, synth
    from oracle import smpl_ref
    assets = synth.smpl_model(1, 6890)
    m, o = S.SMPL(model=assets).cuda(), smpl_ref.SMPL(assets)
    pose, betas = synth.smpl_pose(5, batch=2, scale=0.6)
    pose_t, betas_t = torch.from_numpy(pose), torch.from_numpy(betas)
    verts, T = m(pose_t.cuda(), betas_t.cuda()), m.get_vertex_transformation(pose_t.cuda(), betas_t.cuda())
    assert relerr(verts, o.forward(pose_t, betas_t)) <= 1e-5
    T_o = o.get_vertex_transformation(pose_t, betas_t)
    assert relerr(T, T_o) <= 1e-5
    Q = 20000                                        # 2 x num_sampled of c2f.yaml
    query = torch.from_numpy(((synth.uniform01(3, Q * 3).reshape(Q, 3) - 0.5) * 3.0).astype(np.float32))
    inside, outside = S.frame_correspondence(m, pose_t[:1].cuda(), betas_t[:1].cuda(), pose_t[1:].cuda(), betas_t[1:].cuda(),
                                             query.cuda(), thickness=0.2)
    src = o.forward(pose_t[:1], betas_t[:1])[0]
    d2 = torch.cdist(query.double(), src.double())
    dist, ind = d2.min(1)
    cano = smpl_ref.apply_vertex_transforms(smpl_ref.frame_transforms(T_o[0], T_o[1]), ind, query)
    ins_o, out_o = smpl_ref.split_inside_outside(query, cano, dist.float(), 0.2)
    assert inside.shape == ins_o.shape and outside.shape == out_o.shape and inside.shape[0] > 100
    assert relerr(inside, ins_o) <= 1e-4 and relerr(outside, out_o) <= 1e-4
    assert S.apply_vertex_transforms(T[0], torch.zeros(0, dtype=torch.int64, device="cuda"), torch.zeros(0, 3, device="cuda")).shape == (0, 3)
    assert m(pose_t[:0].cuda(), betas_t[:0].cuda()).shape == (0, 6890, 3)
    with pytest.raises(RuntimeError):
        m(pose_t[:, :10].cuda(), betas_t.cuda())


# (kept last in the file: a failed stream capture can leave the process unable to run further GPU work)
@pytest.mark.parametrize("mode", ["bf16_inference", "f32_training"])
def test_pass_and_training_step_replay_in_a_hip_graph(M, mode):
    """With the loss fast path nothing in a pass synchronises with the host: a gradient-free pass and a whole training step
    (HIP forward with dumps, mf_loss_partials, the HIP backward nodes) capture into a torch.cuda.CUDAGraph and replay
    bit for bit (outputs / every parameter gradient).  Guards against a host sync, a pageable copy or a launch on a
    foreign stream creeping into the path."""
    from moco_flow_amd import rendering, synth, losses
    c = dict(RENDER_CASES["r_moco_global_fine"])
    n = 512
    rays_np, bg_np = synth.rays(0, n, chained=True)
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    gt = torch.rand(n, 3, device="cuda")
    embs, nerfs, kw = build_case(M, c, 0, device="cuda")
    kw = dict(kw, perturb=0, noise_std=0)
    params = [p for m in list(nerfs) + list(kw["nof_models"]) for p in m.parameters()]
    train = mode == "f32_training"
    strict = rendering.STRICT_RNG

    def step():
        if train:
            res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)
            t = losses.from_partials(res["loss_partials"])
            (t["img_loss"] + 0.1 * (t["nof_local"] + t["nof_global"])).backward()
            return res
        with torch.no_grad():
            return M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)

    def clear():
        for p in params:
            p.grad = None

    try:
        rendering.STRICT_RNG = False
        rendering.set_precision("f32" if train else "bf16")
        clear()
        ref = step()
        ref = {k: v.detach().clone() for k, v in ref.items() if torch.is_tensor(v)}
        ref_g = [p.grad.clone() for p in params] if train else []
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                clear()
                step()
        torch.cuda.current_stream().wait_stream(s)
        clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        for k, v in ref.items():
            assert torch.equal(out[k].detach(), v), k
        for p, q in zip(params, ref_g):
            assert torch.equal(p.grad, q)
    finally:
        rendering.set_precision("f32")
        rendering.STRICT_RNG = strict
        clear()


@pytest.mark.parametrize("P", [1, 37, 5000, 70001])
def test_weight_grads_three_products_vs_float64(M, P):
    """mf_weight_grads_p(MF_PREC_BF16X3) alone: the NeRF's and the NoF's block shapes on column slices of strided dumps,
    ragged sample counts (not a multiple of the 16- / 32-sample step, fewer samples than workgroups), against a float64
    GEMM: the three-product blocks (round 5: every block but the heads' 4x640 -- 256x256, 128x256, 256x64 and 128x32 of the
    NeRF, 128x128, 128x80 and 12x128 of the NoF, the narrow ones as zero-padded 256x128 / 128x128 blocks) to 3e-5 l2-rel /
    1e-4 max-rel (measured 6e-6 .. 1.6e-5; the fp32 MFMA 4e-6), the heads block (fp32 either way) to 2e-5 and equal to the
    fp32 launch's; db = column sums to 1e-5; bit-identical between runs; nothing written outside a block's (rows, n_in)
    layout (the padded columns / rows of the wider three-product block are dropped by the reduction)."""
    from moco_flow_amd import autograd as A
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(7 + P)
    W, stride = 256, 9 * 256 + 128
    acts = torch.randn(P, stride, generator=g).to(dev)
    gpre = torch.randn(P, stride, generator=g).to(dev)
    emb = torch.randn(P, 64, generator=g).to(dev)
    ext = torch.randn(P, 32, generator=g).to(dev)
    ghead = torch.randn(P, 4, generator=g).to(dev)
    nstride = 4 * 128 + 16
    nacts = torch.randn(P, nstride, generator=g).to(dev)
    ngpre = torch.randn(P, nstride, generator=g).to(dev)
    emb80 = torch.randn(P, 80, generator=g).to(dev)
    sl = lambda t, l, w=W: t[:, l * W:l * W + w]
    nsl = lambda t, l, w=128: t[:, l * 128:l * 128 + w]
    jobs = [(sl(gpre, 1), sl(acts, 0), 256, 256, True), (sl(gpre, 5), sl(acts, 4), 256, 256, False),
            (sl(gpre, 9, 128), sl(acts, 8), 128, 256, True), (sl(gpre, 0), emb, 256, 64, True),
            (sl(gpre, 9, 128), ext, 128, 32, False), (ghead, acts[:, 7 * W:7 * W + 640], 4, 640, True),
            (nsl(ngpre, 1), nsl(nacts, 0), 128, 128, True), (nsl(ngpre, 0), emb80, 128, 80, True),
            (nsl(ngpre, 2), emb80, 128, 80, False), (ngpre[:, 512:524], nsl(nacts, 3), 12, 128, True)]
    old = A.WGRAD_PRECISION
    try:
        A.set_wgrad_precision("bf16x3")
        res = A.weight_grads(jobs, P, dev)
        res2 = A.weight_grads(jobs, P, dev)
        A.set_wgrad_precision("f32")
        ref32 = A.weight_grads(jobs, P, dev)
    finally:
        A.set_wgrad_precision(old)
    for (G, X, no, ni, b), (dW, db), (dW2, db2), (dWf, dbf) in zip(jobs, res, res2, ref32):
        assert torch.equal(dW, dW2)                                   # deterministic (fixed-order partial sums)
        assert dW.shape == dWf.shape and dW.shape[1] == ni            # the fp32 shape's layout under either arithmetic
        want = G.double().t() @ X.double()
        if ni == 640:        # the heads block: no head reads the `final` columns -- not fetched, dW there is 0 by contract
            want[:, 256:512] = 0
            assert not dW[:, 256:512].any()
        x3 = ni != 640
        l2 = float((dW[:no].double() - want).norm() / want.norm().clamp_min(1e-30))
        mr = float((dW[:no].double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
        print(f"P={P} {no}x{ni} {'x3' if x3 else 'f32'}: l2-rel {l2:.2e} max-rel {mr:.2e}")
        assert l2 <= (3e-5 if x3 else 2e-5) and mr <= 1e-4, (no, ni, l2, mr)
        if not x3:       # same arithmetic under either setting (only the split of the sample range over workgroups moves)
            assert float((dW - dWf).abs().max()) <= 2e-6 * float(dWf.abs().max().clamp_min(1e-30))
        if b:
            wb = G.double().sum(0)
            assert float((db[:no].double() - wb).abs().max() / wb.abs().max().clamp_min(1e-30)) <= 1e-5


@pytest.mark.parametrize("P", [1, 300, 20000])
def test_nerf_backward_three_products_vs_fp32_chain(M, P):
    """mf_nerf_backward3 alone: on the SAME activation dump and output gradients as the fp32 chain (mf_nerf_backward_x),
    every pre-activation gradient [d z_0 .. d z_{D-1} | d final | d extra] agrees to 1e-4 max-rel per layer block (measured
    ~1e-5: 16-bit operands, identical ReLU masks), ghead bit for bit (same VALU arithmetic), the embedded-input gradient
    to 1e-4, ragged sample counts, both head types of the extra block, run-to-run bit-identical."""
    from moco_flow_amd import autograd as A, synth
    dev = torch.device("cuda")
    for ext_type, ext_dim in (("dir", 27), ("ind", 5)):
        sd = synth.nerf_state(31, extra_feat_type=ext_type, extra_feat_dim=ext_dim, regime="dense", tag="bwd3")
        nerf = M.NeRF(8, 256, 63, [4], ext_type, ext_dim)
        nerf.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nerf = nerf.cuda()
        g = torch.Generator(device="cpu").manual_seed(11 + P)
        x = torch.randn(P, 63 + ext_dim, generator=g).to(dev)
        from moco_flow_amd import _lib as L
        import ctypes as C
        desc, buf = nerf.packed(L.MF_PREC_F32)
        stride = 9 * 256 + 128
        out = torch.empty(P, 4, device=dev)
        acts = torch.empty(P, stride, device=dev)
        L.check(L.lib().mf_nerf_forward_dump(C.byref(desc), buf.data_ptr(), x.data_ptr(), x.stride(0), P, out.data_ptr(),
                                             acts.data_ptr(), stride, L.current_stream(dev)), "mf_nerf_forward_dump")
        g_out = torch.randn(P, 4, generator=g).to(dev)
        old = A.DX_PRECISION
        try:
            A.set_dx_precision("f32")
            gp32, gh32, ge32 = A.nerf_backward_hip(nerf, g_out, acts, out, want_emb=True)
            A.set_dx_precision("bf16x3")
            gp3, gh3, _ = A.nerf_backward_hip(nerf, g_out, acts, out)
            gp3b, _, ge3 = A.nerf_backward_hip(nerf, g_out, acts, out, want_emb=True)
        finally:
            A.set_dx_precision(old)
        assert torch.equal(gp3, gp3b) and torch.equal(gh32, gh3)
        # the gradient of the embedded input (two more 64-row layers: W_0^T d_z_0 + W_skip^T d_z_skip), column 63 = 0
        assert relerr(ge3, ge32) <= TOL and float(ge3[:, 63].abs().max()) == 0.0, relerr(ge3, ge32)
        worst = 0.0
        for l in range(10):
            w = 128 if l == 9 else 256
            a, b = gp3[:, l * 256:l * 256 + w], gp32[:, l * 256:l * 256 + w]
            worst = max(worst, relerr(a, b))
            assert torch.equal(a == 0, b == 0) or l == 8            # same ReLU masks (xyz_encoding_final has none)
        print(f"P={P} extra={ext_type}: three-product dX chain vs the fp32 chain, worst max-rel per layer block {worst:.2e}")
        assert worst <= TOL


class _GuardedTorch:
    """`torch` as rendering.py sees it, with every device allocation of `empty` carved out of an arena that carries a sentinel
    band on either side (test_training_forward_dumps_stay_inside_their_tensors)."""
    GUARD = 2048                                            # elements on either side

    def __init__(self, real):
        self._real, self.arenas = real, []

    def __getattr__(self, k):
        return getattr(self._real, k)

    def empty(self, *shape, device=None, dtype=None, **kw):
        real = self._real
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, real.Size)):
            shape = tuple(shape[0])
        dt = dtype or real.float32
        n = int(np.prod(shape)) if len(shape) else 1
        if device is None or real.device(device).type != "cuda" or dt not in (real.float32, real.int32, real.uint8) or n == 0:
            return real.empty(shape, device=device, dtype=dtype, **kw)
        fill = 1234567.0 if dt == real.float32 else (0x5A5A5A5A if dt == real.int32 else 0x5A)
        arena = real.full((n + 2 * self.GUARD,), fill, device=device, dtype=dt)
        self.arenas.append((arena, n, fill))
        return arena[self.GUARD:self.GUARD + n].view(shape)

    def check(self):
        for arena, n, fill in self.arenas:
            lo, hi = arena[:self.GUARD], arena[self.GUARD + n:]
            assert bool((lo == fill).all()) and bool((hi == fill).all()), (tuple(arena.shape), n)
        return len(self.arenas)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_training_forward_dumps_stay_inside_their_tensors(M, prec):
    """The training forward writes its dump planes (activations, ReLU bit rows, per-step NoF rows, rgb / sigma, points) and its
    per-ray outputs INSIDE their tensors also where a tile is ragged: 7 rays x 64 samples (448 samples: three and a half 128-sample
    tiles, the last waves without a single row) with local + global chains, every output carved out of an arena with 8 KiB
    sentinel bands on both sides (round 5: the three-product kernel's row stores are buffer stores whose row-less lanes carry an
    out-of-range offset instead of an exec mask).  The valid part is fully written: no sentinel left inside the activations."""
    from moco_flow_amd import rendering, _lib as L
    c = dict(RENDER_CASES["r_moco_global"])
    seed = int(load_golden("r_moco_global")["meta_seed"])
    n, S = 7, c["S"]
    rays, bg = case_inputs(c, seed, n=n)
    rays, bg = rays.cuda(), bg.cuda()
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    t = torch.linspace(0, 1, S, device="cuda")
    z = (rays[:, 6:7] * (1 - t) + rays[:, 7:8] * t).contiguous()
    g = _GuardedTorch(torch)
    rendering.torch = g
    try:
        with torch.no_grad():
            out = rendering._render_pass(rays, bg, z, None, False, None, L.MF_ACT_RELU, nerfs[0], embs, kw["nof_models"],
                                         kw["nof_embeddings"], True, True, False, True, dump=True, precision=prec)
            torch.cuda.synchronize()
    finally:
        rendering.torch = torch
    assert g.check() >= 8                                    # outputs + dump planes all went through the guarded allocator
    acts = out["acts"]
    assert acts.shape[0] == n * S and not bool((acts == 1234567.0).any())     # every row of the valid part written
    assert bool(torch.isfinite(out["rgb"]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("P", [37, 4099])
def test_weight_grads_outputs_and_scratch_stay_inside(M, P):
    """mf_weight_grads_p(MF_PREC_BF16X3): the narrow blocks run on zero-padded 128 x 128 / 256 x 128 three-product blocks whose
    partials are wider than the results -- the reduction must write the (rows, n_in) layout of the fp32 shapes only, the partials
    must fit the planned scratch: every dW / db / scratch allocation of autograd.weight_grads inside sentinel bands."""
    from moco_flow_amd import autograd as A
    dev = torch.device("cuda")
    gen = torch.Generator(device="cpu").manual_seed(5 + P)
    rn = lambda *s: torch.randn(*s, generator=gen).to(dev)
    W, stride, ns = 256, 9 * 256 + 128, 4 * 128 + 16
    acts, gpre, emb, ext, ghead = rn(P, stride), rn(P, stride), rn(P, 64), rn(P, 32), rn(P, 4)
    nacts, ngpre, emb80 = rn(P, ns), rn(P, ns), rn(P, 80)
    sl = lambda t, l, w=W: t[:, l * W:l * W + w]
    nsl = lambda t, l, w=128: t[:, l * 128:l * 128 + w]
    jobs = [(sl(gpre, 1), sl(acts, 0), 256, 256, True), (sl(gpre, 9, 128), sl(acts, 8), 128, 256, True), (sl(gpre, 0), emb, 256, 64, True),
            (sl(gpre, 9, 128), ext, 128, 32, True), (ghead, acts[:, 7 * W:7 * W + 640], 4, 640, True),
            (nsl(ngpre, 1), nsl(nacts, 0), 128, 128, True), (nsl(ngpre, 0), emb80, 128, 80, True), (ngpre[:, 512:524], nsl(nacts, 3), 12, 128, True)]
    g = _GuardedTorch(torch)
    old = A.WGRAD_PRECISION
    A.torch = g
    try:
        A.set_wgrad_precision("bf16x3")
        res = A.weight_grads(jobs, P, dev)
        torch.cuda.synchronize()
    finally:
        A.torch = torch
        A.set_wgrad_precision(old)
    assert g.check() >= 2 * len(jobs) + 1 and any(a.dtype == torch.uint8 for a, _, _ in g.arenas)      # (+ the scratch buffer of the partials)
    for (G, X, no, ni, b), (dW, db) in zip(jobs, res):
        want = G.double().t() @ X.double()
        if ni == 640:
            want[:, 256:512] = 0
        assert float((dW[:no].double() - want).norm() / want.norm()) <= 3e-5 and not bool((dW == 1234567.0).any())


@pytest.mark.gpu
def test_training_step_frees_its_dumps_without_the_garbage_collector(M):
    """A training step through render_rays (local + global chains, coarse + fine) must not leave its dump planes to Python's
    CYCLIC collector: with gc disabled, the allocated device memory is back at its base after every step.  (Round 5: the
    consensus bookkeeping's callbacks closed over the pass object -- pass -> callback -> cell -> pass -- and 8.9 GB per joint-stage
    step stayed alive until a collection happened to run; the caching allocator then went to the driver for every step's dumps.)"""
    import gc
    c = dict(RENDER_CASES["r_moco_global_fine"])
    seed = int(load_golden("r_moco_global_fine")["meta_seed"])
    rays, bg = case_inputs(c, seed, n=64)
    rays, bg = rays.cuda(), bg.cuda()
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    nets = list(nerfs) + list(kw["nof_models"])

    def step():
        for m in nets:
            m.zero_grad(set_to_none=True)
        res = M.render_rays(rays, bg, embs, nerfs, **kw)
        loss = res["rgb_fine"].mean() + res["rgb_coarse"].mean()
        for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
            loss = loss + 0.1 * res[k].mean()
        loss.backward()
        del res, loss

    step(); step()
    torch.cuda.synchronize()
    gc.collect()
    base = torch.cuda.memory_allocated()
    gc.disable()
    try:
        for _ in range(3):
            step()
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated() <= base + (1 << 20), (torch.cuda.memory_allocated(), base)
    finally:
        gc.enable()
