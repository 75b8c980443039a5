"""Pin the oracle (oracle/cpu_ref.py) against every golden vector produced by the
reference itself (tests/golden/gen_golden.py). CPU only; tolerance 1e-6 relative
(same library, same op order)."""
import numpy as np
import pytest
import torch

from cases import RENDER_CASES
from helpers import build_case, case_inputs, load_golden, relerr
from oracle import cpu_ref as R

TOL = 1e-6


@pytest.mark.parametrize("name", sorted(RENDER_CASES))
def test_render_rays_matches_reference(name):
    c = RENDER_CASES[name]
    g = load_golden(name)
    seed = int(g["meta_seed"])
    embs, nerfs, kw = build_case(R, c, seed)
    rays = torch.from_numpy(g["in_rays"])
    bg = torch.from_numpy(g["in_background"]) if c.get("bg", True) else None
    cap = {}
    with torch.no_grad():
        res = R.render_rays(rays, bg, embs, nerfs, _capture=cap, **kw)
    want = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    assert sorted(res) == sorted(want)
    for k, v in want.items():
        assert tuple(res[k].shape) == v.shape, k
        assert res[k].dtype == torch.float32
        assert relerr(res[k], v) <= TOL, (k, relerr(res[k], v))
    for tag in ("coarse", "fine"):
        if f"mid_z_{tag}" in g:
            assert relerr(cap[f"z_{tag}"], g[f"mid_z_{tag}"]) <= TOL
            assert relerr(cap[f"weights_{tag}"], g[f"mid_weights_{tag}"]) <= TOL
            assert relerr(cap[f"alphas_{tag}"], g[f"mid_alphas_{tag}"]) <= TOL


def test_embedding_vectors():
    g = load_golden("u_embedding")
    x3, x1 = torch.from_numpy(g["in_x3"]), torch.from_numpy(g["in_x1"])
    for nf in (0, 2, 4, 5, 10, 16):
        assert torch.equal(R.Embedding(3, nf)(x3), torch.from_numpy(g[f"out_x3_f{nf}"]))
        assert torch.equal(R.Embedding(1, nf)(x1), torch.from_numpy(g[f"out_x1_f{nf}"]))
        assert R.Embedding(3, nf).out_channels == 3 * (2 * nf + 1)
    e = R.Embedding(3, 10)
    e.set_weights(0)
    assert torch.equal(e(x3), torch.from_numpy(g["out_x3_f10_w0"]))
    e.weights = list(g["in_ramp"])
    assert torch.equal(e(x3), torch.from_numpy(g["out_x3_f10_ramp"]))
    assert torch.equal(R.Embedding(3, 6, logscale=False)(x3), torch.from_numpy(g["out_x3_f6_linear"]))
    with pytest.raises(AssertionError):
        e.set_weights([1, 2])


def test_network_vectors():
    from moco_flow_amd import synth
    g = load_golden("u_networks")
    for extra, dim in (("dir", 27), ("ind", 5), ("none", 0)):
        sd = synth.nerf_state(11, extra_feat_type=extra, extra_feat_dim=dim, regime="dense", tag="unit")
        m = R.NeRF(8, 256, 63, [4], extra, dim, state=sd)
        inp = torch.from_numpy(g[f"in_nerf_{extra}"])
        assert relerr(m(inp), g[f"out_nerf_{extra}_full"]) <= TOL
        assert relerr(m(inp[:, :63].contiguous(), sigma_only=True), g[f"out_nerf_{extra}_sigma"]) <= TOL
    for quat in (True, False):
        m = R.NoF(4, 128, 33, [2], "ind", 33, quat, state=synth.nof_state(13, use_quat=quat, tag="unit"))
        out = m(torch.from_numpy(g["in_nof_inputs"]), torch.from_numpy(g["in_nof_xyz"]))
        assert relerr(out, g[f"out_nof_{'quat' if quat else 'flow'}"]) <= TOL


def test_sample_pdf_vectors_bit_exact_indices():
    g = load_golden("u_sample_pdf")
    bins, w = torch.from_numpy(g["in_bins"]), torch.from_numpy(g["in_weights"])
    full = R.sample_pdf_full(bins, w, 128, det=True)
    assert torch.equal(full["inds"], torch.from_numpy(g["mid_inds_det"]))       # int64, bit-exact
    assert torch.equal(full["cdf"], torch.from_numpy(g["mid_cdf"]))
    assert relerr(full["samples"], g["out_samples_det"]) <= TOL
    full = R.sample_pdf_full(bins, w, 128, det=False, u=torch.from_numpy(g["in_u_rand"]))
    assert torch.equal(full["inds"], torch.from_numpy(g["mid_inds_rand"]))
    assert relerr(full["samples"], g["out_samples_rand"]) <= TOL
    assert w.sum() == torch.from_numpy(g["in_weights"]).sum()                   # input not mutated


def test_trainer_glue_vectors():
    from moco_flow_amd import synth
    g = load_golden("u_trainer_glue")
    xyz = torch.from_numpy(g["in_xyz"])
    nerf = R.NeRF(8, 256, 63, [4], "ind", 5, state=synth.nerf_state(
        32, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag="glue"))
    a = R.forward_nerf_alpha(xyz, float(g["in_delta"]), R.Embedding(3, 10), nerf)
    assert relerr(a, g["out_alphas"]) <= TOL
    nof = R.NoF(4, 128, 33, [2], "ind", 33, True, state=synth.nof_state(33, use_quat=True, tag="glue"))
    out = R.forward_nof_points(xyz, torch.from_numpy(g["in_ind"]), int(g["in_num_frames"]),
                               R.Embedding(3, 5), R.Embedding(1, 16), nof)
    assert relerr(out, g["out_nof_xyz"]) <= TOL


def test_error_conventions():
    with pytest.raises(AssertionError):
        R.NeRF(extra_feat_type="bogus")
    with pytest.raises(AssertionError):
        R.NoF(extra_feat_type="dir")
    with pytest.raises(ValueError):
        R.composite(torch.zeros(1, 4), None, torch.arange(4.).view(1, 4), torch.ones(1, 3),
                    torch.zeros(1, 4), activate_type="tanh")
    with pytest.raises(NotImplementedError):
        from moco_flow_amd import synth
        sd = synth.nerf_state(1, D=2, W=32, in_channels_xyz=33, skips=(), extra_feat_type="latent_code",
                              extra_feat_dim=4)
        R.NeRF(2, 32, 33, [], "latent_code", 4, state=sd)(torch.zeros(2, 37))


def test_camera_vectors():
    g = load_golden("u_camera")
    H, W = [int(v) for v in g["in_HW"]]
    K, c2w = g["in_K"], g["in_c2w"]
    d = np.sqrt(np.sum((g["in_aabb_verts"] - c2w[:3, 3]) ** 2, axis=-1))
    rays = R.make_rays(H, W, [K[0][0], K[1][1]], [K[0][2], K[1][2]], c2w, min(d), max(d), float(g["in_idx"]))
    assert rays.shape == (H * W, 9)
    assert relerr(rays, g["out_rays"]) <= TOL
    assert torch.equal(R.gen_ray_directions(H, W, [K[0][0], K[1][1]], [K[0][2], K[1][2]]),
                       torch.from_numpy(g["out_directions"]))
    cam = R.make_rays(H, W, [K[0][0]], [K[0][2], K[1][2]], None, 0.0, 1.0, 0.0)
    assert relerr(cam[:, 3:6], g["out_dirs_cam"]) <= TOL


def test_valid_rays_mask_oracle():
    """Camera.get_valid_rays_mask (utils/camera.py:119-132).  The projection (calculate_2d_projections, :83-103) is
    pinned to the reference's own output; the hull + fill half is cv2 (absent): PARITY UNPINNED, the oracle restates
    OpenCV's FillConvexPoly -- outline by Line() (clipLine + 8-connected LineIterator), then 16.16 fixed-point scan-line
    spans up to, not including, the hull's last row -- checked here on hand-countable polygons."""
    g = load_golden("u_camera")
    pix = R.project_aabb(g["in_aabb_verts"], g["in_c2w"], g["in_K"])
    assert pix.dtype == np.int32 and np.array_equal(pix, g["out_projected_pixels"])
    H, W = [int(v) for v in g["in_HW"]]
    m = R.valid_rays_mask(pix, H, W).reshape(H, W)
    assert m.sum() == 120 and m[0, 34] and not m[0, 33] and m[:, -1].all()      # (115 without the clipped outline, round 2)
    # axis-aligned rectangle: closed on all four sides
    sq = R.valid_rays_mask(np.array([[2, 1], [5, 1], [5, 3], [2, 3], [3, 2]]), 6, 8).reshape(6, 8)
    assert sq.sum() == 12 and sq[1:4, 2:6].all()
    # right triangle (0,0) (4,0) (0,4): row y spans x = 0 .. round_half_up(4 - y)
    tri = R.valid_rays_mask(np.array([[0, 0], [4, 0], [0, 4]]), 5, 5).reshape(5, 5)
    assert [int(r.sum()) for r in tri] == [5, 4, 3, 2, 1]
    # slanted edge with half-pixel intersections: (0,0) (3,0) (0,2): row 1 ends at x = 1.5 -> rounds up to 2
    t2 = R.valid_rays_mask(np.array([[0, 0], [3, 0], [0, 2]]), 3, 4).reshape(3, 4)
    assert [int(r.sum()) for r in t2] == [4, 3, 1]
    # shallow top edge (0,0)-(9,1): the scan-line span of row 0 is the single pixel x = 0, the OUTLINE's Bresenham run adds
    # x = 1 .. 4 (m_k = (2k + 8) // 18 turns 1 at k = 5); the last row is the outline of the bottom edge alone
    sh = R.valid_rays_mask(np.array([[0, 0], [9, 1], [9, 3], [0, 3]]), 4, 10).reshape(4, 10)
    assert [int(r.sum()) for r in sh] == [5, 10, 10, 10] and sh[0, :5].all() and not sh[0, 5:].any()
    # degenerate inputs: a single point, a segment, everything off-screen
    assert R.valid_rays_mask(np.array([[1, 1]] * 8), 3, 3).sum() == 1
    assert R.valid_rays_mask(np.array([[0, 0], [3, 3]]), 4, 4).reshape(4, 4).diagonal().all()
    assert R.valid_rays_mask(np.array([[-9, -9], [-5, -9], [-7, -3]]), 4, 4).sum() == 0
    assert R.convex_hull_int([(0, 0), (2, 0), (1, 0), (2, 2), (0, 2), (1, 1)]) == [(0, 0), (2, 0), (2, 2), (0, 2)]


def test_knn1_semantics():
    ref = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 0, 0], [0, 2, 0]])
    q = torch.tensor([[0.9, 0, 0], [0, 0.9, 0], [0, 1.1, 0], [5, 5, 5]])
    d, i = R.knn1(ref, q)
    assert i.view(-1).tolist() == [1, 0, 3, 3]          # first minimum on the duplicated point
    assert relerr(d.view(-1), torch.tensor([0.1, 0.9, 0.9, (25 + 9 + 25) ** 0.5])) <= 1e-6


def _image_case(backend, device):
    """u_render_image's networks on ``backend`` (oracle or product): synth seeds + the fixture's sigma shift."""
    from moco_flow_amd import synth
    g = load_golden("u_render_image")
    nerfs = []
    for tag in ("coarse", "fine"):
        m = backend.NeRF(8, 256, 63, [4], "dir", 27)
        sd = synth.nerf_state(int(g["meta_seed"]), regime="default", tag=tag)
        sd["sigma.bias"] = (sd["sigma.bias"] + np.float32(float(g["in_sigma_shift"]))).astype(np.float32)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nerfs.append(m.to(device) if hasattr(m, "to") else m)
    embs = [backend.Embedding(3, 10, True), None, backend.Embedding(3, 4, True)]
    return g, embs, nerfs


def test_render_image_vectors():
    """oracle.render_image (trainer_moco_flow.py:226-268 restated) against the fixture made from the
    reference's render_rays + the same scatter-back lines: chunk loop with an empty last chunk, rays with
    opacity exactly 0, masked-out rays."""
    g, embs, nerfs = _image_case(R, "cpu")
    rays, bg = torch.from_numpy(g["in_rays"]), torch.from_numpy(g["in_background"])

    def fwd(r, b):
        with torch.no_grad():
            return R.render_rays(r, b, embs, nerfs, N_samples=int(g["in_S"]), N_importance=int(g["in_M"]), perturb=0, noise_std=0)

    res = R.render_image(rays, bg, fwd, int(g["in_N_rand"]), g["in_rays_msk"])
    for k in ("rgb_fine", "depth_fine", "opacity_fine", "rgb_coarse", "opacity_coarse"):
        assert relerr(res[k], g["out_" + k]) <= 1e-6, (k, relerr(res[k], g["out_" + k]))
    assert res["rgb_fine"].shape == (rays.shape[0], 3) and res["opacity_fine"].shape[0] == int(g["in_rays_msk"].sum())


def test_smpl_oracle_vs_reference_golden():
    """oracle/smpl_ref.py against tests/golden/u_smpl.npz = the REFERENCE's own SMPL.forward /
    get_vertex_transformation / correspondence lines run on moco_flow_amd.synth.smpl_model assets
    (utils/smpl/smpl_model.py:96-186, datasets/moco_flow_dataset.py:96-99,127-129)."""
    from moco_flow_amd import synth
    from oracle import smpl_ref
    g = load_golden("u_smpl")
    m = smpl_ref.SMPL(synth.smpl_model(int(g["meta_seed"]), int(g["meta_V"])))
    pose, betas = torch.from_numpy(g["in_pose"]), torch.from_numpy(g["in_betas"])
    assert relerr(smpl_ref.rodrigues(pose.view(-1, 3)).view(-1, 24, 3, 3), g["out_R"]) <= 1e-6
    assert relerr(m.forward(pose, betas), g["out_verts"]) <= 1e-6
    T = m.get_vertex_transformation(pose, betas)
    assert relerr(T, g["out_T"]) <= 1e-6
    assert relerr(m.forward(torch.from_numpy(g["out_R"]), betas), g["out_verts_from_R"]) <= 1e-6
    trans = smpl_ref.frame_transforms(T[0], T[1])
    assert relerr(trans, g["out_trans"]) <= 1e-5
    cano = smpl_ref.apply_vertex_transforms(trans, torch.from_numpy(g["out_ind"]), torch.from_numpy(g["in_query"]))
    assert relerr(cano, g["out_cano"]) <= 1e-5


def test_bf16_ref_hooks_off_is_cpu_ref():
    """oracle/bf16_ref.py -- the oracle with rounding hooks at the bf16 kernels' choices -- degenerates, hooks off, to
    the very calls of oracle/cpu_ref.py: torch.equal on whole render passes (MoCo coarse + fine with both chains, flow
    head, NeRF alone with dir / none), so it inherits cpu_ref's pinning to the reference-generated fixtures."""
    from oracle import bf16_ref as B
    for name in ("r_moco_global_fine", "r_moco_global_flowhead", "r_nerf_dir_dense", "r_nerf_none_dense", "r_moco_local_test"):
        c = RENDER_CASES[name]
        rays, bg = case_inputs(c, 0)
        embs, nerfs, kw = build_case(R, c, 0)
        want = R.render_rays(rays, bg, embs, nerfs, **kw)
        embs2, nerfs2, kw2 = build_case(B.Backend(B.F32), c, 0)
        got = R.render_rays(rays, bg, embs2, nerfs2, **kw2)
        assert list(got) == list(want)
        for k in want:
            assert torch.equal(got[k], want[k]), (name, k)


def test_bf16_ref_modes_sit_where_their_arithmetic_puts_them():
    """The hooks on: each mode's distance to the fp32 oracle on the MoCo fixture is the size of its operand rounding --
    bf16 operands 35-60 dB, three bf16 products >= 95 dB, float64 accumulation of fp32 operands >= 115 dB -- and every
    sin / cos variant of the encodings agrees with torch's to the accuracy the kernels rely on (transcendental-unit path
    <= 7.5e-8 of the angle, doubling chains <= 5e-6: four doublings of a seed's 1e-7)."""
    from oracle import bf16_ref as B
    c = RENDER_CASES["r_moco_global"]
    rays, bg = case_inputs(c, 0)
    embs, nerfs, kw = build_case(R, c, 0)
    want = R.render_rays(rays, bg, embs, nerfs, **kw)
    band = {"bf16": (35.0, 60.0), "bf16x3": (95.0, 125.0), "bf16x3_r3": (90.0, 125.0)}
    for name, (lo, hi) in band.items():
        e2, n2, kw2 = build_case(B.Backend(B.ARITH[name]), c, 0)
        got = R.render_rays(rays, bg, e2, n2, **kw2)
        ps = B.psnr_equiv(got["rgb_coarse"], want["rgb_coarse"])
        assert lo <= ps <= hi, (name, ps)
    from dataclasses import replace
    e2, n2, kw2 = build_case(B.Backend(replace(B.F32, acc="f64")), c, 0)
    got = R.render_rays(rays, bg, e2, n2, **kw2)
    assert B.psnr_equiv(got["rgb_coarse"], want["rgb_coarse"]) >= 115.0
    x = (torch.rand(2000, 3) - 0.5) * 12
    for nf in (10, 5, 4):
        ref = R.Embedding(3, nf)(x)
        for mode, tol in (("hw", 7.5e-8 * 2 ** (nf - 1) * 6 + 2e-7), ("chain", 5e-6)):
            e = B.Embedding(3, nf)
            e.mode = mode
            assert float((e(x) - ref).abs().max()) <= tol, (nf, mode, float((e(x) - ref).abs().max()))
