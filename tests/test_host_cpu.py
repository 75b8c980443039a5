"""CPU-side checks (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, the drop-in modules keep the reference's constructor / attribute / state_dict
contract, errors are loud, and the N > 1 sharding + reduction logic is right (2-rank gloo)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import moco_flow_amd._lib as L
    header = open(os.path.join(ROOT, "include", "mocoflow_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 14, names
    lib = L.lib()                                   # raises if the .so is missing
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in L.SYMBOLS, f"{n} has no ctypes prototype"
    assert lib.mf_version() == L.MF_ABI_VERSION == 16
    # argument validation is host-side and must not need a GPU
    d = L.mf_nerf_desc()
    d.D, d.W, d.in_channels_xyz, d.skip_mask = 8, 256, 63, 1 << 4
    d.extra_feat_type, d.extra_feat_dim = L.MF_EXTRA_DIR, 27
    assert lib.mf_nerf_packed_bytes(ctypes.byref(d)) > 2 * 1024 * 1024
    d.W = 192
    assert lib.mf_nerf_packed_bytes(ctypes.byref(d)) == 0
    assert b"unsupported" in lib.mf_last_error()
    n = L.mf_nof_desc()
    n.D, n.W, n.in_channels_xyz, n.extra_feat_dim, n.skip_mask, n.use_quat = 4, 128, 33, 33, 1 << 2, 1
    assert lib.mf_nof_packed_bytes(ctypes.byref(n)) > 0
    assert lib.mf_render_pass(None, None) == -1


def test_backward_entry_points_validate_on_the_host():
    """ABI v4-v7 (mf_nerf_backward, mf_nof_*, mf_weight_grads, mf_image_compose): sizes, the scheduler's
    scratch plan and argument validation are host-side and must not need a GPU."""
    import moco_flow_amd._lib as L
    lib = L.lib()
    d = L.mf_nerf_desc()
    d.D, d.W, d.in_channels_xyz, d.skip_mask = 8, 256, 63, 1 << 4
    d.extra_feat_type, d.extra_feat_dim = L.MF_EXTRA_DIR, 27
    # transposed stream: (W/2 -> W) + (W+sigma -> W) + 7 x (W -> W) panels, 1 KiB groups
    want = ((16 + 34 + 32 * 7) * 8 + 2 * 32 * 2) * 1024     # + the two embedded-input-gradient layers (64 rows each)
    got = lib.mf_nerf_bwd_packed_bytes(ctypes.byref(d))
    assert want < got <= want + 4096
    d.W = 128
    assert lib.mf_nerf_bwd_packed_bytes(ctypes.byref(d)) == 0 and b"unsupported" in lib.mf_last_error()
    assert lib.mf_nerf_backward(None, None, 0, None, None, 0, None, None, None, None) == -1
    n = L.mf_nof_desc()
    n.D, n.W, n.in_channels_xyz, n.extra_feat_dim, n.skip_mask, n.use_quat = 4, 128, 33, 33, 1 << 2, 1
    assert lib.mf_nof_bwd_packed_bytes(ctypes.byref(n)) > (3 * 4 * 16 + 2 * 32) * 1024
    n.skip_mask = (1 << 1) | (1 << 2)                      # two skip layers: not built
    assert lib.mf_nof_bwd_packed_bytes(ctypes.byref(n)) == 0
    assert lib.mf_nof_backward(None, None, None, 0, None, None, 0, None, None, None, None) == -1
    # weight-gradient plan: every supported block, partial slots bounded by the workgroup count
    items = (L.mf_wgrad_item * 3)()
    buf = (ctypes.c_float * 4096)()
    base = ctypes.addressof(buf) & ~15
    for it, (no, ni) in zip(items, [(256, 256), (128, 80), (4, 640)]):
        it.G, it.g_stride, it.n_out, it.X, it.x_stride, it.n_in, it.dW, it.db = base, 2432, no, base, 2432, ni, base, None
    P = 100000
    nbytes = lib.mf_weight_grads_scratch_bytes(items, 3, P)
    lo = (256 * 256 + 256 + 128 * 80 + 128 + 16 * 640 + 16) * 4
    assert lo <= nbytes <= 300 * lo
    assert lib.mf_weight_grads_scratch_bytes(items, 3, 0) >= 0
    items[1].n_in = 77                                       # unsupported block
    assert lib.mf_weight_grads_scratch_bytes(items, 3, P) == -1 and b"unsupported" in lib.mf_last_error()
    items[1].n_in, items[1].x_stride = 80, 2431             # stride not a multiple of 4 floats
    assert lib.mf_weight_grads_scratch_bytes(items, 3, P) == -1
    assert lib.mf_weight_grads(items, L.MF_WG_MAX_ITEMS + 1, P, None, None) < 0
    assert lib.mf_image_compose(None, None, 5, None, None, None, None, None, None, None) == -1
    assert lib.mf_image_compose(None, None, 0, None, None, None, None, None, None, None) == 0


def test_three_product_entry_points_validate_on_the_host():
    """ABI v13 (mf_weight_grads_p, mf_nerf_backward3 and its packer, MF_PREC_BF16X3 in the point query): sizes, scratch
    plans and argument validation are host-side and must not need a GPU."""
    import moco_flow_amd._lib as L
    lib = L.lib()
    d = L.mf_nerf_desc()
    d.D, d.W, d.in_channels_xyz, d.skip_mask = 8, 256, 63, 1 << 4
    d.extra_feat_type, d.extra_feat_dim = L.MF_EXTRA_DIR, 27
    # transposed (hi, lo) stream: 3 KiB resident + extra^T (K = 128: 8 tiles x 16 groups) + 8 layers x (8 x 32) + the two
    # 64-row layers of the embedded-input gradient (2 x 32 each)
    assert lib.mf_nerf_bwd3_packed_bytes(ctypes.byref(d)) == 3 * 1024 + (8 * 16 + 8 * 8 * 32 + 2 * 2 * 32) * 1024
    d.skip_mask = 0
    assert lib.mf_nerf_bwd3_packed_bytes(ctypes.byref(d)) == 3 * 1024 + (8 * 16 + 8 * 8 * 32 + 1 * 2 * 32) * 1024
    d.skip_mask = (1 << 2) | (1 << 4)          # two skip layers: the chain is built, the embedded-input gradient is not
    assert lib.mf_nerf_bwd3_packed_bytes(ctypes.byref(d)) == 3 * 1024 + (8 * 16 + 8 * 8 * 32) * 1024
    d.W = 128
    assert lib.mf_nerf_bwd3_packed_bytes(ctypes.byref(d)) == 0 and b"unsupported" in lib.mf_last_error()
    assert lib.mf_nerf_backward3(None, None, 0, None, None, 0, None, None, None, None, None, 0, None) == -1
    assert lib.mf_nerf_pack_bwd3(None, None, None) == -1
    # weight-gradient plan per precision: the three-product blocks are planned apart from the fp32 ones
    items = (L.mf_wgrad_item * 3)()
    buf = (ctypes.c_float * 4096)()
    base = ctypes.addressof(buf) & ~15
    for it, (no, ni) in zip(items, [(256, 256), (128, 256), (4, 640)]):
        it.G, it.g_stride, it.n_out, it.X, it.x_stride, it.n_in, it.dW, it.db = base, 2432, no, base, 2432, ni, base, None
    P = 100000
    lo = (256 * 256 + 256 + 128 * 256 + 128 + 16 * 640 + 16) * 4
    for prec in (L.MF_PREC_F32, L.MF_PREC_BF16X3):
        nbytes = lib.mf_weight_grads_scratch_bytes_p(prec, items, 3, P)
        assert lo <= nbytes <= 300 * lo, (prec, nbytes)
    assert lib.mf_weight_grads_scratch_bytes_p(L.MF_PREC_F32, items, 3, P) == lib.mf_weight_grads_scratch_bytes(items, 3, P)
    assert lib.mf_weight_grads_scratch_bytes_p(L.MF_PREC_BF16, items, 3, P) == -1          # not an arithmetic of this call
    assert lib.mf_weight_grads_p(L.MF_PREC_BF16, items, 3, P, None, None) < 0
    # round 5: under MF_PREC_BF16X3 the NoF's blocks and the NeRF's narrow ones are planned on the three-product shapes too -- the
    # 128 x 128 block for 128x128 / 128x80 / 12x128 / 128x32, the 256 x 128 block for 256x64: every partial has the padded block's
    # size (the fp32 plan: the narrow shapes' own), one partial per workgroup range that touches the item
    blk128, blk256 = (128 * 128 + 128) * 4, (256 * 128 + 256) * 4
    for (no, ni), xs, blk, f32blk in (((128, 128), 544, blk128, blk128), ((128, 80), 80, blk128, (128 * 80 + 128) * 4),
                                      ((12, 128), 544, blk128, (16 * 128 + 16) * 4), ((128, 32), 32, blk128, (128 * 32 + 128) * 4),
                                      ((256, 64), 64, blk256, (256 * 64 + 256) * 4)):
        one = (L.mf_wgrad_item * 1)()
        one[0].G, one[0].g_stride, one[0].n_out, one[0].X, one[0].x_stride, one[0].n_in, one[0].dW, one[0].db = base, 2432, no, base, xs, ni, base, None
        n3 = lib.mf_weight_grads_scratch_bytes_p(L.MF_PREC_BF16X3, one, 1, P)
        n32 = lib.mf_weight_grads_scratch_bytes_p(L.MF_PREC_F32, one, 1, P)
        assert n3 > 16 and (n3 - 16) % blk == 0 and (n32 - 16) % f32blk == 0, (no, ni, n3, n32)
        assert (n3 - 16) // blk == (n32 - 16) // f32blk                      # the same workgroup ranges: one item, one cost
        one[0].x_stride = xs + 1                                             # the three-product loads are 8 bytes wide
        assert lib.mf_weight_grads_scratch_bytes_p(L.MF_PREC_BF16X3, one, 1, P) == -1 and b"even" in lib.mf_last_error()
    # point query: the workspace (per-point / single NoF bias) exists for both bf16 arithmetics
    n = L.mf_nof_desc()
    n.D, n.W, n.in_channels_xyz, n.extra_feat_dim, n.skip_mask, n.use_quat = 4, 128, 33, 33, 1 << 2, 1
    assert lib.mf_points_sigma_workspace_bytes(L.MF_PREC_F32, ctypes.byref(n), 0, 1000) == 0
    assert lib.mf_points_sigma_workspace_bytes(L.MF_PREC_BF16, ctypes.byref(n), 0, 1000) == 2 * 128 * 4
    assert lib.mf_points_sigma_workspace_bytes(L.MF_PREC_BF16X3, ctypes.byref(n), 0, 1000) == 2 * 128 * 4
    assert lib.mf_points_sigma_workspace_bytes(L.MF_PREC_BF16, ctypes.byref(n), 1, 1000) == 1000 * 2 * 128 * 4


def test_embedding_rows_validates_on_the_host():
    """ABI v14 mf_embedding_forward_rows: row stride below the embedding's width, repeat < 1 and null buffers are refused
    before any launch; zero rows is a no-op."""
    import moco_flow_amd._lib as L
    lib = L.lib()
    e = L.mf_embedding()
    e.in_channels, e.n_freqs = 3, 10                       # width 63
    buf = (ctypes.c_float * 64)()
    ptr = ctypes.addressof(buf)
    assert lib.mf_embedding_forward_rows(ctypes.byref(e), None, 0, 1, None, 64, None) == 0
    assert lib.mf_embedding_forward_rows(ctypes.byref(e), ptr, 1, 1, ptr, 62, None) == -1 and b"out_stride" in lib.mf_last_error()
    assert lib.mf_embedding_forward_rows(ctypes.byref(e), ptr, 1, 0, ptr, 64, None) == -1
    assert lib.mf_embedding_forward_rows(ctypes.byref(e), None, 1, 1, ptr, 64, None) == -1
    assert lib.mf_embedding_forward_rows(None, ptr, 1, 1, ptr, 64, None) == -1


def test_packed_layout_sizes():
    """Packed sizes follow from the panel program: NeRF dir/27 = resident 13 KiB +
    (L0 8 + 3x32 + skip 40 + 3x32 + final 32) groups x 8 panels + extra 36 groups x 4 panels."""
    import moco_flow_amd._lib as L
    lib = L.lib()
    d = L.mf_nerf_desc()
    d.D, d.W, d.in_channels_xyz, d.skip_mask = 8, 256, 63, 1 << 4
    d.extra_feat_type, d.extra_feat_dim = L.MF_EXTRA_DIR, 27
    groups = (8 + 3 * 32 + 40 + 3 * 32 + 32) * 8 + 36 * 4
    assert lib.mf_nerf_packed_bytes(ctypes.byref(d)) == 13 * 1024 + groups * 1024
    d.extra_feat_type, d.extra_feat_dim = L.MF_EXTRA_IND, 5
    groups = (8 + 3 * 32 + 40 + 3 * 32 + 32) * 8 + 34 * 4
    assert lib.mf_nerf_packed_bytes(ctypes.byref(d)) == 13 * 1024 + groups * 1024
    n = L.mf_nof_desc()
    n.D, n.W, n.in_channels_xyz, n.extra_feat_dim, n.skip_mask, n.use_quat = 4, 128, 33, 33, 1 << 2, 1
    groups = (10 + 16 + 26 + 16) * 4
    assert lib.mf_nof_packed_bytes(ctypes.byref(n)) == 7 * 1024 + groups * 1024
    # MF_PREC_BF16 (mf_bf16.hpp): a panel is ONE 32-row tile, a group one A fragment of v_mfma_f32_32x32x16_bf16;
    # the NoF's matrix input is its xyz block only -- 3 embedded k-steps as (hi, lo) group pairs -- its head is one more
    # 16-group panel, and the fp32 image-index columns of its two embedded layers (2 x 128 rows x 36 floats = 36 KiB: the
    # per-ray bias is made from them) follow the panels; the NeRF's encodings are single groups (plain bf16 operands:
    # 4 k-steps of the xyz block, 2 of the direction block)
    assert lib.mf_nof_packed_bytes_p(ctypes.byref(n), L.MF_PREC_BF16) == 7 * 1024 + ((6 + 8 + 14 + 8) * 4 + 8) * 1024 + 36 * 1024      # (round 6: the head panel is 8 groups, its (hi, lo) terms are tile rows)
    d.extra_feat_type, d.extra_feat_dim = L.MF_EXTRA_DIR, 27
    # (round 6: + the sigma head panel, 16 groups in front of xyz_encoding_final, and the rgb head panel, 8 groups behind
    #  extra_encoding -- NetLayout::head_tiles: the fast mode's heads run on the matrix pipe)
    groups = (4 + 3 * 16 + 20 + 3 * 16 + 16) * 8 + (16 + 2) * 4 + 16 + 8
    assert lib.mf_nerf_packed_bytes_p(ctypes.byref(d), L.MF_PREC_BF16) == 13 * 1024 + groups * 1024
    # MF_PREC_BF16X3: the NeRF's k-steps as (hi, lo) group pairs -- twice the groups of the bf16 layout (its encodings split
    # too); the NoF's as IEEE-half (hi, lo) pairs (round 5: three products per k-step on the f16 matrix instruction, 22
    # significand bits -- its output point feeds sin(512 x); round 4 packed bf16 triples): 3 embedded k-steps x 2, 8 hidden
    # k-steps x 2, a 16-group head panel
    groups = (8 + 3 * 32 + 40 + 3 * 32 + 32) * 8 + (32 + 4) * 4
    assert lib.mf_nerf_packed_bytes_p(ctypes.byref(d), L.MF_PREC_BF16X3) == 13 * 1024 + groups * 1024
    assert lib.mf_nof_packed_bytes_p(ctypes.byref(n), L.MF_PREC_BF16X3) == 7 * 1024 + ((6 + 16 + 22 + 16) * 4 + 16) * 1024 + 36 * 1024
    # the envelope is the reference's constructors, not only its YAMLs: NeRF() defaults to in_channels_xyz = 33 and no
    # extra block (models/nerf.py:6-12), NoF() to extra_feat_dim = 0 (models/nof.py:7-15) -- narrower input blocks pack
    # into the same slots (same sizes); wider ones are refused
    d2 = L.mf_nerf_desc()
    d2.D, d2.W, d2.in_channels_xyz, d2.skip_mask = 8, 256, 33, 1 << 4
    d2.extra_feat_type, d2.extra_feat_dim = L.MF_EXTRA_NONE, 0
    groups = (8 + 3 * 32 + 40 + 3 * 32 + 32) * 8 + 32 * 4
    assert lib.mf_nerf_packed_bytes(ctypes.byref(d2)) == 13 * 1024 + groups * 1024
    d2.in_channels_xyz = 65
    assert lib.mf_nerf_packed_bytes(ctypes.byref(d2)) == 0
    n2 = L.mf_nof_desc()
    n2.D, n2.W, n2.in_channels_xyz, n2.extra_feat_dim, n2.skip_mask, n2.use_quat = 4, 128, 33, 0, 0, 0
    assert lib.mf_nof_packed_bytes(ctypes.byref(n2)) == 4 * 1024 + (10 + 3 * 16) * 4 * 1024      # (flow head: 3 rows resident)
    n2.in_channels_xyz, n2.extra_feat_dim = 21, 17
    assert lib.mf_nof_packed_bytes_p(ctypes.byref(n2), L.MF_PREC_BF16) > 0
    n2.extra_feat_dim = 34
    assert lib.mf_nof_packed_bytes(ctypes.byref(n2)) == 0
    assert lib.mf_loss_partials_scratch_bytes() == 256 * 12 * 8
    assert lib.mf_loss_partials(None, None, None, 0, None, None, None, None) == -1


def test_modules_keep_reference_contract():
    import moco_flow_amd as M
    from moco_flow_amd import synth
    nerf = M.get_model(dict(type="NeRF", D=8, W=256, in_channels_xyz=63, skips=[4], extra_feat_type="dir",
                            extra_feat_dim=27))
    want = synth.nerf_state(0)
    sd = nerf.state_dict()
    assert list(sd.keys()) == list(want.keys())              # same keys, same order as the reference
    for k, v in want.items():
        assert tuple(sd[k].shape) == v.shape, k
    assert sum(p.numel() for p in nerf.parameters()) == 595844
    for attr in ("in_channels_xyz", "extra_feat_type", "extra_feat_dim", "D", "W", "skips"):
        assert hasattr(nerf, attr)
    for sub in ("rgb", "xyz_encoding_final", "extra_encoding", "sigma"):       # trainer_moco_flow.py:395-401
        assert isinstance(getattr(nerf, sub), torch.nn.Module)
    nof = M.get_model(dict(type="NoF", D=4, W=128, in_channels_xyz=33, skips=[2], extra_feat_type="ind",
                           extra_feat_dim=33, use_quat=True))
    want = synth.nof_state(0)
    assert list(nof.state_dict().keys()) == list(want.keys())
    assert sum(p.numel() for p in nof.parameters()) == 67721
    assert M.NoF(4, 128, 33, [2], "ind", 33, False).nof_encoding_final.out_features == 3
    emb = M.get_model(dict(type="Embedding", in_channels=3, N_freqs=10, logscale=True))
    assert emb.out_channels == 63 and len(emb.state_dict()) == 0 and emb.weights == [1] * 10
    emb.set_weights(0)
    assert emb.weights == [0] * 10
    with pytest.raises(AssertionError):
        emb.set_weights([1, 2, 3])
    assert torch.equal(emb.freq_bands, 2 ** torch.linspace(0, 9, 10))
    assert torch.equal(M.Embedding(3, 6, False).freq_bands, torch.linspace(1, 32, 6))
    # stage hand-off filter of trainer_moco_flow.py:54-55 keeps only xyz / sigma keys
    kept = [k for k in nerf.state_dict() if "xyz" in k or "sigma" in k]
    assert len(kept) == 20
    with pytest.raises(ValueError):
        M.get_model(dict(type="Bogus"))
    with pytest.raises(ValueError):
        M.get_loss(dict(type="Bogus"))
    assert isinstance(M.get_loss(dict(type="MSE")), M.MSELoss)
    with pytest.raises(AssertionError):
        M.NeRF(extra_feat_type="bogus")
    with pytest.raises(AssertionError):
        M.NoF(extra_feat_type="dir")


def test_no_cpu_fallback_and_loud_errors():
    import moco_flow_amd as M
    nerf = M.NeRF(8, 256, 63, [4], "dir", 27)
    with pytest.raises(RuntimeError, match="no CPU"):
        nerf(torch.zeros(4, 90))
    with pytest.raises(RuntimeError, match="no CPU"):
        M.Embedding(3, 4)(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU"):
        M.render_rays(torch.zeros(4, 9), None, [M.Embedding(3, 10), None, M.Embedding(3, 4)], [nerf])
    with pytest.raises(NotImplementedError):
        M.NeRF(2, 256, 63, [], "latent_code", 4)._build_desc()
    # the product never imports the oracle
    src = ""
    pkg = os.path.join(ROOT, "moco_flow_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src += open(os.path.join(pkg, f)).read()
    import re as _re
    assert not _re.search(r"^\s*(from|import)\s+oracle", src, flags=_re.M)
    assert "cpu_ref" not in src


def test_shard_bounds_cover_and_order():
    from moco_flow_amd.dist import shard_bounds
    for n in (0, 1, 7, 4096, 32768, 1000003):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert lo == prev and hi >= lo and hi - lo in (n // world, n // world + 1)
                prev = hi
            assert prev == n
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _gloo_worker(rank, world, port, n, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from moco_flow_amd import dist as D, synth
    from oracle import cpu_ref as R
    sd = synth.nerf_state(3, regime="dense")
    rays, bg = synth.rays(3, n)
    rays, bg = torch.from_numpy(rays), torch.from_numpy(bg)
    gt = torch.from_numpy(synth.uniform01(9, n * 3).reshape(n, 3).astype(np.float32))
    embs, nerfs = [R.Embedding(3, 10), None, R.Embedding(3, 4)], [R.build_nerf(sd)]
    with torch.no_grad():
        out, (lo, hi) = D.render_sharded(R.render_rays, rays, bg, embs, nerfs, N_samples=16, noise_std=0)
        loss = D.reduce_loss(D.loss_partials(out, gt[lo:hi]))
        full_rgb = D.gather_pixels(out["rgb_coarse"], n)
    q.put((rank, lo, hi, loss, full_rgb.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process():
    """world_size 2 on CPU (gloo): each rank renders its contiguous block (the oracle stands in for
    the kernel here -- this test is about the sharding / reduction logic), the all-reduced loss
    equals the single-process loss and the gathered pixels keep the global ray order bit-exactly."""
    import torch.multiprocessing as mp
    from moco_flow_amd import dist as D, synth
    from oracle import cpu_ref as R
    n, world, port = 37, 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = synth.nerf_state(3, regime="dense")
    rays, bg = synth.rays(3, n)
    gt = torch.from_numpy(synth.uniform01(9, n * 3).reshape(n, 3).astype(np.float32))
    with torch.no_grad():
        ref = R.render_rays(torch.from_numpy(rays), torch.from_numpy(bg),
                            [R.Embedding(3, 10), None, R.Embedding(3, 4)], [R.build_nerf(sd)],
                            N_samples=16, noise_std=0)
    want = D.reduce_loss(D.loss_partials(ref, gt))
    assert (got[0][1], got[0][2], got[1][1], got[1][2]) == (0, 19, 19, 37)      # index bookkeeping
    for r in range(world):
        assert got[r][3]["img_loss"] == pytest.approx(want["img_loss"], rel=1e-12)
        assert np.array_equal(got[r][4], ref["rgb_coarse"].numpy())


def _trainer_losses(res, gt):
    """models/losses.py:4-14 (MSELoss over both passes) and trainer_moco_flow.py:317-328 (consensus means,
    coarse + fine), op for op."""
    mse = torch.nn.MSELoss(reduction="mean")
    img = mse(res["rgb_coarse"], gt)
    if "rgb_fine" in res:
        img = img + mse(res["rgb_fine"], gt)
    out = {"img_loss": float(img)}
    for key in ("nof_local", "nof_global"):
        if f"{key}_disp_coarse" in res:
            v = torch.mean(res[f"{key}_disp_coarse"])
            if f"{key}_disp_fine" in res:
                v = v + torch.mean(res[f"{key}_disp_fine"])
            out[key] = float(v)
    return out


def _moco_fine_oracle(n, lo=0, hi=None):
    from helpers import build_case, case_inputs
    from oracle import cpu_ref as R
    case = dict(extra="ind", regime="dense", nof="global", S=12, M=8, n=n)
    embs, nerfs, kw = build_case(R, case, 5)
    rays, bg = case_inputs(case, 5)
    hi = n if hi is None else hi
    with torch.no_grad():
        return R.render_rays(rays[lo:hi], bg[lo:hi], embs, nerfs, **kw)


def _moco_loss_worker(rank, world, port, n, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from moco_flow_amd import dist as D, synth
    lo, hi = D.shard_bounds(n, rank, world)
    gt = torch.from_numpy(synth.uniform01(11, n * 3).reshape(n, 3).astype(np.float32))
    res = _moco_fine_oracle(n, lo, hi)
    q.put((rank, D.reduce_loss(D.loss_partials(res, gt[lo:hi]))))
    dist.barrier()
    dist.destroy_process_group()


def test_loss_partials_match_the_trainer_formulas_moco_fine():
    """the consensus terms are mean(coarse) + mean(fine) (trainer_moco_flow.py:317-328), not one
    pooled mean.  Single process and 2-rank gloo (ray-sharded) against MSELoss + the trainer's formula on a
    MoCo coarse+fine result of the oracle."""
    import torch.multiprocessing as mp
    from moco_flow_amd import dist as D, synth
    n = 21
    gt = torch.from_numpy(synth.uniform01(11, n * 3).reshape(n, 3).astype(np.float32))
    res = _moco_fine_oracle(n)
    assert {"nof_local_disp_fine", "nof_global_disp_fine", "rgb_fine"} <= set(res)
    want = _trainer_losses(res, gt)
    got = D.reduce_loss(D.loss_partials(res, gt))
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-6), k
    assert D.loss_partials(res, gt).shape == (D.N_PARTIALS,)
    world, port = 2, 33500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_moco_loss_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, loss in outs:
        for k, v in want.items():
            assert loss[k] == pytest.approx(v, rel=1e-6), k


def _reducer_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moco_flow_amd.dist import OverlappedLossReducer
    red = OverlappedLossReducer(2, "cpu", depth=2)
    got = []
    for step in range(5):
        part = torch.tensor([float(step * 10 + rank + 1), 1.0], dtype=torch.float64)
        r = red.push(part, collect=True, donate=step % 2 == 1)      # (odd steps: reduced in place, no staging copy)
        if step % 2 == 1:
            red.work[(red.i - 1) % 2].wait()
            assert part.tolist() == [float(2 * step * 10 + 3), 2.0]   # the donated tensor IS the ring slot
        if r is not None:
            got.append(r.tolist())
    got += [r.tolist() for r in red.finish()]
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_loss_reducer_two_rank_gloo():
    """bench.py's per-step loss all-reduce (asynchronous, rotating buffers): totals come back complete, in
    issue order, on both ranks; a single process degenerates to the identity."""
    import torch.multiprocessing as mp
    from moco_flow_amd.dist import OverlappedLossReducer
    world, port = 2, 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[float(2 * (s * 10) + 3), 2.0] for s in range(5)]          # (10 s + 1) + (10 s + 2), counts 1 + 1
    assert got[0][1] == want and got[1][1] == want
    solo = OverlappedLossReducer(2, "cpu", depth=3)
    outs = [solo.push(torch.tensor([float(i), 1.0], dtype=torch.float64), collect=True) for i in range(4)]
    assert outs[:3] == [None, None, None] and outs[3].tolist() == [0.0, 1.0]
    assert [r.tolist() for r in solo.finish()] == [[1.0, 1.0], [2.0, 1.0], [3.0, 1.0]]


def _dp_case(n, lo=0, hi=None):
    """The oracle's networks + one MoCo coarse + fine pass over rays [lo, hi) WITH gradients (the oracle stands in for the
    HIP modules: this is about the reduction logic, and HIP modules do not run on the CPU)."""
    from helpers import build_case, case_inputs
    from oracle import cpu_ref as R
    case = dict(extra="ind", regime="dense", nof="global", S=12, M=8, n=n)
    embs, nerfs, kw = build_case(R, case, 5)
    rays, bg = case_inputs(case, 5)
    hi = n if hi is None else hi
    nets = list(nerfs) + list(kw["nof_models"])
    plists = []
    for i, m in enumerate(nets):                                # the oracle's networks are tensor containers (.p)
        plists.append([m.p[k].requires_grad_(not (i == 1 and k.startswith("rgb."))) for k in sorted(m.p)])
    return plists, (lambda: R.render_rays(rays[lo:hi], bg[lo:hi], embs, nerfs, **kw))


def _dp_total(parts):
    from moco_flow_amd import losses
    t = losses.from_partials(parts)
    return t["img_loss"] + 0.1 * t["nof_local"] + 0.1 * t["nof_global"]


def _grad_worker(rank, world, port, n, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from moco_flow_amd import dist as D, synth
    lo, hi = D.shard_bounds(n, rank, world)
    gt = torch.from_numpy(synth.uniform01(11, n * 3).reshape(n, 3).astype(np.float32))
    # (the fine NeRF's rgb head is frozen -- the reference's frozen-sub-module phases, trainer_moco_flow.py:391-404: no
    #  gradient, no hook, and the bucket still goes out, at wait())
    nets, render = _dp_case(n, lo, hi)
    red = D.GradReducer([(f"net{i}", m) for i, m in enumerate(nets)], average=False)
    opt = torch.optim.SGD([p for m in nets for p in m if p.requires_grad], lr=0.05)
    snap = []
    for step in range(2):
        opt.zero_grad(set_to_none=(step == 0))
        res = render()
        total = _dp_total(D.global_partials(D.loss_partials(res, gt[lo:hi])))
        total.backward()
        issued_in_backward = red.issued
        red.wait()
        if step == 0:
            snap = [None if p.grad is None else p.grad.clone() for m in nets for p in m]
            first = float(total.detach())
        opt.step()
        if step == 0:
            w_first = [p.detach().clone().numpy() for m in nets for p in m]
    w = [p.detach().clone() for m in nets for p in m]
    q.put((rank, (first, float(total.detach()), w_first), issued_in_backward, [None if g is None else g.numpy() for g in snap], [t.numpy() for t in w]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_two_rank_gloo_equals_single_process_step():
    """SURVEY 8(e), training half: two ranks, different rays (a ragged 11 / 10 split), the global loss from the all-reduced
    partials (dist.global_partials), per-network gradient buckets all-reduced from the post-accumulate hooks
    (dist.GradReducer, SUM) -> after two SGD steps both ranks hold IDENTICAL weights, equal to the single-process steps on
    the concatenated batch within fp32 reduction order; masked consensus means with data-dependent counts included."""
    import torch.multiprocessing as mp
    from moco_flow_amd import dist as D, synth
    n = 21
    gt = torch.from_numpy(synth.uniform01(11, n * 3).reshape(n, 3).astype(np.float32))
    nets, render = _dp_case(n)
    opt = torch.optim.SGD([p for m in nets for p in m if p.requires_grad], lr=0.05)
    for step in range(2):
        opt.zero_grad()
        total = _dp_total(D.loss_partials(render(), gt))
        total.backward()
        if step == 0:
            want_g = [None if p.grad is None else p.grad.clone() for m in nets for p in m]
            first = float(total.detach())
        opt.step()
        if step == 0:
            want_w = [p.detach().clone() for m in nets for p in m]
    world, port = 2, 35500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, i0, g0, w0), (_, l1, i1, g1, w1) = outs
    # every rank forms the same global loss; step 0: float64 sums of the same fp32 terms; step 1: behind one update (the MoCo
    # chain carries sin(512 x) of a NoF output: a 1e-6 difference in a gradient moves the next loss by ~1e-5)
    assert l0[:2] == l1[:2]
    assert l0[0] == pytest.approx(first, rel=1e-9) and l0[1] == pytest.approx(float(total.detach()), rel=1e-3)
    # all four buckets go out INSIDE backward(), from the hooks (a frozen parameter is not waited for), in both steps
    assert i0 == i1 == 8
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)                                     # identical weights on both ranks
    for a, b, w in zip(g0, g1, want_g):
        assert (a is None) == (w is None) == (b is None)
        if w is not None:
            assert np.array_equal(a, b)
            denom = max(float(w.abs().max()), 1e-12)
            assert float(np.abs(a - w.numpy()).max()) / denom < 2e-5
    # the weights after the FIRST update against the single-process step (the second gradient is taken at weights 1e-6
    # apart, on a loss with ReLU kinks and sin(512 x): compared between the ranks only, above)
    for a, b, w in zip(l0[2], l1[2], want_w):
        assert np.array_equal(a, b)
        assert float(np.abs(a - w.numpy()).max()) / max(float(w.abs().max()), 1e-12) < 2e-5


def test_grad_reducer_single_process_semantics():
    """No process group: the reducer is a flat-buffer view manager.  .grad aliases the flat slots; zero_grad(set_to_none)
    is survived; average=True at world 1 is the identity; a second backward without wait() raises (a bucket is reduced
    once per step); the donate path of the loss reducer refuses tensors that autograd saved."""
    from moco_flow_amd.dist import GradReducer, OverlappedLossReducer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(4, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    red = GradReducer([m[0], m[2]])
    x = torch.randn(7, 4)
    m(x).square().sum().backward()
    assert red.issued == 2
    flat = red.wait().clone()
    for p in m.parameters():
        assert p.grad.data_ptr() == red.view_of(p).data_ptr()
    ref = torch.autograd.grad(m(x).square().sum(), list(m.parameters()))
    for p, g in zip(m.parameters(), ref):
        assert torch.allclose(p.grad, g)
    m.zero_grad(set_to_none=True)
    m(x).square().sum().backward()
    assert torch.equal(red.wait(), flat)
    m.zero_grad(set_to_none=True)
    m(x).square().sum().backward()
    with pytest.raises(RuntimeError, match="after its all-reduce was issued"):
        m(x).square().sum().backward()
    red.wait()
    red.remove()
    with pytest.raises(ValueError):
        GradReducer([])
    # a parameter UNFROZEN after construction has no hook for one step -- its bucket must then be left to
    # wait() as a whole (issued from the other parameters' hooks it would go out before that gradient exists and the
    # gradient would land in a fresh .grad outside the flat buffer, never reduced): the reference's coarse2fine transition
    # (trainer_moco_flow.py:391-404) unfreezes the trunk beside already-trainable heads of the same network
    m = torch.nn.Sequential(torch.nn.Linear(4, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    m[0].weight.requires_grad_(False)
    red = GradReducer([("net", m)])
    m(x).square().sum().backward()
    assert red.issued == 1 and m[0].weight.grad is None
    red.wait()
    m[0].weight.requires_grad_(True)                                   # unfrozen between two steps: no hook yet
    m.zero_grad(set_to_none=True)
    m(x).square().sum().backward()
    assert red.issued == 1                                             # NOT issued from the hooks this step
    red.wait()
    assert red.issued == 2
    want = torch.autograd.grad(m(x).square().sum(), list(m.parameters()))
    for q, g in zip(m.parameters(), want):
        assert q.grad.data_ptr() == red.view_of(q).data_ptr() and torch.allclose(red.view_of(q), g)
    m.zero_grad(set_to_none=True)
    m(x).square().sum().backward()
    assert red.issued == 3                                             # from the hooks again (registered by wait())
    red.wait()
    red.remove()
    # the collectives go out in ONE order whatever order the hooks complete in.  Three independent
    # networks, backward through them in different orders: before the first wait() reverse registration; afterwards the
    # order observed in the first backward -- also when a later step's backward runs the other way round
    nets = [torch.nn.Linear(3, 3) for _ in range(3)]
    red = GradReducer([(f"n{i}", n) for i, n in enumerate(nets)])
    y = torch.randn(2, 3)
    loss_of = lambda order: sum((k + 1.0) * nets[i](y).square().sum() for k, i in enumerate(order))
    for i in (1, 0, 2):                                                # hooks complete in the order 1, 0, 2
        nets[i](y).square().sum().backward()
    assert red.issue_log == [2, 1, 0]                                  # nothing went out before bucket 2 was ready
    red.wait()
    assert red._order == [1, 0, 2]                                     # adopted from the first backward
    for n in nets:
        n.zero_grad(set_to_none=True)
    for i in (2, 0, 1):                                                # a step whose hooks complete the other way round
        nets[i](y).square().sum().backward()
    red.wait()
    assert red.issue_log[3:] == [1, 0, 2]
    # gradient accumulation: all but the last backward under no_sync() (the old message recommended a
    # sequence that raised)
    for n in nets:
        n.zero_grad(set_to_none=True)
    before = red.issued
    with red.no_sync():
        for n in nets:
            n(y).square().sum().backward()
    assert red.issued == before
    for n in nets:
        n(y).square().sum().backward()
    flat2 = red.wait().clone()
    for n in nets:
        g = torch.autograd.grad(n(y).square().sum(), list(n.parameters()))
        for q, gg in zip(n.parameters(), g):
            assert torch.allclose(red.view_of(q), 2 * gg)
    assert red.issued == before + 3 and flat2.abs().sum() > 0
    red.remove()
    # donation only for gradient-free partials
    lr = OverlappedLossReducer(3, "cpu", depth=2)
    w = torch.ones(3, dtype=torch.float64, requires_grad=True)
    part = w * 2.0
    lr.push(part, donate=True)
    assert lr.bufs[0] is not part and lr.bufs[0].tolist() == [2.0, 2.0, 2.0]
    part.sum().backward()                                               # autograd's saved tensors are untouched
    free = torch.full((3,), 5.0, dtype=torch.float64)
    lr.push(free, donate=True)
    assert lr.bufs[1] is free


def test_packed_cache_does_not_travel():
    """the packed-weights cache holds ctypes structures with device pointers; deepcopy / pickle of a
    module that has rendered must work (the copy re-packs) and invalidate_packed() must drop every cache."""
    import copy
    import io
    import pickle
    import moco_flow_amd as M
    import moco_flow_amd._lib as L
    for m in (M.NeRF(8, 256, 63, [4], "dir", 27), M.NoF(4, 128, 33, [2], "ind", 33, True)):
        for c in (m._packed, m._packed_bf16, m._packed_bwd):       # what a first forward leaves behind
            c.key, c.desc, c.buf, c.keep = ("k",), (L.mf_nerf_desc(), ctypes.pointer(L.mf_nerf_desc())), object(), [1]
        m2 = copy.deepcopy(m)
        assert m2._packed.key is None and m2._packed.desc is None and m2._packed is not m._packed
        m3 = pickle.loads(pickle.dumps(m))
        assert m3._packed_bwd.key is None and m3._packed_bf16.buf is None
        buf = io.BytesIO()
        torch.save(m, buf)
        assert set(m3.state_dict()) == set(m.state_dict())
        m.invalidate_packed()
        assert all(c.key is None and c.buf is None for c in (m._packed, m._packed_bf16, m._packed_bwd))


def test_descriptor_and_slot_caches_follow_edits():
    """The per-call host work of a render pass is cached (embedding descriptors, the parameter slots of the packed-weights
    key): the caches must follow what the trainer does to these objects -- `emb.weights` re-assigned or edited in place
    (trainer_moco_flow.py:289,301), parameters replaced -- and must survive deepcopy / pickle."""
    import copy
    import io
    import moco_flow_amd as M
    from moco_flow_amd.packing import PackedWeights
    e = M.Embedding(3, 10)
    d = e.descriptor()
    assert e.descriptor() is d and d.n_freqs == 10 and d.freq[9] == 512.0 and d.weight[9] == 1.0
    e.weights = [0.5] * 10
    assert e.descriptor() is not d and e.descriptor().weight[3] == 0.5
    e.weights[2] = 0.25
    assert e.descriptor().weight[2] == 0.25
    e.set_weights(0)
    assert e.descriptor().weight[2] == 0.0
    e.freq_bands = torch.linspace(1, 512, 10)              # a re-assigned table (logscale=False spacing) is seen
    assert abs(e.descriptor().freq[1] - float(e.freq_bands[1])) < 1e-6
    e.freq_bands[1] = 3.0                                  # and so is an in-place edit (version counter)
    assert e.descriptor().freq[1] == 3.0
    assert copy.deepcopy(e).descriptor().weight[2] == 0.0
    buf = io.BytesIO()
    torch.save(e, buf)
    buf.seek(0)
    assert torch.load(buf, weights_only=False).descriptor().n_freqs == 10
    # the packed-weights key walks the module tree itself (no named_modules()): same tensors, same order as
    # module.parameters(), and a replaced parameter or sub-module is seen on the next call
    from moco_flow_amd.packing import _collect_params
    for m in (M.NoF(4, 128, 33, [2], "ind", 33, True), M.NeRF(8, 256, 63, [4], "dir", 27)):
        got = []
        _collect_params(m, got)
        assert len(got) == len(list(m.parameters())) and all(a is b for a, b in zip(got, m.parameters()))
    m = M.NoF(4, 128, 33, [2], "ind", 33, True)
    m.nof_encoding_1 = torch.nn.Sequential(torch.nn.Linear(66, 128), torch.nn.ReLU(True))
    got = []
    _collect_params(m, got)
    assert any(p is m.nof_encoding_1[0].weight for p in got)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        PackedWeights().get(m, None, None, None, "NoF")


def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` run BARE (no torch.distributed.run) starts its own two workers before touching
    the GPU; control flow on CPU with MF_BENCH_DRYRUN (gloo rendezvous on 127.0.0.1, overlapped reducer,
    MAX-over-ranks timing, exactly one JSON line from rank 0, exit code = worst worker)."""
    import json
    import subprocess
    env = dict(os.environ, MF_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dryrun"][0] == 3.0      # ranks contribute 1 + 2
    # the driver's largest world: 8 ranks (1 + 2 + ... + 8 = 36)
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                        env=env, capture_output=True, text=True, timeout=400)
    assert r8.returncode == 0, r8.stderr[-2000:]
    d8 = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][-1])
    assert d8["n_gpus"] == 8 and d8["dryrun"][0] == 36.0
    # round 6: at N > 1 the MAIN line is the sharded workload with its collective, and the ranks the
    # process group's all-reduce reached are on the line
    for dd, n in ((d, 2), (d8, 8)):
        assert dd["rccl_ranks_seen"] == n
        assert dd["config"]["main_has_collective"] is True and dd["config"]["loss_allreduce"] is True
        assert dd["config"]["sharding"] == f"rays{n}" and dd["config"]["global_rays"] == 4096 * n and dd["dtype"] == "f32"
        assert set(dd["legs"]) == {"C2", "C4", "C5"}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "nope"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


REQUIRED_LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config", "roofline", "legs"}


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_last_stdout_line_is_compact(gpus):
    """round 5's 27.9 KB line left the driver's record with parsed = null.  The LAST stdout line of bench.py
    must be ONE strict-JSON object of at most 1800 bytes (it then fits whole in the driver's 2000-character tail) that
    carries the contract keys + roofline + cpu_baseline + legs; the tens of KB of detail go to stderr / a side file.  Run
    through MF_BENCH_DRYRUN: the same assemble() / emit() as a real run, fed results shaped like run_config()'s with
    full-length floats for the complete leg set of that world size."""
    import json
    import subprocess
    env = dict(os.environ, MF_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last.encode()) <= 1800, len(last.encode())

    def no_constants(name):
        raise AssertionError("non-strict JSON constant " + name)
    d = json.loads(last, parse_constant=no_constants)
    assert REQUIRED_LINE_KEYS <= set(d), REQUIRED_LINE_KEYS - set(d)
    assert d["n_gpus"] == gpus and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "ray-samples/s"
    assert {"workload", "rays_per_gpu", "samples_per_ray", "global_rays", "sharding", "loss_allreduce", "main_has_collective"} <= set(d["config"])
    assert {"bound", "achieved", "peak", "unit", "frac", "frac_step", "traffic", "traffic_algorithmic", "kernel", "kernel_ms",
            "from_profiles"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample", "cpu"} <= set(d["cpu_baseline"])
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] <= 1
    want = {"C2x", "C3", "C3x", "C3g", "C5", "C5x", "C5full", "C5xfull"} if gpus == 1 else {"C2", "C4", "C5"}
    assert set(d["legs"]) == want and all(len(v) == len(d["leg_fields"]) for v in d["legs"].values())
    if gpus == 1:
        assert {"s1", "joint"} <= set(d["train_ms"])
    # no prose: every string on the line is short
    def strings(o):
        if isinstance(o, str):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
    assert max(len(x) for x in strings(d)) <= 64
    # the detail went to stderr as one line that no '{'-line parser picks up
    det = [l for l in r.stderr.splitlines() if l.startswith("bench_detail {")]
    assert len(det) == 1 and json.loads(det[0][len("bench_detail "):])["roofline"]["from_profiles"] is not None


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 launch line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with MF_BENCH_DRYRUN (gloo, CPU): ranks from the launcher's environment (no
    self-spawn), the compact object as the LAST stdout line."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MF_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.strip()][-1]
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["rccl_ranks_seen"] == 2 and d["config"]["main_has_collective"] is True
    assert d["dryrun"][0] == 3.0 and len(last.encode()) <= 1800


def test_profile_summary_splits_coarse_and_fine_launches():
    """tools/summarize_prof.py: the dispatches of ONE kernel that fall into two duration classes -- the coarse and
    the fine pass of a render_rays call: same grid, dynamic LDS the trace does not record -- are reported per class, never as one
    average; a kernel whose dispatches agree within a factor 1.6 is left alone."""
    import re
    src = open(os.path.join(ROOT, "tools", "summarize_prof.py")).read()
    ns = {}
    exec(re.search(r"def two_clusters\(d\):.*?return a, b\n", src, re.S).group(0), ns)
    two = ns["two_clusters"]
    a, b = two([850, 2520, 845, 2530, 2490, 860, 3500, 2510])
    assert sorted(a) == [845, 850, 860] and len(b) == 5 and min(b) == 2490
    assert two([318, 320, 330, 400, 298]) is None and two([100, 300]) is None


def test_scripts_compile():
    """bench.py, __graft_entry__.py and every tools/*.py at least byte-compile (they only run on the GPU box)."""
    import glob
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) >= 8
    for f in files:
        with open(f) as fh:
            compile(fh.read(), f, "exec")


def test_lazy_vectors_host_logic():
    """lazy.MaskedVector / ConsensusPass on CPU tensors (no kernel involved): (i) the eager opt-out under autograd asks for
    "local" before the "global" plane exists -- the keyed cache must serve both (KeyError 'global' with
    LAZY_CONSENSUS off and chain_global); (ii) torch.cat of vectors along dim 0 -- the reference trainer's per-chunk
    concatenation, trainer/trainer_moco_flow.py:199-223 -- stays lazy, and its mean / sum are those of the concatenated
    tensors; (iii) the mean of a gradient-free vector is the caller's own scalar: the trainer's in-place `+=`
    (trainer_moco_flow.py:318-321) must not reach what a later torch.mean of the same vector returns."""
    from moco_flow_amd.lazy import ConsensusPass, MaskedVector
    torch.manual_seed(0)

    def diff_group(n, s):
        alphas = torch.rand(n, s) * 0.05
        planes = {}
        sel = lambda: {k: torch.masked_select(v, alphas.ge(0.01)) for k, v in planes.items()}
        return alphas, planes, ConsensusPass(alphas, planes, None, sel, True)

    # (i)
    alphas, planes, g = diff_group(5, 7)
    planes["local"] = torch.rand(5, 7, requires_grad=True)
    local = MaskedVector(g, "local").materialize()
    planes["global"] = torch.rand(5, 7, requires_grad=True)
    glob = MaskedVector(g, "global").materialize()
    assert torch.equal(glob, planes["global"][alphas.ge(0.01)])
    assert MaskedVector(g, "local").materialize() is local          # what was handed out stays
    (local.sum() + glob.sum()).backward()
    assert planes["local"].grad is not None and planes["global"].grad is not None

    # (ii) differentiable parts
    parts, eager = [], []
    for n in (3, 4, 2):
        a, pl, gg = diff_group(n, 6)
        pl["local"] = torch.rand(n, 6)
        parts.append(MaskedVector(gg, "local"))
        eager.append(pl["local"][a.ge(0.01)])
    one = torch.cat([parts[0]], 0)
    assert isinstance(one, MaskedVector) and one._g._vectors is None
    cat = torch.cat(parts, 0)
    assert isinstance(cat, MaskedVector) and all(p._g._vectors is None for p in parts)
    want = torch.cat(eager, 0)
    assert torch.mean(cat).item() == pytest.approx(want.mean().item(), rel=1e-6)
    assert cat.sum().item() == pytest.approx(want.sum().item(), rel=1e-6)
    assert torch.equal(cat.materialize(), want) and len(cat) == want.shape[0]
    assert torch.is_tensor(torch.cat(parts, 0) * 2)                  # anything else materialises
    # a mixed list (tensor + vector) is an ordinary cat
    assert torch.equal(torch.cat([parts[0], want[:2]], 0), torch.cat([eager[0], want[:2]], 0))

    # (iii) gradient-free: stats() is the kernel's (sum, count, mean) triple
    cache = {"local": (torch.tensor(6.0, dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64), torch.tensor(2.0))}
    gf = ConsensusPass(torch.rand(2, 2), {"local": torch.rand(2, 2)}, lambda: cache, lambda: {}, False)
    v = MaskedVector(gf, "local")
    m = torch.mean(v)                                                # the kernel's scalar itself, handed out once (no copy)
    assert m is cache["local"][2]
    m += 10.0
    m2 = torch.mean(v)                                               # later requests: the same float from (sum, count)
    assert float(m2) == 2.0 and m2.dtype == torch.float32 and m2 is not m
    m2 += 1.0
    assert float(torch.mean(v)) == 2.0 and float(cache["local"][0]) == 6.0
    two = torch.cat([v, v], 0)                                       # chunks: sum of sums / sum of counts
    assert float(torch.mean(two)) == pytest.approx(2.0)
