"""Case table shared by gen_golden.py (build container) and the tests (both boxes).
Pure data: no reference import here."""

# name -> dict(n, S, M, nof: none|local|global, quat, extra, regime, act, disp, test, ...)
RENDER_CASES = {
    "r_nerf_dir_default":      dict(n=48, S=64, M=0, extra="dir", regime="default"),
    "r_nerf_dir_dense":        dict(n=48, S=64, M=0, extra="dir", regime="dense"),
    "r_nerf_dir_dense_nobg":   dict(n=16, S=64, M=0, extra="dir", regime="dense", bg=False),
    "r_nerf_dir_softplus":     dict(n=32, S=64, M=0, extra="dir", regime="dense", act="softplus"),
    "r_nerf_dir_disp":         dict(n=32, S=64, M=0, extra="dir", regime="dense", disp=True),
    "r_nerf_dir_S128":         dict(n=16, S=128, M=0, extra="dir", regime="dense"),
    "r_nerf_dir_S40":          dict(n=20, S=40, M=0, extra="dir", regime="dense"),
    "r_nerf_none_dense":       dict(n=32, S=64, M=0, extra="none", regime="dense"),
    "r_nerf_ind_dense":        dict(n=32, S=64, M=0, extra="ind", regime="dense"),
    "r_nerf_dir_fine_train":   dict(n=32, S=64, M=128, extra="dir", regime="dense"),
    "r_nerf_dir_fine_test":    dict(n=32, S=64, M=128, extra="dir", regime="dense", test=True),
    "r_nerf_dir_c2f_weights":  dict(n=32, S=64, M=0, extra="dir", regime="dense",
                                    xyz_w=[1, 1, 1, 0.375, 0, 0, 0, 0, 0, 0]),
    "r_nerf_dir_nfreq0":       dict(n=32, S=64, M=0, extra="dir", regime="dense", xyz_freqs=0),
    "r_moco_bw_only":          dict(n=32, S=64, M=0, extra="ind", regime="dense", nof="bw"),
    "r_moco_local":            dict(n=32, S=64, M=0, extra="ind", regime="dense", nof="local"),
    "r_moco_global":           dict(n=32, S=64, M=0, extra="ind", regime="dense", nof="global"),
    "r_moco_global_default":   dict(n=32, S=64, M=0, extra="ind", regime="default", nof="global"),
    "r_moco_global_flowhead":  dict(n=32, S=64, M=0, extra="ind", regime="dense", nof="global", quat=False),
    "r_moco_global_fine":      dict(n=16, S=64, M=128, extra="ind", regime="dense", nof="global"),
    "r_moco_global_fine_test": dict(n=16, S=64, M=128, extra="ind", regime="dense", nof="global", test=True),
    "r_moco_local_test":       dict(n=16, S=64, M=0, extra="ind", regime="dense", nof="local", test=True),
    "r_empty":                 dict(n=0, S=64, M=0, extra="dir", regime="dense"),
    # N=0 with NoF is not a fixture: the reference itself raises there
    # (rendering.py:83 ``view(0, S, -1)`` is ambiguous) -- see DESIGN.md.
}


