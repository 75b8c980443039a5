#!/usr/bin/env python3
"""Experiment (pricing, not product): do a WRITER and a READER of the training backward run faster side by side on disjoint CUs
than one after the other?  Stage-1 shapes: the dX chain of the coarse pass (mf_nerf_backward3: 5120 x 128 samples, writes ~6 GB of
gradient rows, half matrix-bound) and the weight gradients of the fine pass (mf_weight_grads_p: 5120 x 256 samples, reads ~26 GB
at the ~4 TB/s a reader gets).  Sequential on the default stream vs two streams created with hipExtStreamCreateWithCUMask
(a fraction f of every XCD's CUs for the dX chain, the rest for the weight gradients).
    python tools/try_overlap.py [f ...]          (default f = 0.25 0.375 0.5)"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import moco_flow_amd as M
from moco_flow_amd import autograd as A

dev = torch.device("cuda")
hip = C.CDLL("libamdhip64.so")
N_CU, N_XCD = 256, 8


def masked_stream(pred):
    """stream restricted to the CUs i with pred(xcd = i % 8, cu_in_xcd = i // 8)"""
    words = (C.c_uint32 * (N_CU // 32))()
    for i in range(N_CU):
        if pred(i % N_XCD, i // N_XCD):
            words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(N_CU // 32), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def main():
    fracs = [float(a) for a in sys.argv[1:]] or [0.25, 0.375, 0.5]
    A.set_dx_precision("bf16x3")
    A.set_wgrad_precision("bf16x3")
    D, W = 8, 256
    stride = (D + 1) * W + W // 2
    nerf = M.NeRF(8, 256, 63, [4], "dir", 27).cuda()
    Pc, Pf = 5120 * 128, 5120 * 256
    acts_c = torch.randn(Pc, stride, device=dev).relu_()
    acts_c._mf_mask = torch.randint(-2**31, 2**31 - 1, (Pc, 80), device=dev, dtype=torch.int32)
    rgbsig, g_out = torch.rand(Pc, 4, device=dev), torch.randn(Pc, 4, device=dev)
    acts_f = torch.randn(Pf, stride, device=dev)
    gpre_f = torch.randn((Pf + 127) // 128 * 128, stride, device=dev)[:Pf]
    sl = lambda t, l, w=W: t[:, l * W:l * W + w]
    jobs = [(sl(gpre_f, l), sl(acts_f, l - 1), 256, 256, True) for l in range(1, 9)] + [(sl(gpre_f, 8), sl(acts_f, 7), 256, 256, True)]

    def dx():
        return A.nerf_backward_hip(nerf, g_out, acts_c, rgbsig)

    def wg():
        return A.weight_grads(jobs, Pf, dev)

    def timed(f, n=5):
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    t_dx, t_wg = timed(dx), timed(wg)
    t_seq = timed(lambda: (dx(), wg()))
    print(f"dX chain (coarse, {Pc} samples) {t_dx:.3f} ms; weight gradients (fine, nine 256 x 256 blocks, {Pf} samples) {t_wg:.3f} ms; "
          f"one after the other {t_seq:.3f} ms")
    for f in fracs:
        k = int(round(32 * f))
        sa = masked_stream(lambda x, c: c < k)
        sb = masked_stream(lambda x, c: c >= k)

        def both():
            with torch.cuda.stream(sa):
                dx()
            with torch.cuda.stream(sb):
                wg()

        def alone(s, fn):
            def run():
                with torch.cuda.stream(s):
                    fn()
            return run

        a_dx, a_wg = timed(alone(sa, dx)), timed(alone(sb, wg))
        t_par = timed(both)
        print(f"  dX on {8 * k} CUs alone {a_dx:.3f} ms, weight gradients on {256 - 8 * k} CUs alone {a_wg:.3f} ms, side by side {t_par:.3f} ms "
              f"({t_par / t_seq:.2f} x the sequential time)")


if __name__ == "__main__":
    main()
