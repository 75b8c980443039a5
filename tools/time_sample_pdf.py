"""Time the inverse-CDF resample + merge launch (mf_sample_pdf_eps) in pieces on cuda:0:
full (pdf/cdf + search + rank sort), without the sort (z_sorted_out NULL), with a given cdf (no pdf/cdf sums).
    python tools/time_sample_pdf.py [n_rays ...]"""
import sys
import torch
from moco_flow_amd import _lib as L

def timed(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def main():
    dev = torch.device("cuda:0")
    S, M = 64, 128
    for N in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]:
        g = torch.Generator(device=dev).manual_seed(0)
        z = torch.linspace(2.0, 6.0, S, device=dev).expand(N, S).contiguous()
        w = torch.rand(N, S, device=dev, generator=g)
        u = torch.linspace(0, 1, M, device=dev)
        cdf = torch.cumsum(torch.rand(N, S - 1, device=dev, generator=g), -1)
        cdf = (cdf / cdf[:, -1:]).contiguous()
        zo = torch.empty(N, S + M, device=dev)
        zn = torch.empty(N, M, device=dev)
        st = L.current_stream(dev)
        lib = L.lib()
        def call(cdf_in, z_sorted, z_new):
            L.check(lib.mf_sample_pdf_eps(None, L.ptr(z), L.ptr(w[:, 1:]), S, N, S - 1, M, L.ptr(u), 0,
                                          None if cdf_in is None else L.ptr(cdf_in), None if z_new is None else L.ptr(z_new),
                                          None, None if z_sorted is None else L.ptr(z_sorted), 1e-5, st), "sample_pdf")
        print(f"N={N}: full {timed(lambda: call(None, zo, None)):.1f} us, no sort {timed(lambda: call(None, None, zn)):.1f} us, "
              f"cdf given + sort {timed(lambda: call(cdf, zo, None)):.1f} us, cdf given no sort {timed(lambda: call(cdf, None, zn)):.1f} us")

if __name__ == "__main__":
    main()
