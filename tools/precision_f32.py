#!/usr/bin/env python3
"""Full-matrix error of the fp32 variants at BASELINE size against an fp64 product of the same
operands (torch.float64 matmul on the GPU = rocBLAS dgemm, used here ONLY as an independent
cross-check of precision; the parity tests use CPU fp64).  Reports max and rms of
|C - C64| / C64 over all N*M outputs, and speed, per variant."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402  (MM_LIB=lab selects the lab build)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--variants", default="0,1,3,8,9,10", help="f32_variant values; s<v> = MM_PATH_SPLIT with split_variant v")
    ap.add_argument("--lo", type=float, default=None, help="with --hi: uniform [lo, hi) operands instead of the seeded [1,10); "
                    "errors are then normwise, |c - c64| / (|a| . |b|)")
    ap.add_argument("--hi", type=float, default=None)
    args = ap.parse_args()
    n = args.size
    dev = torch.device("cuda:0")
    a = torch.empty((n, n), dtype=torch.float32, device=dev)
    b = torch.empty((n, n), dtype=torch.float32, device=dev)
    L = g.lib()
    g._check(L.mm_fill_device(0, 0, a.data_ptr(), a.numel(), 1))
    g._check(L.mm_fill_device(0, 0, b.data_ptr(), b.numel(), 2))
    if args.lo is not None:
        a.uniform_(args.lo, args.hi)
        b.uniform_(args.lo, args.hi)
    ref = torch.empty((n, n), dtype=torch.float64, device=dev)
    den = ref if args.lo is None else torch.empty((n, n), dtype=torch.float64, device=dev)
    rows = 2048
    b64 = b.double()
    for r0 in range(0, n, rows):
        ref[r0:r0 + rows] = a[r0:r0 + rows].double() @ b64
        if args.lo is not None:
            den[r0:r0 + rows] = a[r0:r0 + rows].double().abs() @ b64.abs()
    del b64
    c = torch.empty((n, n), dtype=torch.float32, device=dev)
    for v in args.variants.split(","):
        path = g.PATH_SPLIT if v.startswith("s") else g.PATH_AUTO
        g.set_tuning("split_variant" if path == g.PATH_SPLIT else "f32_variant", int(v.lstrip("s")))
        g.matmul(a, b, out=c, path=path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g.matmul(a, b, out=c, path=path)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        worst, sq = 0.0, 0.0
        for r0 in range(0, n, rows):
            rel = ((c[r0:r0 + rows].double() - ref[r0:r0 + rows]) / den[r0:r0 + rows]).abs()
            worst = max(worst, float(rel.max()))
            sq += float((rel * rel).sum())
        print(f"v{v} {g.kernel_name(g.make_config('float', path=path), n, n, n):36s} {2.0*n**3/dt/1e12:7.2f} TF  "
              f"max rel err {worst:.3e}  rms {(sq/(n*n))**0.5:.3e}", flush=True)


if __name__ == "__main__":
    main()
