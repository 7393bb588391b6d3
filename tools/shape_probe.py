#!/usr/bin/env python3
"""Times arbitrary N x K x M shapes through the C ABI (HIP-event kernel time), several variants round-robin.
  python tools/shape_probe.py float 16384x4096x16384 32768x4096x32768 [--variants 8,3] [--reps 7] [--path split]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402  (MM_LIB=lab selects the lab build)

KNOB = {"float": "f32_variant", "double": "f64_variant", "half": "f16_variant", "uint8_t": "i8_variant"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dtype")
    ap.add_argument("shapes", nargs="+")
    ap.add_argument("--variants", default="-1")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--path", choices=["auto", "split"], default="auto")
    args = ap.parse_args()
    L = g.lib()
    es = L.mm_dtype_size(g.DTYPES[args.dtype])
    path = g.PATH_SPLIT if args.path == "split" else g.PATH_AUTO
    knob = "split_variant" if args.path == "split" else KNOB[args.dtype]
    variants = [int(v) for v in args.variants.split(",")]
    for shape in args.shapes:
        n, k, m = (int(x) for x in shape.split("x"))
        ptrs = [ctypes.c_void_p() for _ in range(3)]
        for p, cnt in zip(ptrs, (n * k, k * m, n * m)):
            g._check(L.mm_alloc(0, cnt * es, ctypes.byref(p)))
        g._check(L.mm_fill_device(0, g.DTYPES[args.dtype], ptrs[0], n * k, 1))
        g._check(L.mm_fill_device(0, g.DTYPES[args.dtype], ptrs[1], k * m, 2))
        cfg = g.make_config(args.dtype, path=path)
        t = ctypes.c_double(0)
        times = {v: [] for v in variants}
        for rnd in range(args.reps + 1):
            for v in variants:
                g.set_tuning(knob, v)
                g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], n, k, m, ctypes.byref(t)))
                if rnd:
                    times[v].append(t.value)
        for p in ptrs:
            L.mm_free(0, p)
        for v in variants:
            g.set_tuning(knob, v)
            ts = sorted(times[v])
            med = ts[len(ts) // 2]
            print(f"{args.dtype} {shape:>22s} v{v:<4d} {g.kernel_name(cfg, n, k, m):36s} med {med*1e3:9.3f} ms  "
                  f"{2.0*n*k*m/med/1e12:8.2f} TOp/s  best {2.0*n*k*m/ts[0]/1e12:8.2f}", flush=True)
        g.set_tuning(knob, -1)


if __name__ == "__main__":
    main()
