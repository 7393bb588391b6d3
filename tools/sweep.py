#!/usr/bin/env python3
"""Times kernel variants through the C ABI (mm_gemm_launch: HIP-event kernel time) on
device-filled operands (uniform [1,10), the reference's input distribution).

  python tools/sweep.py f32 [--sizes 4096,8192,16384] [--variants 0,1,2] [--reps 5]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402  (MM_LIB=lab selects the lab build)

PEAK = {"float": 157.3, "double": 78.6, "half": 2500.0}


def time_config(dtype, mp, rd, n, k, m, reps, path=g.PATH_AUTO):
    L = g.lib()
    es = L.mm_dtype_size(g.DTYPES[dtype])
    cfg = g.make_config(dtype, mp, rd, path)
    ptrs = [ctypes.c_void_p() for _ in range(3)]
    for p, cnt in zip(ptrs, (n * k, k * m, n * m)):
        g._check(L.mm_alloc(0, cnt * es, ctypes.byref(p)))
    g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[0], n * k, 1))
    g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[1], k * m, 2))
    t = ctypes.c_double(0)
    times = []
    for i in range(reps + 1):
        g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], n, k, m, ctypes.byref(t)))
        if i:
            times.append(t.value)
    for p in ptrs:
        L.mm_free(0, p)
    times.sort()
    return times[len(times) // 2], times[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["f32", "f64", "f16", "minplus", "minplus_f64", "uint8", "split", "all"])
    ap.add_argument("--sizes", default="4096,8192,16384")
    ap.add_argument("--variants", default="")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    sizes = [int(s) for s in args.sizes.split(",")]
    if args.what == "split":   # MM_PATH_SPLIT next to the native fp32 kernel: ROUND-ROBIN launches (one of each per round) so
        # that clock / thermal drift hits every candidate alike; "n" = the native kernel
        variants = args.variants.split(",") if args.variants else ["n", "-1"]
        L = g.lib()
        for s in sizes:
            ptrs = [ctypes.c_void_p() for _ in range(3)]
            for p, cnt in zip(ptrs, (s * s, s * s, s * s)):
                g._check(L.mm_alloc(0, cnt * 4, ctypes.byref(p)))
            g._check(L.mm_fill_device(0, 0, ptrs[0], s * s, 1))
            g._check(L.mm_fill_device(0, 0, ptrs[1], s * s, 2))
            t = ctypes.c_double(0)
            times = {v: [] for v in variants}
            for rnd in range(args.reps + 1):
                for v in variants:
                    path = g.PATH_AUTO if v == "n" else g.PATH_SPLIT
                    if v != "n":
                        g.set_tuning("split_variant", int(v))
                    cfg = g.make_config("float", path=path)
                    g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], s, s, s, ctypes.byref(t)))
                    if rnd:
                        times[v].append(t.value)
            for p in ptrs:
                L.mm_free(0, p)
            g.set_tuning("split_variant", -1)
            for v in variants:
                ts = sorted(times[v])
                med = ts[len(ts) // 2]
                tf = 2.0 * s ** 3 / med / 1e12
                print(f"f32 {'native' if v == 'n' else 'split v' + v:12s} {s:6d}^3  med {med*1e3:9.3f} ms  {tf:7.2f} TF  "
                      f"({100*tf/PEAK['float']:5.1f}% of 157.3)  best {2.0*s**3/ts[0]/1e12:7.2f} TF", flush=True)
    elif args.what == "f32":
        variants = [int(v) for v in args.variants.split(",")] if args.variants else list(range(8))
        for v in variants:
            g.set_tuning("f32_variant", v)
            for s in sizes:
                name = g.kernel_name(g.make_config("float"), s, s, s)
                med, best = time_config("float", "Multiply", "Add", s, s, s, args.reps)
                tf = 2.0 * s ** 3 / med / 1e12
                print(f"f32 v{v} {name:36s} {s:6d}^3  med {med*1e3:9.3f} ms  {tf:7.2f} TF  "
                      f"({100*tf/PEAK['float']:5.1f}% of 157.3)  best {2.0*s**3/best/1e12:7.2f} TF", flush=True)
    else:
        table = {"f64": ("double", "Multiply", "Add"), "f16": ("half", "Multiply", "Add"),
                 "minplus": ("float", "Add", "Min"), "minplus_f64": ("double", "Add", "Min"), "uint8": ("uint8_t", "Multiply", "Add")}
        knob = {"f64": "f64_variant", "f16": "f16_variant", "uint8": "i8_variant", "minplus": "valu_variant", "minplus_f64": "valu_variant"}
        for key in ([args.what] if args.what != "all" else list(table)):
            dtype, mp, rd = table[key]
            variants = [int(v) for v in args.variants.split(",")] if (args.variants and key in knob) else [-1]
            for s in sizes:
                # variants interleaved per size, in one process: within-probe A/B (guide rule 24)
                for v in variants:
                    if key in knob:
                        g.set_tuning(knob[key], v)
                    name = g.kernel_name(g.make_config(dtype, mp, rd), s, s, s)
                    med, best = time_config(dtype, mp, rd, s, s, s, args.reps)
                    tf = 2.0 * s ** 3 / med / 1e12
                    print(f"{key} v{v:<3d} {name:24s} {s:6d}^3  med {med*1e3:9.3f} ms  {tf:8.2f} TOp/s  best {2.0*s**3/best/1e12:8.2f}",
                          flush=True)
            if key in knob:
                g.set_tuning(knob[key], -1)


if __name__ == "__main__":
    main()
