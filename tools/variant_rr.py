#!/usr/bin/env python3
"""Round-robin A/B of kernel variants on ONE problem in ONE process: each round launches every variant once (HIP-event
kernel time through mm_gemm_launch), so clock and thermal drift hit all of them alike; medians over the rounds.
  python tools/variant_rr.py f16 --variants 200,203,204 --size 32768 --rounds 12"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402

TABLE = {"f16": ("half", "Multiply", "Add", "f16_variant"), "uint8": ("uint8_t", "Multiply", "Add", "i8_variant"),
         "f32": ("float", "Multiply", "Add", "f32_variant"), "f64": ("double", "Multiply", "Add", "f64_variant"),
         "minplus": ("float", "Add", "Min", "valu_variant"), "minplus_f64": ("double", "Add", "Min", "valu_variant")}
ap = argparse.ArgumentParser()
ap.add_argument("what", choices=list(TABLE))
ap.add_argument("--variants", required=True)
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--rounds", type=int, default=12)
args = ap.parse_args()
dtype, mp, rd, knob = TABLE[args.what]
variants = [int(v) for v in args.variants.split(",")]
L = g.lib()
s = args.size
es = L.mm_dtype_size(g.DTYPES[dtype])
cfg = g.make_config(dtype, mp, rd)
ptrs = [ctypes.c_void_p() for _ in range(3)]
for p in ptrs:
    g._check(L.mm_alloc(0, s * s * es, ctypes.byref(p)))
g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[0], s * s, 1))
g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[1], s * s, 2))
t = ctypes.c_double(0)
times = {v: [] for v in variants}
names = {}
for rnd in range(args.rounds + 2):          # two warm-up rounds
    order = variants if rnd % 2 == 0 else variants[::-1]
    for v in order:
        g.set_tuning(knob, v)
        names[v] = g.kernel_name(cfg, s, s, s)
        g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], s, s, s, ctypes.byref(t)))
        if rnd >= 2:
            times[v].append(t.value)
g.set_tuning(knob, -1)
base = None
for v in variants:
    ts = sorted(times[v])
    med = ts[len(ts) // 2]
    base = base or med
    print(f"{args.what} {s}^3 v{v:<4d} {names[v]:46s} median {med * 1e3:9.3f} ms {2.0 * s ** 3 / med / 1e12:9.2f} TOp/s  "
          f"best {2.0 * s ** 3 / ts[0] / 1e12:9.2f}  {100.0 * base / med:6.2f} % of the first", flush=True)
