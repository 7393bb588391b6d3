#!/usr/bin/env python3
"""The k-ordered contract has two kernels (mm_ordered.hip 64 x 64; the register-tiled "ordered_tile" of
mm_valu_tile_fp_exact.hip / the integer valu_tile): this tool checks that they give the SAME BITS over the type x operator
matrix on ragged and aligned shapes (device-side comparison of the whole output), and times both.

  python tools/ordered_tile_check.py [--sizes 4096,8192] [--reps 3]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402

NP = {"float": np.float32, "double": np.float64, "half": np.float16, "int8_t": np.int8, "uint8_t": np.uint8, "int16_t": np.int16,
      "uint16_t": np.uint16, "int": np.int32, "unsigned": np.uint32, "long": np.int64, "unsigned long": np.uint64}


def operands(dtype, n, k, m, transposed, rng):
    if dtype in ("float", "double", "half"):
        a = rng.uniform(-4, 10, size=(k, n) if transposed else (n, k)).astype(NP[dtype])
        b = rng.uniform(-4, 10, size=(k, m)).astype(NP[dtype])
        # specials: the ordered contract is std::min / std::max to the letter (NaN, signed zeros) and IEEE on inf
        for arr in (a, b):
            flat = arr.reshape(-1)
            idx = rng.integers(0, flat.size, size=max(1, flat.size // 97))
            flat[idx] = rng.choice(np.array([np.nan, np.inf, -np.inf, 0.0, -0.0], dtype=NP[dtype]), size=idx.size)
        return a, b
    info = np.iinfo(NP[dtype])
    a = rng.integers(info.min, int(info.max) + 1, size=(k, n) if transposed else (n, k), dtype=NP[dtype])
    b = rng.integers(info.min, int(info.max) + 1, size=(k, m), dtype=NP[dtype])
    return a, b


def bits(x):
    return x.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[x.dtype.itemsize])


def parity(shapes, seed=6):
    rng = np.random.default_rng(seed)
    bad = cases = tiled = 0
    for dtype in NP:
        for mp in ("Multiply", "Add", "Min", "Max", "And"):
            for rd in ("Add", "Min", "Max", "Multiply", "And"):
                for (n, k, m) in shapes:
                    for transposed in (False, True):
                        a, b = operands(dtype, n, k, m, transposed, rng)
                        name = g.kernel_name(g.make_config(dtype, mp, rd, g.PATH_ORDERED, transposed), n, k, m)
                        g.set_tuning("ordered_variant", 0)
                        c_old, _ = g.matmul_capi(a, b, dtype, mp, rd, g.PATH_ORDERED, transposed)
                        g.set_tuning("ordered_variant", -1)
                        c_new, _ = g.matmul_capi(a, b, dtype, mp, rd, g.PATH_ORDERED, transposed)
                        cases += 1
                        tiled += name == "ordered_tile"
                        # all NaNs count as one value: which of two NaN operands an add hands on follows the instruction's operand
                        # order (IEEE 754 leaves it open), which two compilations of one expression need not share
                        differ = bits(c_old) != bits(c_new)
                        if c_old.dtype.kind == "f":
                            differ &= ~(np.isnan(c_old) & np.isnan(c_new))
                        if differ.any():
                            bad += 1
                            w = np.argwhere(differ)
                            print(f"MISMATCH {dtype} ({mp},{rd}) {n}x{k}x{m} kxn={transposed} {name}: {len(w)} elements, first {w[0]} "
                                  f"{c_old[tuple(w[0])]} vs {c_new[tuple(w[0])]}", flush=True)
    print(f"ordered_tile vs ordered: {cases} cases ({tiled} on the tile kernel), {bad} with differing bits", flush=True)
    return bad


def timing(sizes, reps):
    from sweep import time_config
    for dtype, mp, rd in (("half", "Multiply", "Add"), ("float", "Multiply", "Add"), ("double", "Multiply", "Add"), ("int", "Multiply", "Add"),
                          ("float", "Add", "Min")):
        for s in sizes:
            row = []
            for ov in (0, -1):
                g.set_tuning("ordered_variant", ov)
                name = g.kernel_name(g.make_config(dtype, mp, rd, g.PATH_ORDERED), s, s, s)
                med, best = time_config(dtype, mp, rd, s, s, s, reps, path=g.PATH_ORDERED)
                row.append(f"{name:13s} med {med*1e3:9.2f} ms {2.0*s**3/med/1e12:7.2f} TOp/s")
            g.set_tuning("ordered_variant", -1)
            print(f"ORDERED {dtype:7s}({mp},{rd}) {s:6d}^3  " + "  |  ".join(row), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,8192")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    rc = 0
    if not args.no_parity:
        rc = parity([(128, 64, 128), (130, 68, 132), (257, 264, 272), (5, 4, 4), (513, 528, 528)])
    timing([int(s) for s in args.sizes.split(",")], args.reps)
    sys.exit(1 if rc else 0)
