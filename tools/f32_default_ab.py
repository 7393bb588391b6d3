#!/usr/bin/env python3
"""Steady-state throughput of the two fp32 geometries the large-problem default can be, float 16384^3 (BASELINE
configs[1]) and the 65536 x 16384 x 16384 job: R launches back to back between two events on one stream (what bench.py
times), f32_variant 33 (128 x 256, two workgroups per CU) and 8 (256 x 256) alternating in one process, so that clock and
thermal drift hit both alike.  The energy side of the comparison is tools/f32_energy.py.
  python tools/f32_default_ab.py [--rounds 4] [--steps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gemm_hls_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--variants", default="33,8")
ap.add_argument("--shapes", default="16384x16384x16384,65536x16384x16384")
args = ap.parse_args()
dev = torch.device("cuda:0")
variants = [int(v) for v in args.variants.split(",")]
for shape in args.shapes.split(","):
    n, k, m = (int(x) for x in shape.split("x"))
    a = torch.empty((n, k), device=dev)
    b = torch.empty((k, m), device=dev)
    c = torch.empty((n, m), device=dev)
    g._check(g.lib().mm_fill_device(0, 0, a.data_ptr(), a.numel(), 1))
    g._check(g.lib().mm_fill_device(0, 0, b.data_ptr(), b.numel(), 2))
    steps = args.steps if n <= 16384 else max(5, args.steps // 4)
    res = {v: [] for v in variants}
    for rnd in range(args.rounds + 1):          # round 0 warms up
        for v in variants:
            g.set_tuning("f32_variant", v)
            g.matmul(a, b, out=c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                g.matmul(a, b, out=c)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[v].append(2.0 * n * k * m * steps / (1e-3 * e0.elapsed_time(e1)) / 1e12)
    g.set_tuning("f32_variant", -1)
    for v in variants:
        r = sorted(res[v])
        print(f"float {shape} f32_variant {v:3d}: TFLOP/s per round {[round(x, 2) for x in res[v]]}  median {r[len(r) // 2]:.2f}", flush=True)
    del a, b, c
    torch.cuda.empty_cache()
