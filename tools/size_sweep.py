#!/usr/bin/env python3
"""fp32 (Multiply, Add) across problem sizes: shape-adaptive geometry (default) against the fixed
256x256 tile.  Shows the round-quantisation effect the adaptive pick removes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep  # noqa: E402
from _lib import g  # noqa: E402  (MM_LIB=lab selects the lab build)

sizes = [512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 12288, 16384]
print(f"{'size':>6}  {'adaptive kernel':38s} {'TF':>7}   {'fixed 256x256 TF':>16}")
for s in sizes:
    g.set_tuning("f32_variant", -1)
    name = g.kernel_name(g.make_config("float"), s, s, s)
    med, _ = sweep.time_config("float", "Multiply", "Add", s, s, s, 5)
    g.set_tuning("f32_variant", 8)
    med8, _ = sweep.time_config("float", "Multiply", "Add", s, s, s, 5)
    print(f"{s:6d}  {name:38s} {2.0*s**3/med/1e12:7.1f}   {2.0*s**3/med8/1e12:16.1f}", flush=True)

for key, dtype, env, fixed in (("f64", "double", "f64_variant", 0), ("f16", "half", "f16_variant", 0)):
    print(f"\n{key}: adaptive vs fixed large tile")
    for s_ in [1024, 2048, 3072, 4096, 6144, 8192]:
        g.set_tuning(env, -1)
        med, _ = sweep.time_config(dtype, "Multiply", "Add", s_, s_, s_, 5)
        g.set_tuning(env, fixed)
        med0, _ = sweep.time_config(dtype, "Multiply", "Add", s_, s_, s_, 5)
        g.set_tuning(env, -1)
        print(f"{s_:6d}  adaptive {2.0*s_**3/med/1e12:8.1f}   fixed {2.0*s_**3/med0/1e12:8.1f}", flush=True)
