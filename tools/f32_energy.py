#!/usr/bin/env python3
"""fp32 default, decided by energy: the 128 x 256 two-workgroups-per-CU geometry (f32_variant 33, the
default: +1 % speed, 1.5-2x the fabric traffic) against the 256 x 256 geometry (8), float 16384^3, through the power-meter
build of the runner (bin/RunHardware_power.exe N K M hw off: the kernel keeps running until the GPU's power sensor has
been sampled for MM_POWER_WINDOW_MS).  Interleaved runs on one box; GFLOP/s per watt per variant.
  python tools/f32_energy.py [--rounds 4] [--window-ms 2000]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PERF = re.compile(r"([\d\.]+) seconds[^\d]+([\d\.]+) GOp/s")      # the reference's regex (scripts/build_manager.py:601-602)
POWER = re.compile(r"Measured an average power of ([\d\.]+) W")

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--window-ms", type=int, default=2000)
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--variants", default="33,8")
args = ap.parse_args()
exe = os.path.join(ROOT, "bin", "RunHardware_power.exe")
variants = [int(v) for v in args.variants.split(",")]
rows = {v: [] for v in variants}
for rnd in range(args.rounds):
    for v in variants:
        env = dict(os.environ, MM_F32_VARIANT=str(v), MM_POWER_WINDOW_MS=str(args.window_ms))
        r = subprocess.run([exe, str(args.size), str(args.size), str(args.size), "hw", "off"], capture_output=True, text=True, env=env, timeout=600)
        if r.returncode != 0:
            sys.exit(r.stdout + r.stderr)
        mt, mp = PERF.search(r.stdout), POWER.search(r.stdout)
        kern = re.search(r"Executing kernel \(([^)]+)\)", r.stdout)
        gops, watts = float(mt.group(2)), float(mp.group(1)) if mp else float("nan")
        rows[v].append((gops, watts))
        print(f"round {rnd} f32_variant {v:3d} {kern.group(1) if kern else '?':40s} {gops / 1e3:8.2f} TFLOP/s {watts:8.1f} W {gops / watts:8.2f} GFLOP/s/W", flush=True)
print()
for v in variants:
    gs = sorted(x[0] for x in rows[v])
    ws = sorted(x[1] for x in rows[v])
    es = sorted(x[0] / x[1] for x in rows[v])
    mid = len(gs) // 2
    print(f"f32_variant {v:3d}: median {gs[mid] / 1e3:8.2f} TFLOP/s, {ws[mid]:8.1f} W, {es[mid]:8.2f} GFLOP/s/W  (n = {len(gs)})")
