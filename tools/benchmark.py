#!/usr/bin/env python3
"""Benchmark driver with the contract of the reference's `scripts/build_manager.py benchmark` /
`extract_benchmarks` (:578-669): run `RunHardware*.exe N K M hw off` `repetitions` times per
configuration, parse `([\\d\\.]+) seconds[^\\d]+([\\d\\.]+) GOp/s`, append rows to benchmark.csv.
The FPGA columns (tile sizes, frequency, resources) are replaced by what identifies a GPU
configuration: data type, map/reduce op, sizes, kernel name.  Power (the reference's PSU meter,
host/RunHardware.cpp:156-185) is sampled from `rocm-smi --showpower` around each run when
available, otherwise left empty.

  python tools/benchmark.py [--repetitions 3] [--configs baseline] [--out benchmark.csv]
"""
import argparse
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PERF = re.compile(r"([\d\.]+) seconds[^\d]+([\d\.]+) GOp/s")
KERNEL = re.compile(r"Executing kernel \(([^)]+)\)")
BASELINE_CONFIGS = [  # BASELINE.json configs that fit one GPU
    ("float", "Multiply", "Add", 16384, 16384, 16384),
    ("half", "Multiply", "Add", 32768, 32768, 32768),
    ("double", "Multiply", "Add", 16384, 16384, 16384),
    ("float", "Add", "Min", 8192, 8192, 8192),
]
QUICK_CONFIGS = [(t, m, r, 2048, 2048, 2048) for (t, m, r, *_rest) in BASELINE_CONFIGS]


def power_watts():
    try:
        out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"Power \(W\):\s*([\d\.]+)", out)
        return float(m.group(1)) if m else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repetitions", type=int, default=3)
    ap.add_argument("--configs", choices=["baseline", "quick"], default="baseline")
    ap.add_argument("--out", default="benchmark.csv")
    ap.add_argument("--timeout", type=float, default=600)
    args = ap.parse_args()
    configs = BASELINE_CONFIGS if args.configs == "baseline" else QUICK_CONFIGS
    with open(args.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["data_type", "map_op", "reduce_op", "size_n", "size_k", "size_m", "kernel", "time", "performance",
                    "power", "power_efficiency"])
        for (dt, mp, rd, n, k, m) in configs:
            exe = os.path.join(ROOT, "bin", f"RunHardware_{dt}_{mp}_{rd}.exe")
            done = timeouts = 0
            while done < args.repetitions:
                print(f"Running {dt} {mp}/{rd} {n}x{k}x{m}, iteration {done + 1} / {args.repetitions}...", flush=True)
                try:
                    r = subprocess.run([exe, str(n), str(k), str(m), "hw", "off"], capture_output=True, text=True,
                                       timeout=args.timeout)
                except subprocess.TimeoutExpired:
                    timeouts += 1
                    if timeouts > 10:
                        print("exceeded maximum number of timeouts. Skipping.")
                        break
                    continue
                if r.returncode != 0:
                    raise SystemExit(f"{exe}: kernel execution failed.\n{r.stdout}{r.stderr}")
                mt = PERF.search(r.stdout)
                kn = KERNEL.search(r.stdout)
                watts = power_watts()
                w.writerow([dt, mp, rd, n, k, m, kn.group(1) if kn else "", mt.group(1), mt.group(2),
                            "" if watts is None else watts,
                            "" if watts is None else float(mt.group(2)) / watts])
                f.flush()
                done += 1
    print(open(args.out).read())


if __name__ == "__main__":
    main()
