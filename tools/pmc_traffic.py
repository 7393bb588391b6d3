#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes, as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
  * FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots: FETCH_SIZE costs 3,
    WRITE_SIZE 2), never combined with trace domains other than kernel-trace;
  * both are reported in KiB;
  * FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced streams -> doubled
    here (this kernel reads only through global_load_lds_dwordx4, 16 B per lane);
  * WRITE_SIZE is uncalibrated on gfx950; it is checked against the one quantity we know exactly,
    the C matrix (N*M*4 bytes, each byte written once), and reported as measured.
Run on the GPU box:  python tools/pmc_traffic.py --out profiles/r01_traffic.json
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


WHAT = {"f32": ("mfma_f32_kernel", 4, "float"), "f16": ("mfma_f16_", 2, "half"), "f64": ("mfma_f64_kernel", 8, "double"),
        "uint8": ("mfma_i8_", 1, "uint8_t"), "minplus": ("valu_tile_", 4, "float (Add,Min)"), "minplus_f64": ("valu_tile_", 8, "double (Add,Min)"),
        "split": ("mfma_f32_split_kernel", 4, "float via MM_PATH_SPLIT (GEMM kernel only; the pre-pass moves 10 B per element of A and B on top)")}


def run_pass(counter, size, workdir, what="f32", variant=None):
    d = os.path.join(workdir, counter)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "tools", "sweep.py"), what, "--sizes", str(size), "--reps", "3"]
    if variant is not None:
        cmd += ["--variants", str(variant)]
    elif what == "f32":
        cmd += ["--variants", os.environ.get("MM_F32_VARIANT", "8")]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
        raise SystemExit(f"rocprofv3 pass for {counter} failed")
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if WHAT[what][0] in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for the mfma kernel under {d}")
    return vals


def dispatched_kernel_name(what, size, variant):
    """mm_kernel_name() of the launch the passes profile (asked in a child process: the knob is process state)."""
    table = {"f32": ("float", "Multiply", "Add", "f32_variant", 0), "f16": ("half", "Multiply", "Add", "f16_variant", 0),
             "f64": ("double", "Multiply", "Add", "f64_variant", 0), "uint8": ("uint8_t", "Multiply", "Add", "i8_variant", 0),
             "minplus": ("float", "Add", "Min", "valu_variant", 0), "minplus_f64": ("double", "Add", "Min", "valu_variant", 0), "split": ("float", "Multiply", "Add", "split_variant", 2)}
    dtype, mp, rd, knob, path = table[what]
    code = ("import gemm_hls_amd as g\n"
            f"v = {variant!r}\n"
            f"if v is not None: g.set_tuning({knob!r}, v)\n"
            f"print(g.kernel_name(g.make_config({dtype!r}, {mp!r}, {rd!r}, {path}), {size}, {size}, {size}))\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    return r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--what", choices=list(WHAT), default="f32")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_traffic.json"))
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    args = ap.parse_args()
    fetch = run_pass("FETCH_SIZE", args.size, args.workdir, args.what, args.variant)
    write = run_pass("WRITE_SIZE", args.size, args.workdir, args.what, args.variant)
    n = args.size
    es = WHAT[args.what][1]
    fetch_kib = sum(fetch) / len(fetch)
    write_kib = sum(write) / len(write)
    res = {
        "kernel": WHAT[args.what][0], "kernel_name": dispatched_kernel_name(args.what, n, args.variant),
        "dtype": {"f32": "float", "f16": "half", "f64": "double", "uint8": "uint8_t", "minplus": "float", "minplus_f64": "double", "split": "float"}[args.what], "workload": {"f32": "float"}.get(args.what, args.what), "variant": args.variant,
        "shape": [n, n, n], "launches_profiled": len(fetch),
        "FETCH_SIZE_KiB_raw_per_launch": fetch_kib, "WRITE_SIZE_KiB_raw_per_launch": write_kib,
        "fetch_bytes_corrected": 2.0 * fetch_kib * 1024.0,  # gfx950: 128-B requests tallied as 64 B
        "write_bytes": write_kib * 1024.0,
        "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
        "algorithmic_bytes_compulsory": 3.0 * n * n * es,
        "c_bytes_exact": 1.0 * n * n * es,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; WRITE_SIZE as reported "
                "(compare with c_bytes_exact for its calibration)",
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
