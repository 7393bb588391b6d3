#!/usr/bin/env python3
"""MFMA utilisation, effective clock, wait breakdown, LDS conflicts and L2 hit rate of a kernel
family from rocprofv3 PMC passes (each pass only --pmc + --kernel-trace, as gpurun requires).
  python tools/pmc_mfma.py f32|f64|f16|minplus|uint8|split [--size 16384] [--out profiles/r01_pmc_f32.json]
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs); effective clock =
GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md, DVFS note)."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VALU", "SQ_INSTS_LDS"],
]
KERNEL = {"f32": "mfma_f32_", "f64": "mfma_f64_kernel", "f16": "mfma_f16_", "minplus": "valu_tile_", "minplus_f64": "valu_tile_",
          "uint8": "mfma_i8_", "split": "mfma_f32_split_kernel"}
VARIANT = None  # --variant: pins the family's tuning knob (sweep.py --variants) for every pass


def run_pass(counters, what, size, workdir, idx):
    d = os.path.join(workdir, f"pass{idx}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "tools", "sweep.py"), what, "--sizes", str(size), "--reps", "3"]
    if VARIANT is not None:
        cmd += ["--variants", str(VARIANT)]
    elif what == "f32" and os.environ.get("MM_F32_VARIANT"):
        cmd += ["--variants", os.environ["MM_F32_VARIANT"]]
    elif what == "f32":
        cmd += ["--variants", "8"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-1500:] + r.stderr[-1500:])
        raise SystemExit("rocprofv3 pass failed: " + " ".join(counters))
    vals, durs = {}, []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL[what] in row.get("Kernel_Name", ""):
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL[what] in row.get("Kernel_Name", ""):
                durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {k: sum(v) / len(v) for k, v in vals.items()}
    out["_duration_ns"] = sum(durs) / max(1, len(durs))
    return out


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
    ap.add_argument("what", choices=list(KERNEL))
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--out", default=None)
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "pmc_mfma"))
    args = ap.parse_args()
    global VARIANT
    VARIANT = args.variant
    dtype_of = {"f32": "float", "f16": "half", "f64": "double", "uint8": "uint8_t", "minplus": "float", "minplus_f64": "double", "split": "float"}
    res = {"kernel": KERNEL[args.what], "kernel_name": dispatched_kernel_name(args.what, args.size, args.variant), "dtype": dtype_of[args.what],
           "variant": args.variant, "size": args.size, "passes": []}
    flat = {}
    for i, p in enumerate(PASSES):
        r = run_pass(p, args.what, args.size, os.path.join(args.workdir, args.what), i)
        res["passes"].append(r)
        flat.update(r)
    simds = 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in flat and flat.get("GRBM_GUI_ACTIVE"):
        gui = flat["GRBM_GUI_ACTIVE"] / 8.0  # rocprofv3 reports the sum over the 8 XCDs
        res["GRBM_GUI_ACTIVE_per_XCD"] = gui
        res["MfmaUtil_pct"] = 100.0 * flat["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * simds)
        res["effective_clock_GHz_profiled"] = gui / res["passes"][0]["_duration_ns"]
    if flat.get("TCC_HIT_sum") is not None and flat.get("TCC_MISS_sum") is not None:
        res["L2_hit_rate"] = flat["TCC_HIT_sum"] / max(1.0, flat["TCC_HIT_sum"] + flat["TCC_MISS_sum"])
    if flat.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in flat:
                res[k + "_over_WAVE_CYCLES"] = flat[k] / flat["SQ_WAVE_CYCLES"]
    out = args.out or os.path.join(ROOT, "gpurun_out", f"pmc_{args.what}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
