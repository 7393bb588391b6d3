#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE on THIS kernel's access pattern, as MI355X_MICROARCH.md asks
("calibrate on a known byte count in your own access pattern before trusting an absolute").
A single-tile problem (N = M = 256, K large) reads every byte of A and B exactly once and nothing
can be re-read: known bytes = (N*K + K*M)*4.  Run for the BK=16 kernel (A fetched as 64-B row
segments, B as 1-KiB rows) and the BK=32 kernel (A as 128-B segments)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = '''
import ctypes, os, sys
sys.path.insert(0, %r)
import gemm_hls_amd as g
L = g.lib()
n, k, m = %d, %d, %d
cfg = g.make_config("float")
ptrs = [ctypes.c_void_p() for _ in range(3)]
for p, cnt in zip(ptrs, (n*k, k*m, n*m)):
    g._check(L.mm_alloc(0, cnt*4, ctypes.byref(p)))
g._check(L.mm_fill_device(0, 0, ptrs[0], n*k, 1)); g._check(L.mm_fill_device(0, 0, ptrs[1], k*m, 2))
t = ctypes.c_double(0)
for _ in range(3):
    g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], n, k, m, ctypes.byref(t)))
print(t.value)
'''


def measure(variant, n, k, m, workdir):
    d = os.path.join(workdir, f"cal_v{variant}_{n}x{k}x{m}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    script = os.path.join(d, "drv.py")
    open(script, "w").write(DRIVER % (ROOT, n, k, m))
    env = dict(os.environ, TMPDIR="/tmp", MM_F32_VARIANT=str(variant))
    r = subprocess.run(["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "-d", d, "-o", "cal", "--output-format",
                        "csv", "--", sys.executable, script], cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-1000:] + r.stderr[-1000:])
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mfma_f32_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "FETCH_SIZE":
                vals.append(float(row["Counter_Value"]))
    raw = sum(vals) / len(vals) * 1024.0
    known = (n * k + k * m) * 4.0
    return {"variant": variant, "shape": [n, k, m], "FETCH_SIZE_bytes_raw": raw, "known_read_bytes": known,
            "known_over_raw": known / raw}


def main():
    workdir = os.path.join(ROOT, "gpurun_out", "pmc_cal")
    out = []
    for variant, shape in [(8, (256, 262144, 256)), (10, (256, 262144, 256)), (8, (2048, 32768, 2048))]:
        out.append(measure(variant, *shape, workdir))
        print(json.dumps(out[-1]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fetch_calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
