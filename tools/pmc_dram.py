#!/usr/bin/env python3
"""What rocprofv3 on this box can say about bytes that go to DRAM ("achieved HBM GB/s").

FETCH_SIZE / WRITE_SIZE are derived from ALL of the L2's memory-side requests (rocprofv3 -L: FETCH_SIZE = (TCC_BUBBLE*128 +
(TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B)*64 + TCC_EA0_RDREQ_32B*32) / 1024 -- the gfx94x formula, which tallies gfx950's
128-byte requests at 64 bytes: the guide's "double it").  gfx950 also exposes the requests by SIZE (TCC_EA0_RDREQ_32B / _64B / _128B)
and by DESTINATION (TCC_EA0_RDREQ_DRAM_32B / _GMI_32B / _IO_32B in 32-byte units; TCC_EA0_WRREQ_WRITE_DRAM_32B).  This tool collects
them, in passes of at most 4 TCC counters, for (i) a single-tile float problem whose every operand byte is read exactly once and
whose operands (512 MiB) do not fit the 256 MiB Infinity Cache -- so the byte count that must come from HBM is KNOWN -- and
(ii) BASELINE shapes, and prints exact request bytes, DRAM-destined bytes and the FETCH_SIZE-style figure side by side.

  python tools/pmc_dram.py --out profiles/r06_pmc_dram_side_counters.json
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
DRIVER = '''
import ctypes, sys
sys.path.insert(0, %r)
import gemm_hls_amd as g
L = g.lib()
dtype, n, k, m, reps = %r, %d, %d, %d, %d
es = L.mm_dtype_size(g.DTYPES[dtype])
cfg = g.make_config(dtype)
ptrs = [ctypes.c_void_p() for _ in range(3)]
for p, cnt in zip(ptrs, (n*k, k*m, n*m)):
    g._check(L.mm_alloc(0, cnt*es, ctypes.byref(p)))
g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[0], n*k, 1)); g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[1], k*m, 2))
t = ctypes.c_double(0)
for _ in range(reps):
    g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], n, k, m, ctypes.byref(t)))
print(g.kernel_name(cfg, n, k, m), t.value)
'''
PASSES = [
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    ["TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_RDREQ_GMI_32B_sum", "TCC_EA0_RDREQ_IO_32B_sum", "TCC_EA0_RDREQ_DRAM_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "TCC_EA0_WRREQ_DRAM_sum"],
    ["TCC_BUBBLE_sum", "TCC_EA0_RD_UNCACHED_32B_sum"],
]
SYMBOL = {"float": "mfma_f32_kernel", "half": "mfma_f16_", "double": "mfma_f64_kernel"}


def measure(dtype, n, k, m, workdir, reps=3, env_extra=None):
    vals, name, seconds = {}, None, None
    for i, counters in enumerate(PASSES):
        d = os.path.join(workdir, f"{dtype}_{n}x{k}x{m}_p{i}")
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        script = os.path.join(d, "drv.py")
        open(script, "w").write(DRIVER % (ROOT, dtype, n, k, m, reps))
        r = subprocess.run(["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                            sys.executable, script], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", **(env_extra or {})),
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(f"pass {counters} failed: {(r.stdout + r.stderr)[-600:]}", file=sys.stderr)
            continue
        name, seconds = r.stdout.split()[-2], float(r.stdout.split()[-1])
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if SYMBOL[dtype] in row.get("Kernel_Name", ""):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for c, v in acc.items():
            vals[c] = sum(v) / len(v)
    es = {"float": 4, "half": 2, "double": 8}[dtype]
    g = lambda c: vals.get(c)   # noqa: E731
    out = {"dtype": dtype, "shape": [n, k, m], "kernel_name": name, "unprofiled_launch_ms": None if seconds is None else round(1e3 * seconds, 3), "counters_per_launch": vals,
           "operand_bytes_read_once": (n * k + k * m) * es, "c_bytes": n * m * es}
    if all(g(c) is not None for c in PASSES[0]):
        out["read_bytes_exact_by_request_size"] = g("TCC_EA0_RDREQ_128B_sum") * 128 + g("TCC_EA0_RDREQ_64B_sum") * 64 + g("TCC_EA0_RDREQ_32B_sum") * 32
        out["read_bytes_FETCH_SIZE_formula_x2"] = 2 * ((g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum")) * 64 + g("TCC_EA0_RDREQ_32B_sum") * 32)
    if g("TCC_EA0_RDREQ_DRAM_32B_sum") is not None:
        out["read_bytes_destined_for_dram"] = g("TCC_EA0_RDREQ_DRAM_32B_sum") * 32
        out["read_bytes_destined_for_gmi_io"] = (g("TCC_EA0_RDREQ_GMI_32B_sum") or 0) * 32 + (g("TCC_EA0_RDREQ_IO_32B_sum") or 0) * 32
    if g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") is not None:
        out["write_bytes_destined_for_dram"] = g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") * 32
        out["write_bytes_WRITE_SIZE_formula"] = (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum")) * 32 + g("TCC_EA0_WRREQ_64B_sum") * 64
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_dram.json"))
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "pmc_dram"))
    ap.add_argument("--quick", action="store_true", help="the calibration shape and float 16384^3 only")
    args = ap.parse_args()
    jobs = [("float", 256, 262144, 256, {"MM_F32_VARIANT": "8"}), ("float", 16384, 16384, 16384, None)]
    if not args.quick:
        jobs += [("half", 32768, 32768, 32768, None), ("double", 16384, 16384, 16384, None)]
    res = []
    for dtype, n, k, m, env in jobs:
        res.append(measure(dtype, n, k, m, args.workdir, env_extra=env))
        print(json.dumps(res[-1]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
