#!/usr/bin/env python3
"""A DRAM-side look at the kernels, from the only place this stack offers one to an unprivileged user: the amdgpu driver's
`mem_busy_percent` (and `gpu_busy_percent`) in sysfs -- the memory controller's activity as the SMU reports it -- sampled every few
milliseconds WHILE a kernel is launched back to back for a couple of seconds.  rocprofv3 on this part has no counter behind the
Infinity Cache (tools/pmc_dram.py: the L2's requests by destination equal all of its requests), so fabric traffic is an upper bound
on HBM traffic; this tool asks the other side.  It is calibrated in the same run on workloads whose HBM bytes per second are KNOWN
and cannot be served by the 256-MiB Infinity Cache: a device-to-device copy of 4 GiB (read + write) and a write-only fill.

  python tools/hbm_busy.py [--seconds 2.0] [--out gpurun_out/hbm_busy.json]
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g, ROOT  # noqa: E402
import torch  # noqa: E402


def sysfs_device():
    buf = ctypes.create_string_buffer(32)
    g._check(g.lib().mm_device_pci_bus_id(0, buf, 32))
    bdf = buf.value.decode().lower()
    for base in (f"/sys/bus/pci/devices/{bdf}",) + tuple(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(base, "mem_busy_percent")):
            return bdf, base
    return bdf, None


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


def sample_while(fn, seconds, base, period=0.005):
    """Run fn() back to back in a worker thread for `seconds` (ctypes / torch release the GIL inside the device calls) and sample the
    two busy figures meanwhile; returns (launches, wall seconds, mem samples, gpu samples)."""
    stop = threading.Event()
    count = [0]

    def worker():
        while not stop.is_set():
            fn()
            count[0] += 1
        torch.cuda.synchronize()

    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = threading.Thread(target=worker)
    t0 = time.perf_counter()
    t.start()
    mem, gpu = [], []
    time.sleep(min(0.3, seconds / 4))                      # let the SMU's averaging window fill with this workload
    while time.perf_counter() - t0 < seconds:
        mem.append(read_int(os.path.join(base, "mem_busy_percent")))
        gpu.append(read_int(os.path.join(base, "gpu_busy_percent")))
        time.sleep(period)
    stop.set()
    t.join()
    wall = time.perf_counter() - t0
    return count[0], wall, [x for x in mem if x is not None], [x for x in gpu if x is not None]


def gemm_job(dtype, mp, rd, n, path=0):
    L = g.lib()
    es = L.mm_dtype_size(g.DTYPES[dtype])
    ptrs = [ctypes.c_void_p() for _ in range(3)]
    for p in ptrs:
        g._check(L.mm_alloc(0, n * n * es, ctypes.byref(p)))
    g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[0], n * n, 1))
    g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[1], n * n, 2))
    cfg = g.make_config(dtype, mp, rd, path)
    t = ctypes.c_double(0)

    def fn():
        g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], n, n, n, ctypes.byref(t)))
    return fn, ptrs, 3.0 * n * n * es, g.kernel_name(cfg, n, n, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hbm_busy.json"))
    args = ap.parse_args()
    bdf, base = sysfs_device()
    out = {"pci_bus_id": bdf, "sysfs": base, "rows": []}
    if base is None:
        out["error"] = "no mem_busy_percent in sysfs for this device"
        print(json.dumps(out))
        json.dump(out, open(args.out, "w"), indent=1)
        return
    dev = torch.device("cuda", 0)

    def row(name, fn, seconds, known_bytes_per_call=None, algorithmic_bytes=None, kernel=None):
        calls, wall, mem, gpu = sample_while(fn, seconds, base)
        r = {"workload": name, "kernel": kernel, "calls": calls, "ms_per_call": round(1e3 * wall / max(calls, 1), 3),
             "mem_busy_percent_mean": round(sum(mem) / len(mem), 2) if mem else None, "mem_busy_percent_max": max(mem) if mem else None,
             "gpu_busy_percent_mean": round(sum(gpu) / len(gpu), 2) if gpu else None, "samples": len(mem)}
        if known_bytes_per_call:
            r["known_hbm_GBps"] = round(known_bytes_per_call * calls / wall / 1e9, 1)
        if algorithmic_bytes:
            r["algorithmic_GBps"] = round(algorithmic_bytes * calls / wall / 1e9, 1)
        out["rows"].append(r)
        print(json.dumps(r), flush=True)
        return r

    # idle
    time.sleep(0.5)
    out["idle_mem_busy_percent"] = read_int(os.path.join(base, "mem_busy_percent"))
    # calibration: 4 GiB device-to-device copy (4 GiB read + 4 GiB written per call; far beyond the Infinity Cache), and a write-only fill
    src = torch.empty(1 << 30, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    src.fill_(1.0)
    cal_copy = row("calibration: d2d copy 4 GiB (read + write)", lambda: dst.copy_(src), args.seconds, known_bytes_per_call=2.0 * src.numel() * 4)
    cal_fill = row("calibration: fill 4 GiB (write only)", lambda: dst.fill_(2.0), args.seconds, known_bytes_per_call=1.0 * src.numel() * 4)
    half = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    row("calibration: d2d copy 1 GiB (read + write)", lambda: half.copy_(src[: 1 << 28]), args.seconds, known_bytes_per_call=2.0 * half.numel() * 4)
    del src, dst, half
    torch.cuda.empty_cache()
    for name, (dtype, mp, rd, n, path) in {"C2 float 16384^3": ("float", "Multiply", "Add", 16384, 0), "C3 half 32768^3": ("half", "Multiply", "Add", 32768, 0),
                                           "C4 double 16384^3": ("double", "Multiply", "Add", 16384, 0), "C5b float (Add,Min) 8192^3": ("float", "Add", "Min", 8192, 0),
                                           "uint8 32768^3": ("uint8_t", "Multiply", "Add", 32768, 0)}.items():
        fn, ptrs, alg, kernel = gemm_job(dtype, mp, rd, n, path)
        row(name, fn, args.seconds, algorithmic_bytes=alg, kernel=kernel)
        for p in ptrs:
            g.lib().mm_free(0, p)
    # busy % -> GB/s through the copy calibration (if the figure is a linear activity measure at all -- the rows above say)
    if cal_copy["mem_busy_percent_mean"]:
        k = cal_copy["known_hbm_GBps"] / cal_copy["mem_busy_percent_mean"]
        out["GBps_per_busy_percent_from_copy"] = round(k, 1)
        out["GBps_per_busy_percent_from_fill"] = round(cal_fill["known_hbm_GBps"] / cal_fill["mem_busy_percent_mean"], 1) if cal_fill["mem_busy_percent_mean"] else None
        for r in out["rows"]:
            if r.get("kernel") and r["mem_busy_percent_mean"] is not None:
                r["hbm_GBps_estimate_from_busy_percent"] = round(k * r["mem_busy_percent_mean"], 1)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}))


if __name__ == "__main__":
    main()
