#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config.

N = 1 (default):  fp32 GEMM N=K=M=16384 on one MI355X ("float 16384x16384x16384 on 1 MI355X, MFMA
                  fp32, LDS-tiled outer product", BASELINE configs[1]): GFLOP/s and fraction of the
                  fp32 MFMA peak.  The same JSON line carries a "workloads" array with the other
                  single-GPU BASELINE configs (half 32768^3, double 16384^3, min-plus 8192^3, plus
                  uint8 32768^3 and the 65536 x 16384 x 16384 job on ONE GPU), a few steps each, so
                  that they are timed by the driver too.
N > 1:            one process per GPU (python -m torch.distributed.run ...), rows of C split over
                  the ranks with gemm_hls_amd.partition.row_slab, B replicated, NO data-path
                  collective (the control plane -- barrier and max over ranks -- is the only
                  communication).  --scaling strong (default): the FIXED job BASELINE configs[4]
                  names, float 65536 x 16384 x 16384, split along N across the G ranks (8192 rows
                  per GPU at G = 8).  --scaling weak: every rank owns a 16384-row slab (the job
                  grows with G).  The strong run also reports the weak figure ("weak_scaling").

A step = one pass of the hot path (one mm_gemm_enqueue through the C ABI) over operands that are
already resident in HBM.  Timing = W untimed steps, barrier + synchronize, EXACTLY K steps,
synchronize + barrier, MAX over ranks.  GOp/s = 1e-9 * 2*N*K*M / t as host/RunHardware.cpp:174-180.

Besides the contract keys the line carries
  roofline     the dominant kernel against its roof, from HIP events on the launch stream
  cpu_baseline the reference's OWN hlslib CPU-simulation path (oracle/_ref, compiled from the
               reference's kernel sources) timed on this host's cores on float 1024^3 (BASELINE C1)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SIZE = 16384
C5A_ROWS = 65536  # BASELINE configs[4]: float 65536 x 16384 x 16384 split along N
# Roofs.  MFMA peaks: /opt/skills/guides/MI355X_MICROARCH.md (dense); fp64 78.6 TF is the datasheet
# figure, confirmed by tools/probes/probe_mfma_rate.hip (64 cycles per v_mfma_f64_16x16x4_f64).
# min-plus: SURVEY.md 8(d)'s VALU ceiling, 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s
# counting add + min as 2 operations; tools/probes/probe_valu_rate.hip measures the instruction mix
# the kernel actually issues (2 v_add + 1 v_min3 per 4 operations) at the delivered clock.
#            dtype     map         reduce  size   roof  peak (T op/s)
WORKLOADS = {
    "float": ("float", "Multiply", "Add", 16384, "mfma", 157.3),
    "half": ("half", "Multiply", "Add", 32768, "mfma", 2500.0),
    "double": ("double", "Multiply", "Add", 16384, "mfma", 78.6),
    "minplus": ("float", "Add", "Min", 8192, "valu", 78.6),
    # the same semiring on an 8-byte Data_t (round 3: the DMA-staged VALU kernel serves every element size).  fp64 VALU
    # instructions issue at half the fp32 rate: 1024 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3 T lane-ops/s, add and min = 2 ops
    "minplus_f64": ("double", "Add", "Min", 8192, "valu", 39.3),
    "uint8": ("uint8_t", "Multiply", "Add", 32768, "mfma", 5000.0),
    # MM_PATH_SPLIT (opt-in): the same fp32 problem on the bf16 matrix cores, six bf16 MFMAs per 16-deep fp32
    # multiply-add block -> the roof for ALGORITHMIC fp32 flops is the bf16 dense peak / 6
    "float_split": ("float", "Multiply", "Add", 16384, "mfma", round(2500.0 / 6.0, 1)),
    # BASELINE C3's problem in the REFERENCE's half arithmetic (binary16 products accumulated in binary16, k ascending:
    # kernel/Compute.cpp:129-133 -- what its hosts compare with exactly), on the k-ordered register-tile kernel ("ordered_tile";
    # MM_PATH_ORDERED == what half_contract = reference routes MM_PATH_AUTO to).  VALU-bound: one v_pk_mul_f16 + one v_pk_add_f16 per
    # two multiply-adds, each two elements per lane at 4-cycle issue = 128 operations per clock per CU = 78.6 TOp/s
    # (tools/probes/probe_valu_rate.hip measures the pair at 70.8: profiles/r06a_probe_valu_issue_rates_unfused_pairs.txt)
    "half_exact": ("half", "Multiply", "Add", 32768, "valu", 78.6),
}
PATHS = {"float_split": 2, "half_exact": 1}  # workload -> mm_path_t (default MM_PATH_AUTO)
STEPS = {"half_exact": (3, 1)}               # workload -> (timed steps, warm-up) of the extras leg when not 5 / 2: one step takes ~0.95 s
DTYPE_TAG = {"float": "f32", "half": "f16 (f32 accumulate)", "double": "f64", "uint8_t": "u8 (i32 accumulate)"}
# Context for the two power-limited workloads (not a roof this file prices against: `peak` stays the
# guide's dense MFMA peak): what the matrix cores sustain from registers alone, no LDS and no memory, on
# different random operands per MFMA at this board's power limit -- for the instruction the SHIPPED kernel issues
# (v_mfma_f32_16x16x32_f16 / v_mfma_i32_16x16x64_i8 over the kernel's 128 x 64 wavefront tile, two waves per SIMD:
# tools/probes/probe_mfma_power.hip).  Round 2's figures for the retired 32x32 forms were 1778 / 3532.
POWER_CEILING = {
    "half": {"register_only_mfma_on_random_operands_TOps": 2060.0, "clock_GHz": 1.98, "instruction": "v_mfma_f32_16x16x32_f16",
             "source": "profiles/r03b_probe_mfma_power_by_shape_and_operand_order.txt"},
    "uint8_t": {"register_only_mfma_on_random_operands_TOps": 4090.0, "clock_GHz": 1.97, "instruction": "v_mfma_i32_16x16x64_i8",
                "source": "profiles/r03b_probe_mfma_power_by_shape_and_operand_order.txt"},
}
# MM_PATH_SPLIT runs v_mfma_f32_32x32x16_bf16 (same probe file)
SPLIT_BF16_REGISTER_ONLY_TOPS = 1930.0


# What a loop of nothing but independent MFMAs on registers sustains on this part (same probe, constant operands):
# the clock the chip holds under that load, not the nominal 2.4 GHz the `peak` column assumes.
MFMA_SUSTAINED = {
    "float": {"register_only_mfma_TOps": 155.5, "clock_GHz": 2.374,
              "source": "profiles/r02s_probe_mfma_issue_rates_and_power_ceiling.txt"},
    "double": {"register_only_mfma_TOps": 78.1, "clock_GHz": 2.382,
               "source": "profiles/r02s_probe_mfma_issue_rates_and_power_ceiling.txt"},
}


# What the VALU sustains on the fp64 min-plus instruction pair (tools/probes/probe_valu_rate.hip, cycles measured with s_memtime):
# v_add_f64 / v_min_f64 issue every 4.26 cycles at 4 waves per SIMD (4.51 at the 2 waves this kernel's 128 accumulator
# registers allow), not every 4.0 as the nominal 39.3 TOp/s `peak` assumes.
VALU_SUSTAINED = {
    "minplus_f64": {"probe_mix_TOps_4_waves_per_simd": 35.2, "probe_mix_TOps_2_waves_per_simd": 32.7, "clock_GHz": 2.36,
                    "source": "profiles/r03s_probe_valu_issue_rates_incl_f64.txt"},
}


def cpu_baseline(sample_n=1024):
    """The repo's own hlslib simulation path (reference kernel sources + test-only shim) on
    float sample_n^3; this is the checker's side of the house, never the product."""
    import _oracle
    # the reference's dataflow graph is 32 ProcessingElement threads + 7 data movers (kernel/Top.cpp:67-116)
    info = {"value": None, "unit": "GFLOP/s", "cores": min(39, os.cpu_count() or 1), "host_cores": os.cpu_count(),
            "kind": "reference",
            "sample": f"float {sample_n}x{sample_n}x{sample_n} (BASELINE config C1), one call of the reference's "
                      "MatrixMultiplicationKernel compiled from /root/reference/kernel/*.cpp against "
                      "oracle/hlslib_shim: 32 ProcessingElement threads + 7 data movers"}
    try:
        with open("/proc/cpuinfo") as f:
            info["cpu_model"] = next((line.split(":", 1)[1].strip() for line in f if line.startswith("model name")), None)
    except OSError:
        info["cpu_model"] = None
    a, b = _oracle.fill("float", sample_n, sample_n, sample_n)
    # The yardsticks FIRST, on a quiet host (timed right after the 39-thread simulation, and with the BLAS's
    # first-call thread-pool start-up inside the clock, they read 36.9 GFLOP/s on the driver's box against ~1 700 on the builder's):
    # the BLAS reference (ReferenceImplementation, include/Utility.h:76-89) on the same sample -- one untimed call, then the BEST of 5
    import numpy as np
    np.matmul(a, b)
    best = min(_timed(lambda: np.matmul(a, b)) for _ in range(5))
    info["blas_sgemm_gflops_same_sample"] = round(2.0 * sample_n ** 3 / best / 1e9, 1)
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        info["blas_threads"] = pools[0].get("num_threads") if pools else None
        info["blas_library"] = (pools[0].get("internal_api") or "") + " " + str(pools[0].get("version") or "") if pools else None
    except Exception:
        info["blas_threads"] = None
    # and the semantic spec, Naive (include/Utility.h:18-42), single-threaded: best of 2
    best = min(_timed(lambda: _oracle.naive("float", "Multiply", "Add", a, b, threads=1)) for _ in range(2))
    info["naive_1thread_gflops_same_sample"] = round(2.0 * sample_n ** 3 / best / 1e9, 2)
    if not _oracle.ref_available():
        info["sample"] += " -- oracle/_ref not built here, not measured"
        return info
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    sys.stdout.flush()
    os.dup2(devnull, 1)  # WriteC prints one progress line per tile (kernel/Memory.cpp:384-389)
    try:
        t0 = time.perf_counter()
        _oracle.ref_kernel("float", "Multiply", "Add", a, b)
        dt = time.perf_counter() - t0
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
    info["value"] = round(2.0 * sample_n ** 3 / dt / 1e9, 4)
    info["seconds"] = round(dt, 3)
    return info


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def replayed_counters(kernel_name, shape, dtype=None):
    """Counters of the kernel this run dispatched, REPLAYED from committed rocprofv3 PMC files (the counter passes
    serialise and slow the kernel, so they are collected separately: tools/pmc_traffic.py = FETCH_SIZE / WRITE_SIZE in
    separate --pmc passes with the guide's gfx950 correction, tools/pmc_mfma.py = SQ / GRBM / TCC passes).  A file is
    used only if it says it profiled THIS kernel (`kernel_name` == mm_kernel_name of the run) on THIS shape; the
    newest matching file of each kind wins.  Returns {} when nothing matches -- never another kernel's numbers."""
    import glob
    out = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*.json")))  # rNN[x]_ prefixes sort by round
    for path in reversed(files):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if not isinstance(d, dict) or d.get("kernel_name") != kernel_name:
            continue
        dshape = list(d["shape"]) if "shape" in d else [d.get("size")] * 3
        if dshape != list(shape):
            continue
        # one kernel NAME can serve several element types (valu_tile is the generic-semiring family): a file that says
        # which Data_t it profiled only speaks for that one; files from before the field existed profiled float
        fdtype = d.get("dtype") or ("float" if kernel_name == "valu_tile" else None)
        if dtype is not None and fdtype is not None and fdtype != dtype:
            continue
        rel = os.path.relpath(path, ROOT)
        if "hbm_bytes_per_launch" in d and "traffic" not in out:
            out["traffic"] = d["hbm_bytes_per_launch"]
            out["traffic_source"] = rel
        if "MfmaUtil_pct" in d and "mfma_util_pct" not in out:
            out["mfma_util_pct"] = round(d["MfmaUtil_pct"], 2)
            out["profiled_clock_GHz"] = round(d.get("effective_clock_GHz_profiled", 0.0), 3)
            if "L2_hit_rate" in d:
                out["l2_hit_rate"] = round(d["L2_hit_rate"], 4)
            out["mfma_util_source"] = rel
    return out


# Counters MEASURED IN THIS RUN.  key -> (mm_kernel_name the passes must be profiling, tools/sweep.py arguments, substring of the
# kernel's symbol in rocprofv3's CSVs, has an MFMA pass).  The passes run tools/sweep.py -- the same C-ABI launch on device-filled operands -- as children.
LIVE = {
    "float": ("mfma_f32_256x256x16_w8_flush4096", ["f32", "--variants", "8"], "mfma_f32_kernel", True),
    "half": ("mfma_f16_256x256_pingpong_16x16x32", ["f16"], "mfma_f16_", True),
    "double": ("mfma_f64_256x128x16_w8", ["f64"], "mfma_f64_kernel", True),
    "minplus": ("valu_tile", ["minplus"], "valu_tile_", False),
    "minplus_f64": ("valu_tile", ["minplus_f64"], "valu_tile_", False),
    "uint8": ("mfma_i8_256x256_pingpong_16x16x64", ["uint8"], "mfma_i8_", True),
}
LIVE_KERNEL = LIVE["float"][0]
DRAM_SIDE_PASS = ("float", "half")   # BASELINE C2 / C3: one more pass each, the L2's requests by destination


LIVE_STATE = {"deadline": None, "dead": False}   # one wall-clock budget for ALL counter passes of a run; the first failure ends them
LIVE_BUDGET_S = 170.0


def live_counters(kernel_name, shape, avg_launch_ms, key="float", timeout=75):
    """rocprofv3 child passes on the same shape after the timed legs (they serialise and slow the kernel): FETCH_SIZE, WRITE_SIZE in
    separate --pmc passes (gfx950 correction: fetch doubled, MI355X_MICROARCH.md HBM section) and, for the matrix-core kernels,
    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -- each with --kernel-trace only.  Returns None when rocprofv3 is absent, the kernel
    is not the one the passes profile, or a pass fails: the caller then keeps the replayed figures."""
    import shutil
    import tempfile
    want, sweep_args, symbol, has_mfma = LIVE[key]
    size = shape[0]
    if (os.environ.get("MM_BENCH_NO_PMC") or shutil.which("rocprofv3") is None or kernel_name != want or list(shape) != [size] * 3
            or size != WORKLOADS[key][3]):
        return None
    if LIVE_STATE["dead"]:
        return None
    if LIVE_STATE["deadline"] is None:
        LIVE_STATE["deadline"] = time.perf_counter() + LIVE_BUDGET_S
    workdir = tempfile.mkdtemp(prefix="mm_bench_pmc_", dir="/tmp")
    try:
        import csv
        import glob
        import subprocess

        def one_pass(counters, tag):
            left = LIVE_STATE["deadline"] - time.perf_counter()
            if left < 10.0:
                raise RuntimeError("the run's budget for counter passes is used up")
            d = os.path.join(workdir, tag)
            os.makedirs(d, exist_ok=True)
            cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "tools", "sweep.py"), sweep_args[0], "--sizes", str(size), "--reps", "2", *sweep_args[1:]]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=min(timeout, left))
            if r.returncode != 0:
                raise RuntimeError(f"rocprofv3 {tag}: rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}")
            vals, durs = {}, []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if symbol in row.get("Kernel_Name", ""):
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if symbol in row.get("Kernel_Name", ""):
                        durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            if not vals:
                raise RuntimeError(f"rocprofv3 {tag}: no counter rows for the kernel")
            return {k: sum(v) / len(v) for k, v in vals.items()}, (sum(durs) / len(durs) if durs else None), max(len(v) for v in vals.values())

        t0 = time.perf_counter()
        fetch, _, launches = one_pass(["FETCH_SIZE"], "fetch")
        write, _, _ = one_pass(["WRITE_SIZE"], "write")
        traffic = 2.0 * fetch["FETCH_SIZE"] * 1024.0 + write["WRITE_SIZE"] * 1024.0
        out = {"traffic": int(traffic), "fetch_bytes_corrected": int(2.0 * fetch["FETCH_SIZE"] * 1024.0), "write_bytes": int(write["WRITE_SIZE"] * 1024.0),
               "achieved_fabric_GBps": round(traffic / (1e-3 * avg_launch_ms) / 1e9, 1),
               "counters_measured_in_this_run": True, "launches_profiled_per_pass": launches}
        if key in DRAM_SIDE_PASS:
            # the L2's memory-side requests BY DESTINATION (32-byte units): local memory vs GMI / IO.  This is as far towards the
            # HBM as rocprofv3 -L goes on this part -- see notes.counters and profiles/r06b_pmc_dram_side_counters.json
            try:
                dram, _, _ = one_pass(["TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"], "dram")
                db = 32.0 * (dram["TCC_EA0_RDREQ_DRAM_32B_sum"] + dram["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"])
                out["dram_destined_bytes"] = int(db)
                out["achieved_dram_destined_GBps"] = round(db / (1e-3 * avg_launch_ms) / 1e9, 1)
            except Exception as exc:   # a soft pass: the figures above stand without it
                out["dram_destined_bytes"] = None
                sys.stderr.write(f"[bench] DRAM-destination counters for {key} unavailable: {exc!r}\n")
        if has_mfma:
            busy, dur_ns, _ = one_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], "mfma")
            gui = busy["GRBM_GUI_ACTIVE"] / 8.0          # rocprofv3 sums the 8 XCDs
            out["mfma_util_pct"] = round(100.0 * busy["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024), 2)
            if dur_ns:
                out["profiled_clock_GHz"] = round(gui / dur_ns, 3)
                out["profiled_launch_ms"] = round(dur_ns * 1e-6, 3)
        out["counter_passes_s"] = round(time.perf_counter() - t0, 1)
        return out
    except Exception as exc:   # a counter pass must never take the line down with it -- nor hold it up: the first failure ends all passes
        LIVE_STATE["dead"] = True
        sys.stderr.write(f"[bench] live counters for {key} unavailable (no further passes in this run): {exc!r}\n")
        return None
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


class HbmBusy:
    """The DRAM-side figure ("achieved HBM GB/s").  rocprofv3 on this part counts nothing behind the Infinity Cache, but
    the amdgpu driver exports the memory controller's activity as the SMU reports it: /sys/bus/pci/devices/<bdf>/mem_busy_percent.  It is a
    LINEAR measure of HBM bytes per second -- a 4-GiB device copy (4.9 TB/s known) and a 4-GiB fill (6.9 TB/s known) give the same 82.0 GB/s
    per percent, i.e. 100 % = 8.2 TB/s (profiles/r06g_hbm_busy_percent_calibrated_copy_fill_and_kernels.json) -- so sampled while a kernel
    is launched back to back, and calibrated IN THIS RUN on the same two known streams, it says what the kernel pulls from HBM (resolution:
    one percent = 82 GB/s).  Untimed legs after the timed regions; None wherever sysfs does not offer the file."""

    def __init__(self, g, torch, dev, local_rank):
        self.torch, self.dev, self.path, self.k, self.cal = torch, dev, None, None, None
        if os.environ.get("MM_BENCH_NO_HBM_PROBE"):
            return
        try:
            buf = ctypes.create_string_buffer(32)
            g._check(g.lib().mm_device_pci_bus_id(local_rank, buf, 32))
            path = f"/sys/bus/pci/devices/{buf.value.decode().lower()}/mem_busy_percent"
            if self._read(path) is not None:
                self.path = path
        except Exception:
            pass

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def busy(self, launch, seconds=1.0, depth=4):
        """Mean mem_busy_percent over ~`seconds` of back-to-back `launch()`es (at most `depth` in flight), the first 0.3 s discarded."""
        import threading
        samples, stop = [], threading.Event()

        def sampler():
            t0 = time.perf_counter()
            while not stop.is_set():
                v = self._read(self.path)
                if v is not None and time.perf_counter() - t0 > 0.3:
                    samples.append(v)
                time.sleep(0.005)

        th = threading.Thread(target=sampler)
        th.start()
        t0, calls = time.perf_counter(), 0
        try:
            while time.perf_counter() - t0 < seconds or calls < depth:
                for _ in range(depth):
                    launch()
                calls += depth
                self.torch.cuda.synchronize()
        finally:
            stop.set()
            th.join()
        wall = time.perf_counter() - t0
        return (sum(samples) / len(samples) if samples else None), calls, wall

    def calibrate(self):
        if self.path is None or self.k is not None:
            return
        torch = self.torch
        src = torch.empty(1 << 30, dtype=torch.float32, device=self.dev)      # 4 GiB: far beyond the 256-MiB Infinity Cache
        dst = torch.empty_like(src)
        src.fill_(1.0)
        pc, calls, wall = self.busy(lambda: dst.copy_(src), 0.8, depth=16)
        copy_gbps = 2.0 * src.numel() * 4 * calls / wall / 1e9
        pf, calls, wall = self.busy(lambda: dst.fill_(2.0), 0.8, depth=16)
        fill_gbps = 1.0 * src.numel() * 4 * calls / wall / 1e9
        del src, dst
        torch.cuda.empty_cache()
        if pc and pf:
            self.k = 0.5 * (copy_gbps / pc + fill_gbps / pf)
            self.cal = {"copy_4GiB": {"known_GBps": round(copy_gbps, 1), "mem_busy_pct": round(pc, 2), "GBps_per_pct": round(copy_gbps / pc, 1)},
                        "fill_4GiB": {"known_GBps": round(fill_gbps, 1), "mem_busy_pct": round(pf, 2), "GBps_per_pct": round(fill_gbps / pf, 1)}}

    def attach(self, roofline, launch, depth=4):
        """Adds the DRAM-side figures of the kernel behind `launch` to a roofline object (`depth` launches in flight at a time)."""
        if self.path is None:
            return
        try:
            self.calibrate()
            pct, _, _ = self.busy(launch, 1.0, depth)
            if pct is None or self.k is None:
                return
            roofline["hbm_busy_pct"] = round(pct, 2)
            roofline["achieved_HBM_GBps"] = round(self.k * pct, 1)
            roofline["achieved_HBM_frac_of_8TBps"] = round(self.k * pct / 8000.0, 4)
            if roofline.get("avg_launch_ms"):      # per launch, like `traffic`: what one launch pulls from / pushes to HBM
                roofline["hbm_bytes_per_launch"] = int(self.k * pct * 1e9 * 1e-3 * roofline["avg_launch_ms"])
        except Exception as exc:          # never the line's problem
            sys.stderr.write(f"[bench] HBM busy probe failed: {exc!r}\n")


def attach_replayed(roofline, kernel_name, shape, dtype=None):
    """Adds the replayed counters to a roofline object; `achieved_fabric_GBps` = replayed bytes per launch over the
    launch time measured live in this run."""
    rc = replayed_counters(kernel_name, shape, dtype)
    if not rc:
        roofline["traffic_note"] = "no committed PMC file names this kernel at this shape: not replayed from another kernel"
        return roofline
    if roofline.get("bound") != "mfma":            # a VALU kernel has no MFMA utilisation to report
        for key in ("mfma_util_pct", "mfma_util_source"):
            rc.pop(key, None)
    roofline.update(rc)
    if "traffic" in rc:
        roofline["achieved_fabric_GBps"] = round(rc["traffic"] / (1e-3 * roofline["avg_launch_ms"]) / 1e9, 1)
    roofline["counters_measured_in_this_run"] = False      # replayed: see the top-level "notes"
    return roofline


def timed_steps(g, torch, a, b, c, dtype, map_op, reduce_op, steps, warmup, barrier, path=0, transposed_a=False):
    """W untimed steps, barrier + synchronize, EXACTLY `steps` steps, synchronize + barrier.
    Returns (wall seconds of the timed region, sorted per-launch ms from events on the launch stream)."""
    def step():
        g.matmul(a, b, dtype, map_op, reduce_op, path=path, transposed_a=transposed_a, out=c)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()  # same stream as the launch: per-launch duration without host gaps
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))


def alloc_fill(g, torch, dev, local_rank, dtype, rows, k, m, seed_a, seed_b):
    tdt = g.torch_dtype(dtype)
    a = torch.empty((rows, k), dtype=tdt, device=dev)
    b = torch.empty((k, m), dtype=tdt, device=dev)
    c = torch.empty((rows, m), dtype=tdt, device=dev)
    L = g.lib()
    # synthetic operands of the reference's distribution, generated in HBM
    g._check(L.mm_fill_device(local_rank, g.DTYPES[dtype], a.data_ptr(), a.numel(), seed_a))
    g._check(L.mm_fill_device(local_rank, g.DTYPES[dtype], b.data_ptr(), b.numel(), seed_b))
    torch.cuda.set_device(local_rank)
    return a, b, c


def roofline_obj(dtype, roof, peak, rows, k, m, launch_ms):
    avg_s = 1e-3 * sum(launch_ms) / len(launch_ms)
    achieved = 2.0 * rows * k * m / avg_s / 1e12
    es = {"float": 4, "half": 2, "double": 8, "uint8_t": 1}[dtype]
    return {"bound": roof, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s" if roof == "mfma" and dtype != "uint8_t" else "TOp/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "algorithmic_flops_per_launch": 2.0 * rows * k * m,
            "algorithmic_bytes_per_launch": float((rows * k + k * m + rows * m) * es),
            "avg_launch_ms": round(1e3 * avg_s, 4)}   # (what the part sustains from registers alone: top-level "context")


def extra_workloads(g, torch, dev, local_rank, default_steps=5, hbm=None):
    """The other single-GPU BASELINE configs, a few steps each, in the same process."""
    out = []
    # (MM_EXTRA_KXN=1 adds half_kxn / uint8_kxn, round 4's one-round extras: the same problems with A handed over K x N,
    # MM_TRANSPOSED_A, kernel/Memory.cpp:205-261 -- a transposition pre-pass inside the timed step, then the row-major default)
    jobs = [("half", None), ("double", None), ("minplus", None), ("minplus_f64", None), ("uint8", None), ("float", C5A_ROWS),
            ("float_split", None), ("half_exact", None)] + ([("half_kxn", None), ("uint8_kxn", None)] if os.environ.get("MM_EXTRA_KXN") == "1" else [])
    for key, rows_override in jobs:
        kxn = key.endswith("_kxn")
        dtype, map_op, reduce_op, size, roof, peak = WORKLOADS[key[:-4] if kxn else key]
        rows = rows_override or size
        k = m = size
        try:
            a, b, c = alloc_fill(g, torch, dev, local_rank, dtype, rows, k, m, 2000 + len(out), 3000 + len(out))
            if kxn:
                a = a.view(k, rows)      # the same HBM bytes read as a K x N matrix (rows == k here)
            path = PATHS.get(key, 0)
            steps, warm = STEPS.get(key, (default_steps, 2))
            elapsed, launch_ms = timed_steps(g, torch, a, b, c, dtype, map_op, reduce_op, steps, warm, lambda: None, path, kxn)
            value = 1e-9 * 2.0 * rows * k * m * steps / elapsed
            entry = {"workload": f"{dtype} {rows}x{k}x{m} ({map_op},{reduce_op}) on 1 MI355X"
                                 + ("; BASELINE configs[4]'s job without the split" if rows_override else "")
                                 + ("; A stored K x N (MM_TRANSPOSED_A): pre-pass + kernel both timed" if kxn else ""),
                     "kernel": g.kernel_name(g.make_config(dtype, map_op, reduce_op, path, kxn), rows, k, m),
                     "dtype": DTYPE_TAG[dtype], "value": round(value, 1), "unit": "GOp/s", "steps": steps, "warmup": warm,
                     "ms_per_step": round(1e3 * elapsed / steps, 4),
                     "roofline": roofline_obj(dtype, roof, peak, rows, k, m, launch_ms)}
            entry["key"] = key
            if hbm is not None and key in ("half", "double", "minplus", "uint8", "half_exact") and not kxn:
                hbm.attach(entry["roofline"], lambda: g.matmul(a, b, dtype, map_op, reduce_op, path=path, out=c),
                           depth=1 if entry["ms_per_step"] > 250.0 else 4)      # (a ~1-s launch: two of them are sample enough)
            for drop in ("algorithmic_flops_per_launch", "algorithmic_bytes_per_launch"):   # derivable from the shape; the headline keeps them
                entry["roofline"].pop(drop, None)
            if not kxn:   # (two kernels per step there: no single kernel's counters apply)
                attach_replayed(entry["roofline"], entry["kernel"], (rows, k, m), dtype)
            if key == "half_exact":
                entry["workload"] += "; the REFERENCE's half contract (binary16 accumulating in binary16, k ascending): MM_PATH_ORDERED == MM_HALF_CONTRACT=reference"
                entry["dtype"] = "f16 (f16 accumulate, unfused: kernel/Compute.cpp:129-133)"
                entry["roofline"]["probe_pair_sustained_TOps"] = 70.8
            if key == "float_split":
                entry["workload"] += "; MM_PATH_SPLIT (opt-in), pre-pass timed"
                entry["dtype"] = "f32 in/out; 3 bf16 planes, 6 bf16 MFMAs per pair, f32 accumulate"
                entry["roofline"]["frac_of_fp32_mfma_peak"] = round(entry["roofline"]["achieved"] / 157.3, 3)
            out.append(entry)
        except Exception as exc:  # an extra must never take the headline line down with it
            out.append({"workload": key, "error": repr(exc)})
        a = b = c = None
        torch.cuda.empty_cache()
    return out


def baseline_summary(out):
    """BASELINE.json's configs, one compact entry each: value (GOp/s), ms per step, fraction of the roof, kernel.  Well under
    1 KB, so that it survives in the tail of the line that the driver's record keeps."""
    def brief(value, ms, rl, kernel):
        b = {"value": value, "ms": round(ms, 3), "frac": rl.get("frac"), "kernel": kernel}
        if rl.get("achieved_HBM_GBps") is not None:
            b["hbm_GBps"] = rl["achieved_HBM_GBps"]       # DRAM side (sysfs mem_busy_percent, calibrated in the run): notes.hbm
        return b
    summary = {"unit": "GOp/s", "C2_float_16384": brief(out["value"], out["ms_per_step"], out["roofline"], out["config"]["kernel"])}
    names = {"half": "C3_half_32768", "half_exact": "C3_half_32768_reference_contract", "double": "C4_double_16384", "minplus": "C5b_minplus_8192",
             "float": "C5a_float_65536_rows_1gpu"}
    for w in out.get("workloads", []):
        if w.get("key") in names and "error" not in w:
            summary[names[w["key"]]] = brief(w["value"], w["ms_per_step"], w["roofline"], w["kernel"])
        elif w.get("workload") in names:          # an extra that failed is listed by its key
            summary[names[w["workload"]]] = {"error": str(w.get("error"))[:60]}
    cb = out.get("cpu_baseline")
    if cb:
        summary["C1_float_1024_ref_cpu_sim"] = {"value": cb.get("value"), "threads": cb.get("cores"), "host_cores": cb.get("host_cores")}
    return summary


def self_launch(gpus):
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stderr.write("[bench] --gpus %d without a launcher: re-running as `%s`\n" % (gpus, " ".join(cmd[1:])))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="float")
    ap.add_argument("--size", type=int, default=0, help="K = M (and N per GPU for weak scaling); default: the workload's BASELINE size")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 only. strong: a fixed job (--total-rows, default BASELINE's 65536 x 16384 x 16384) split "
                         "along N; weak: every rank owns --size rows")
    ap.add_argument("--total-rows", type=int, default=0, help="rows of C of the strong-scaling job")
    ap.add_argument("--scale-base", type=float, default=float(os.environ.get("MM_SCALE_BASE", "0") or 0),
                    help="N > 1: GFLOP/s of the SAME job on one GPU (the N = 1 line's scale_base.value); the line then carries "
                         "strong_scaling_vs_scale_base = value / this.  Also read from MM_SCALE_BASE")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the workloads[] / weak_scaling extras")
    args = ap.parse_args()

    import torch
    import gemm_hls_amd as g

    dtype, map_op, reduce_op, base_size, roof, peak = WORKLOADS[args.workload]
    if not args.size:
        args.size = base_size
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly (`python bench.py --gpus G ...`): become the launcher -- one rank per GPU under
        # torch.distributed.run on a free local port -- and hand its exit status back.  Under a launcher
        # (WORLD_SIZE set) this branch is never taken.
        return self_launch(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # MM_BENCH_DEVICE_MOD / MM_BENCH_BACKEND exist only so that the N>1 control flow can be smoke-tested
    # on a 1-GPU box (both ranks on device 0, gloo instead of RCCL); the driver never sets them.
    local_rank %= int(os.environ.get("MM_BENCH_DEVICE_MOD", torch.cuda.device_count()))
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # MM_BENCH_FORCE_DIST=1 (tests): form the control plane even for one rank, so that the RCCL init / all-reduce / barrier
    # sequence the N > 1 runs depend on is exercised on a 1-GPU box (tests/test_gpu_multi_device.py)
    use_dist = world > 1 or (os.environ.get("MM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist
        # control plane only: barrier + max(t).  The data path has no collective.
        if backend == "nccl":
            try:
                import datetime
                # (a bounded wait: a communicator that cannot form must end in the gloo fall-back below, not in a hung bench)
                dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)            # builds the RCCL communicator now, outside the timed region
                torch.cuda.synchronize()
            except Exception as exc:              # control plane only: a host-side barrier serves as well
                sys.stderr.write(f"[bench] RCCL control plane unavailable ({exc!r}); using gloo for barrier/max\n")
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if use_dist:
            dist.barrier()

    ctl_dev = dev if backend == "nccl" else "cpu"

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def rows_of_every_rank(rows):
        """What every rank actually owns, as the formed communicator reports it (one slot per rank, summed)."""
        if not use_dist:
            return [rows]
        t = torch.zeros(world, dtype=torch.float64, device=ctl_dev)
        t[rank] = rows
        dist.all_reduce(t)
        return [int(x) for x in t.tolist()]

    def gather_per_rank(record):
        """Every rank's own account of its leg, on rank 0 -- over the same control plane as the barrier (all_reduce of a
        [world, slot] byte matrix in which each rank fills its row: works on RCCL and gloo alike, no pickling of device tensors)."""
        if not use_dist:
            return [record]
        slot = 1024
        raw = json.dumps(record).encode()[:slot]
        t = torch.zeros((world, slot), dtype=torch.int32, device=ctl_dev)
        t[rank, :len(raw)] = torch.tensor(list(raw), dtype=torch.int32, device=ctl_dev)
        dist.all_reduce(t)
        out = []
        for r in range(world):
            row = bytes(int(x) for x in t[r].tolist()).rstrip(b"\0")
            try:
                out.append(json.loads(row.decode()))
            except Exception:
                out.append({"rank": r, "error": "record did not arrive"})
        return out

    def device_identity():
        ident = {"pci_bus_id": None, "sclk_mhz_after_run": None}
        try:
            buf = ctypes.create_string_buffer(32)
            g._check(g.lib().mm_device_pci_bus_id(local_rank, buf, 32))
            ident["pci_bus_id"] = buf.value.decode()
        except Exception as exc:
            ident["pci_bus_id"] = f"unavailable: {exc!r}"[:60]
        try:
            ident["sclk_mhz_after_run"] = int(torch.cuda.clock_rate(dev))
        except Exception:
            pass
        return ident

    ranks_seen = 1
    if use_dist:
        one = torch.ones(1, dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(one)                      # every rank of the communicator contributes 1
        ranks_seen = int(one.item())

    k = m = args.size
    job_cfg = g.make_config(dtype, map_op, reduce_op, PATHS.get(args.workload, 0))

    def row_slab(n_total, world_size, r):
        # the library's own partition (mm_row_slab == what mm_gemm_multi_device does): slabs aligned to the tile
        # rows of the kernel that will run on them
        return g.row_slab(job_cfg, n_total, k, m, world_size, r)

    headline = args.workload == "float" and args.size == SIZE
    hbm = HbmBusy(g, torch, dev, local_rank) if (world == 1 and not args.no_extra) else None
    hbm_box = {}

    def run_job(n_total, seed_base):
        """One timed region of the contract over the job of n_total rows; rank r owns row_slab(r)."""
        row0, rows = row_slab(n_total, world, rank)
        a, b, c = alloc_fill(g, torch, dev, local_rank, dtype, rows, k, m, seed_base + rank, 7)
        own_elapsed, launch_ms = timed_steps(g, torch, a, b, c, dtype, map_op, reduce_op, args.steps, args.warmup, barrier,
                                             PATHS.get(args.workload, 0))
        elapsed = max_over_ranks(own_elapsed)
        if hbm is not None and world == 1:    # the DRAM-side figure of this kernel: an untimed leg of back-to-back launches, after the timed region
            hbm_box.clear()
            hbm_box["avg_launch_ms"] = round(sum(launch_ms) / len(launch_ms), 4)
            hbm.attach(hbm_box, lambda: g.matmul(a, b, dtype, map_op, reduce_op, path=PATHS.get(args.workload, 0), out=c))
        a = b = c = None
        torch.cuda.empty_cache()
        # what THIS rank did, in its own words: if the job lands at 6.6x instead of 8x the line says which rank, on which PCI
        # device, running which kernel, at what launch time -- not only rank 0's launches and a maximum (VERDICT r5 missing 4)
        record = {"rank": rank, "local_rank": local_rank, "row0": row0, "rows": rows,
                  "kernel": g.kernel_name(job_cfg, rows, k, m) if rows else None,
                  "wall_ms_per_step": round(1e3 * own_elapsed / args.steps, 4),
                  "launch_ms_median": round(launch_ms[len(launch_ms) // 2], 4), "launch_ms_min": round(launch_ms[0], 4),
                  "launch_ms_max": round(launch_ms[-1], 4),
                  "control_plane": ("rccl" if backend == "nccl" else backend) if use_dist else "none"}
        record.update(device_identity())
        return rows, elapsed, launch_ms, rows_of_every_rank(rows), gather_per_rank(record)

    if world == 1:
        scaling = args.scaling
        n_total = args.total_rows or args.size
    elif args.scaling == "strong":
        scaling = "strong"
        n_total = args.total_rows or (C5A_ROWS if headline else 4 * args.size)
    else:
        scaling = "weak"
        n_total = args.size * world
    rows, elapsed, launch_ms, rows_all, per_rank = run_job(n_total, 1000)

    weak = None
    if world > 1 and scaling == "strong" and not args.no_extra:
        w_rows, w_elapsed, w_launch, _, w_per_rank = run_job(args.size * world, 5000)
        weak = (w_rows, w_elapsed, w_launch, w_per_rank)

    if rank == 0:
        flops_job = 2.0 * n_total * k * m
        value = 1e-9 * flops_job * args.steps / elapsed
        rl = roofline_obj(dtype, roof, peak, rows, k, m, launch_ms)
        rl.update(hbm_box)
        if world == 1:
            attach_replayed(rl, g.kernel_name(g.make_config(dtype, map_op, reduce_op, PATHS.get(args.workload, 0)), rows, k, m),
                            (rows, k, m), dtype)
        if world == 1:
            what = f"{dtype} {n_total}x{k}x{m} ({map_op},{reduce_op}) on 1 MI355X" + ("; BASELINE configs[1]" if headline else "")
        elif scaling == "strong":
            what = (f"{dtype} {n_total}x{k}x{m} ({map_op},{reduce_op}) split along N over {world} MI355X "
                    f"({rows} rows on rank 0), B replicated, no collective"
                    + ("; BASELINE configs[4]" if (headline and n_total == C5A_ROWS) else ""))
        else:
            what = (f"{dtype} {n_total}x{k}x{m} ({map_op},{reduce_op}): {args.size} rows per GPU over {world} MI355X, "
                    "B replicated, no collective")
        out = {
            "metric": ("GFLOP/s, fp32 GEMM N=K=M=16384 per MI355X (tiled outer-product C=A.B), and % of fp32 MFMA peak"
                       if headline else f"GOp/s, {dtype} ({map_op},{reduce_op}) K=M={args.size} per MI355X"),
            "value": round(value, 1),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": DTYPE_TAG[dtype],
            "data": "synthetic: uniform [1,10) (the reference generator's distribution), generated on device",
            "pct_of_mfma_peak": round(100.0 * value / 1e3 / (peak * world), 2),
            "config": {"workload": what,
                       "kernel": g.kernel_name(g.make_config(dtype, map_op, reduce_op, PATHS.get(args.workload, 0)), rows, k, m),
                       "rows_total": n_total, "rows_per_gpu": rows_all if world > 1 else rows},
            # the data path has no collective; this is the barrier / max-over-ranks plane only
            "control_plane": {"backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if use_dist else "none",
                              "ranks_seen": ranks_seen, "devices_visible": torch.cuda.device_count()},
            "roofline": rl,
        }
        if world > 1 or use_dist:
            out["per_rank"] = per_rank
            walls = [(r.get("wall_ms_per_step") or 0.0, r.get("rank")) for r in per_rank]
            out["slowest_rank"] = max(walls)[1]
            out["fastest_over_slowest_wall"] = round(min(w for w, _ in walls) / max(w for w, _ in walls), 4) if max(walls)[0] > 0 else None
            # the partition every rank computed for itself (mm_row_slab: configuration, shape and knobs only) must tile the job
            spans = sorted((r.get("row0", 0), r.get("rows", 0)) for r in per_rank if r.get("rows"))
            tiled = bool(spans) and spans[0][0] == 0 and all(a0 + an == b0 for (a0, an), (b0, _) in zip(spans, spans[1:])) \
                and spans[-1][0] + spans[-1][1] == n_total
            out["config"]["row_slabs_tile_the_job"] = tiled
            if not tiled:
                raise SystemExit(f"the ranks' row slabs do not tile the {n_total}-row job: {spans} -- ranks disagree on the partition "
                                 "(different knobs in their environments?)")
            if args.scale_base > 0:
                out["strong_scaling_vs_scale_base"] = {"scale_base_GFLOPs": args.scale_base, "factor": round(value / args.scale_base, 3),
                                                       "efficiency_note": "the driver computes efficiency itself; this is value / the 1-GPU value of the same job handed in with --scale-base"}
        if weak is not None:
            w_rows, w_elapsed, w_launch, w_per_rank = weak
            out["weak_scaling"] = {
                "workload": f"{dtype} {args.size * world}x{k}x{m}: {args.size} rows per GPU over {world} MI355X",
                "value": round(1e-9 * 2.0 * args.size * world * k * m * args.steps / w_elapsed, 1), "unit": "GFLOP/s",
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * w_elapsed / args.steps, 4),
                "roofline": roofline_obj(dtype, roof, peak, w_rows, k, m, w_launch), "per_rank": w_per_rank}
        if world == 1 and headline and not args.no_extra:
            out["workloads"] = extra_workloads(g, torch, dev, local_rank, hbm=hbm)
            # the strong-scaling job (BASELINE configs[4], float 65536 x 16384 x 16384) on ONE GPU, already timed above:
            # an N = G line's value / scale_base.value is the strong-scaling factor of that job without a second run
            for w in out["workloads"]:
                if w.get("roofline") and w["workload"].startswith(f"float {C5A_ROWS}x{SIZE}x{SIZE} "):
                    out["scale_base"] = {"workload": w["workload"], "value": w["value"], "unit": w["unit"], "n_gpus": 1,
                                         "steps": w["steps"], "warmup": w["warmup"], "ms_per_step": w["ms_per_step"],
                                         "note": "the job `bench.py --gpus G` (scaling strong) splits along N, run unsplit on this GPU: "
                                                 "strong-scaling factor at G = that line's value / this value"}
        if world == 1 and not args.no_cpu_baseline and headline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and headline and not args.no_extra:
            # counters measured in this run, after every timed leg: the headline kernel, BASELINE's C3 / C4 / C5b kernels and the two extras
            targets = [("float", rl, out["config"]["kernel"], (rows, k, m))]
            targets += [(w["key"], w["roofline"], w["kernel"], (WORKLOADS[w["key"]][3],) * 3) for w in out["workloads"]
                        if w.get("key") in ("half", "double", "minplus", "minplus_f64", "uint8") and "roofline" in w]
            for key, obj, kernel, shape in targets:
                live = live_counters(kernel, shape, obj["avg_launch_ms"], key)
                if live:
                    for stale in ("traffic_source", "mfma_util_source", "l2_hit_rate", "profiled_clock_GHz", "mfma_util_pct"):
                        obj.pop(stale, None)
                    obj.update(live)
        if world == 1 and headline:
            out["notes"] = {
                "counters": "roofline objects with counters_measured_in_this_run = true carry rocprofv3 PMC figures collected by this run: child passes "
                            "(--pmc X --kernel-trace) of tools/sweep.py on the same shape after the timed legs; traffic = FETCH_SIZE x2 (gfx950 correction) + "
                            "WRITE_SIZE, KiB -> bytes per launch; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).  The passes profile "
                            "launches of the SAME C-ABI call on the same shape made by tools/sweep.py children (2 + 1 warm-up each), not the timed launches "
                            "themselves: profiled_launch_ms next to avg_launch_ms says how much the profiler slows them.  `traffic` is L2 <-> FABRIC bytes, "
                            "Infinity-Cache hits included: an UPPER bound on HBM bytes.  rocprofv3 -L on this part has no counter behind the Infinity Cache "
                            "(no MALL / HBM-channel counters); the nearest are TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B, the L2's requests whose "
                            "DESTINATION is local memory (as opposed to GMI / IO), collected for C2 and C3 as dram_destined_bytes: they equal the all-request "
                            "byte count to 0.05 % (profiles/r06b_pmc_dram_side_counters.json, incl. a single-tile problem whose 537 MB of operands are read once: "
                            "553 MB by request size, 553 MB DRAM-destined, FETCH_SIZE x2 553 MB), i.e. they confirm the x2 correction and say nothing about "
                            "Infinity-Cache hits.  With false, traffic (L2 <-> fabric bytes per launch, Infinity-Cache hits included), mfma_util_pct, "
                            "profiled_clock_GHz and l2_hit_rate are REPLAYED from the committed PMC files named in *_source, which profiled the "
                            "same kernel on the same shape on an earlier box",
                "hbm": ("achieved_HBM_GBps / hbm_busy_pct: the DRAM side.  amdgpu's sysfs mem_busy_percent (the memory controller's activity as the SMU reports it) "
                        "sampled every 5 ms over ~1 s of back-to-back launches of the same call AFTER the timed region, times a GB/s-per-percent factor "
                        "calibrated in this run on two streams with known HBM bytes that the 256-MiB Infinity Cache cannot serve -- a 4-GiB device copy and a "
                        "4-GiB fill (hbm_calibration; both give ~82 GB/s per percent: the figure is linear, 100 % = 8.2 TB/s).  Resolution one percent = 82 GB/s.  "
                        "Compare with achieved_fabric_GBps (L2 <-> fabric, Infinity-Cache hits included): the difference is what the Infinity Cache serves"
                        if hbm is not None and hbm.cal else "no mem_busy_percent in sysfs on this box (or MM_BENCH_NO_HBM_PROBE): no DRAM-side figure"),
                "peaks": "MI355X_MICROARCH.md dense MFMA peaks (fp32 157.3, fp16 2500, i8 5000 T/s; fp64 78.6 datasheet); min-plus: SURVEY 8(d) "
                         "VALU ceiling 78.6 T lane-ops/s (fp64: 39.3); float_split: bf16 peak / 6 MFMAs per fp32 multiply-add block"}
            if hbm is not None and hbm.cal:
                out["hbm_calibration"] = dict(hbm.cal, GBps_per_pct_used=round(hbm.k, 1), sysfs=hbm.path)
            out["context"] = {"what_the_part_sustains_from_registers_alone": {
                "float": MFMA_SUSTAINED["float"], "double": MFMA_SUSTAINED["double"], "half": POWER_CEILING["half"], "uint8_t": POWER_CEILING["uint8_t"],
                "float_split": {"register_only_bf16_mfma_on_random_operands_TOps_div_6": round(SPLIT_BF16_REGISTER_ONLY_TOPS / 6.0, 1),
                                "instruction": "v_mfma_f32_32x32x16_bf16", "source": POWER_CEILING["half"]["source"]},
                "minplus_f64": VALU_SUSTAINED["minplus_f64"]}}
            # LAST key on purpose: the driver's record keeps the tail of this line -- the BASELINE configs, one short entry each
            out["baseline_summary"] = baseline_summary(out)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
