#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: fp32 GEMM N=K=M=16384 per MI355X
("float 16384x16384x16384 on 1 MI355X, MFMA fp32, LDS-tiled outer product"), GFLOP/s and fraction of
the fp32 MFMA peak; with --gpus G the rows of C are split into G slabs, one process per GPU, B
replicated, no data-path collective (weak scaling: every rank multiplies its own 16384-row slab,
i.e. the job is (G*16384) x 16384 x 16384 -- at G=4 exactly BASELINE's 65536x16384x16384).

A step = one pass of the hot path (one mm_gemm_enqueue through the C ABI) over operands that are
already resident in HBM.  Timing = W untimed steps, barrier + synchronize, EXACTLY K steps,
synchronize + barrier, MAX over ranks.  GOp/s = 1e-9 * 2*N*K*M / t as host/RunHardware.cpp:174-180.

One JSON line on rank 0; besides the contract keys it carries
  roofline     the dominant kernel against the MFMA roof, from HIP events on the launch stream
  cpu_baseline the reference's OWN hlslib CPU-simulation path (oracle/_ref, compiled from the
               reference's kernel sources) timed on this host's cores on float 1024^3 (BASELINE C1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_TFLOPS_F32_MFMA = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
SIZE = 16384
# The default workload is BASELINE.json's metric (configs[1]).  The other single-GPU BASELINE
# configs can be timed with --workload; they are parity-test cases, not the headline line.
#            dtype     map         reduce  size   roof  peak (T op/s)  unit of the roof
WORKLOADS = {
    "float": ("float", "Multiply", "Add", 16384, "mfma", 157.3),
    "half": ("half", "Multiply", "Add", 32768, "mfma", 2500.0),
    "double": ("double", "Multiply", "Add", 16384, "mfma", 78.6),
    "minplus": ("float", "Add", "Min", 8192, "valu", 65.0),   # measured VALU issue ceiling for 2 add + 1 min3
    "uint8": ("uint8_t", "Multiply", "Add", 32768, "mfma", 5000.0),
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
    if not _oracle.ref_available():
        info["sample"] += " -- oracle/_ref not built here, not measured"
        return info
    a, b = _oracle.fill("float", sample_n, sample_n, sample_n)
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
    # for scale: the BLAS reference (ReferenceImplementation, include/Utility.h:76-89) on the same sample
    import numpy as np
    t0 = time.perf_counter()
    for _ in range(5):
        np.matmul(a, b)
    info["blas_sgemm_gflops_same_sample"] = round(5 * 2.0 * sample_n ** 3 / (time.perf_counter() - t0) / 1e9, 1)
    # and the semantic spec, Naive (include/Utility.h:18-42), single-threaded
    t0 = time.perf_counter()
    _oracle.naive("float", "Multiply", "Add", a, b, threads=1)
    info["naive_1thread_gflops_same_sample"] = round(2.0 * sample_n ** 3 / (time.perf_counter() - t0) / 1e9, 2)
    return info


def hbm_traffic_per_launch():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic.json,
    written by tools/pmc_traffic.py with the guide's gfx950 correction); None if not collected."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1])).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="float")
    ap.add_argument("--size", type=int, default=0, help="N=K=M per GPU (default: the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # MM_BENCH_DEVICE_MOD / MM_BENCH_BACKEND exist only so that the N>1 control flow can be smoke-tested
    # on a 1-GPU box (both ranks on device 0, gloo instead of RCCL); the driver never sets them.
    local_rank %= int(os.environ.get("MM_BENCH_DEVICE_MOD", torch.cuda.device_count()))
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        # control plane only: barrier + max(t).  The data path has no collective.
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=dev)
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

    from gemm_hls_amd.partition import row_slab
    n_total = args.size * world
    row0, rows = row_slab(n_total, world, rank)
    k = m = args.size
    # synthetic operands of the reference's distribution, generated in HBM
    tdt = g.torch_dtype(dtype)
    a = torch.empty((rows, k), dtype=tdt, device=dev)
    b = torch.empty((k, m), dtype=tdt, device=dev)
    c = torch.empty((rows, m), dtype=tdt, device=dev)
    L = g.lib()
    g._check(L.mm_fill_device(local_rank, g.DTYPES[dtype], a.data_ptr(), a.numel(), 1000 + rank))
    g._check(L.mm_fill_device(local_rank, g.DTYPES[dtype], b.data_ptr(), b.numel(), 7))
    torch.cuda.set_device(local_rank)

    def step():
        g.matmul(a, b, dtype, map_op, reduce_op, out=c)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()  # same stream as the launch: per-launch duration without host gaps
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    avg_launch_s = 1e-3 * sum(launch_ms) / len(launch_ms)

    if rank == 0:
        flops_job = 2.0 * n_total * k * m
        value = 1e-9 * flops_job * args.steps / elapsed
        achieved_tf = 2.0 * rows * k * m / avg_launch_s / 1e12
        headline = args.workload == "float"
        out = {
            "metric": ("GFLOP/s, fp32 GEMM N=K=M=16384 per MI355X (tiled outer-product C=A.B), and % of fp32 MFMA peak"
                       if headline else f"GOp/s, {dtype} ({map_op},{reduce_op}) N=K=M={args.size} per MI355X"),
            "value": round(value, 1),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"float": "f32", "half": "f16 (f32 accumulate)", "double": "f64", "uint8_t": "u8 (i32 accumulate)"}[dtype],
            "data": "synthetic: uniform [1,10) (the reference generator's distribution), generated on device",
            "pct_of_mfma_peak": round(100.0 * value / 1e3 / (peak * world), 2),
            "config": {"workload": f"{dtype} {n_total}x{k}x{m} ({map_op},{reduce_op}), rows of C split over {world} GPU(s), "
                                   "B replicated, no collective" + ("; BASELINE configs[1] per GPU" if headline else ""),
                       "kernel": g.kernel_name(g.make_config(dtype, map_op, reduce_op), rows, k, m),
                       "rows_per_gpu": rows},
            "roofline": {"bound": roof, "achieved": round(achieved_tf, 2), "peak": peak,
                         "unit": "TFLOP/s" if roof == "mfma" and dtype != "uint8_t" else "TOp/s",
                         "frac": round(achieved_tf / peak, 4),
                         "traffic": hbm_traffic_per_launch() if (headline and args.size == SIZE) else None,
                         "algorithmic_flops_per_launch": 2.0 * rows * k * m,
                         "avg_launch_ms": round(1e3 * avg_launch_s, 4)},
        }
        if world == 1 and not args.no_cpu_baseline and headline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
