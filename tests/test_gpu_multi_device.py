"""SURVEY.md 8(e), the N split over the GPUs of one node, EXECUTED with G > 1 on a 1-GPU box.

`md_virtual_devices` = V (MM_MD_VIRTUAL_DEVICES) makes mm_gemm_multi_device treat the physical devices as V logical
ones, dealt out round-robin: on this box G streams, G A / C slabs and G copies of B on device 0, the B fan-out taking
the hipMemcpyPeerAsync branch behind the "B is on device 0" event.  Every line a real G-GPU node runs -- row0 offsets,
ragged last slab, empty trailing slabs, n_total propagation, the strided column slabs of a K x N A -- runs here, and the
result is compared BITWISE with the one-device launch (kernel/Compute.cpp:53-60: outer tiles of C are independent, so a
row's value may not depend on how the rows were dealt out).  Where >= 2 real GPUs are visible the same tests run on them
(the knob only raises the limit; logical device g lands on physical device g % count)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle  # noqa: E402
import gemm_hls_amd as g  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def virtual_devices():
    g.set_tuning("md_virtual_devices", 8)
    yield 8
    g.set_tuning("md_virtual_devices", -1)


def _uniform(rng, shape, dtype):
    if np.issubdtype(dtype, np.integer):
        return rng.integers(1, 11, size=shape).astype(dtype)
    return rng.uniform(1, 10, size=shape).astype(dtype)


@pytest.mark.parametrize("devices", [2, 3, 8])
@pytest.mark.parametrize("dtype,npdt,shape", [
    ("float", np.float32, (2513, 1024, 1040)),     # ragged N: split-K of a small job is decided on the WHOLE job
    ("float", np.float32, (1001, 512, 272)),       # 64 x 64 geometry: more devices than 64-row tile rows at G = 8? (1001 / 8 -> 128)
    ("float", np.float32, (5, 64, 48)),            # fewer rows than devices: trailing devices own nothing
    ("int", np.int32, (1001, 64, 96)),
    ("int", np.int32, (5, 16, 16)),
    ("double", np.float64, (777, 256, 130)),
    ("half", np.float16, (1300, 256, 264)),
    ("uint8_t", np.uint8, (1000, 256, 272)),
], ids=lambda v: v if isinstance(v, str) else ("x".join(map(str, v)) if isinstance(v, tuple) else None))
def test_multi_device_virtual_split_is_bit_identical_to_one_device(virtual_devices, devices, dtype, npdt, shape):
    n, k, m = shape
    rng = np.random.default_rng(n + devices)
    a, b = _uniform(rng, (n, k), npdt), _uniform(rng, (k, m), npdt)
    c1, _ = g.matmul_host(a, b, dtype, devices=1)
    cg, t = g.matmul_host(a, b, dtype, devices=devices)
    assert t > 0
    assert np.array_equal(cg.view(np.uint8), c1.view(np.uint8)), (dtype, shape, devices)
    # and the one-device result is the contract's: exact for integers, the oracle's rule for floating types
    if np.issubdtype(npdt, np.integer):
        assert np.array_equal(c1, _oracle.naive(dtype, "Multiply", "Add", a, b))
    elif dtype != "half":
        exact = a.astype(np.float64) @ b.astype(np.float64)
        assert np.max(np.abs(c1 - exact) / exact) < (1e-5 if dtype == "float" else 1e-12)


@pytest.mark.parametrize("devices", [2, 3, 8])
def test_multi_device_slab_bookkeeping_matches_mm_row_slab(virtual_devices, devices):
    """Every slab lands at ITS rows of C: A's row r is r + 1 everywhere, B is the identity-like selector, so C[r, :] names the
    row it was computed from; min-plus on the VALU family for a second kernel family behind the same split."""
    n, k, m = 1500, 64, 128
    a = np.repeat(np.arange(1, n + 1, dtype=np.float32)[:, None], k, axis=1)
    b = np.zeros((k, m), np.float32)
    b[0, :] = 1.0
    c, _ = g.matmul_host(a, b, devices=devices)
    assert np.array_equal(c[:, 0], np.arange(1, n + 1, dtype=np.float32))
    cfg = g.make_config("float")
    slabs = [g.row_slab(cfg, n, k, m, devices, r) for r in range(devices)]
    assert [s[0] for s in slabs] == [min(r * slabs[0][1], n) for r in range(devices)]
    assert sum(s[1] for s in slabs) == n
    am, bm = _oracle.fill("float", 700, 48, 64)
    cm, _ = g.matmul_host(am, bm, "float", "Add", "Min", devices=devices)
    assert np.array_equal(cm, _oracle.naive("float", "Add", "Min", am, bm))


@pytest.mark.parametrize("devices", [2, 3])
@pytest.mark.parametrize("dtype,npdt,shape", [
    ("float", np.float32, (2516, 512, 528)),      # N % 4 == 0: the K x N matrix-core kernels serve the job and every slab
    ("float", np.float32, (2513, 512, 528)),      # N % 4 != 0: the job's family is the generic one -- and so is every slab's
    ("float", np.float32, (4096, 512, 4096)),     # one device: a whole round of the K x N kernel's tiles; slabs: transposed first, row-major geometry -- same bits
    ("int", np.int32, (1001, 64, 96)),
    ("half", np.float16, (1304, 256, 264)),
    ("double", np.float64, (778, 256, 130)),
], ids=lambda v: v if isinstance(v, str) else ("x".join(map(str, v)) if isinstance(v, tuple) else None))
def test_multi_device_transposed_a_column_slabs(virtual_devices, devices, dtype, npdt, shape):
    """MM_TRANSPOSED_A (A stored K x N, kernel/Memory.cpp:205-261) through the split: device g gets COLUMNS
    [row0, row0 + rows) of A with one strided copy.  Bitwise equal to the one-device K x N launch, and equal in value to
    the row-major product."""
    n, k, m = shape
    rng = np.random.default_rng(7 * n + devices)
    a, b = _uniform(rng, (n, k), npdt), _uniform(rng, (k, m), npdt)
    at = np.ascontiguousarray(a.T)
    c1, _ = g.matmul_host(at, b, dtype, devices=1, transposed_a=True)
    cg, _ = g.matmul_host(at, b, dtype, devices=devices, transposed_a=True)
    assert np.array_equal(cg.view(np.uint8), c1.view(np.uint8)), (dtype, shape, devices)
    if np.issubdtype(npdt, np.integer):
        assert np.array_equal(cg, _oracle.naive(dtype, "Multiply", "Add", a, b))
    else:
        exact = a.astype(np.float64) @ b.astype(np.float64)
        tol = {"float": 1e-5, "double": 1e-12, "half": 2.0 ** -10}[dtype]
        assert np.max(np.abs(cg.astype(np.float64) - exact) / exact) < tol


def test_multi_device_large_default_geometry_and_tile_aligned_slabs(virtual_devices):
    """The 256-row tile of the large fp32 default: slabs are aligned to the tile height of the kernel that runs on them
    (VERDICT r4 weak 1), checked through the library's own arithmetic; and a job big enough to take that geometry split 3
    ways keeps the one-device bits."""
    cfg = g.make_config("float")
    # N = 64512 on 8: ceil = 8064 rows = 31.5 tiles of the 256-row kernel that serves 8064 x 16384 x 16384 -> 8192-row slabs
    rows = [g.row_slab(cfg, 64512, 16384, 16384, 8, r)[1] for r in range(8)]
    assert rows == [8192] * 7 + [7168]
    assert g.kernel_info(cfg, 8064, 16384, 16384).tile_n == 256
    # N = 33792 on 8 (the verdict's example): 4224-row slabs are served by a 128-row geometry -> nothing to round
    assert [g.row_slab(cfg, 33792, 16384, 16384, 8, r)[1] for r in range(8)] == [4224] * 8
    assert g.kernel_info(cfg, 4224, 16384, 16384).tile_n == 128
    assert [g.row_slab(cfg, 65536, 16384, 16384, 8, r) for r in range(8)] == [(8192 * r, 8192) for r in range(8)]
    n, k, m = 16384, 256, 4096
    assert g.kernel_info(cfg, n, k, m).tile_n == 256
    rng = np.random.default_rng(11)
    a, b = _uniform(rng, (n, k), np.float32), _uniform(rng, (k, m), np.float32)
    c1, _ = g.matmul_host(a, b, devices=1)
    c3, _ = g.matmul_host(a, b, devices=3)
    assert np.array_equal(c3, c1)
    rows = np.random.default_rng(2).integers(0, n, 64)
    exact = a[rows].astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(c3[rows] - exact) / exact) < 1e-5


def test_multi_device_timed_reports_every_devices_own_kernel_time():
    """mm_gemm_multi_device_timed ("max over devices of kernel time"): HIP events around each device's launch on its
    own stream; the job's time is their maximum, a device without rows reports 0, the host clock is the cross-check -- and
    `MM_GPUS=G RunHardware.exe` prints them."""
    g.set_tuning("md_virtual_devices", 8)
    try:
        rng = np.random.default_rng(5)
        n, k, m = 1000, 512, 528
        a, b = _uniform(rng, (n, k), np.float32), _uniform(rng, (k, m), np.float32)
        c1, _ = g.matmul_host(a, b, devices=1)
        c, t, per_device, wall = g.matmul_host(a, b, devices=3, timing=True)
        assert np.array_equal(c, c1)
        assert len(per_device) == 3 and all(x > 0 for x in per_device) and t == max(per_device) and wall >= t > 0
        # 5 rows over 8 devices: one 5-row slab (rounded up to a tile), seven devices idle
        a5 = a[:5].copy()
        c5, t5, per5, _ = g.matmul_host(a5, b, devices=8, timing=True)
        assert np.array_equal(c5, c1[:5]) and per5[0] == t5 > 0 and per5[1:] == [0.0] * 7
    finally:
        g.set_tuning("md_virtual_devices", -1)
    exe = os.path.join(ROOT, "bin", "RunHardware.exe")
    env = dict(os.environ, MM_GPUS="3", MM_MD_VIRTUAL_DEVICES="3")
    r = subprocess.run([exe, "1000", "528", "528", "hw", "off"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = re.findall(r"device (\d): rows \[(\d+), (\d+)\) in ([\d\.e\-\+]+) seconds", r.stdout)
    assert [(int(d), int(lo), int(hi)) for d, lo, hi, _ in got] == [(0, 0, 384), (1, 384, 768), (2, 768, 1000)], r.stdout
    assert r.stdout.count("<- slowest") == 1 and "host clock, first dispatch to last completion" in r.stdout


def test_multi_device_count_is_never_silently_reduced():
    import torch
    have = torch.cuda.device_count()
    a = np.ones((64, 32), np.float32)
    b = np.ones((32, 48), np.float32)
    g.set_tuning("md_virtual_devices", -1)
    with pytest.raises(g.MMError, match="device_count"):
        g.matmul_host(a, b, devices=have + 1)
    g.set_tuning("md_virtual_devices", 4)
    try:
        c, _ = g.matmul_host(a, b, devices=4)
        assert np.array_equal(c, np.full((64, 48), 32.0, np.float32))
        with pytest.raises(g.MMError, match="device_count"):
            g.matmul_host(a, b, devices=5)
    finally:
        g.set_tuning("md_virtual_devices", -1)


def test_multi_device_run_hardware_env_split(tmp_path):
    """The runner's multi-GPU mode (MM_GPUS=G, gemm_hls_amd/host/RunHardware.cpp) over virtual devices: verified
    against the host reference like any other run."""
    exe = os.path.join(ROOT, "bin", "RunHardware.exe")
    env = dict(os.environ, MM_GPUS="3", MM_MD_VIRTUAL_DEVICES="3")
    r = subprocess.run([exe, "1000", "528", "528", "hw", "on"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "Successfully verified." in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_multi_device_bench_eight_ranks_on_one_device_dry_run():
    """The driver's exact N = 8 command shape, `python bench.py --gpus 8 --steps 20 --warmup 5`, launched plainly, all
    eight ranks on device 0 (MM_BENCH_DEVICE_MOD=1; 8 x 2 GiB of operands fit one 288-GB device) and gloo as the control
    plane: the BASELINE configs[4] job, 8192 rows per rank, the large fp32 default kernel, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MM_BENCH_DEVICE_MOD="1", MM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--scale-base", "150000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "strong"
    assert out["control_plane"]["ranks_seen"] == 8
    assert out["config"]["rows_total"] == 65536 and out["config"]["rows_per_gpu"] == [8192] * 8
    assert out["config"]["kernel"] == "mfma_f32_256x256x16_w8_flush4096"
    assert "BASELINE configs[4]" in out["config"]["workload"]
    gops = 1e-9 * 2.0 * 65536 * 16384 * 16384 / (1e-3 * out["ms_per_step"])
    assert abs(out["value"] - gops) / out["value"] < 1e-3
    assert out["weak_scaling"]["value"] > 0 and "16384 rows per GPU" in out["weak_scaling"]["workload"]
    # eight ranks time-share ONE device here: the whole job cannot run faster than one device's roof
    assert out["value"] < 157.3e3 * 1.02
    # the line explains itself: one record per rank -- PCI device, slab, kernel, its own wall time and launch
    # times from the stream events, the control plane it ended up on -- the slowest rank named, the partition cross-checked
    assert [r["rank"] for r in out["per_rank"]] == list(range(8))
    for i, rec in enumerate(out["per_rank"]):
        assert (rec["row0"], rec["rows"]) == (8192 * i, 8192) and rec["kernel"] == "mfma_f32_256x256x16_w8_flush4096"
        assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9]", rec["pci_bus_id"]), rec
        assert 0 < rec["launch_ms_min"] <= rec["launch_ms_median"] <= rec["launch_ms_max"] and rec["wall_ms_per_step"] > 0
        assert rec["control_plane"] == "gloo"
    assert len({rec["pci_bus_id"] for rec in out["per_rank"]}) == 1          # the dry run: all eight on the one device
    assert out["slowest_rank"] in range(8) and 0 < out["fastest_over_slowest_wall"] <= 1
    assert max(rec["wall_ms_per_step"] for rec in out["per_rank"]) == pytest.approx(out["ms_per_step"], rel=1e-3)
    assert out["config"]["row_slabs_tile_the_job"] is True
    assert out["strong_scaling_vs_scale_base"]["factor"] == pytest.approx(out["value"] / 150000.0, rel=1e-3)
    assert len(out["weak_scaling"]["per_rank"]) == 8


def test_multi_device_bench_control_plane_on_rccl_with_one_rank():
    """The control plane of the N > 1 runs is RCCL (torch.distributed 'nccl'): init with device_id, an all-reduce that builds
    the communicator outside the timed region, barrier, MAX over ranks, the per-rank row census, destroy.  Two ranks cannot
    share one device under RCCL, so the 8-rank dry run above uses gloo; this runs the SAME code path on the real backend
    with the one rank a 1-GPU box allows, under the launcher the driver uses."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MM_BENCH_BACKEND")}
    env.update(MM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--size", "4096",
           "--no-extra", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["control_plane"]["ranks_seen"] == 1
    if not out["control_plane"]["backend"].startswith("rccl"):
        # bench.py falls back to gloo for barrier / max when RCCL cannot form (by design: the data path has no collective)
        pytest.skip("RCCL control plane did not form on this box; bench.py fell back to " + out["control_plane"]["backend"] + ": "
                    + r.stderr[-400:])


def test_multi_device_c5a_real_size_eight_virtual_devices_sampled_rows():
    """BASELINE configs[4] at its REAL size -- float 65536 x 16384 x 16384 split along N over G = 8 -- through
    mm_gemm_multi_device itself (the one configuration of the split that had only ever run through
    bench.py): eight logical devices dealt out over the box's GPU(s), 8192-row slabs on the 256 x 256 kernel, B fanned out
    device 0 -> g, per-device kernel times from HIP events.  Checked on sampled rows of every slab (first, last, the rows either
    side of each slab boundary) against fp64, and -- whole output, 4 GiB -- bit for bit against the one-device launch."""
    import torch
    n, k, m, G = 65536, 16384, 16384, 8
    try:
        import psutil
        free_gib = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        free_gib = None
    if free_gib is not None and free_gib < 24:
        pytest.skip(f"needs ~14 GiB of host memory for A, B and two copies of C (a strike-proof margin of 24 GiB); {free_gib:.0f} GiB available")
    rng = np.random.default_rng(65)
    b = rng.uniform(1, 10, size=(k, m)).astype(np.float32)
    a = np.empty((n, k), np.float32)
    for r0 in range(0, n, 8192):                      # filled slab by slab: no 8-GiB fp64 temporary
        a[r0:r0 + 8192] = rng.uniform(1, 10, size=(8192, k)).astype(np.float32)
    g.set_tuning("md_virtual_devices", G)
    try:
        cfg = g.make_config("float")
        assert [g.row_slab(cfg, n, k, m, G, r) for r in range(G)] == [(8192 * r, 8192) for r in range(G)]
        assert g.kernel_name(cfg, 8192, k, m) == "mfma_f32_256x256x16_w8_flush4096"
        c8, t, per_device, wall = g.matmul_host(a, b, devices=G, timing=True)
    finally:
        g.set_tuning("md_virtual_devices", -1)
    assert len(per_device) == G and all(x > 0 for x in per_device) and t == max(per_device) and wall >= t
    rows = sorted({0, 1, n - 1} | {r for s in range(1, G) for r in (8192 * s - 1, 8192 * s)} | set(rng.integers(0, n, 8).tolist()))
    b64 = b.astype(np.float64)
    exact = a[rows].astype(np.float64) @ b64
    assert np.max(np.abs(c8[rows] - exact) / exact) < 1e-5
    del b64, exact
    c1, _ = g.matmul_host(a, b, devices=1)             # the same job on ONE device: the split gives its bits
    assert np.array_equal(c8, c1)
    # on one physical GPU the eight slabs time-share it: the job cannot beat one device's roof, and each device's own kernel
    # time is at least a slab's worth at that roof
    flops = 2.0 * n * k * m
    assert flops / wall / 1e12 < 157.3 * 1.02 * max(1, torch.cuda.device_count())
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "multi_device_c5a_real_size_8_virtual_devices.txt"), "w") as f:
            f.write(f"float {n}x{k}x{m} over {G} logical devices on {torch.cuda.device_count()} physical: max per-device kernel time {t:.4f} s, "
                    f"host clock {wall:.4f} s, per device {[round(x, 4) for x in per_device]}, {flops / wall / 1e12:.1f} TF by the host clock; "
                    f"sampled rows {rows} within 1e-5 of fp64; whole output bit-identical to the one-device launch\n")
