"""GPU tests of the C-ABI's contract beyond the arithmetic: one half (x,+) contract for every
shape, pointer alignment, argument checking of the torch binding, concurrent use from several
host threads / streams, the multi-device row split, and the one-process-per-GPU bench launch."""
import ctypes
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import _oracle
import gemm_hls_amd as g

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape", [(129, 4104, 264), (65, 40, 8), (33, 4096, 12), (70, 2056, 20), (1, 2, 2)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("transposed_a", [False, True], ids=["rowmajorA", "KxN_A"])
def test_half_auto_contract_does_not_depend_on_shape(shape, transposed_a):
    """half (Multiply, Add) under MM_PATH_AUTO on shapes the matrix-core kernel
    does not take (K % 16 != 0 or M % 8 != 0) keeps the f32-accumulate / round-once contract instead
    of dropping to half accumulation (which would turn K = 4104 into inf where K = 4096 is finite)."""
    n, k, m = shape
    rng = np.random.default_rng(k * 7 + m)
    a = rng.uniform(0.5, 2.0, size=(n, k)).astype(np.float16)
    b = rng.uniform(0.5, 2.0, size=(k, m)).astype(np.float16)
    assert g.kernel_name(g.make_config("half", transposed_a=transposed_a), n, k, m) == "ordered_wide_f16"
    arg = np.ascontiguousarray(a.T) if transposed_a else a
    c, _ = g.matmul_capi(arg, b, "half", transposed_a=transposed_a)
    wide = _oracle.naive("half", "Multiply", "Add", a, b, wide_half=True)
    assert np.all(np.isfinite(c.astype(np.float32)))
    ulps = np.abs(c.view(np.int16).astype(np.int32) - wide.view(np.int16).astype(np.int32))
    assert ulps.max() <= 1
    if k >= 2056:  # the half-accumulating Naive is visibly different at this depth (stagnation)
        narrow = _oracle.naive("half", "Multiply", "Add", a, b)
        assert not np.array_equal(narrow, wide)


def test_half_auto_is_continuous_across_the_k_multiple_of_16_boundary():
    n, m = 64, 64
    a, b = _oracle.fill("half", n, 4112, m)   # [1,10): sums pass 65504 near k ~ 2000
    for k in (4096, 4104, 4112):
        c, _ = g.matmul_capi(np.ascontiguousarray(a[:, :k]), np.ascontiguousarray(b[:k]), "half")
        assert np.all(np.isinf(c.astype(np.float32))), k   # same (IEEE-correct) answer on both kernels
    a2 = (a.astype(np.float32) / 64).astype(np.float16)
    b2 = (b.astype(np.float32) / 64).astype(np.float16)
    for k in (4096, 4104):
        c, _ = g.matmul_capi(np.ascontiguousarray(a2[:, :k]), np.ascontiguousarray(b2[:k]), "half")
        exact = a2[:, :k].astype(np.float64) @ b2[:k].astype(np.float64)
        assert np.max(np.abs(c.astype(np.float64) - exact) / exact) <= 2.0 ** -11 * 1.02, k


def test_unaligned_pointers_are_refused_by_the_fast_path_and_served_by_the_ordered_one():
    import torch
    dev = torch.device("cuda:0")
    n = k = m = 256
    base = torch.rand(n * k + 1, device=dev) + 1.0
    a = base[1:].view(n, k)                       # contiguous, but 4 bytes past a 16-B boundary
    b = torch.rand((k, m), device=dev) + 1.0
    assert a.is_contiguous() and a.data_ptr() % 16 == 4
    with pytest.raises(g.MMError, match="16-byte aligned"):
        g.matmul(a, b)
    c = g.matmul(a, b, path=g.PATH_ORDERED)
    ref = a.double() @ b.double()
    assert float(((c.double() - ref).abs() / ref).max()) < 1e-5
    out = torch.empty(n * m + 1, device=dev)[1:].view(n, m)
    with pytest.raises(g.MMError, match="16-byte aligned"):
        g.matmul(b, b, out=out)


def test_torch_binding_checks_out_and_operands():
    import torch
    dev = torch.device("cuda:0")
    a = torch.ones((64, 32), device=dev)
    b = torch.ones((32, 48), device=dev)
    with pytest.raises(g.MMError, match="out must be"):
        g.matmul(a, b, out=torch.empty((64, 47), device=dev))
    with pytest.raises(g.MMError, match="out must be"):
        g.matmul(a, b, out=torch.empty((64, 48), device=dev, dtype=torch.float64))
    with pytest.raises(g.MMError, match="out must be"):
        g.matmul(a, b, out=torch.empty((48, 64), device=dev).t())
    with pytest.raises(g.MMError, match="inner dimensions"):
        g.matmul(a, torch.ones((31, 48), device=dev))
    with pytest.raises(g.MMError, match="contiguous"):
        g.matmul(a.t(), torch.ones((64, 48), device=dev))
    with pytest.raises(g.MMError, match="do not match"):
        g.matmul(a.double(), b.double())
    with pytest.raises(g.MMError, match="no CPU path"):
        g.matmul(a.cpu(), b.cpu())


def test_concurrent_launches_from_two_threads_on_two_streams():
    """The header promises every entry point is callable concurrently.  Two host threads, each with
    its own stream, interleave different configurations (MFMA fp32 incl. first-use LDS opt-in of
    several geometries, VALU min-plus, int MFMA, timed blocking launches) on ONE device; results
    must equal the serial ones bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    jobs = []
    rng = np.random.default_rng(99)
    for i, (dtype, ops, shape) in enumerate([
            ("float", ("Multiply", "Add"), (1024, 1024, 1024)), ("float", ("Add", "Min"), (640, 256, 384)),
            ("uint8_t", ("Multiply", "Add"), (512, 512, 512)), ("float", ("Multiply", "Add"), (300, 64, 272)),
            ("double", ("Multiply", "Add"), (513, 528, 528)), ("half", ("Multiply", "Add"), (520, 528, 528)),
            ("float", ("Multiply", "Add"), (2048, 512, 2048)), ("int", ("Multiply", "Add"), (257, 64, 96)),
            # MM_PATH_SPLIT: stream-ordered workspace from both threads at once, both tile sizes
            ("float", ("Multiply", "Add"), (4096, 300, 4100, g.PATH_SPLIT)), ("float", ("Multiply", "Add"), (300, 520, 260, g.PATH_SPLIT))]):
        path = shape[3] if len(shape) == 4 else g.PATH_AUTO
        n, k, m = shape[:3]
        tdt = g.torch_dtype(dtype)
        if dtype in ("uint8_t", "int"):
            a = torch.randint(1, 10, (n, k), device=dev).to(tdt)
            b = torch.randint(1, 10, (k, m), device=dev).to(tdt)
        else:
            a = (torch.rand((n, k), device=dev) * 9 + 1).to(tdt)
            b = (torch.rand((k, m), device=dev) * 9 + 1).to(tdt)
        jobs.append((dtype, ops, a, b, path))
    serial = [g.matmul(a, b, dtype, *ops, path=path) for (dtype, ops, a, b, path) in jobs]
    torch.cuda.synchronize()
    results = {}
    errors = []

    def worker(tid):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for rep in range(6):
                    order = list(range(len(jobs)))
                    if tid:
                        order.reverse()
                    for j in order:
                        dtype, ops, a, b, path = jobs[j]
                        out = g.matmul(a, b, dtype, *ops, path=path)
                        results[(tid, rep, j)] = out
                    # a blocking, timed C-ABI launch from this thread while the other one enqueues
                    dtype, ops, a, b, _ = jobs[3]
                    cfg = g.make_config(dtype, *ops)
                    t = ctypes.c_double(0)
                    c = torch.empty((a.shape[0], b.shape[1]), dtype=a.dtype, device=dev)
                    g._check(g.lib().mm_gemm_launch(0, ctypes.byref(cfg), a.data_ptr(), b.data_ptr(), c.data_ptr(),
                                                    a.shape[0], a.shape[1], b.shape[1], ctypes.byref(t)))
                    assert t.value > 0
                    results[(tid, rep, "timed")] = c
            stream.synchronize()
        except Exception as exc:  # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for (tid, rep, j), out in results.items():
        want = serial[3] if j == "timed" else serial[j]
        assert torch.equal(out, want), (tid, rep, j)


def test_tuning_set_switches_geometry_in_process():
    n = k = m = 512
    a, b = _oracle.fill("float", n, k, m)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    try:
        for v in (3, 0, 35):
            g.set_tuning("f32_variant", v)
            assert g.get_tuning("f32_variant") == v
            c, _ = g.matmul_capi(a, b)
            assert np.max(np.abs(c - exact) / exact) < 1e-5
        g.set_tuning("band_rows", 8)
        c8, _ = g.matmul_capi(a, b)
        g.set_tuning("band_rows", -1)
        c4, _ = g.matmul_capi(a, b)
        assert np.array_equal(c8, c4)           # rasterisation order never changes a result
    finally:
        g.set_tuning("f32_variant", -1)
        g.set_tuning("band_rows", -1)


def test_host_pointer_entry_with_explicit_config():
    n, k, m = 130, 64, 96
    a, b = _oracle.fill("int", n, k, m)
    c = np.zeros((n, m), np.int32)
    cfg = g.make_config("int")
    g._check(g.lib().mm_gemm_host(ctypes.byref(cfg), a.ctypes.data, b.ctypes.data, c.ctypes.data, n, k, m))
    assert np.array_equal(c, _oracle.naive("int", "Multiply", "Add", a, b))
    # K x N A through the same entry (what an MM_TRANSPOSED_A build of the kernel shim forwards)
    cfg_t = g.make_config("int", transposed_a=True)
    at = np.ascontiguousarray(a.T)
    c2 = np.zeros((n, m), np.int32)
    g._check(g.lib().mm_gemm_host(ctypes.byref(cfg_t), at.ctypes.data, b.ctypes.data, c2.ctypes.data, n, k, m))
    assert np.array_equal(c2, c)


def test_empty_and_degenerate_problems_at_every_entry_point():
    """An empty C (N = 0 or M = 0) is a successful no-op at every entry point -- nothing is launched, C is not touched,
    null pointers are fine -- and K = 0 is refused (the reference's kernel would leave C unwritten, kernel/Compute.cpp:54-149
    never runs its k loop; here that is an argument error rather than silently returned garbage).  The device info the
    library reports is the device's."""
    import torch
    L = g.lib()
    dev = torch.device("cuda:0")
    cfg = g.make_config("float")
    a = torch.ones((64, 32), device=dev)
    b = torch.ones((32, 48), device=dev)
    c = torch.full((64, 48), 7.0, device=dev)
    t = ctypes.c_double(-1.0)
    for (n, m) in ((0, 48), (64, 0), (0, 0)):
        g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), a.data_ptr(), b.data_ptr(), c.data_ptr(), n, 32, m, ctypes.byref(t)))
        g._check(L.mm_gemm_enqueue(None, ctypes.byref(cfg), None, None, None, n, 32, m))
        g._check(L.mm_gemm_host(ctypes.byref(cfg), None, None, None, n, 32, m))
        t2 = ctypes.c_double(-1.0)
        g._check(L.mm_gemm_multi_device(1, ctypes.byref(cfg), None, None, None, n, 32, m, ctypes.byref(t2)))
        assert t2.value == 0.0
    torch.cuda.synchronize()
    assert bool((c == 7.0).all())
    for call in (lambda: L.mm_gemm_launch(0, ctypes.byref(cfg), a.data_ptr(), b.data_ptr(), c.data_ptr(), 64, 0, 48, None),
                 lambda: L.mm_gemm_enqueue(None, ctypes.byref(cfg), a.data_ptr(), b.data_ptr(), c.data_ptr(), 64, 0, 48),
                 lambda: L.mm_gemm_host(ctypes.byref(cfg), a.data_ptr(), b.data_ptr(), c.data_ptr(), 64, 0, 48)):
        assert call() != 0 and b"size_k must be positive" in L.mm_last_error()
    assert L.mm_gemm_launch(0, ctypes.byref(cfg), None, b.data_ptr(), c.data_ptr(), 64, 32, 48, None) != 0     # a null operand of a non-empty problem
    assert b"null matrix pointer" in L.mm_last_error()
    # single-row / single-column-block / single-k-chunk problems: the smallest shapes every fast family serves
    for dtype, shape in (("float", (1, 8, 4)), ("double", (1, 8, 2)), ("half", (1, 16, 8)), ("uint8_t", (1, 32, 16)), ("int", (1, 4, 4))):
        n, k, m = shape
        aa, bb = _oracle.fill(dtype, n, k, m)
        cc, _ = g.matmul_capi(aa, bb, dtype)
        want = _oracle.naive(dtype, "Multiply", "Add", aa, bb, wide_half=True) if dtype == "half" else _oracle.naive(dtype, "Multiply", "Add", aa, bb)
        if dtype in ("float", "double"):
            assert np.allclose(cc, want, rtol=1e-6), dtype
        elif dtype == "half":       # exact products, fp32 accumulation, one rounding: within one binary16 ulp of the wide oracle
            assert np.abs(cc.view(np.int16).astype(np.int32) - want.view(np.int16).astype(np.int32)).max() <= 1
        else:
            assert np.array_equal(cc, want), dtype
    info = g.kernel_info(cfg, 16384, 16384, 16384)
    assert info.compute_units == torch.cuda.get_device_properties(0).multi_processor_count == 256


# ---- multi-GPU ------------------------------------------------------------------------------------
def test_multi_device_ragged_split_is_bit_identical_to_one_device():
    """Runs wherever >= 2 GPUs are visible (the driver's 8-GPU node; skipped on the 1-GPU box):
    2 and ALL devices, N not divisible by the device count, against the one-device bits."""
    import torch
    have = torch.cuda.device_count()
    if have < 2:
        pytest.skip("needs at least 2 GPUs")
    n, k, m = 2500 + 13, 1024, 1040
    rng = np.random.default_rng(5)
    a = rng.uniform(1, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(1, 10, size=(k, m)).astype(np.float32)
    c1, _ = g.matmul_host(a, b, devices=1)
    for devices in sorted({2, have}):
        cg, t = g.matmul_host(a, b, devices=devices)
        assert np.array_equal(cg, c1), devices
        assert t > 0
    ai, bi = _oracle.fill("int", 1001, 64, 96)
    want = _oracle.naive("int", "Multiply", "Add", ai, bi)
    for devices in sorted({2, have}):
        ci, _ = g.matmul_host(ai, bi, "int", devices=devices)
        assert np.array_equal(ci, want)


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_on_one_device_smoke(scaling):
    """The one-process-per-GPU launch of bench.py exactly as the driver issues it, with both ranks on
    device 0 (MM_BENCH_DEVICE_MOD=1) and gloo as the control plane: checks the N > 1 control flow
    (row_slab ownership, barrier, max over ranks, one JSON line from rank 0)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MM_BENCH_DEVICE_MOD="1", MM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--scaling", scaling, "--size", "2048", "--total-rows", "6000", "--no-extra"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == scaling
    rows_total = 6000 if scaling == "strong" else 2 * 2048
    assert out["config"]["rows_total"] == rows_total
    gops = 1e-9 * 2.0 * rows_total * 2048 * 2048 / (1e-3 * out["ms_per_step"])   # whole job / time of one step
    assert abs(out["value"] - gops) / out["value"] < 1e-3


def test_bench_gpus_2_launched_plainly_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (the shape of the driver's recorded N = 1
    command): bench.py must start its two ranks itself and print one line that says what the control plane saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MM_BENCH_DEVICE_MOD="1", MM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--size", "2048", "--total-rows", "6000", "--no-extra"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["rows_total"] == 6000
    assert out["control_plane"]["ranks_seen"] == 2 and out["control_plane"]["backend"] == "gloo"
    # tile-aligned slabs: ceil(6000 / 2) = 3000 -> 3072 rows on rank 0, the ragged rest on rank 1
    assert out["config"]["rows_per_gpu"] == [3072, 2928]


def test_race_screen_of_the_hand_synchronised_kernels():
    """tools/soak.py: every DMA-ring kernel repeated under an uneven background load gives the same bits
    every time and equals an independently scheduled kernel of the same arithmetic (a few seconds)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py")], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert " 0 differing results" in r.stdout and "differing results, equal to independent schedule: True" in r.stdout


def test_library_loaded_before_torch_still_sees_the_gpu():
    """PyTorch-ROCm bundles its own HIP runtime; two runtimes in one process leave the second one
    without a device.  The binding imports torch first for that reason, so loading it (or running
    __graft_entry__.build()) BEFORE the first torch import must work -- the order the driver may use."""
    code = ("import __graft_entry__ as e, sys\n"
            "import gemm_hls_amd as g\n"
            "L = g.lib(); assert g.device_count() >= 1\n"
            "import torch\n"
            "a = torch.full((64, 64), 2.0, device='cuda')\n"
            "assert float(g.matmul(a, a)[3, 5]) == 256.0\n"
            "e.smoke()\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_lab_variant_ids_are_not_in_the_product():
    """The retired schedules and the work-skipping ablations (f16_variant 27 / 43 / 59 / 75: no DMA / no fragment reads /
    L2-resident sources, wrong results on purpose; 300: the 384 x 256 tile; f32_variant 28-32; split 16 / 32) are
    built into tools/lab/libmm_gemm_amd_lab.so only.  The product refuses their ids -- whatever "ablations" says --
    instead of running something else under that name."""
    n = k = m = 512
    a = np.ones((n, k), np.float16)
    b = np.ones((k, m), np.float16)
    try:
        for unlocked in (-1, 1):
            g.set_tuning("ablations", unlocked)
            for v in (27, 43, 59, 75, 300, 13):
                g.set_tuning("f16_variant", v)
                with pytest.raises(g.MMError):
                    g.matmul_capi(a, b, "half")
            g.set_tuning("f16_variant", -1)
            for v in (28, 29, 30, 31, 32, 20, 13):
                g.set_tuning("f32_variant", v)
                with pytest.raises(g.MMError):
                    g.matmul_capi(a.astype(np.float32), b.astype(np.float32))
            g.set_tuning("f32_variant", -1)
            for v in (16, 32):
                g.set_tuning("split_variant", v)
                with pytest.raises(g.MMError):
                    g.matmul_capi(a.astype(np.float32), b.astype(np.float32), path=g.PATH_SPLIT)
            g.set_tuning("split_variant", -1)
    finally:
        for knob in ("ablations", "f16_variant", "f32_variant", "split_variant"):
            g.set_tuning(knob, -1)


def test_lab_library_runs_the_retired_schedules_and_gates_its_ablations():
    """tools/lab/libmm_gemm_amd_lab.so (MM_LIB=lab for the measurement tools; built only by MM_BUILD_LAB=1 python
    gemm_hls_amd/build.py): same C ABI, the lab editions of the matrix-core kernels.  Retired schedules still compute the
    product; ablations need MM_ABLATIONS=1."""
    if not os.path.exists(os.path.join(ROOT, "tools", "lab", "libmm_gemm_amd_lab.so")):
        pytest.skip("the lab library is built on request only (MM_BUILD_LAB=1)")
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})\n"
        "from _lib import g\n"
        "assert g.LIB_PATH.endswith('libmm_gemm_amd_lab.so')\n"
        "rng = np.random.default_rng(3)\n"
        "a = rng.uniform(1, 2, (300, 512)).astype(np.float32); b = rng.uniform(1, 2, (512, 272)).astype(np.float32)\n"
        "exact = a.astype(np.float64) @ b.astype(np.float64)\n"
        "for v in (1, 13, 20, 24, 34):\n"
        "    g.set_tuning('f32_variant', v); c, _ = g.matmul_capi(a, b)\n"
        "    assert np.max(np.abs(c - exact) / exact) < 1e-5, v\n"
        "g.set_tuning('f32_variant', 29)\n"
        "try:\n"
        "    g.matmul_capi(a, b); raise SystemExit('ablation ran without MM_ABLATIONS')\n"
        "except g.MMError: pass\n"
        "g.set_tuning('ablations', 1); c, _ = g.matmul_capi(a, b); assert c.shape == (300, 272)\n"
        "g.set_tuning('ablations', -1); g.set_tuning('f32_variant', -1)\n"
        "ah, bh = a.astype(np.float16), b.astype(np.float16)\n"
        "eh = ah.astype(np.float64) @ bh.astype(np.float64)\n"
        "for v in (10, 13, 300, 302):\n"
        "    g.set_tuning('f16_variant', v); c, _ = g.matmul_capi(ah, bh, 'half')\n"
        "    assert np.max(np.abs(c.astype(np.float64) - eh) / eh) < 2.0 ** -10, v\n"
        "print('lab ok')\n")
    env = dict(os.environ, MM_LIB="lab")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "lab ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("path_name", ["auto", "split"])
@pytest.mark.parametrize("shape", [(384, 256, 320), (512, 1024, 512), (384, 4096, 256), (2304, 256, 2304)], ids=lambda s: "x".join(map(str, s)))
def test_enqueue_is_capturable_into_a_hip_graph(path_name, shape):
    """mm_gemm_enqueue issues only stream-ordered work (kernels; for MM_PATH_SPLIT also hipMallocAsync / hipFreeAsync),
    so a launch-bound loop of small products can be captured once and replayed as a hipGraph (here through torch's
    graph capture, which puts its capture stream in hipStreamCaptureModeGlobal)."""
    import torch
    path = g.PATH_SPLIT if path_name == "split" else g.PATH_AUTO
    n, k, m = shape     # whole tiles (64 x 64 geometry twice) / split-K (8 chunks + the ordered reduction) / stream-K (ONE teams kernel, the last part to arrive gathers: Combine::LastArriver) under AUTO
    if path_name == "split" and shape != (384, 256, 320):
        pytest.skip("one shape is enough for the opt-in path")
    a, b = _oracle.fill("float", n, k, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    if path_name == "auto":
        expect = {(384, 256, 320): "64x64x32", (512, 1024, 512): "64x64x32", (384, 4096, 256): "splitk8", (2304, 256, 2304): "streamk"}[shape]
        assert expect in g.kernel_name(g.make_config("float"), n, k, m)
    outs = [torch.zeros((n, m), dtype=torch.float32, device="cuda") for _ in range(4)]
    # Stream-K under MM_PATH_AUTO is the one-kernel last-arriver form with epoch flags, eager and captured alike.  A captured
    # launch has its epoch BAKED INTO the graph, so every replay raises the same value: replay correctness rests on
    # flags_alloc() (mm_capi.hip) adding a memset node for the flag block whenever the stream is capturing -- the flags are
    # cleared on EVERY replay, in stream order before the kernel -- instead of relying on "never held this epoch before" as
    # eager launches do.  The four captured launches below share one graph and three replays: if that memset node were
    # missing, the second replay would find all flags already raised, the first part to arrive would gather slots the other
    # parts have not written yet, and `o` would differ from `want`.
    want = g.matmul(ta, tb, path=path).clone()      # also warms up: function attributes, pool configuration
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for o in outs:
            g.matmul(ta, tb, path=path, out=o)
    for o in outs:
        o.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)
