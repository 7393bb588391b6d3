"""MM_PATH_SPLIT (gemm_hls_amd/csrc/mm_mfma_f32_split.hip): fp32 (Multiply, Add) computed on the bf16
matrix cores from three bf16 planes per operand.  It is a tolerance path like every fp32 fast path
(BASELINE.json north_star: 1e-5 relative of the BLAS reference, test/TestSimulation.cpp:75-92), so
it is checked (i) with the reference's rule on the reference's seeded inputs, (ii) with a normwise
bound on mixed-sign data that is TIGHTER than what one bf16 pass (2^-9) or two planes (2^-17) could
meet -- which pins that all six products are present and every plane lands in the right fragment
-- and (iii) for the pack step's edge handling: any N, K, M, K x N layout of A, unaligned views."""
import ctypes

import numpy as np
import pytest

import _bounds
import _oracle
import gemm_hls_amd as g

pytestmark = pytest.mark.gpu

SHAPES = [(513, 528, 528), (1, 16, 16), (37, 32, 48), (300, 64, 272), (129, 80, 260), (256, 8, 4), (1024, 1024, 1024),
          (1, 1, 1), (3, 5, 7), (257, 17, 255), (300, 1000, 70), (511, 2049, 513)]   # the last five: no divisibility at all
# (odd M also exercises the scalar write-back: the 8-byte one needs M even)


@pytest.fixture(autouse=True)
def _default_variant():
    g.set_tuning("split_variant", -1)
    yield
    g.set_tuning("split_variant", -1)


def _normwise(c, a, b):
    exact = a.astype(np.float64) @ b.astype(np.float64)
    den = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    return float(np.max(np.abs(c - exact) / np.maximum(den, 1e-300)))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_split_vs_blas_and_exact_on_reference_inputs(shape):
    n, k, m = shape
    a, b = _oracle.fill("float", n, k, m)
    assert g.kernel_name(g.make_config("float", path=g.PATH_SPLIT), n, k, m) == "mfma_f32_split_bf16x3"
    c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    bad, first, worst = _oracle.compare("float", c, a @ b, 1e-5)
    assert bad == 0, (first, worst)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    rel = float(np.max(np.abs(c - exact) / exact))
    _bounds.north_star(rel, f"MM_PATH_SPLIT {n}x{k}x{m} vs fp64")
    _bounds.guard(rel, 2e-6, f"MM_PATH_SPLIT {n}x{k}x{m} vs fp64")


@pytest.mark.parametrize("k", [16, 32, 48])
def test_split_keeps_all_six_products(k):
    """With a short chain there is next to no accumulation noise, so the bound can sit at 2^-22 -- 16x below what a
    missing product of weight 2^-17 leaves behind (the three-product variant measures 5e-6 here), and only reachable
    when every plane of every fragment is the right one."""
    n, m = 300, 272
    rng = np.random.default_rng(k)
    a = rng.uniform(-3, 3, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 3, size=(k, m)).astype(np.float32)
    c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert _normwise(c, a, b) < 2.0 ** -22


@pytest.mark.parametrize("variant", [-1, 1, 4, 64, 128])
@pytest.mark.parametrize("shape", [(513, 528, 528), (300, 1000, 70), (260, 4200, 300)], ids=lambda s: "x".join(map(str, s)))
def test_split_mixed_sign_no_worse_than_the_fp32_kernels(shape, variant):
    """Mixed signs and a few entries 10^4 larger / smaller than the rest (exponents differ inside a fragment): the
    normwise error |c - exact| / (|a| . |b|) stays within 2x of the native fp32 matrix-core kernel's AND of the
    k-ordered fp32 chain's (the reference's own arithmetic) on the same operands -- measured: it is smaller than both
    (profiles/r02y_split_error_vs_native.log).  4200 = one flush chunk + a remainder."""
    g.set_tuning("split_variant", variant)
    n, k, m = shape
    rng = np.random.default_rng(n * 31 + k)
    a = rng.uniform(-3, 3, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 3, size=(k, m)).astype(np.float32)
    a[rng.integers(0, n, 40), rng.integers(0, k, 40)] *= 1e4
    b[rng.integers(0, k, 40), rng.integers(0, m, 40)] *= 1e-4
    c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    native, _ = g.matmul_capi(a, b)
    ordered, _ = g.matmul_capi(a, b, path=g.PATH_ORDERED)
    e = _normwise(c, a, b)
    assert e <= 2 * _normwise(native, a, b) and e <= 2 * _normwise(ordered, a, b) and e < 1e-5


@pytest.mark.parametrize("variant", [-1, 64])
def test_split_flush_chunks_are_deterministic_and_bound_the_drift(variant):
    """K = 12424 on the reference's all-positive inputs = 2 chunks of the default (flush every 8256 k: one plain store
    + the final read-modify-write) and 4 chunks of variant 64 (every 4128 k: read-modify-write inside the loop too).
    Same bits on every launch; the chunked result is at least as close to fp64 as the single chain (variant 4)."""
    n, k, m = 260, 3 * 4128 + 40, 264
    a, b = _oracle.fill("float", n, k, m)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    g.set_tuning("split_variant", variant)
    c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    c2, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert np.array_equal(c, c2)
    g.set_tuning("split_variant", 4)
    c1, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    e, e1 = np.max(np.abs(c - exact) / exact), np.max(np.abs(c1 - exact) / exact)
    _bounds.north_star(float(e), "MM_PATH_SPLIT flush chunks vs fp64")
    _bounds.guard(float(e), 3e-6, "MM_PATH_SPLIT flush chunks vs fp64")
    assert e <= e1 * 1.25, (e, e1)


@pytest.mark.parametrize("shape", [(513, 528, 528), (300, 8300, 272), (257, 17, 255)], ids=lambda s: "x".join(map(str, s)))
def test_split_schedules_are_bit_identical(shape):
    """Every schedule of the six-product kernel issues the same MFMAs on the same operands in the same order per
    accumulator: ping-pong (default), one barrier per stage with reads ahead (1) / interleaved (128) must agree bit
    for bit."""
    n, k, m = shape
    a, b = _oracle.fill("float", n, k, m)
    ref, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    for variant in (1, 128):
        g.set_tuning("split_variant", variant)
        c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
        assert np.array_equal(c, ref), variant


@pytest.mark.parametrize("shape", [(513, 528, 528), (1024, 1024, 1024), (300, 8300, 272), (3, 5, 7), (2048, 4200, 1920)],
                         ids=lambda s: "x".join(map(str, s)))
def test_split_both_tiles_are_bit_identical(shape):
    """The 128 x 128 tile (picked for problems that would leave compute units idle) runs the same per-accumulator MFMA
    sequence as the 256 x 256 one: pinning either (variants 512 / 256) gives the same bits."""
    n, k, m = shape
    a, b = _oracle.fill("float", n, k, m)
    g.set_tuning("split_variant", 256)
    big, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    g.set_tuning("split_variant", 512)
    small, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert np.array_equal(big, small)
    info = g.kernel_info(g.make_config("float", path=g.PATH_SPLIT), n, k, m)
    assert (info.tile_n, info.wavefronts) == (128, 4)
    g.set_tuning("split_variant", -1)
    auto, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert np.array_equal(auto, big)


def test_split_three_product_variant_is_the_coarser_class():
    """split_variant 2 keeps a1b1 + a1b2 + a2b1 only: passes 1e-5 on the reference's inputs, but not the 2^-21 bound
    machinery above by construction -- it must stay a knob, and it must be measurably coarser than the default."""
    n, k, m = 300, 1000, 272
    rng = np.random.default_rng(7)
    a = rng.uniform(-3, 3, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 3, size=(k, m)).astype(np.float32)
    c6, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    g.set_tuning("split_variant", 2)
    c3, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    e6, e3 = _normwise(c6, a, b), _normwise(c3, a, b)
    assert e3 < 2.0 ** -15 and e6 < e3


def test_split_transposed_a_is_bit_identical_to_row_major():
    n, k, m = 300, 200, 272
    a, b = _oracle.fill("float", n, k, m)
    c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    ct, _ = g.matmul_capi(np.ascontiguousarray(a.T), b, path=g.PATH_SPLIT, transposed_a=True)
    assert np.array_equal(c, ct)


def test_split_is_deterministic_and_leaves_neighbours_alone():
    import torch
    n, k, m = 300, 520, 260
    a, b = _oracle.fill("float", n, k, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    guard = torch.full((n + 1 + 256, m), 7.0, dtype=torch.float32, device="cuda")   # a whole tile of rows below C
    out = guard[1:n + 1]
    g.matmul(ta, tb, path=g.PATH_SPLIT, out=out)
    first = out.clone()
    for _ in range(5):
        g.matmul(ta, tb, path=g.PATH_SPLIT, out=out)
        assert torch.equal(out, first)
    assert bool((guard[0] == 7).all()) and bool((guard[n + 1:] == 7).all())


def test_split_takes_unaligned_views():
    """The pack step reads scalars when a row is not 16-byte aligned: element-aligned pointers are enough."""
    import torch
    n, k, m = 130, 72, 96
    a, b = _oracle.fill("float", n, k, m)
    buf_a = torch.zeros(n * k + 1, dtype=torch.float32, device="cuda")
    buf_a[1:] = torch.from_numpy(a).reshape(-1).cuda()
    va = buf_a[1:].view(n, k)
    assert va.data_ptr() % 16 == 4
    c = g.matmul(va, torch.from_numpy(b).cuda(), path=g.PATH_SPLIT).cpu().numpy()
    ref, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert np.array_equal(c, ref)


def test_split_non_finite_operands_stay_non_finite():
    """An inf operand meets the (signed) residual planes of the other matrix, so an output that fp32 arithmetic
    gives as +inf may come out as nan here -- never as a finite number (include/mm_gemm.h, MM_PATH_SPLIT)."""
    n, k, m = 64, 64, 64
    a, b = _oracle.fill("float", n, k, m)
    a[3, 5] = np.inf
    a[10, 0] = np.nan
    c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert not np.any(np.isfinite(c[3])) and np.all(np.isnan(c[10]))
    keep = np.ones(n, bool)
    keep[[3, 10]] = False
    exact = a[keep].astype(np.float64) @ b.astype(np.float64)
    rel = float(np.max(np.abs(c[keep] - exact) / exact))
    _bounds.north_star(rel, "MM_PATH_SPLIT rows without non-finite operands")
    _bounds.guard(rel, 2e-6, "MM_PATH_SPLIT rows without non-finite operands")


def test_split_through_the_multi_device_driver():
    """mm_gemm_multi_device with path = MM_PATH_SPLIT: every device packs its own row slab of A and all of B."""
    n, k, m = 300, 200, 272
    a, b = _oracle.fill("float", n, k, m)
    c1, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    cm, t = g.matmul_host(a, b, path=g.PATH_SPLIT, devices=1)
    assert np.array_equal(c1, cm) and t > 0


def test_release_workspace_returns_the_pool_memory_and_the_path_keeps_working():
    """The packed planes come from a memory pool the LIBRARY owns (it used to raise the release threshold of the
    process's default pool): freed workspace stays cached there between launches, the application's default pool is
    never touched, and mm_release_workspace hands the cache back to the driver."""
    import torch
    hip = ctypes.CDLL("libamdhip64.so")   # the runtime this process already uses

    def default_pool_state():
        pool, reserved, threshold = ctypes.c_void_p(), ctypes.c_uint64(0), ctypes.c_uint64(0)
        assert hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), 0) == 0
        assert hip.hipMemPoolGetAttribute(pool, 5, ctypes.byref(reserved)) == 0    # hipMemPoolAttrReservedMemCurrent
        assert hip.hipMemPoolGetAttribute(pool, 4, ctypes.byref(threshold)) == 0   # hipMemPoolAttrReleaseThreshold
        return reserved.value, threshold.value

    n = 2048
    a, b = _oracle.fill("float", n, 256, n)
    c0, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    g._check(g.lib().mm_release_workspace(0))
    big = torch.empty((8192, 8192), dtype=torch.float32, device="cuda").uniform_(1, 10)
    out = torch.empty_like(big)
    torch.cuda.synchronize()
    before = default_pool_state()
    free_before = torch.cuda.mem_get_info()[0]
    g.matmul(big, big, path=g.PATH_SPLIT, out=out)  # 2 x 8192^2 x 6 B = 768 MiB of workspace now sit in the library's pool
    torch.cuda.synchronize()
    assert free_before - torch.cuda.mem_get_info()[0] >= 760 << 20
    assert default_pool_state() == before            # not in the application's pool, and its threshold is as it was
    g._check(g.lib().mm_release_workspace(0))
    assert free_before - torch.cuda.mem_get_info()[0] <= 32 << 20
    assert default_pool_state() == before
    c1, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
    assert np.array_equal(c0, c1)
    with pytest.raises(g.MMError):
        g._check(g.lib().mm_release_workspace(99))


def test_split_refuses_other_configurations():
    a, b = _oracle.fill("int", 32, 32, 32)
    with pytest.raises(g.MMError, match="MM_PATH_SPLIT"):
        g.matmul_capi(a, b, "int", path=g.PATH_SPLIT)
    af, bf = _oracle.fill("float", 32, 32, 32)
    with pytest.raises(g.MMError, match="MM_PATH_SPLIT"):
        g.matmul_capi(af, bf, "float", "Add", "Min", path=g.PATH_SPLIT)
    assert g.lib().mm_config_supported(ctypes.byref(g.make_config("float", path=g.PATH_SPLIT))) == 1
    assert g.lib().mm_config_supported(ctypes.byref(g.make_config("double", path=g.PATH_SPLIT))) == 0


def test_split_full_size_sampled_rows():
    """16384^3 (BASELINE configs[1]) through MM_PATH_SPLIT: sampled rows against fp64, and against the native kernel."""
    import torch
    n = 16384
    ta = torch.empty((n, n), dtype=torch.float32, device="cuda")
    tb = torch.empty((n, n), dtype=torch.float32, device="cuda")
    g._check(g.lib().mm_fill_device(0, g.DTYPES["float"], ta.data_ptr(), ta.numel(), 5))
    g._check(g.lib().mm_fill_device(0, g.DTYPES["float"], tb.data_ptr(), tb.numel(), 6))
    c = g.matmul(ta, tb, path=g.PATH_SPLIT)
    rows = [0, 1, 255, 256, 4097, 8191, 12345, n - 1]
    exact = ta[rows].double() @ tb.double()
    rel = ((c[rows].double() - exact).abs() / exact).max().item()
    assert rel < 4e-6, rel
    native = g.matmul(ta, tb)
    assert ((c - native).abs() / native).max().item() < 1e-5


def test_split_c5a_shape_offsets_beyond_4gib():
    """65536 x 16384 x 16384 (BASELINE configs[4] on one GPU): 6 GiB of packed A, C offsets past 4 GiB."""
    import torch
    n, k = 65536, 16384
    ta = torch.empty((n, k), dtype=torch.float32, device="cuda")
    tb = torch.empty((k, k), dtype=torch.float32, device="cuda")
    g._check(g.lib().mm_fill_device(0, g.DTYPES["float"], ta.data_ptr(), ta.numel(), 5))
    g._check(g.lib().mm_fill_device(0, g.DTYPES["float"], tb.data_ptr(), tb.numel(), 6))
    c = g.matmul(ta, tb, path=g.PATH_SPLIT)
    rows = [0, 255, 256, 16383, 16384, 40000, 65279, 65280, n - 1]
    exact = ta[rows].double() @ tb.double()
    rel = ((c[rows].double() - exact).abs() / exact).max().item()
    assert rel < 4e-6, rel
    # a slab computed alone equals the same rows of the full product, bitwise (same tiles, same chunking)
    part = g.matmul(ta[32768:32768 + 512], tb, path=g.PATH_SPLIT)
    assert torch.equal(part, c[32768:32768 + 512])
