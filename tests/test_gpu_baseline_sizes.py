"""Parity at BASELINE.json's FULL sizes for the configs that are not the fp32 headline
(tests/test_gpu_parity.py covers float 16384^3 and 65536 x 16384 x 16384):

    C3  half   32768^3   (Multiply, Add)  mfma_f16
    C4  double 16384^3   (Multiply, Add)  mfma_f64
    C5b float  8192^3    (Add, Min)       valu_tile, the non-MFMA path

A CPU oracle cannot finish these sizes, so each test checks (i) sampled rows x sampled column
blocks of the device result against the CPU oracle (tests/_oracle.py: Naive, include/Utility.h:18-42;
wide-accumulate Naive for half; fp64 BLAS for double) and (ii) size-independent properties over the
WHOLE output: the fast family against the independently written k-ordered kernel where the semiring
is order-independent, exact power-of-two linearity, row-permutation equivariance, determinism.
Also here: mixed-sign fp32 with a normwise bound (the seeded reference data is all-positive, which
cannot see a sign or cancellation bug in a fragment path).

Comparison rule and tolerances: test/TestSimulation.cpp:75-92; BASELINE.json north_star (fp32 1e-5,
integer / min-plus semirings bit-exact)."""
import os

import numpy as np
import pytest

import _oracle
import gemm_hls_amd as g

pytestmark = pytest.mark.gpu


def _device_fill(dtype, shape, seed):
    import torch
    t = torch.empty(shape, dtype=g.torch_dtype(dtype), device="cuda:0")
    assert g.lib().mm_fill_device(0, g.DTYPES[dtype], t.data_ptr(), t.numel(), seed) == 0
    return t


def _sample(n, m, block=96):
    """Rows and column blocks that straddle tile, band and XCD-chunk boundaries, first and last."""
    rows = sorted({0, 1, 127, 128, 255, 256, n // 2 - 1, n // 2, n - 257, n - 2, n - 1})
    starts = sorted({0, 256 - block // 2, m // 2 - block // 2, m - 512 - block // 2, m - block})
    cols = np.concatenate([np.arange(s, s + block) for s in starts])
    return rows, cols


def _host_slices(a, b, c, rows, cols):
    import torch
    ci = torch.as_tensor(cols, device=a.device)
    return (a[rows].cpu().numpy(), b[:, ci].contiguous().cpu().numpy(), c[rows][:, ci].contiguous().cpu().numpy())


# ---- C5b: float (Add, Min) 8192^3 ----------------------------------------------------------------
def test_minplus_8192_valu_tile_vs_ordered_and_naive():
    import torch
    n = k = m = 8192
    a = _device_fill("float", (n, k), 31)
    b = _device_fill("float", (k, m), 32)
    assert g.kernel_name(g.make_config("float", "Add", "Min"), n, k, m) == "valu_tile"
    c = g.matmul(a, b, "float", "Add", "Min")
    c_ord = g.matmul(a, b, "float", "Add", "Min", path=g.PATH_ORDERED)
    torch.cuda.synchronize()
    # min over k of (a + b): every a + b is one rounding and min is exact, so any order gives the same bits
    assert torch.equal(c, c_ord)
    rows, cols = _sample(n, m)
    ah, bh, ch = _host_slices(a, b, c, rows, cols)
    assert np.array_equal(ch, _oracle.naive("float", "Add", "Min", ah, bh))
    # tropical "linearity": adding a constant to A adds it to every output only up to rounding, but
    # row-permutation equivariance and determinism are exact
    perm = torch.randperm(n, device=a.device)
    assert torch.equal(g.matmul(a[perm].contiguous(), b, "float", "Add", "Min"), c[perm])
    assert torch.equal(g.matmul(a, b, "float", "Add", "Min"), c)


# ---- C4: double (Multiply, Add) 16384^3 ------------------------------------------------------------
def test_double_16384_sampled_rows_and_properties():
    import torch
    n = k = m = 16384
    a = _device_fill("double", (n, k), 41)
    b = _device_fill("double", (k, m), 42)
    assert g.kernel_name(g.make_config("double"), n, k, m) == "mfma_f64_256x128x16_w8"
    c = g.matmul(a, b, "double")
    torch.cuda.synchronize()
    rows, cols = _sample(n, m)
    ah, bh, ch = _host_slices(a, b, c, rows, cols)
    blas = ah @ bh                                     # cblas_dgemm: the reference's ReferenceImplementation
    bad, first, worst = _oracle.compare("double", ch, blas, 1e-12)
    assert bad == 0, (first, worst)
    # against a wider yardstick (long double accumulation of exact-in-long-double products)
    exact = (ah[:4].astype(np.longdouble) @ bh.astype(np.longdouble))
    assert np.max(np.abs(ch[:4] - exact) / exact) < 1e-13
    # the k-ordered kernel on the same sampled rows (full width): same products, different order
    sub = a[rows].contiguous()
    c_ord = g.matmul(sub, b, "double", path=g.PATH_ORDERED)
    rel = ((c[rows] - c_ord).abs() / c_ord).max().item()
    assert rel < 1e-13, rel
    c2 = g.matmul(a * 2.0, b, "double")
    assert torch.equal(c2, c * 2.0)                    # power-of-two scaling commutes with rounding
    del c2
    perm = torch.randperm(n, device=a.device)
    assert torch.equal(g.matmul(a[perm].contiguous(), b, "double"), c[perm])
    assert torch.equal(g.matmul(a, b, "double"), c)


# ---- C3: half (Multiply, Add) 32768^3 ----------------------------------------------------------------
def test_half_32768_finite_inputs_sampled_rows_and_properties():
    """Inputs = the generator's [1,10) values scaled by 2^-6 (exact in binary16, no subnormals), so that
    sums over K = 32768 stay around 250 instead of overflowing: the contract (exact products, fp32
    accumulation, ONE rounding to binary16) is then visible in every output."""
    import torch
    n = k = m = 32768
    a = _device_fill("half", (n, k), 51)
    b = _device_fill("half", (k, m), 52)
    a.mul_(2.0 ** -6)
    b.mul_(2.0 ** -6)
    assert g.kernel_name(g.make_config("half"), n, k, m) == "mfma_f16_256x256_pingpong_16x16x32"
    c = g.matmul(a, b, "half")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(c).all())
    rows, cols = _sample(n, m)
    ah, bh, ch = _host_slices(a, b, c, rows, cols)
    wide = _oracle.naive("half", "Multiply", "Add", ah, bh, wide_half=True)
    ulps = np.abs(ch.view(np.int16).astype(np.int32) - wide.view(np.int16).astype(np.int32))
    assert ulps.max() <= 1, int(ulps.max())
    exact = ah.astype(np.float64) @ bh.astype(np.float64)
    rel = np.abs(ch.astype(np.float64) - exact) / exact
    # half an ulp of binary16 + the fp32 accumulation error of a 32768-term chain of positive terms
    assert rel.max() <= 2.0 ** -11 + 3e-5, float(rel.max())
    c2 = g.matmul(a * 2.0, b, "half")
    assert torch.equal(c2, c * 2.0)
    del c2
    perm = torch.randperm(n, device=a.device)
    assert torch.equal(g.matmul(a[perm].contiguous(), b, "half"), c[perm])
    assert torch.equal(g.matmul(a, b, "half"), c)


def test_half_32768_reference_inputs_overflow_to_inf_everywhere():
    """On BASELINE's own [1,10) data every sum passes 65504 long before K = 32768: the correctly
    rounded binary16 result is +inf for every element -- which is also what the reference's
    half-accumulating kernel produces -- and nothing may be NaN or negative."""
    import torch
    n = k = m = 32768
    a = _device_fill("half", (n, k), 61)
    b = _device_fill("half", (k, m), 62)
    c = g.matmul(a, b, "half")
    torch.cuda.synchronize()
    assert bool((c == float("inf")).all())
    # the k-ordered kernel (RunHardware hw_emu, half accumulation exactly like Naive) on a row sample
    rows = [0, 255, 256, 16383, 32767]
    c_ord = g.matmul(a[rows].contiguous(), b, "half", path=g.PATH_ORDERED)
    assert bool((c_ord == float("inf")).all())


# ---- fp32 with mixed signs ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(513, 528, 528), (300, 64, 272), (1024, 4104, 1024), (257, 8200, 520), (2048, 2048, 2048)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("transposed_a", [False, True], ids=["rowmajorA", "KxN_A"])
def test_f32_mixed_sign_normwise_bound(shape, transposed_a):
    """|c - exact| <= 1e-5 * (|A| . |B|) elementwise -- the form of BASELINE's 1e-5 bar that stays
    meaningful under cancellation -- on uniform [-1, 1) operands."""
    n, k, m = shape
    if transposed_a and n % 4:
        n += 4 - n % 4
    rng = np.random.default_rng(n * 31 + k)
    a = rng.uniform(-1, 1, size=(n, k)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(k, m)).astype(np.float32)
    assert g.kernel_name(g.make_config("float", transposed_a=transposed_a), n, k, m).startswith("mfma_f32")
    if transposed_a:
        c, _ = g.matmul_capi(np.ascontiguousarray(a.T), b, transposed_a=True)
    else:
        c, _ = g.matmul_capi(a, b)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    exact = a64 @ b64
    bound = 1e-5 * (np.abs(a64) @ np.abs(b64))
    err = np.abs(c - exact)
    assert np.all(err <= bound), float((err / bound).max())
    # sign structure: roughly half the outputs are negative, and every sign agrees wherever the exact
    # value is not within the bound of zero
    sure = np.abs(exact) > bound
    assert np.array_equal(np.sign(c[sure]), np.sign(exact[sure]))
    assert 0.3 < (c < 0).mean() < 0.7


def test_f32_mixed_sign_full_size_sampled():
    import torch
    n = k = m = 16384
    a = _device_fill("float", (n, k), 71)
    b = _device_fill("float", (k, m), 72)
    a.sub_(5.5)     # [-4.5, 4.5): one rounding, still exactly representable inputs for both sides
    b.sub_(5.5)
    c = g.matmul(a, b)
    torch.cuda.synchronize()
    rows, cols = _sample(n, m)
    ah, bh, ch = _host_slices(a, b, c, rows, cols)
    a64, b64 = ah.astype(np.float64), bh.astype(np.float64)
    err = np.abs(ch - a64 @ b64)
    bound = 1e-5 * (np.abs(a64) @ np.abs(b64))
    assert np.all(err <= bound), float((err / bound).max())
    c_neg = g.matmul(-a, b)
    assert torch.equal(c_neg, -c)                      # negation is exact: (-A) B == -(A B) bit for bit


# ---- other dtypes with signs / Max identity ------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["double", "half"])
def test_f64_f16_mixed_sign(dtype):
    n, k, m = 513, 1040, 528
    rng = np.random.default_rng(17)
    npdt = _oracle.NP_DTYPES[dtype]
    a = rng.uniform(-1, 1, size=(n, k)).astype(npdt)
    b = rng.uniform(-1, 1, size=(k, m)).astype(npdt)
    assert g.kernel_name(g.make_config(dtype), n, k, m).startswith("mfma_")
    c, _ = g.matmul_capi(a, b, dtype)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    exact = a64 @ b64
    absum = np.abs(a64) @ np.abs(b64)
    err = np.abs(c.astype(np.float64) - exact)
    if dtype == "double":
        assert np.all(err <= 1e-13 * absum)
    else:
        # one rounding of the result to binary16 + fp32 accumulation noise relative to sum |a||b|
        assert np.all(err <= 2.0 ** -11 * np.abs(exact) + 1e-6 * absum + 2.0 ** -25)


@pytest.mark.parametrize("dtype", ["float", "double", "int", "half"])
@pytest.mark.parametrize("path", [g.PATH_AUTO, g.PATH_ORDERED], ids=["auto", "ordered"])
def test_max_reduce_over_all_negative_values_uses_lowest_identity(dtype, path):
    """(Add, Max) / (Multiply, Max) where every mapped value is negative: the accumulator must be
    seeded below them (identity() = lowest(), as Naive's seed behaves: DESIGN.md 3.4)."""
    n, k, m = 130, 64, 96
    a, b = _oracle.fill(dtype, n, k, m)
    a = (-a).astype(a.dtype)
    b = (-b).astype(b.dtype)
    c, _ = g.matmul_capi(a, b, dtype, "Add", "Max", path=path)
    want = _oracle.naive(dtype, "Add", "Max", a, b)
    assert np.array_equal(c.view(np.uint8), want.view(np.uint8))
    assert np.all(c.astype(np.float64) < 0)
    # independent of the oracle: numpy in the native type (rounding is monotone, so max commutes with it)
    assert np.array_equal(c, (a[:, :, None] + b[None]).max(axis=1))


# ---- extra bench workload: uint8 (Multiply, Add) 32768^3 on the int8 matrix core ---------------------------
def test_uint8_32768_bit_exact_against_ordered_kernel_and_naive():
    """bench.py's workloads[] times uint8 32768^3; parity at that size: the whole output of the ping-pong
    MFMA kernel equals the k-ordered kernel's (wrap-around 8-bit arithmetic is order-independent), sampled
    rows x column blocks equal the CPU Naive, on full-range bytes (sign bit included)."""
    import torch
    n = k = m = 32768
    dev = torch.device("cuda:0")
    g0 = torch.Generator(device=dev)
    g0.manual_seed(91)
    a = torch.randint(0, 256, (n, k), dtype=torch.int16, device=dev, generator=g0).to(torch.uint8)
    b = torch.randint(0, 256, (k, m), dtype=torch.int16, device=dev, generator=g0).to(torch.uint8)
    assert g.kernel_name(g.make_config("uint8_t"), n, k, m) == "mfma_i8_256x256_pingpong_16x16x64"
    c = g.matmul(a, b, "uint8_t")
    torch.cuda.synchronize()
    rows, cols = _sample(n, m)
    ah, bh, ch = _host_slices(a, b, c, rows, cols)
    assert np.array_equal(ch, _oracle.naive("uint8_t", "Multiply", "Add", ah, bh))
    # the k-ordered kernel on row slabs (it is ~100x slower than the matrix core: 2048 rows are enough)
    for r0 in (0, n // 2 - 1024, n - 2048):
        c_ord = g.matmul(a[r0:r0 + 2048].contiguous(), b, "uint8_t", path=g.PATH_ORDERED)
        assert torch.equal(c_ord, c[r0:r0 + 2048]), r0
    assert torch.equal(g.matmul(a, b, "uint8_t"), c)
    # the same bytes read as int8_t give the same result bytes (u8 = s8 mod 2^8)
    c_s = g.matmul(a.view(torch.int8), b.view(torch.int8), "int8_t")
    assert torch.equal(c_s.view(torch.uint8), c)


# ---- C1: float 1024^3, the reference's own simulation path, LIVE next to the device ---------------------
def test_c1_float_1024_device_next_to_the_reference_kernel_itself():
    """BASELINE configs[0]: the reference's MatrixMultiplicationKernel (its kernel/*.cpp compiled against the
    test-only hlslib shim into oracle/_ref, the same call bench.py's cpu_baseline times, ~7 s on the box's host
    cores) on the seeded float 1024^3 inputs of test/TestSimulation.cpp:42-55 -- and the device result beside it:
    hw_emu / MM_PATH_ORDERED must give the reference kernel's bits (same k-ordered unfused chain,
    kernel/Compute.cpp:108-142), hw / MM_PATH_AUTO must agree within BASELINE's 1e-5."""
    if not _oracle.ref_available():
        pytest.skip("oracle/_ref not built in this snapshot (needs /root/reference at build time)")
    n = k = m = 1024
    a, b = _oracle.fill("float", n, k, m)
    c_ref = _oracle.ref_kernel("float", "Multiply", "Add", a, b)
    c_ord, _ = g.matmul_capi(a, b, path=g.PATH_ORDERED)
    assert np.array_equal(c_ord, c_ref)
    c, _ = g.matmul_capi(a, b)
    bad, first, worst = _oracle.compare("float", c, c_ref, 1e-5)
    assert bad == 0, (bad, first, worst)
    # the reference's own acceptance rule for this path (TestSimulation: 1e-3 against ReferenceImplementation)
    assert _oracle.compare("float", c_ref, _oracle.naive("float", "Multiply", "Add", a, b), 1e-3)[0] == 0


def test_half_reference_contract_next_to_the_reference_kernel_itself():
    """The reference's kernel for MM_DATA_TYPE=half (its kernel/*.cpp compiled against the test-only shim into oracle/_ref), run LIVE
    on its seeded 512 x 1024 x 544 inputs -- K = 1024 on [1,10): running sums reach ~31 000, where one binary16 ulp is 16, so
    every one of the 1024 roundings of each output matters -- and the device beside it: MM_PATH_AUTO under half_contract =
    reference and MM_PATH_ORDERED must both give the reference kernel's bits (kernel/Compute.cpp:129-133), from the tile kernel
    and from the anchor; the default contract (f32 accumulation, one rounding) is the more accurate one and so differs."""
    if not _oracle.ref_available("half", "Multiply", "Add"):
        pytest.skip("oracle/_ref (half build) not in this snapshot (needs /root/reference at build time)")
    n, k, m = 512, 1024, 544
    a, b = _oracle.fill("half", n, k, m)
    c_ref = _oracle.ref_kernel("half", "Multiply", "Add", a, b)
    assert np.isfinite(c_ref.astype(np.float32)).all()
    bits = lambda x: x.view(np.uint16)   # noqa: E731
    try:
        g.set_tuning("half_contract", 1)
        assert g.kernel_name(g.make_config("half"), n, k, m) == "ordered_tile"
        c_auto, _ = g.matmul_capi(a, b, "half")
        g.set_tuning("half_contract", -1)
        c_ord, _ = g.matmul_capi(a, b, "half", path=g.PATH_ORDERED)
        g.set_tuning("ordered_variant", 0)
        c_anchor, _ = g.matmul_capi(a, b, "half", path=g.PATH_ORDERED)
    finally:
        g.set_tuning("half_contract", -1)
        g.set_tuning("ordered_variant", -1)
    assert np.array_equal(bits(c_auto), bits(c_ref)) and np.array_equal(bits(c_ord), bits(c_ref)) and np.array_equal(bits(c_anchor), bits(c_ref))
    c_wide, _ = g.matmul_capi(a, b, "half")
    exact = a.astype(np.float64) @ b.astype(np.float64)
    err_ref = np.abs(c_ref.astype(np.float64) - exact) / exact
    err_wide = np.abs(c_wide.astype(np.float64) - exact) / exact
    assert err_wide.max() <= 2.0 ** -11 < err_ref.max()        # one rounding vs 1024 of them: which is why the default is not the reference's bits


def test_f32_rows_of_c_longer_than_the_32_bit_reach_of_the_interior_write_back():
    """the straight-line interior write-back of the shipped fp32 geometries addresses C with 32-bit
    byte offsets from a wavefront's first row (up to 64 rows x M x 4 B).  With M = 17 Mi floats that passes 4 GiB; the
    kernel must notice (wavefront-uniform guard) and take the 64-bit predicated path instead of wrapping around."""
    import torch
    n, k, m = 192, 16, 17 * 1024 * 1024
    gen = torch.Generator(device="cuda").manual_seed(5)
    a = torch.empty((n, k), dtype=torch.float32, device="cuda").uniform_(1, 10, generator=gen)
    b = torch.empty((k, m), dtype=torch.float32, device="cuda").uniform_(1, 10, generator=gen)
    for variant in (-1, 8, 35):
        g.set_tuning("f32_variant", variant)
        c = g.matmul(a, b)
        g.set_tuning("f32_variant", -1)
        cols = torch.tensor([0, 1, 4095, 8 * 1024 * 1024 + 5, 16 * 1024 * 1024 - 1, 16 * 1024 * 1024, m - 4, m - 1], device="cuda")
        want = a.double() @ b[:, cols].double()
        got = c[:, cols].double()
        assert float(((got - want).abs() / want).max()) < 1e-5, variant
        # and every row block got its own data: row r of C depends on row r of A only
        assert float((c[100, ::65537].double() - (a[100].double() @ b[:, ::65537].double())).abs().max() / 1e3) < 1e-5
        del c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,ops,size,floor_tops,kernel", [
    ("float", ("Multiply", "Add"), 16384, 0.92 * 157.3, "mfma_f32_256x256x16_w8_flush4096"),     # measured 150.2-150.8 (the 128 x 256 geometry: 151.2-152.6) across boxes
    ("double", ("Multiply", "Add"), 16384, 0.92 * 78.6, "mfma_f64_256x128x16_w8"),               # measured 75.7-76.4
    ("half", ("Multiply", "Add"), 32768, 1250.0, "mfma_f16_256x256_pingpong_16x16x32"),          # measured 1.44-1.51 PF (power-limited: box-dependent)
    ("float", ("Add", "Min"), 8192, 0.75 * 78.6, "valu_tile"),                                   # measured 65-75 TOp/s (cold / warm box)
], ids=lambda v: str(v) if not isinstance(v, tuple) else "_".join(v))
def test_throughput_floor_of_the_baseline_workloads(dtype, ops, size, floor_tops, kernel):
    """A regression guard, not a benchmark, and not a parity test: the kernels BASELINE.json's configs dispatch should
    stay within reach of their measured rates (floors 5-15 % below the slowest box seen), timed by HIP events through
    mm_gemm_launch after two warm-up launches, best of five.  WHICH kernel the configuration dispatches is asserted
    always; the rate is power-limited and box-dependent (clocks are not pinned, a box may be shared), so a rate under the
    floor FAILS only with MM_PERF_FLOORS=1 in the environment (the builder's own runs set it) and is a warning otherwise --
    a cold or busy box must not block the functional suite (collected last for the same reason: tests/conftest.py)."""
    import ctypes
    L = g.lib()
    cfg = g.make_config(dtype, *ops)
    assert g.kernel_name(cfg, size, size, size) == kernel
    es = L.mm_dtype_size(g.DTYPES[dtype])
    ptrs = [ctypes.c_void_p() for _ in range(3)]
    try:
        for p in ptrs:
            g._check(L.mm_alloc(0, size * size * es, ctypes.byref(p)))
        g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[0], size * size, 11))
        g._check(L.mm_fill_device(0, g.DTYPES[dtype], ptrs[1], size * size, 12))
        t, best = ctypes.c_double(0), 1e30
        for i in range(7):
            g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], size, size, size, ctypes.byref(t)))
            if i >= 2:
                best = min(best, t.value)
    finally:
        for p in ptrs:
            if p.value:
                L.mm_free(0, p)
    tops = 2.0 * size ** 3 / best / 1e12
    msg = f"REGRESSION GUARD (self-imposed throughput floor): {kernel}: {tops:.1f} TOp/s, floor {floor_tops:.1f}"
    if os.environ.get("MM_PERF_FLOORS") == "1":
        assert tops >= floor_tops, msg
    elif tops < floor_tops:
        import warnings
        warnings.warn(msg)
