"""CPU tests (no GPU): pin the oracle (oracle/mm_oracle.c) against
  * the committed golden draws of the reference's generator (tests/golden/rng_golden.json,
    produced by libstdc++'s <random> exactly as host/RunHardware.cpp:31-35 uses it),
  * the committed outputs of the reference's OWN kernel (tests/golden/ref_*.npz, produced from
    /root/reference/kernel/*.cpp via oracle/_ref),
  * the reference's own kernel run live, when oracle/_ref is present,
  * independent numpy arithmetic.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_rng_real_draws_match_libstdcxx_golden():
    g = json.load(open(os.path.join(GOLD, "rng_golden.json")))
    want = np.array([float.fromhex(h) for h in g["real_hex"]])
    got = _oracle.draws_real(len(want))
    assert np.array_equal(got, want)
    # the four draws SURVEY.md quotes
    assert [repr(float(x)) for x in got[:4]] == [
        "6.919200465008035", "3.639255936853937", "1.8531633837525048", "4.548912257521801"]


def test_rng_int_draws_match_libstdcxx_golden():
    g = json.load(open(os.path.join(GOLD, "rng_golden.json")))
    want = np.array(g["int"], dtype=np.uint64)
    got = _oracle.draws_int(len(want))
    assert np.array_equal(got, want)
    assert list(got[:12]) == [1, 7, 8, 3, 7, 1, 3, 4, 4, 7, 10, 6]


def test_fill_is_a_then_b_from_one_stream():
    a, b = _oracle.fill("float", 5, 16, 16)
    draws = _oracle.draws_real(5 * 16 + 16 * 16)
    assert np.array_equal(a.ravel(), draws[:80].astype(np.float32))
    assert np.array_equal(b.ravel(), draws[80:].astype(np.float32))
    ai, bi = _oracle.fill("int", 3, 16, 16)
    di = _oracle.draws_int(3 * 16 + 16 * 16)
    assert np.array_equal(ai.ravel(), di[:48].astype(np.int32))
    assert np.array_equal(bi.ravel(), di[48:].astype(np.int32))
    # half is neither is_integral nor is_floating_point -> REAL distribution, rounded to binary16
    ah, _ = _oracle.fill("half", 2, 16, 16)
    assert np.array_equal(ah.ravel(), draws[:32].astype(np.float16))


def test_half_conversion_matches_numpy_rne():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-70000, 70000, 5000), rng.uniform(-1e-4, 1e-4, 5000),
                         np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25])])
    L = _oracle.lib()
    L.mm_oracle_double_to_half.argtypes = [__import__("ctypes").c_double]
    L.mm_oracle_double_to_half.restype = __import__("ctypes").c_uint16
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.mm_oracle_double_to_half(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,k,m", [(1, 16, 16), (37, 32, 48), (65, 80, 96)])
def test_naive_float_multiply_add_is_k_ordered_unfused_chain(n, k, m):
    a, b = _oracle.fill("float", n, k, m)
    c = _oracle.naive("float", "Multiply", "Add", a, b)
    want = np.zeros((n, m), dtype=np.float32)
    for kk in range(k):  # acc = acc + (a*b): two float32 roundings per step, k ascending
        want = (want + (a[:, kk:kk + 1] * b[kk:kk + 1, :]).astype(np.float32)).astype(np.float32)
    assert np.array_equal(c, want)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(c - exact) / exact) < 1e-5


@pytest.mark.parametrize("dtype", ["float", "double", "int", "unsigned", "uint8_t", "int8_t", "long", "int16_t"])
@pytest.mark.parametrize("ops", [("Multiply", "Add"), ("Add", "Min"), ("Add", "Max"), ("Min", "Max"), ("Max", "Min")])
def test_naive_matches_numpy_semiring(dtype, ops):
    mp, rd = ops
    n, k, m = 9, 32, 16
    a, b = _oracle.fill(dtype, n, k, m)
    c = _oracle.naive(dtype, mp, rd, a, b, threads=3)
    npdt = _oracle.NP_DTYPES[dtype]
    wide = np.float64 if npdt in (np.float32, np.float64) else np.int64
    A = a.astype(wide)[:, :, None]
    B = b.astype(wide)[None, :, :]
    mapped = {"Multiply": A * B, "Add": A + B, "Min": np.minimum(A, B), "Max": np.maximum(A, B)}[mp]
    if npdt not in (np.float32, np.float64):
        mapped = mapped.astype(npdt).astype(wide)  # Map result is a Data_t
    red = {"Add": mapped.sum(axis=1), "Min": mapped.min(axis=1), "Max": mapped.max(axis=1)}[rd]
    if npdt in (np.float32, np.float64):
        assert np.allclose(c, red, rtol=1e-5)
    else:
        assert np.array_equal(c, red.astype(npdt))  # modular wrap-around like C++ narrowing


def test_naive_transposed_a_indexing():
    n, k, m = 7, 16, 16
    a, b = _oracle.fill("int", n, k, m)
    c = _oracle.naive("int", "Multiply", "Add", a, b)
    ct = _oracle.naive("int", "Multiply", "Add", np.ascontiguousarray(a.T), b, transposed_a=True)
    assert np.array_equal(c, ct)


def test_naive_half_reference_vs_wide_contract():
    n, k, m = 8, 64, 32
    a, b = _oracle.fill("half", n, k, m)
    ref = _oracle.naive("half", "Multiply", "Add", a, b)            # half accumulator (reference)
    wide = _oracle.naive("half", "Multiply", "Add", a, b, wide_half=True)  # one final rounding
    exact = a.astype(np.float64) @ b.astype(np.float64)
    assert np.array_equal(wide, exact.astype(np.float16))
    # the reference's half-accumulate drifts by a few half-ulps at K=64, never by more than 1%
    assert np.max(np.abs(ref.astype(np.float64) - exact) / exact) < 1e-2


def test_compare_rule():
    ref = np.full((2, 16), 100.0, dtype=np.float32)
    test = ref.copy()
    test[1, 3] = 100.2
    bad, first, worst = _oracle.compare("float", test, ref, 1e-3)
    assert (bad, first) == (1, 16 + 3) and abs(worst - 2e-3) < 1e-5
    assert _oracle.compare("float", test, ref, 1e-2)[0] == 0
    ri = np.arange(32, dtype=np.int32).reshape(2, 16)
    ti = ri.copy()
    assert _oracle.compare("int", ti, ri, 0)[0] == 0
    ti[0, 5] += 1
    assert _oracle.compare("int", ti, ri, 0)[:2] == (1, 5)


# ---- against the reference's own kernel -------------------------------------------------------
def _golden_cases():
    for path in sorted(glob.glob(os.path.join(GOLD, "ref_*_*x*x*.npz"))):
        stem = os.path.basename(path)[4:-4]
        ta = stem.startswith("transposedA_")          # the reference's -DMM_TRANSPOSED_A build: A is K x N
        dtype, mp, rd, shape = stem[len("transposedA_") if ta else 0:].rsplit("_", 3)
        yield pytest.param(path, dtype, mp, rd, tuple(int(x) for x in shape.split("x")), ta, id=stem)


@pytest.mark.parametrize("path,dtype,mp,rd,shape,ta", list(_golden_cases()))
def test_oracle_reproduces_reference_kernel_golden(path, dtype, mp, rd, shape, ta):
    n, k, m = shape
    z = np.load(path)
    a, b = _oracle.fill(dtype, n, k, m, transposed_a=ta)
    assert a.shape == ((k, n) if ta else (n, k))
    assert hashlib.sha256(a.tobytes()).hexdigest() == str(z["a_sha256"])
    assert hashlib.sha256(b.tobytes()).hexdigest() == str(z["b_sha256"])
    c_ref = z["c"]
    if (mp, rd) == ("Add", "Min"):
        # Known reference defect: the HLS kernel seeds k=0 with literal 0
        # (kernel/Compute.cpp:116-118) instead of OperatorReduce::identity() as Naive does
        # (include/Utility.h:29), so min-plus on positive inputs collapses to 0 there.
        assert np.all(c_ref == 0)
        c = _oracle.naive(dtype, mp, rd, a, b)
        want = (a.astype(np.float64)[:, :, None] + b.astype(np.float64)[None]).min(axis=1)
        assert np.array_equal(c, want.astype(np.float32))
        return
    c = _oracle.naive(dtype, mp, rd, a, b, transposed_a=ta)
    # The HLS kernel accumulates k = 0..K-1 into one accumulator starting from 0, unfused:
    # the same chain as Naive -> bit-identical, floats and doubles included; uint8_t wraps mod 2^8
    # in both; with MM_TRANSPOSED_A only the indexing of `a` changes (include/Utility.h:31-35).
    assert np.array_equal(c, c_ref)
    if ta:  # and the K x N layout is the same product as the row-major one
        assert np.array_equal(_oracle.naive(dtype, mp, rd, np.ascontiguousarray(a.T), b), c_ref)


def test_goldens_cover_every_reference_build_the_oracle_makefile_names():
    have = {os.path.basename(p)[4:-4].rsplit("_", 1)[0] for p in glob.glob(os.path.join(GOLD, "ref_*_*x*x*.npz"))}
    assert have >= {"float_Multiply_Add", "float_Add_Min", "int_Multiply_Add", "half_Multiply_Add", "double_Multiply_Add",
                    "uint8_t_Multiply_Add", "transposedA_float_Multiply_Add", "transposedA_int_Multiply_Add"}


def test_oracle_reproduces_reference_ctest_shape_checksum():
    checks = json.load(open(os.path.join(GOLD, "ref_checksums.json")))
    assert set(checks) >= {"float_Multiply_Add", "int_Multiply_Add", "half_Multiply_Add", "double_Multiply_Add",
                           "uint8_t_Multiply_Add", "transposedA_float_Multiply_Add", "transposedA_int_Multiply_Add"}
    for cfg, chk in checks.items():
        ta = cfg.startswith("transposedA_")
        dtype, mp, rd = cfg[len("transposedA_") if ta else 0:].rsplit("_", 2)
        if (mp, rd) == ("Add", "Min"):
            continue   # reference defect H4, see above
        n, k, m = chk["shape"]
        a, b = _oracle.fill(dtype, n, k, m, transposed_a=ta)
        assert hashlib.sha256(a.tobytes()).hexdigest() == chk["a_sha256"]
        c = _oracle.naive(dtype, mp, rd, a, b, transposed_a=ta)
        assert hashlib.sha256(c.tobytes()).hexdigest() == chk["c_sha256"], cfg


@pytest.mark.skipif(not _oracle.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_kernel_live_ragged_tiles():
    # ragged last N tile and one bus word into a second M tile, like the reference's CTest sizing
    n, k, m = 257, 64, 272
    a, b = _oracle.fill("float", n, k, m)
    c_ref = _oracle.ref_kernel("float", "Multiply", "Add", a, b)
    assert np.array_equal(_oracle.naive("float", "Multiply", "Add", a, b), c_ref)
    bad, _, worst = _oracle.compare("float", c_ref, (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32), 1e-5)
    assert bad == 0 and worst < 1e-5


@pytest.mark.parametrize("dtype,ta,shape", [("double", False, (257, 64, 264)), ("uint8_t", False, (257, 128, 320)),
                                            ("float", True, (272, 64, 272)), ("int", True, (16, 32, 272))])
def test_reference_kernel_live_other_builds(dtype, ta, shape):
    """The double, uint8_t and MM_TRANSPOSED_A builds of the reference's kernel, run live next to the restatement."""
    if not _oracle.ref_available(dtype, "Multiply", "Add", transposed_a=ta):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m, transposed_a=ta)
    c_ref = _oracle.ref_kernel(dtype, "Multiply", "Add", a, b, transposed_a=ta)
    assert np.array_equal(_oracle.naive(dtype, "Multiply", "Add", a, b, transposed_a=ta), c_ref)
