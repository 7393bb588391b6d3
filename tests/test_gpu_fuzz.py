"""Randomised shape fuzz on the GPU: every fast family against an independent result, over shapes
that straddle tile edges, partial k-slabs, flush boundaries (fp32 flushes its accumulators every
4096 k) and the minimum sizes each family accepts.  Seeded, so failures reproduce."""
import os
import zlib

import numpy as np
import pytest

import _oracle
import gemm_hls_amd as g

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("MM_FUZZ_SCALE", "1"))  # MM_FUZZ_SCALE=10 for a long soak

#            dtype      k-multiple m-multiple  tolerance (None = bit-exact vs oracle Naive)
FAMILIES = {
    "float": (8, 4, 1e-5),
    "double": (8, 2, 1e-12),
    "half": (16, 8, 2.0 ** -10),
    "uint8_t": (32, 16, None),
    "int8_t": (32, 16, None),
}


def _shapes(rng, kmul, mmul, count):
    out = []
    for i in range(count):
        n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700]))
        m = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 32, 33, 40, 64, 65])) * mmul
        if i % 7 == 0:
            k = int(rng.choice([4096, 4096 + kmul, 8192 - kmul, 8192 + 2 * kmul]))  # around flush boundaries
            n, m = min(n, 300), min(m, 40 * mmul)
        else:
            k = int(rng.integers(1, 40)) * kmul
        out.append((n, k, m))
    return out


@pytest.mark.parametrize("dtype", list(FAMILIES))
def test_fuzz_fast_family_shapes(dtype):
    kmul, mmul, tol = FAMILIES[dtype]
    rng = np.random.default_rng(zlib.crc32(dtype.encode()))
    npdt = _oracle.NP_DTYPES[dtype]
    for (n, k, m) in _shapes(rng, kmul, mmul, 40 * SCALE):
        name = g.kernel_name(g.make_config(dtype), n, k, m)
        assert name.startswith("mfma_"), (dtype, n, k, m, name)
        if tol is None:
            a = rng.integers(0, 256, size=(n, k), dtype=np.uint8).view(npdt)
            b = rng.integers(0, 256, size=(k, m), dtype=np.uint8).view(npdt)
            want = _oracle.naive(dtype, "Multiply", "Add", a, b)
            c, _ = g.matmul_capi(a, b, dtype)
            assert np.array_equal(c, want), (dtype, n, k, m)
        else:
            lo, hi = (0.5, 2.0) if dtype == "half" else (1.0, 10.0)  # keep half sums finite at K = 8192
            a = rng.uniform(lo, hi, size=(n, k)).astype(npdt)
            b = rng.uniform(lo, hi, size=(k, m)).astype(npdt)
            exact = a.astype(np.float64) @ b.astype(np.float64)
            c, _ = g.matmul_capi(a, b, dtype)
            rel = np.abs(c.astype(np.float64) - exact) / exact
            assert rel.max() <= tol, (dtype, n, k, m, float(rel.max()))


@pytest.mark.parametrize("dtype,ops", [("float", ("Add", "Min")), ("int", ("Multiply", "Add")), ("double", ("Add", "Max")),
                                        ("long", ("Min", "Max")), ("int16_t", ("Multiply", "Add"))])
def test_fuzz_valu_tile_vs_ordered_and_oracle(dtype, ops):
    rng = np.random.default_rng(11)
    for (n, k, m) in _shapes(rng, 4, 4, 25 * SCALE):
        k = min(k, 600)
        a, b = _oracle.fill(dtype, n, k, m)
        c_fast, _ = g.matmul_capi(a, b, dtype, *ops)
        c_ord, _ = g.matmul_capi(a, b, dtype, *ops, path=g.PATH_ORDERED)
        assert np.array_equal(c_fast.view(np.uint8), c_ord.view(np.uint8)), (dtype, ops, n, k, m)
        assert np.array_equal(c_ord, _oracle.naive(dtype, ops[0], ops[1], a, b))


def test_fuzz_unaligned_shapes_take_the_predicated_path_and_stay_exact():
    rng = np.random.default_rng(5)
    for _ in range(20 * SCALE):
        n, k, m = (int(x) for x in rng.integers(1, 200, size=3))
        for dtype, ops in (("float", ("Multiply", "Add")), ("int", ("Multiply", "Add")), ("float", ("Add", "Min"))):
            a, b = _oracle.fill(dtype, n, k, m)
            c, _ = g.matmul_capi(a, b, dtype, *ops)
            want = _oracle.naive(dtype, ops[0], ops[1], a, b)
            if dtype == "float" and ops == ("Multiply", "Add"):
                assert _oracle.compare("float", c, want, 1e-5)[0] == 0, (n, k, m)
            else:
                assert np.array_equal(c, want), (dtype, ops, n, k, m)


def test_fuzz_split_path_any_shape_mixed_sign():
    """MM_PATH_SPLIT takes any N, K, M (the pack step pads): odd sizes straddling fragment (32), tile (128 / 256), slab
    (16) and flush (8256 k) boundaries, mixed-sign operands, both layouts of A, both tiles -- normwise against fp64 and
    bitwise between the two tile sizes."""
    rng = np.random.default_rng(2024)
    sizes = [1, 2, 31, 32, 33, 63, 65, 127, 129, 255, 256, 257, 300, 383, 511, 513, 700]
    for i in range(40 * SCALE):
        n, m = int(rng.choice(sizes)), int(rng.choice(sizes))
        k = int(rng.choice([8256, 8257, 8272, 16511])) if i % 8 == 0 else int(rng.integers(1, 700))
        if i % 8 == 0:
            n, m = min(n, 257), min(m, 257)
        a = rng.uniform(-3, 3, size=(n, k)).astype(np.float32)
        b = rng.uniform(-3, 3, size=(k, m)).astype(np.float32)
        exact = a.astype(np.float64) @ b.astype(np.float64)
        den = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
        out = []
        for variant in (256, 512):                       # pinned 256 x 256 / 128 x 128 tile
            g.set_tuning("split_variant", variant)
            c, _ = g.matmul_capi(a, b, path=g.PATH_SPLIT)
            out.append(c)
        g.set_tuning("split_variant", -1)
        assert np.array_equal(out[0], out[1]), (n, k, m)
        assert float(np.max(np.abs(out[0] - exact) / np.maximum(den, 1e-300))) < 2.0 ** -19, (n, k, m)
        if i % 4 == 0:
            ct, _ = g.matmul_capi(np.ascontiguousarray(a.T), b, path=g.PATH_SPLIT, transposed_a=True)
            assert np.array_equal(ct, out[0]), (n, k, m)


@pytest.mark.parametrize("dtype", ["half", "float", "double"])
def test_fuzz_k_ordered_tile_kernel_is_naive_bit_for_bit(dtype):
    """MM_PATH_ORDERED (Multiply, Add) on the floating types -- the unfused, k-ascending contract of the reference's kernel
    (kernel/Compute.cpp:129-133), binary16 accumulating in binary16 for half -- through the register-tile kernel
    (mm_valu_tile_fp_exact.hip), on shapes with ragged N, one-chunk M, K below / not a multiple of the slab depth (synchronous
    staging, the shifted last slab) and K x N A: the bits of the CPU oracle's Naive and of the 64 x 64 anchor kernel."""
    rng = np.random.default_rng(zlib.crc32(("ordered" + dtype).encode()))
    for i, (n, k, m) in enumerate(_shapes(rng, 4, 4, 30 * SCALE)):
        k = min(k, 700)
        ta = i % 3 == 2
        if ta:
            n = (n + 3) // 4 * 4                              # a K x N A is served by the tile kernel when N % 4 == 0
        a, b = _oracle.fill(dtype, n, k, m)
        if dtype == "half":                                   # keep most binary16 sums finite so that the bits say something
            a, b = (a * np.float16(0.125)).astype(np.float16), (b * np.float16(0.25)).astype(np.float16)
        want = _oracle.naive(dtype, "Multiply", "Add", a, b)
        a_dev = np.ascontiguousarray(a.T) if ta else a
        assert g.kernel_name(g.make_config(dtype, path=g.PATH_ORDERED, transposed_a=ta), n, k, m) == "ordered_tile"
        c_tile, _ = g.matmul_capi(a_dev, b, dtype, path=g.PATH_ORDERED, transposed_a=ta)
        assert np.array_equal(c_tile.view(np.uint8), want.view(np.uint8)), (dtype, n, k, m, ta)
        if i % 5 == 0:
            g.set_tuning("ordered_variant", 0)
            try:
                c_anchor, _ = g.matmul_capi(a_dev, b, dtype, path=g.PATH_ORDERED, transposed_a=ta)
            finally:
                g.set_tuning("ordered_variant", -1)
            assert np.array_equal(c_anchor.view(np.uint8), want.view(np.uint8)), (dtype, n, k, m, ta)
