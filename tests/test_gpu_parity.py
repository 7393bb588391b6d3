"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(include/mm_gemm.h): the HIP path is compared with the CPU oracle on the reference's seeded
inputs, with the committed outputs of the reference's own kernel, and -- at BASELINE.json's full
size -- through size-independent properties.

Tolerances (BASELINE.json north_star): float within 1e-5 relative of the BLAS reference, applied
with the reference's own rule |test-ref|/ref (test/TestSimulation.cpp:75-92); integer semirings,
min/max semirings and the ORDERED path bit-exact."""
import glob
import os

import numpy as np
import pytest

import _bounds
import _oracle
import gemm_hls_amd as g

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32_TOL = 1e-5

# reference CTest shape (CMakeLists.txt:155-159) + edge shapes: single row, ragged N tile,
# ragged M tile, K with a partial slab, tiny
SHAPES = [(513, 528, 528), (1, 16, 16), (37, 32, 48), (300, 64, 272), (129, 80, 260), (256, 8, 4), (1024, 1024, 1024)]


@pytest.fixture(autouse=True)
def _default_tuning():
    """Every test starts and ends on the library's own geometry choice (knobs are process-wide)."""
    knobs = ("f32_variant", "f64_variant", "f16_variant", "i8_variant", "band_rows", "valu_variant", "f32_splitk", "ordered_variant",
             "half_contract")
    for knob in knobs:
        g.set_tuning(knob, -1)
    yield
    for knob in knobs:
        g.set_tuning(knob, -1)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_f32_mfma_default_vs_blas_and_exact(shape):
    n, k, m = shape
    a, b = _oracle.fill("float", n, k, m)
    assert g.kernel_name(g.make_config("float"), n, k, m).startswith("mfma_f32")
    c, _ = g.matmul_capi(a, b)
    blas = a @ b  # numpy float32 matmul == cblas_sgemm, the reference's ReferenceImplementation
    bad, first, worst = _oracle.compare("float", c, blas, F32_TOL)
    assert bad == 0, (first, worst)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(c - exact) / exact) < F32_TOL


F32_VARIANTS = [33, 8, 35, 64, 0, 3]   # the product's geometries (tests/test_capi_symbols.py pins this list to the library's)


@pytest.mark.parametrize("variant", F32_VARIANTS)
@pytest.mark.parametrize("shape", [(513, 528, 528), (300, 64, 272), (256, 8, 4), (37, 32, 48), (300, 8208, 272)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f32_mfma_every_variant(variant, shape):
    g.set_tuning("f32_variant", variant)
    n, k, m = shape
    a, b = _oracle.fill("float", n, k, m)
    c, _ = g.matmul_capi(a, b)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    rel = np.abs(c - exact) / exact
    assert np.max(rel) < F32_TOL, (variant, np.unravel_index(np.argmax(rel), rel.shape), rel.max())


@pytest.mark.parametrize("shape", [(1024, 8208, 512), (513, 4112, 528), (300, 64, 272), (257, 12304, 260), (300, 4128, 272), (260, 8216, 132), (130, 4120, 256),
                                   (129, 12320, 260)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f32_shipped_geometries_are_bit_identical_to_each_other(shape):
    """The four geometries MM_PATH_AUTO picks from -- 128x256 (33), 256x256 (8), 128x128x32 (35), 64x64x32 (64) -- issue the same MFMAs
    per output element in the same k order and flush into C at the same k (every 4096; one, two and three flushes in
    these shapes), so whichever the shape-adaptive pick takes, the bits are the same; also with mixed signs.  The
    compiler-scheduled single-chain geometry (3) has the same order without the flush: identical while K <= 4096 + a slab."""
    n, k, m = shape
    rng = np.random.default_rng(k)
    a = rng.uniform(-3, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 10, size=(k, m)).astype(np.float32)
    out = {}
    for v in (8, 33, 35, 64, 3):
        g.set_tuning("f32_variant", v)
        out[v], _ = g.matmul_capi(a, b)
    g.set_tuning("f32_variant", -1)
    assert np.array_equal(out[8], out[33]) and np.array_equal(out[8], out[35]) and np.array_equal(out[8], out[64])
    if k <= 4096 + 32:
        assert np.array_equal(out[8], out[3])
    else:
        assert not np.array_equal(out[8], out[3])      # a different (unbounded) chain: close, not equal
        assert np.max(np.abs(out[8] - out[3]) / np.maximum(np.abs(out[3]), 1e-3)) < 1e-3


@pytest.mark.parametrize("shape", [(1024, 1024, 1024), (513, 1032, 520), (1000, 96, 1500), (768, 768, 768), (1061, 512, 1024), (512, 512, 512),
                                   (1280, 1280, 1280), (1024, 4128, 1024)], ids=lambda s: "x".join(map(str, s)))
def test_f32_small_problems_take_the_64x64_geometry_with_the_bits_of_the_others(shape):
    """Below a round of 128 x 128 tiles the shape-adaptive pick takes the 64 x 64 geometry (one 32 x 32 accumulator per
    wavefront, whole K, no second kernel) where its time model beats split-K.  Same MFMA chain per output element and
    the same flush rule as the other shipped geometries: identical bits to the unsplit 128 x 128 and 256 x 256 kernels."""
    n, k, m = shape
    rng = np.random.default_rng(n + k)
    a = rng.uniform(-3, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 10, size=(k, m)).astype(np.float32)
    name = g.kernel_name(g.make_config("float"), n, k, m)
    info = g.kernel_info(g.make_config("float"), n, k, m)
    assert (info.tile_n, info.tile_m, info.tile_k, info.wavefronts) == (64, 64, 32, 4)
    c, _ = g.matmul_capi(a, b)
    out = {}
    try:
        g.set_tuning("f32_splitk", 1)
        for v in (35, 8):
            g.set_tuning("f32_variant", v)
            out[v], _ = g.matmul_capi(a, b)
    finally:
        g.set_tuning("f32_splitk", -1)
        g.set_tuning("f32_variant", -1)
    assert name == "mfma_f32_64x64x32_w4x2_flush4096", name
    assert np.array_equal(c, out[35]) and np.array_equal(c, out[8])
    exact = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    _bounds.north_star(_bounds.normwise(c, exact, scale), name)
    _bounds.guard(_bounds.normwise(c, exact, scale), _bounds.f32_chain_guard(k), name)   # longer unsplit chains drift further


@pytest.mark.parametrize("shape", [(64, 8, 64), (1, 8, 4), (65, 24, 68), (70, 16, 132), (129, 40, 260), (64, 32, 64), (200, 8224, 136), (3, 4120, 8)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f32_64x64_geometry_edges(shape):
    """f32_variant 64 pinned on shapes its own rule would not take: K shorter than a slab (the per-lane clamped staging),
    a partial last slab (fetched as the last 32 k of the matrix and consumed from the end of its stage), ragged N / M,
    flush boundaries; bit-identical to the 128 x 128 geometry."""
    n, k, m = shape
    rng = np.random.default_rng(n * k)
    a = rng.uniform(-3, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 10, size=(k, m)).astype(np.float32)
    try:
        g.set_tuning("f32_splitk", 1)
        g.set_tuning("f32_variant", 64)
        assert g.kernel_name(g.make_config("float"), n, k, m) == "mfma_f32_64x64x32_w4x2_flush4096"
        c64, _ = g.matmul_capi(a, b)
        g.set_tuning("f32_variant", 35)
        c35, _ = g.matmul_capi(a, b)
    finally:
        g.set_tuning("f32_splitk", -1)
        g.set_tuning("f32_variant", -1)
    assert np.array_equal(c64, c35)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    _bounds.north_star(_bounds.normwise(c64, exact, scale), "64x64 geometry")
    _bounds.guard(_bounds.normwise(c64, exact, scale), _bounds.f32_chain_guard(k), "64x64 geometry")


@pytest.mark.parametrize("shape,splitk,expect", [((512, 4096, 512), -1, "mfma_f32_64x64x32_w4x2_splitk4"), ((256, 8192, 256), -1, "mfma_f32_64x64x32_w4x2_splitk8"),
                                                 ((300, 2048, 272), 3, "mfma_f32_64x64x32_w4x2_splitk3"), ((129, 4104, 132), 8, "mfma_f32_64x64x32_w4x2_splitk8"),
                                                 ((70, 200, 68), 2, "mfma_f32_64x64x32_w4x2_splitk2")],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_f32_64x64_geometry_splits_k_for_few_tiles_and_long_k(shape, splitk, expect):
    """Few tiles of 64 x 64 and a long K (512 x 4096 x 512: 64 tiles for 256 CUs): the small geometry runs up to 8 copies of its
    tile grid on K chunks of >= 512 and the ordered reduce kernel adds the planes, when its time model says the second kernel
    pays (auto cases) or when forced (f32_variant 64 + f32_splitk 2..8; chunks that end mid-slab included).  Deterministic;
    close to the unsplit kernel, not the same summation order."""
    n, k, m = shape
    rng = np.random.default_rng(k + n)
    a = rng.uniform(-3, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 10, size=(k, m)).astype(np.float32)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    try:
        if splitk > 1:
            g.set_tuning("f32_variant", 64)
            g.set_tuning("f32_splitk", splitk)
        name = g.kernel_name(g.make_config("float"), n, k, m)
        c1, _ = g.matmul_capi(a, b)
        c2, _ = g.matmul_capi(a, b)
        g.set_tuning("f32_variant", 64)
        g.set_tuning("f32_splitk", 1)
        c_one, _ = g.matmul_capi(a, b)
    finally:
        g.set_tuning("f32_splitk", -1)
        g.set_tuning("f32_variant", -1)
    assert name == expect, name
    assert np.array_equal(c1, c2)
    _bounds.north_star(_bounds.normwise(c1, exact, scale), "split")
    _bounds.guard(_bounds.normwise(c1, exact, scale), _bounds.f32_chain_guard(k), "split vs exact")
    _bounds.guard(_bounds.normwise(c1, c_one, scale), 5e-6, "split vs unsplit")


@pytest.mark.parametrize("shape,splitk", [((512, 4096, 512), -1), ((1024, 1024, 1024), 2), ((300, 2048, 272), 8), ((129, 4104, 132), 3),
                                          ((257, 4104, 260), -1), ((1536, 1536, 1536), -1), ((640, 520, 384), 2), ((1024, 1024, 1024), 4)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else f"splitk{v}")
def test_f32_split_k_for_small_problems_is_deterministic_and_accurate(shape, splitk):
    """problems that cannot fill the chip with whole 128 x 128 tiles are cut along K into up to 8
    chunks; the copies of the tile grid run side by side and a second kernel adds the partial planes in ascending
    order.  Same bits on every launch, inside 1e-5 of fp64 like every fp32 fast path, K chunks that end mid-slab
    (4104 / 3, 520 / 2) included; the kernel name says what ran."""
    n, k, m = shape
    rng = np.random.default_rng(k + n)
    a = rng.uniform(-3, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 10, size=(k, m)).astype(np.float32)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    try:
        g.set_tuning("f32_splitk", splitk)
        if splitk > 1:
            g.set_tuning("f32_variant", 35)
        name = g.kernel_name(g.make_config("float"), n, k, m)
        c1, _ = g.matmul_capi(a, b)
        c2, _ = g.matmul_capi(a, b)
        g.set_tuning("f32_splitk", 1)
        c_one, _ = g.matmul_capi(a, b)
    finally:
        g.set_tuning("f32_splitk", -1)
        g.set_tuning("f32_variant", -1)
    assert "splitk" in name, name                      # every case here is small enough to split
    assert np.array_equal(c1, c2)
    _bounds.north_star(_bounds.normwise(c1, exact, scale), name)
    _bounds.guard(_bounds.normwise(c1, exact, scale), _bounds.f32_chain_guard(k), name + " vs exact")
    _bounds.guard(_bounds.normwise(c1, c_one, scale), 5e-6, name + " vs unsplit")   # close to the unsplit kernel, not the same summation order


@pytest.mark.parametrize("shape,forced", [((2304, 2304, 2304), False), ((3072, 1056, 520), True), ((1000, 96, 3000), True),
                                          ((2560, 512, 2560), False), ((129, 4096, 132), True), ((3584, 256, 3584), False), ((100, 96, 120), True), ((300, 64, 272), True),
                                          ((5120, 256, 5120), False), ((2341, 2304, 2304), True), ((128, 32768, 128), True), ((2432, 288, 3712), False),
                                          ((2304, 8448, 2304), False)],   # a workgroup's part of a tile > 4128 k: the chain is flushed INTO the scratch slot
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else ("forced" if v else "auto"))
def test_f32_stream_k_for_partial_rounds_is_deterministic_and_accurate(shape, forced):
    """Problems of a few partial rounds of 128 x 128 tiles are dealt out to 512 persistent workgroups in equal ranges of
    (tile, slab) units -- to teams of up to 4 x 4 workgroups on neighbouring tiles, so that an XCD's L2 still shares the
    slabs.  A tile cut by a range boundary is finished in ascending k from scratch slots, in two interchangeable ways:
    the LAST PART TO ARRIVE gathers (what MM_PATH_AUTO runs, f32_splitk 0: one kernel, every part raises a flag and then
    looks at the others' -- nobody waits), or a small second kernel gathers (11, the cross-check), or every part draws a
    TICKET from a per-tile counter with one agent-scope acq_rel read-modify-write after a release fence and the last ticket
    gathers (12: the canonical last-block pattern, correct by the language's memory model alone; 0.2-30 % slower than the
    shipped form -- its release fence writes the L2 back -- profiles/r06d_*).  All perform the same additions in the same
    order: BIT-IDENTICAL.  (Rounds 3-4's form, in which the owner of the lowest-k part waited for the
    others inside the launch, is retired: it gave these bits too, was no faster, and could not be made safe next to other
    processes.)  Same bits on every launch; inside the fp32 bound; picked by the shape-adaptive rule where it pays (`auto` cases) and
    forcible (ragged N / M, K of 3 slabs, ranges shorter than a tile, one tile cut 512 ways).  f32_splitk = 9 is the
    single-range form with its own fix-up kernel: an independent implementation of the same idea (different cut points,
    different bits)."""
    n, k, m = shape
    rng = np.random.default_rng(k + n)
    a = rng.uniform(-3, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(-3, 10, size=(k, m)).astype(np.float32)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    try:
        if forced:
            g.set_tuning("f32_variant", 35)
            g.set_tuning("f32_splitk", 0)
        name = g.kernel_name(g.make_config("float"), n, k, m)
        runs = [g.matmul_capi(a, b)[0] for _ in range(3)]
        g.set_tuning("f32_variant", 35)
        g.set_tuning("f32_splitk", 11)
        name_two = g.kernel_name(g.make_config("float"), n, k, m)
        c_two = [g.matmul_capi(a, b)[0] for _ in range(2)]
        g.set_tuning("f32_splitk", 12)      # the counter-ticket form: the last-arriver idea in the language's memory model
        name_ticket = g.kernel_name(g.make_config("float"), n, k, m)
        c_ticket = [g.matmul_capi(a, b)[0] for _ in range(2)]
        g.set_tuning("f32_splitk", 9)
        name_fixup = g.kernel_name(g.make_config("float"), n, k, m)
        c_fixup, _ = g.matmul_capi(a, b)
        g.set_tuning("f32_splitk", 1)
        c_one, _ = g.matmul_capi(a, b)
    finally:
        g.set_tuning("f32_splitk", -1)
        g.set_tuning("f32_variant", -1)
    assert name == "mfma_f32_128x128x32_w4x2_streamk", name
    assert name_fixup == "mfma_f32_128x128x32_w4x2_streamk_fixup", name_fixup
    c1 = runs[0]
    assert np.array_equal(c1, runs[1]) and np.array_equal(c1, runs[2])
    assert name_two == "mfma_f32_128x128x32_w4x2_streamk_two_kernels", name_two
    assert np.array_equal(c1, c_two[0]) and np.array_equal(c1, c_two[1]), "last-arriver form != two-kernel form"
    assert name_ticket == "mfma_f32_128x128x32_w4x2_streamk_ticket", name_ticket
    assert np.array_equal(c1, c_ticket[0]) and np.array_equal(c1, c_ticket[1]), "last-arriver form != counter-ticket form"
    # the bar first (BASELINE.json north_star, applied normwise on this mixed-sign data) ...
    for what, c in (("teams + fix-up", c1), ("single ranges + fix-up", c_fixup), ("unsplit", c_one)):
        _bounds.north_star(_bounds.normwise(c, exact, scale), f"{what} {n}x{k}x{m}")
    # ... then the self-imposed guards: chain-length aware (a workgroup's part of a tile can be a chain of up to 4096 k
    # before it is flushed; measured 2.2e-6 at K = 8448, 1.4-1.6e-6 at K <= 2304)
    _bounds.guard(_bounds.normwise(c1, exact, scale), _bounds.f32_chain_guard(k), "teams + fix-up vs exact")
    _bounds.guard(_bounds.normwise(c_fixup, exact, scale), _bounds.f32_chain_guard(k), "single ranges + fix-up vs exact")
    _bounds.guard(_bounds.normwise(c1, c_one, scale), 5e-6, "teams + fix-up vs unsplit")
    _bounds.guard(_bounds.normwise(c1, c_fixup, scale), 5e-6, "teams + fix-up vs single ranges + fix-up")


def test_f32_stream_k_flags_need_no_clearing_between_launches_and_survive_a_release():
    """The stream-K flags hold the launch's epoch (a process-wide 64-bit count) in memory that never holds anything else, so
    a launch does not clear them: back-to-back launches of different shapes reuse the same block, mm_release_workspace hands
    it back to the driver and the next launch clears its fresh block once.  Same bits throughout."""
    import torch
    dev = torch.device("cuda:0")
    shapes = [(2341, 2304, 2304), (3584, 512, 3584), (2560, 256, 2560)]
    ops, alone = [], []
    for i, (n, k, m) in enumerate(shapes):
        a = torch.empty((n, k), device=dev).uniform_(-3, 10, generator=torch.Generator(device=dev).manual_seed(20 + i))
        b = torch.empty((k, m), device=dev).uniform_(-3, 10, generator=torch.Generator(device=dev).manual_seed(30 + i))
        assert g.kernel_name(g.make_config("float"), n, k, m).endswith("streamk")
        ops.append((a, b))
        alone.append(g.matmul(a, b).clone())        # the last-arriver form MM_PATH_AUTO runs: the one with flags
    torch.cuda.synchronize()
    try:
        for round_ in range(3):
            for rep in range(20):
                for i, (a, b) in enumerate(ops):
                    assert torch.equal(g.matmul(a, b), alone[i]), (round_, rep, shapes[i])
            torch.cuda.synchronize()
            g._check(g.lib().mm_release_workspace(0))
    finally:
        g.set_tuning("f32_variant", -1)
        g.set_tuning("f32_splitk", -1)


@pytest.mark.parametrize("form", ["default", "two_kernel"])
def test_f32_stream_k_launches_sharing_the_chip_make_progress_and_keep_their_bits(form):
    """Two stream-K launches on two streams, plus a third stream of whole-tile launches, compete for the same CUs, so
    neither has all of its workgroups resident.  The last part of a cut tile to arrive gathers it and nobody waits inside a
    kernel, so parts of one tile may finish in any order, under any residency: each launch must finish and give the bits it
    gives alone (default; two_kernel = the fix-up-kernel cross-check, f32_splitk 11).  Run under a watchdog all the same."""
    import torch
    dev = torch.device("cuda:0")
    shapes = [(2341, 2304, 2304), (3584, 512, 3584), (2048, 2048, 2048)]
    ops = []
    for i, (n, k, m) in enumerate(shapes):
        a = torch.empty((n, k), device=dev).uniform_(-3, 10, generator=torch.Generator(device=dev).manual_seed(i))
        b = torch.empty((k, m), device=dev).uniform_(-3, 10, generator=torch.Generator(device=dev).manual_seed(10 + i))
        ops.append((a, b))
    names = [g.kernel_name(g.make_config("float"), *sh) for sh in shapes]
    assert names[0].endswith("streamk") and names[1].endswith("streamk") and "streamk" not in names[2], names
    alone = [g.matmul(a, b).clone() for a, b in ops]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in shapes]
    outs = [[] for _ in shapes]
    try:
        for rep in range(25):
            for i, (a, b) in enumerate(ops):
                stream_k = form == "two_kernel" and names[i].endswith("streamk")
                g.set_tuning("f32_variant", 35 if stream_k else -1)      # (read when the launch is enqueued)
                g.set_tuning("f32_splitk", 11 if stream_k else -1)
                with torch.cuda.stream(streams[i]):
                    outs[i].append(g.matmul(a, b))
    finally:
        g.set_tuning("f32_variant", -1)
        g.set_tuning("f32_splitk", -1)
    done = torch.cuda.Event()
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    done.record()
    import time
    t0 = time.time()
    while not done.query():
        assert time.time() - t0 < 60, "stream-K launches sharing the chip did not finish"
        time.sleep(0.01)
    for i in range(len(shapes)):
        for c in outs[i]:
            assert torch.equal(c, alone[i]), names[i]


@pytest.mark.parametrize("native", [True, False], ids=["kxn_kernel", "shape_adaptive"])
@pytest.mark.parametrize("shape", [(516, 528, 528), (4, 16, 16), (300, 64, 272), (1024, 4112, 512), (260, 8, 4), (1028, 1024, 1024), (2308, 512, 2304)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f32_mfma_transposed_a_layout(shape, native):
    """MM_TRANSPOSED_A: A handed over as K x N (kernel/Memory.cpp:205-261, include/Utility.h:31-35).  `kxn_kernel`: the
    geometry pinned (f32_variant 8), which runs the K x N kernel -- slabs of A DMA'd as [k][row] -- whatever the shape.
    `shape_adaptive`: problems that do not fill rounds of that kernel's 256 x 256 tiles are transposed into a workspace
    first and then take the row-major rules (64 x 64 geometry, split-K, stream-K ...).  Either way the bits are those of
    the row-major call under the same knobs."""
    n, k, m = shape
    a, b = _oracle.fill("float", n, k, m)
    at = np.ascontiguousarray(a.T)
    cfg = g.make_config("float", transposed_a=True)
    try:
        if native:
            g.set_tuning("f32_variant", 8)
            g.set_tuning("f32_splitk", 1)
        name = g.kernel_name(cfg, n, k, m)
        name_rm = g.kernel_name(g.make_config("float"), n, k, m)
        c, _ = g.matmul_capi(at, b, transposed_a=True)
        c_rm, _ = g.matmul_capi(a, b)
    finally:
        g.set_tuning("f32_variant", -1)
        g.set_tuning("f32_splitk", -1)
    assert name.startswith("mfma_f32") and name == name_rm, (name, name_rm)
    if native:
        assert name == "mfma_f32_256x256x16_w8_flush4096"
    exact = a.astype(np.float64) @ b.astype(np.float64)
    rel = np.abs(c - exact) / exact
    assert rel.max() < F32_TOL, (np.unravel_index(np.argmax(rel), rel.shape), rel.max())
    assert np.array_equal(c, c_rm)


def test_f32_transposed_a_in_whole_rounds_keeps_the_kxn_kernel():
    """4096^3 is exactly one round of 256 x 256 tiles: no workspace, the K x N kernel itself (and the row-major pick's bits)."""
    import torch
    n = k = m = 4096
    assert g.kernel_name(g.make_config("float", transposed_a=True), n, k, m) == "mfma_f32_256x256x16_w8_flush4096"
    a = torch.empty((n, k), device="cuda").uniform_(1, 10)
    b = torch.empty((k, m), device="cuda").uniform_(1, 10)
    at = a.t().contiguous()
    free0 = torch.cuda.mem_get_info()[0]
    c = g.matmul(at, b, transposed_a=True)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] <= n * m * 4 + (8 << 20)      # C only: no N x K workspace appeared
    assert torch.equal(c, g.matmul(a, b))


def test_f32_mfma_transpose_detecting_inputs():
    # asymmetric operands: A = identity-like selector, B with distinct entries everywhere
    n, k, m = 192, 64, 384
    a = np.zeros((n, k), np.float32)
    a[np.arange(n), np.arange(n) % k] = 1.0
    a[5, 7] = 3.0
    b = (np.arange(k * m, dtype=np.float32).reshape(k, m) % 1021) + 1.0
    c, _ = g.matmul_capi(a, b)
    assert np.array_equal(c, (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32))


@pytest.mark.parametrize("shape", [(513, 528, 528), (300, 1040, 272), (130, 80, 260), (1024, 1024, 1024)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f64_schedules_and_tiles_are_bit_identical(shape):
    """f64_variant 0 / 1 = 256x128 / 128x128 tile with the pinned, software-pipelined schedule; 2 / 3 = the same tiles
    with the compiler-placed schedule of round 1; 4 = the 64x64 geometry of round 3 (32 x 32 per wavefront, for problems
    below a round of the bigger tiles).  Same per-accumulator fma chain in all five, and through the K x N layout of A."""
    n, k, m = shape
    a, b = _oracle.fill("double", n, k, m)
    out = []
    for v in (0, 1, 2, 3, 4):
        g.set_tuning("f64_variant", v)
        c, _ = g.matmul_capi(a, b, "double")
        out.append(c)
        if n % 2 == 0:
            ct, _ = g.matmul_capi(np.ascontiguousarray(a.T), b, "double", transposed_a=True)
            out.append(ct)
    g.set_tuning("f64_variant", -1)
    for c in out[1:]:
        assert np.array_equal(c, out[0])


@pytest.mark.parametrize("shape", [(513, 528, 528), (1, 16, 8), (37, 32, 48), (300, 64, 272), (129, 80, 264), (257, 1040, 520)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f64_mfma_vs_blas(shape):
    n, k, m = shape
    a, b = _oracle.fill("double", n, k, m)
    assert g.kernel_name(g.make_config("double"), n, k, m).startswith("mfma_f64_")
    c, _ = g.matmul_capi(a, b, "double")
    bad, first, worst = _oracle.compare("double", c, a @ b, 1e-12)  # numpy float64 matmul == cblas_dgemm
    assert bad == 0, (first, worst)
    assert np.array_equal(c, _oracle.naive("double", "Multiply", "Add", a, b)) or worst < 1e-13


@pytest.mark.parametrize("shape", [(513, 528, 528), (1, 16, 8), (37, 32, 48), (300, 64, 272), (129, 80, 264), (257, 1040, 520)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f16_mfma_wide_accumulate_contract(shape):
    """half: exact products, f32 accumulation, ONE rounding on store.  Against the
    oracle's wide-accumulate Naive (double accumulation, one rounding) it may differ by at most
    one binary16 ulp; against the exact value by one rounding + the f32 accumulation error."""
    n, k, m = shape
    a, b = _oracle.fill("half", n, k, m)
    assert g.kernel_name(g.make_config("half"), n, k, m).startswith("mfma_f16_")
    c, _ = g.matmul_capi(a, b, "half")
    wide = _oracle.naive("half", "Multiply", "Add", a, b, wide_half=True)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    assert np.all(np.isfinite(c.astype(np.float64)) == np.isfinite(wide.astype(np.float64)))
    fin = np.isfinite(wide.astype(np.float64))
    ulps = np.abs(c.view(np.int16).astype(np.int32) - wide.view(np.int16).astype(np.int32))[fin]
    assert ulps.max() <= 1, ulps.max()
    assert (ulps != 0).mean() < 0.02          # and it is rare
    rel = np.abs(c.astype(np.float64) - exact)[fin] / exact[fin]
    assert rel.max() <= 2.0 ** -11 * 1.01      # half an ulp of binary16 relative to the exact value (+f32 noise)


@pytest.mark.parametrize("shape", [(1024, 1024, 1024), (513, 1040, 528), (65, 80, 264), (1, 16, 8), (130, 8208, 136)], ids=lambda s: "x".join(map(str, s)))
def test_f16_slab64_tiles_are_bit_identical(shape):
    """The slab64 kernel on its three tiles (256 x 256, 128 x 256 and -- round 3, for problems below a round of the others --
    64 x 256 with 32 x 128 per wavefront): the same MFMAs per output element in the same order, identical bits."""
    n, k, m = shape
    rng = np.random.default_rng(n + k)
    a = rng.uniform(-2, 2, size=(n, k)).astype(np.float16)
    b = rng.uniform(-2, 2, size=(k, m)).astype(np.float16)
    out = {}
    try:
        for v in (0, 4, 5):
            g.set_tuning("f16_variant", v)
            out[v], _ = g.matmul_capi(a, b, "half")
    finally:
        g.set_tuning("f16_variant", -1)
    assert np.array_equal(out[0].view(np.uint16), out[4].view(np.uint16)) and np.array_equal(out[0].view(np.uint16), out[5].view(np.uint16))


@pytest.mark.parametrize("variant", [200, 100, 11, 0, 4, 5])
@pytest.mark.parametrize("shape", [(513, 544, 528), (300, 128, 272), (257, 1056, 520), (1024, 4096, 1024), (129, 160, 264),
                                   (770, 2048, 1288), (300, 256, 272), (513, 576, 528), (260, 320, 264)],
                         ids=lambda s: "x".join(map(str, s)))
def test_f16_mfma_every_variant(variant, shape):
    """Every kernel of the half family (200: ping-pong on the 16x16x32 instruction, the default; 100: the same on
    32x32x16; 11: ping-pong with 32-deep A slabs; 0 / 4 / 5: one barrier per 64-deep slab, 256 x 256 / 128 x 256 / 64 x 256 tile)
    against the exact product: same products everywhere, fp32 accumulation, one rounding on store."""
    n, k, m = shape
    rng = np.random.default_rng(n + k)
    a = rng.uniform(-2, 2, size=(n, k)).astype(np.float16)     # mixed signs: cancellation shows layout bugs
    b = rng.uniform(-2, 2, size=(k, m)).astype(np.float16)
    g.set_tuning("f16_variant", variant)
    c, _ = g.matmul_capi(a, b, "half")
    g.set_tuning("f16_variant", -1)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    exact = a64 @ b64
    err = np.abs(c.astype(np.float64) - exact)
    bound = 2.0 ** -11 * np.abs(exact) + 5e-7 * (np.abs(a64) @ np.abs(b64)) + 2.0 ** -25
    assert np.all(err <= bound), (variant, float((err / bound).max()), np.unravel_index(np.argmax(err / bound), err.shape))


@pytest.mark.parametrize("dtype,shape", [("double", (516, 528, 528)), ("double", (2, 8, 2)), ("double", (300, 64, 272)),
                                         ("half", (520, 528, 528)), ("half", (8, 16, 8)), ("half", (304, 64, 272)),
                                         ("uint8_t", (528, 544, 528)), ("int8_t", (16, 32, 16)), ("uint8_t", (304, 4128, 272))],
                         ids=lambda v: str(v))
def test_f64_f16_i8_mfma_transposed_a_layout(dtype, shape):
    """MM_TRANSPOSED_A for the fp64 / fp16 matrix-core paths: same bits as the row-major path."""
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m)
    cfg = g.make_config(dtype, transposed_a=True)
    assert g.kernel_name(cfg, n, k, m).startswith("mfma_")
    c, _ = g.matmul_capi(np.ascontiguousarray(a.T), b, dtype, transposed_a=True)
    c_rm, _ = g.matmul_capi(a, b, dtype)
    assert np.array_equal(c.view(np.uint8), c_rm.view(np.uint8))
    if dtype in ("uint8_t", "int8_t"):
        assert np.array_equal(c, _oracle.naive(dtype, "Multiply", "Add", a, b))
        return
    exact = a.astype(np.float64) @ b.astype(np.float64)
    tol = 1e-12 if dtype == "double" else 2.0 ** -10
    assert np.max(np.abs(c.astype(np.float64) - exact) / exact) <= tol


@pytest.mark.parametrize("dtype", ["int8_t", "uint8_t"])
@pytest.mark.parametrize("shape", [(513, 544, 528), (1, 32, 16), (129, 96, 272), (300, 4128, 272), (257, 64, 1040)],
                         ids=lambda s: "x".join(map(str, s)))
def test_i8_mfma_is_bit_exact_mod_256(dtype, shape):
    """8-bit (Multiply, Add) on the signed-int8 matrix core: u8 = s8 (mod 2^8), sums wrap mod 2^32,
    so the low byte equals the reference's wrap-around Data_t arithmetic for both types."""
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m)
    rng = np.random.default_rng(7)
    # the seeded inputs are 1..10 only; also cover the full 8-bit range incl. the sign bit
    a2 = rng.integers(0, 256, size=a.shape, dtype=np.uint8).view(a.dtype)
    b2 = rng.integers(0, 256, size=b.shape, dtype=np.uint8).view(b.dtype)
    assert g.kernel_name(g.make_config(dtype), n, k, m).startswith("mfma_i8_")
    for aa, bb in ((a, b), (a2, b2)):
        c, _ = g.matmul_capi(aa, bb, dtype)
        assert np.array_equal(c, _oracle.naive(dtype, "Multiply", "Add", aa, bb))


@pytest.mark.parametrize("dtype", ["int8_t", "uint8_t"])
@pytest.mark.parametrize("variant", [-1, 0, 5, 10, 100, 200])
@pytest.mark.parametrize("shape", [(513, 576, 528), (300, 4160, 272), (257, 256, 1040), (1024, 1024, 1024), (1, 320, 16),
                                   (513, 640, 528), (300, 4224, 272), (260, 512, 1040)],
                         ids=lambda s: "x".join(map(str, s)))
def test_i8_mfma_every_schedule_is_bit_exact(dtype, variant, shape):
    """-1 = the default pick (ping-pong with full-line A requests when K % 128 == 0, plain ping-pong
    when K % 64 == 0); 0 = the one-slab-per-barrier kernel, 10 = ping-pong with 64-deep A slabs,
    100 / 200 = full-line A requests on the 32x32x32 / 16x16x64 instruction.  All bit-identical to Naive on full-range bytes."""
    n, k, m = shape
    rng = np.random.default_rng(n * 3 + k)
    a = rng.integers(0, 256, size=(n, k), dtype=np.uint8).view(_oracle.NP_DTYPES[dtype])
    b = rng.integers(0, 256, size=(k, m), dtype=np.uint8).view(_oracle.NP_DTYPES[dtype])
    g.set_tuning("i8_variant", variant)
    c, _ = g.matmul_capi(a, b, dtype)
    g.set_tuning("i8_variant", -1)
    assert np.array_equal(c, _oracle.naive(dtype, "Multiply", "Add", a, b))


@pytest.mark.parametrize("dtype,shape", [("half", (520, 576, 528)), ("half", (264, 4096, 272)), ("half", (1024, 160, 1032)),
                                         ("uint8_t", (528, 576, 528)), ("int8_t", (272, 4160, 272)), ("uint8_t", (1024, 256, 1040))],
                         ids=lambda v: str(v))
def test_f16_i8_transposed_a_on_the_pingpong_schedule(dtype, shape):
    """MM_TRANSPOSED_A on shapes the ping-pong kernels take (K % 32 / % 64): the K x N A is staged and
    gathered like B; results are bit-identical to the row-major ping-pong path AND to the round-1
    K x N kernel (variant 0)."""
    n, k, m = shape
    rng = np.random.default_rng(k + n)
    if dtype == "half":
        a = rng.uniform(-2, 2, size=(n, k)).astype(np.float16)
        b = rng.uniform(-2, 2, size=(k, m)).astype(np.float16)
    else:
        a = rng.integers(0, 256, size=(n, k), dtype=np.uint8).view(_oracle.NP_DTYPES[dtype])
        b = rng.integers(0, 256, size=(k, m), dtype=np.uint8).view(_oracle.NP_DTYPES[dtype])
    at = np.ascontiguousarray(a.T)
    knob = "f16_variant" if dtype == "half" else "i8_variant"
    # the row-major default now runs the 16x16x32 matrix instruction (a different fp32 summation order inside an MFMA);
    # the K x N kernels are on the 32x32x16 form, so for half the bitwise layout check pins the row-major side to it too
    g.set_tuning(knob, 11 if dtype == "half" else -1)
    c_rm, _ = g.matmul_capi(a, b, dtype)
    g.set_tuning(knob, -1)
    c_at, _ = g.matmul_capi(at, b, dtype, transposed_a=True)
    g.set_tuning(knob, 0)
    c_old, _ = g.matmul_capi(at, b, dtype, transposed_a=True)
    g.set_tuning(knob, -1)
    assert np.array_equal(c_at.view(np.uint8), c_rm.view(np.uint8))
    assert np.array_equal(c_at.view(np.uint8), c_old.view(np.uint8))
    if dtype != "half":
        assert np.array_equal(c_at, _oracle.naive(dtype, "Multiply", "Add", a, b))


@pytest.mark.parametrize("dtype,shape,kernel", [("half", (1032, 576, 12296), "mfma_f16_256x256_pingpong_16x16x32"),
                                                ("half", (2048, 256, 16384), "mfma_f16_256x256_pingpong_16x16x32"),
                                                ("uint8_t", (1040, 640, 12304), "mfma_i8_256x256_pingpong_16x16x64"),
                                                ("int8_t", (2048, 512, 16384), "mfma_i8_256x256_pingpong_16x16x64")],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_f16_i8_wide_kxn_problems_are_transposed_first_and_keep_the_row_major_bits(dtype, shape, kernel):
    """Round 4: a K x N A (MM_TRANSPOSED_A, kernel/Memory.cpp:205-261) of a wide problem (M >= 12288) goes through a
    transposition pre-pass into the library's workspace and then runs the row-major default kernel (the K x N kernels
    need twice the LDS gather instructions and run 6-10 % behind).  Ragged N / M tiles included.  Bits: the row-major
    call's, exactly; a pinned variant keeps the K x N kernel, whose bits are the same as well."""
    n, k, m = shape
    rng = np.random.default_rng(n + m)
    if dtype == "half":
        a = rng.uniform(-1, 2, size=(n, k)).astype(np.float16)
        b = rng.uniform(-1, 2, size=(k, m)).astype(np.float16)
    else:
        npdt = np.uint8 if dtype == "uint8_t" else np.int8
        a = rng.integers(np.iinfo(npdt).min, np.iinfo(npdt).max + 1, size=(n, k)).astype(npdt)
        b = rng.integers(np.iinfo(npdt).min, np.iinfo(npdt).max + 1, size=(k, m)).astype(npdt)
    at = np.ascontiguousarray(a.T)
    assert g.kernel_name(g.make_config(dtype, transposed_a=True), n, k, m) == kernel
    c_row, _ = g.matmul_capi(a, b, dtype)
    c_kxn, _ = g.matmul_capi(at, b, dtype, transposed_a=True)
    assert np.array_equal(c_row.view(np.uint8), c_kxn.view(np.uint8))
    knob = "f16_variant" if dtype == "half" else "i8_variant"
    try:
        g.set_tuning(knob, 11 if dtype == "half" else 10)
        assert "KxN" in g.kernel_name(g.make_config(dtype, transposed_a=True), n, k, m)
        c_pinned, _ = g.matmul_capi(at, b, dtype, transposed_a=True)
    finally:
        g.set_tuning(knob, -1)
    assert np.array_equal(c_row.view(np.uint8), c_pinned.view(np.uint8))
    if dtype != "half":
        want = (a[:32].astype(np.int64) @ b.astype(np.int64)).astype(a.dtype)       # wraps like Data_t
        assert np.array_equal(c_row[:32], want)


@pytest.mark.parametrize("dtype", ["float", "double"])
def test_non_finite_operands_propagate_like_ieee_and_leave_other_rows_alone(dtype):
    """inf / NaN in A poison exactly the rows of C they belong to (rows of C are independent, kernel/Compute.cpp:53-60);
    every other row keeps the bits it has without them.  The reference's Naive gives the same by IEEE arithmetic."""
    n, k, m = 300, 528, 272
    a, b = _oracle.fill(dtype, n, k, m)
    clean, _ = g.matmul_capi(a, b, dtype)
    a2 = a.copy()
    a2[7, 3] = np.inf
    a2[130, 500] = np.nan
    a2[299, 0] = -np.inf
    c, _ = g.matmul_capi(a2, b, dtype)
    assert np.all(np.isposinf(c[7])) and np.all(np.isnan(c[130])) and np.all(np.isneginf(c[299]))
    keep = np.ones(n, bool)
    keep[[7, 130, 299]] = False
    assert np.array_equal(c[keep], clean[keep])
    want = _oracle.naive(dtype, "Multiply", "Add", a2, b)
    assert np.array_equal(np.isnan(c), np.isnan(want)) and np.array_equal(np.isinf(c), np.isinf(want))


def test_f16_overflow_behaviour_matches_ieee():
    # K large enough that sums pass 65504: binary16 result is +inf, like a correctly rounded result
    n, k, m = 33, 4096, 64
    a, b = _oracle.fill("half", n, k, m)
    c, _ = g.matmul_capi(a, b, "half")
    assert np.all(np.isinf(c.astype(np.float32)))


ORDERED_CASES = [
    ("float", "Multiply", "Add"), ("float", "Add", "Min"), ("float", "Add", "Max"), ("float", "Min", "Max"),
    ("double", "Multiply", "Add"), ("half", "Multiply", "Add"), ("int", "Multiply", "Add"), ("int", "Add", "Min"),
    ("unsigned", "Multiply", "Add"), ("uint8_t", "Multiply", "Add"), ("int8_t", "Multiply", "Add"),
    ("int16_t", "Multiply", "Add"), ("uint16_t", "Max", "Min"), ("long", "Multiply", "Add"),
    ("unsigned long", "Multiply", "Add"), ("int", "And", "Add"), ("float", "Multiply", "Max"),
]


@pytest.mark.parametrize("dtype,mp,rd", ORDERED_CASES, ids=lambda x: str(x))
@pytest.mark.parametrize("shape", [(65, 48, 80), (129, 33, 70), (257, 264, 272)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("kernel", ["tile_where_it_serves", "anchor_64x64"])
def test_ordered_path_is_bit_identical_to_naive(dtype, mp, rd, shape, kernel):
    """MM_PATH_ORDERED (RunHardware hw_emu) == include/Utility.h:18-42, bit for bit, any shape -- from BOTH kernels that serve
    the contract: the register-tiled one (K % 4 == 0 and M % 4 == 0: the first and third shape) and the 64 x 64 anchor."""
    n, k, m = shape
    g.set_tuning("ordered_variant", 0 if kernel == "anchor_64x64" else -1)
    tiled = kernel != "anchor_64x64" and k % 4 == 0 and m % 4 == 0 and mp != "And" and rd in ("Add", "Min", "Max")
    assert g.kernel_name(g.make_config(dtype, mp, rd, g.PATH_ORDERED), n, k, m) == ("ordered_tile" if tiled else "ordered")
    a, b = _oracle.fill(dtype, n, k, m)
    c, _ = g.matmul_capi(a, b, dtype, mp, rd, path=g.PATH_ORDERED)
    want = _oracle.naive(dtype, mp, rd, a, b)
    assert np.array_equal(c.view(np.uint8), want.view(np.uint8))


def _same_bits_up_to_nan_payload(x, y):
    """Bit equality with all NaNs taken as one value: which of two NaN operands an addition hands on is not specified by
    IEEE 754 and follows the instruction's operand order, which two compilations of the same expression need not share."""
    if x.dtype.kind != "f":
        return np.array_equal(x, y)
    u = {2: np.uint16, 4: np.uint32, 8: np.uint64}[x.dtype.itemsize]
    nx, ny = np.isnan(x), np.isnan(y)
    return np.array_equal(nx, ny) and np.array_equal(x.view(u)[~nx], y.view(u)[~ny])


@pytest.mark.parametrize("dtype", ["half", "float", "double"])
@pytest.mark.parametrize("mp,rd", [("Multiply", "Add"), ("Add", "Min"), ("Add", "Max"), ("Min", "Max"), ("Max", "Min"), ("Min", "Add"),
                                   ("Multiply", "Max"), ("Add", "Add")])
@pytest.mark.parametrize("transposed_a", [False, True], ids=["nxk", "kxn"])
def test_ordered_tile_equals_the_anchor_kernel_on_specials(dtype, mp, rd, transposed_a):
    """The two k-ordered kernels on operands with NaN, +-inf, +-0 and mixed signs sprinkled in: std::min / std::max to the letter
    (a NaN operand, signed-zero ties), IEEE inf arithmetic, overflow of binary16 sums -- the same bits element for element, in
    the DMA-staged and the synchronous staging of the tile kernel (row-major and K x N A)."""
    rng = np.random.default_rng(66)
    n, k, m = 260, 264, 272
    npdt = {"half": np.float16, "float": np.float32, "double": np.float64}[dtype]
    a = rng.uniform(-4, 10, size=(k, n) if transposed_a else (n, k)).astype(npdt)
    b = rng.uniform(-4, 10, size=(k, m)).astype(npdt)
    for arr in (a, b):
        flat = arr.reshape(-1)
        idx = rng.integers(0, flat.size, size=flat.size // 97)
        flat[idx] = rng.choice(np.array([np.nan, np.inf, -np.inf, 0.0, -0.0], dtype=npdt), size=idx.size)
    assert g.kernel_name(g.make_config(dtype, mp, rd, g.PATH_ORDERED, transposed_a), n, k, m) == "ordered_tile"
    c_tile, _ = g.matmul_capi(a, b, dtype, mp, rd, path=g.PATH_ORDERED, transposed_a=transposed_a)
    g.set_tuning("ordered_variant", 0)
    c_anchor, _ = g.matmul_capi(a, b, dtype, mp, rd, path=g.PATH_ORDERED, transposed_a=transposed_a)
    assert _same_bits_up_to_nan_payload(c_tile, c_anchor)
    # the sprinkling reached the outputs and did not drown them (a Min / Max reduction written as std::min / std::max DROPS a
    # NaN operand -- `b < a ? b : a` keeps the accumulator -- so only the Add reductions must show NaNs)
    assert np.isfinite(c_tile).any() and (rd != "Add" or np.isnan(c_tile).any())
    assert not np.isfinite(c_tile).all()


def test_ordered_tile_takes_unaligned_operands_to_the_anchor_kernel():
    """MM_PATH_ORDERED takes any element-aligned pointer (include/mm_gemm.h): an offset view falls back to the predicated
    kernel instead of being refused, with the same bits."""
    import torch
    n, k, m = 128, 64, 128
    a, b = _oracle.fill("float", n, k, m)
    buf = torch.zeros(n * k + 1, dtype=torch.float32, device="cuda")
    buf[1:] = torch.from_numpy(a.reshape(-1)).cuda()
    c = g.matmul(buf[1:].view(n, k), torch.from_numpy(b).cuda(), path=g.PATH_ORDERED).cpu().numpy()
    assert np.array_equal(c, _oracle.naive("float", "Multiply", "Add", a, b))


@pytest.mark.parametrize("shape", [(513, 544, 544), (300, 64, 288), (64, 4096, 64), (129, 30, 70)], ids=lambda s: "x".join(map(str, s)))
def test_half_reference_contract_under_auto_is_the_reference_arithmetic(shape):
    """half_contract = reference (MM_HALF_CONTRACT=reference): half (Multiply, Add) under MM_PATH_AUTO is the reference kernel's
    own arithmetic -- binary16 products, binary16 accumulation, k ascending (kernel/Compute.cpp:129-133) -- bit for bit equal
    to Naive (include/Utility.h:29-37), incl. where the binary16 sum overflows (K = 4096 on [1,10): inf) and on shapes only
    the predicated kernel takes; other configurations are untouched by the knob."""
    n, k, m = shape
    a, b = _oracle.fill("half", n, k, m)
    want = _oracle.naive("half", "Multiply", "Add", a, b)
    wide, _ = g.matmul_capi(a, b, "half")
    g.set_tuning("half_contract", 1)
    assert g.kernel_name(g.make_config("half"), n, k, m) == ("ordered_tile" if k % 4 == 0 and m % 4 == 0 else "ordered")
    c, _ = g.matmul_capi(a, b, "half")
    assert np.array_equal(c.view(np.uint16), want.view(np.uint16))
    if k >= 64 and k < 4096:
        assert not np.array_equal(c.view(np.uint16), wide.view(np.uint16))     # the default contract rounds once: other bits
    assert g.kernel_name(g.make_config("float"), n, 528, 528).startswith("mfma_f32")
    assert g.kernel_name(g.make_config("half", "Add", "Min"), 512, 512, 512) == "valu_tile"


def test_ordered_path_reference_ctest_shape_float():
    n, k, m = 513, 528, 528
    a, b = _oracle.fill("float", n, k, m)
    c, _ = g.matmul_capi(a, b, path=g.PATH_ORDERED)
    assert np.array_equal(c, _oracle.naive("float", "Multiply", "Add", a, b))


def test_ordered_transposed_a_layout():
    n, k, m = 70, 48, 96
    a, b = _oracle.fill("int", n, k, m)
    c, _ = g.matmul_capi(np.ascontiguousarray(a.T), b, "int", path=g.PATH_ORDERED, transposed_a=True)
    assert np.array_equal(c, _oracle.naive("int", "Multiply", "Add", a, b))


AUTO_EXACT_CASES = [("float", "Add", "Min"), ("float", "Add", "Max"), ("int", "Multiply", "Add"),
                    ("uint8_t", "Multiply", "Add"), ("long", "Add", "Min"), ("double", "Min", "Max")]


@pytest.mark.parametrize("dtype,mp,rd", AUTO_EXACT_CASES, ids=lambda x: str(x))
@pytest.mark.parametrize("shape", [(513, 528, 528), (129, 80, 260)], ids=lambda s: "x".join(map(str, s)))
def test_auto_path_exact_semirings(dtype, mp, rd, shape):
    """Integer and min/max semirings are order-independent: the fast path must be bit-exact."""
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m)
    c, _ = g.matmul_capi(a, b, dtype, mp, rd)
    assert np.array_equal(c, _oracle.naive(dtype, mp, rd, a, b))


@pytest.mark.parametrize("dtype,mp,rd", [("float", "Add", "Min"), ("float", "Add", "Max"), ("float", "Min", "Max"),
                                          ("int", "Multiply", "Add"), ("unsigned", "Add", "Min"), ("int", "Max", "Min")],
                         ids=lambda x: str(x))
@pytest.mark.parametrize("shape", [(513, 528, 528), (129, 20, 260), (300, 36, 132), (257, 600, 516), (1, 16, 4), (700, 8, 64),
                                   (130, 1028, 128)], ids=lambda s: "x".join(map(str, s)))
def test_valu_tile_dma_staged_kernel_equals_synchronous_kernel_and_naive(dtype, mp, rd, shape):
    """4-byte types, row-major A: the default VALU kernel stages through LDS-DMA (double-buffered, the
    last partial slab fetched as the last 16 k of the matrix); valu_variant 0 is the synchronous one.
    Same per-output operation sequence -> same bits, and both equal Naive for these semirings."""
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m)
    if dtype == "float":   # signs and magnitudes beyond the generator's [1,10)
        rng = np.random.default_rng(n + k + m)
        a = (a * rng.choice([-1.0, 1.0], size=a.shape)).astype(np.float32)
        b = (b - 5.5).astype(np.float32)
    want = _oracle.naive(dtype, mp, rd, a, b)
    for variant in (-1, 0):
        g.set_tuning("valu_variant", variant)
        c, _ = g.matmul_capi(a, b, dtype, mp, rd)
        assert np.array_equal(c.view(np.uint8), want.view(np.uint8)), (variant, dtype, mp, rd, shape)
    g.set_tuning("valu_variant", -1)


@pytest.mark.parametrize("dtype,mp,rd", [("double", "Add", "Min"), ("double", "Min", "Max"), ("long", "Multiply", "Add"),
                                          ("unsigned long", "Add", "Max"), ("half", "Add", "Min"), ("half", "Max", "Min"),
                                          ("int16_t", "Multiply", "Add"), ("uint16_t", "Add", "Min"), ("int8_t", "Add", "Max"),
                                          ("uint8_t", "Min", "Max"), ("uint8_t", "Add", "Min")], ids=lambda x: str(x))
@pytest.mark.parametrize("shape", [(513, 528, 528), (129, 80, 272), (300, 48, 144), (257, 608, 528), (1, 64, 16), (700, 16, 64),
                                   (130, 1040, 128)], ids=lambda s: "x".join(map(str, s)))
def test_valu_tile_dma_staged_kernel_serves_every_element_size(dtype, mp, rd, shape):
    """Round 3: the DMA-staged VALU kernel for 8-, 2- and 1-byte types too (slab depth 64 / sizeof(T), same byte
    geometry): the reference's ProcessingElement runs any Data_t at full rate (kernel/Compute.cpp:120-139).  Shapes
    include partial last slabs for every slab depth (8 / 32 / 64 k) and ragged N; K and M are multiples of 16 so that
    every type's 16-byte-chunk requirement holds.  Same bits as the synchronous kernel and as Naive."""
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m)
    if dtype in ("double", "half"):
        rng = np.random.default_rng(n + k + m)
        a = (a * rng.choice([-1.0, 1.0], size=a.shape)).astype(a.dtype)
        b = (b - 5.5).astype(b.dtype)
    want = _oracle.naive(dtype, mp, rd, a, b)
    assert g.kernel_name(g.make_config(dtype, mp, rd), n, k, m) == "valu_tile"
    for variant in (-1, 0):
        g.set_tuning("valu_variant", variant)
        c, _ = g.matmul_capi(a, b, dtype, mp, rd)
        assert np.array_equal(c.view(np.uint8), want.view(np.uint8)), (variant, dtype, mp, rd, shape)
    g.set_tuning("valu_variant", -1)


def _golden_cases():
    for path in sorted(glob.glob(os.path.join(GOLD, "ref_*_*x*x*.npz"))):
        stem = os.path.basename(path)[4:-4]
        ta = stem.startswith("transposedA_")          # the reference's -DMM_TRANSPOSED_A build: A is K x N
        dtype, mp, rd, shape = stem[len("transposedA_") if ta else 0:].rsplit("_", 3)
        if (mp, rd) == ("Add", "Min"):
            continue  # reference kernel defect H4 (literal-0 seed); Naive semantics are tested above
        yield pytest.param(path, dtype, mp, rd, tuple(int(x) for x in shape.split("x")), ta, id=stem)


@pytest.mark.parametrize("path,dtype,mp,rd,shape,ta", list(_golden_cases()))
def test_against_reference_kernel_golden_outputs(path, dtype, mp, rd, shape, ta):
    """C committed from the reference's OWN kernel sources (tests/golden/make_golden.py): float, int, half,
    double, uint8_t builds and the MM_TRANSPOSED_A (K x N `a`) builds of float and int."""
    n, k, m = shape
    a, b = _oracle.fill(dtype, n, k, m, transposed_a=ta)
    ref = np.load(path)["c"]
    assert g.kernel_name(g.make_config(dtype, mp, rd, g.PATH_ORDERED, ta), n, k, m) == "ordered_tile"
    c_ord, _ = g.matmul_capi(a, b, dtype, mp, rd, path=g.PATH_ORDERED, transposed_a=ta)
    assert np.array_equal(c_ord, ref)  # same k-ordered unfused chain -> bit-identical, floats too
    g.set_tuning("ordered_variant", 0)  # ... from the 64 x 64 anchor kernel as well
    c_anchor, _ = g.matmul_capi(a, b, dtype, mp, rd, path=g.PATH_ORDERED, transposed_a=ta)
    g.set_tuning("ordered_variant", -1)
    assert np.array_equal(c_anchor, ref)
    if dtype == "half":                 # the reference's half contract under AUTO: the golden bits, not a tolerance
        g.set_tuning("half_contract", 1)
        c_ref_contract, _ = g.matmul_capi(a, b, dtype, mp, rd, transposed_a=ta)
        g.set_tuning("half_contract", -1)
        assert np.array_equal(c_ref_contract.view(np.uint16), ref.view(np.uint16))
    c, _ = g.matmul_capi(a, b, dtype, mp, rd, transposed_a=ta)
    if dtype == "float":
        assert _oracle.compare("float", c, ref, F32_TOL)[0] == 0
    elif dtype == "double":
        assert _oracle.compare("double", c, ref, 1e-13)[0] == 0   # fused vs unfused chain of <= 64 positive terms
    elif dtype == "half":
        # AUTO accumulates in f32 and rounds once (include/mm_gemm.h, half contract); the reference's kernel
        # rounds after every one of the K additions: the two differ by at most K half-ulps of the running sum
        rel = np.abs(c.astype(np.float64) - ref.astype(np.float64)) / np.abs(ref.astype(np.float64))
        assert rel.max() <= k * 2.0 ** -11, rel.max()
    else:
        assert np.array_equal(c, ref)


def test_multi_device_row_split_matches_single():
    n, k, m = 300, 64, 272
    a, b = _oracle.fill("float", n, k, m)
    c1, _ = g.matmul_capi(a, b)
    cm, t = g.matmul_host(a, b, devices=1)
    assert np.array_equal(c1, cm) and t > 0


def test_multi_device_api_contract():
    import torch
    have = torch.cuda.device_count()
    a, b = _oracle.fill("int", 130, 64, 96)
    want = _oracle.naive("int", "Multiply", "Add", a, b)
    for devices in range(1, have + 1):            # every device count that exists on this box
        c, t = g.matmul_host(a, b, "int", devices=devices)
        assert np.array_equal(c, want) and t > 0
    with pytest.raises(g.MMError, match="device_count"):
        g.matmul_host(a, b, "int", devices=have + 1)   # never silently uses fewer devices
    # a K x N A (MM_TRANSPOSED_A) is served too since round 5 (column slabs; G > 1: tests/test_gpu_multi_device.py)
    at = np.ascontiguousarray(a.T)
    ct, _ = g.matmul_host(at, b, "int", devices=1, transposed_a=True)
    assert np.array_equal(ct, want)


def test_reference_entry_point_symbol():
    """extern "C" MatrixMultiplicationKernel(a, b, c, N, K, M) with host pointers (TestSimulation.cpp:66)."""
    n, k, m = 65, 32, 48
    a, b = _oracle.fill("float", n, k, m)
    c = np.zeros((n, m), np.float32)
    g.lib().MatrixMultiplicationKernel(a.ctypes.data, b.ctypes.data, c.ctypes.data, n, k, m)
    assert _oracle.compare("float", c, a @ b, F32_TOL)[0] == 0


def test_reference_entry_point_pipelined_in_row_slabs():
    """Large host-pointer calls are pipelined in row slabs (copy-in / multiply / copy-out overlap);
    the result must be the bits of the one-launch device path (ragged last slab included)."""
    n, k, m = 4100, 4096, 4096
    rng = np.random.default_rng(3)
    a = rng.uniform(1, 10, size=(n, k)).astype(np.float32)
    b = rng.uniform(1, 10, size=(k, m)).astype(np.float32)
    c = np.zeros((n, m), np.float32)
    g.lib().MatrixMultiplicationKernel(a.ctypes.data, b.ctypes.data, c.ctypes.data, n, k, m)
    c_dev, _ = g.matmul_capi(a, b)
    assert np.array_equal(c, c_dev)
    rows = [0, 511, 512, 2047, 4095, 4099]
    exact = a[rows].astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(c[rows] - exact) / exact) < F32_TOL


# ---- BASELINE.json full size: size-independent properties ------------------------------------
def test_f32_full_size_properties():
    import torch
    n = k = m = 16384
    dev = torch.device("cuda:0")
    L = g.lib()
    a = torch.empty((n, k), dtype=torch.float32, device=dev)
    b = torch.empty((k, m), dtype=torch.float32, device=dev)
    assert L.mm_fill_device(0, 0, a.data_ptr(), a.numel(), 11) == 0
    assert L.mm_fill_device(0, 0, b.data_ptr(), b.numel(), 12) == 0
    c = g.matmul(a, b)
    torch.cuda.synchronize()
    # (1) sampled rows against an fp64 host reference, reference comparison rule, 1e-5
    rows = [0, 1, 127, 128, 4095, 8191, 8192, 12345, 16383]
    a_rows = a[rows].double().cpu().numpy()
    exact = a_rows @ b.double().cpu().numpy()
    got = c[rows].cpu().numpy()
    assert np.max(np.abs(got - exact) / exact) < F32_TOL
    # (2) linearity in A: (2A) B == 2 (A B) exactly (power-of-two scaling commutes with rounding)
    c2 = g.matmul(a * 2.0, b)
    assert torch.equal(c2, c * 2.0)
    # (3) row-permutation equivariance: rows of C depend only on the matching rows of A
    perm = torch.randperm(n, device=dev)
    cp = g.matmul(a[perm].contiguous(), b)
    assert torch.equal(cp, c[perm])
    # (4) determinism
    assert torch.equal(g.matmul(a, b), c)


def test_f32_baseline_c5a_shape_on_one_gpu_large_offsets():
    """BASELINE config 5a (65536 x 16384 x 16384) on ONE device: A and C are 4 GiB each, so byte
    offsets exceed 32 bits.  Sampled rows (first, last, around the 4 GiB boundary) against fp64, and
    the N-split property: a row slab computed alone equals the same rows of the full product."""
    import torch
    n, k, m = 65536, 16384, 16384
    dev = torch.device("cuda:0")
    L = g.lib()
    a = torch.empty((n, k), dtype=torch.float32, device=dev)
    b = torch.empty((k, m), dtype=torch.float32, device=dev)
    assert L.mm_fill_device(0, 0, a.data_ptr(), a.numel(), 21) == 0
    assert L.mm_fill_device(0, 0, b.data_ptr(), b.numel(), 22) == 0
    c = g.matmul(a, b)
    torch.cuda.synchronize()
    rows = [0, 1, 32767, 32768, 49151, 65534, 65535]
    b64 = b.double()
    exact = (a[rows].double() @ b64).cpu().numpy()
    got = c[rows].cpu().numpy()
    assert np.max(np.abs(got - exact) / exact) < F32_TOL
    # slab [3/8, 4/8) of an 8-way row split, computed on its own: identical bits
    from gemm_hls_amd.partition import row_slab_for
    row0, cnt = row_slab_for(g.make_config("float"), n, k, m, 8, 3)
    slab = g.matmul(a[row0:row0 + cnt].contiguous(), b)
    assert torch.equal(slab, c[row0:row0 + cnt])


def test_f32_and_f64_rows_longer_than_the_scalar_base_dma_reach():
    """The default kernels address a tile's rows with 32-bit byte offsets from a 64-bit base (256 rows x K x 4 B must
    stay below 4 GiB); a matrix with longer rows must fall back to the vector-address kernels, not wrap around."""
    rng = np.random.default_rng(3)
    k = (1 << 22) + 16                                   # 256 * k * 4 B just above 4 GiB
    a = rng.uniform(1, 2, size=(8, k)).astype(np.float32)
    b = rng.uniform(1, 2, size=(k, 4)).astype(np.float32)
    c, _ = g.matmul_capi(a, b)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(c - exact) / exact) < 1e-5
    kd = (1 << 21) + 16                                  # 256 * k * 8 B just above 4 GiB
    ad, bd = a[:, :kd].astype(np.float64), b[:kd].astype(np.float64)
    cd, _ = g.matmul_capi(ad, bd, "double")
    assert np.max(np.abs(cd - ad @ bd) / (ad @ bd)) < 1e-12


def test_f16_and_i8_rows_longer_than_the_dma_reach():
    """Same reach rule for the half / int8 ping-pong kernels (256 rows x K x element size < 4 GiB): longer rows run the
    round-1 kernels -- same contract, same bits for int8."""
    rng = np.random.default_rng(4)
    k = (1 << 23) + 64                                   # half: 256 * k * 2 B just above 4 GiB
    a = rng.uniform(0.5, 1.0, size=(8, k)).astype(np.float16)
    b = (rng.uniform(0.5, 1.0, size=(k, 8)) * 2.0 ** -12).astype(np.float16)
    c, _ = g.matmul_capi(a, b, "half")
    exact = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(c.astype(np.float64) - exact) / exact) < 2.0 ** -10
    k8 = (1 << 24) + 128                                 # int8: 256 * k B just above 4 GiB
    a8 = rng.integers(0, 256, size=(4, k8), dtype=np.uint8)
    b8 = rng.integers(0, 256, size=(k8, 16), dtype=np.uint8)
    c8, _ = g.matmul_capi(a8, b8, "uint8_t")
    acc = np.zeros((4, 16), np.uint32)
    for i in range(0, k8, 1 << 20):                      # chunked exact reference: sums wrap mod 2^32, then mod 2^8
        acc += (a8[:, i:i + (1 << 20)].astype(np.uint32) @ b8[i:i + (1 << 20)].astype(np.uint32))
    assert np.array_equal(c8, acc.astype(np.uint8))
