"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/mm_gemm.h declares, the enum values in the header / Python binding / oracle agree, and
compute calls fail loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

import _oracle
import gemm_hls_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "mm_gemm.h")).read()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    L = g.lib()
    declared = set(re.findall(r"^\s*(?:int|void|size_t|const char \*)\s*\*?\s*(mm_\w+|MatrixMultiplicationKernel)\s*\(",
                              HEADER, flags=re.M))
    assert declared == set(g.EXPORTS), declared ^ set(g.EXPORTS)
    for sym in declared:
        assert getattr(L, sym) is not None


def test_enum_values_agree_header_binding_oracle():
    def enum_vals(prefix):
        return {k: int(v) for k, v in re.findall(rf"({prefix}\w+)\s*=\s*(\d+)", HEADER)}
    d = enum_vals("MM_DTYPE_")
    want = {"float": "F32", "double": "F64", "half": "F16", "int8_t": "I8", "uint8_t": "U8", "int16_t": "I16",
            "uint16_t": "U16", "int": "I32", "unsigned": "U32", "long": "I64", "unsigned long": "U64"}
    for name, tag in want.items():
        assert d["MM_DTYPE_" + tag] == g.DTYPES[name] == _oracle.DTYPES[name]
    o = enum_vals("MM_OP_")
    for name in ("Add", "Multiply", "And", "Min", "Max"):
        assert o["MM_OP_" + name.upper()] == g.OPS[name] == _oracle.OPS[name]
    for name, code in g.DTYPES.items():
        assert g.lib().mm_dtype_size(code) == _oracle.lib().mm_oracle_dtype_size(code)


def test_kernel_name_dispatch_table():
    assert g.kernel_name(g.make_config("float"), 16384, 16384, 16384).startswith("mfma_f32")
    # the k-ordered contract: the register-tiled kernel where it serves (K % 4 == 0, M % 4 == 0, reductions Add / Min / Max and a
    # map other than And), the fully predicated 64 x 64 kernel for the rest -- and always under ordered_variant = 0
    assert g.kernel_name(g.make_config("float", path=g.PATH_ORDERED), 64, 64, 64) == "ordered_tile"
    assert g.kernel_name(g.make_config("float", path=g.PATH_ORDERED), 64, 62, 64) == "ordered"
    assert g.kernel_name(g.make_config("int", "And", "Add", path=g.PATH_ORDERED), 64, 64, 64) == "ordered"
    assert g.kernel_name(g.make_config("float", "Multiply", "Multiply", path=g.PATH_ORDERED), 64, 64, 64) == "ordered"
    assert g.kernel_name(g.make_config("float", path=g.PATH_ORDERED, transposed_a=True), 62, 64, 64) == "ordered"   # K x N A: N % 4
    try:
        g.set_tuning("ordered_variant", 0)
        assert g.kernel_name(g.make_config("float", path=g.PATH_ORDERED), 64, 64, 64) == "ordered"
    finally:
        g.set_tuning("ordered_variant", -1)
    assert not g.kernel_name(g.make_config("float", "Add", "Min"), 64, 64, 64).startswith("mfma")
    assert not g.kernel_name(g.make_config("float"), 64, 62, 64).startswith("mfma")  # K % 8 != 0
    assert g.kernel_name(g.make_config("float", "Add", "Min"), 64, 64, 64) == "valu_tile"
    assert g.kernel_name(g.make_config("float", "Add", "Min"), 64, 63, 64) == "ordered"      # K % 4 != 0
    assert g.kernel_name(g.make_config("int", "And", "Add"), 64, 64, 64) == "ordered"        # And map
    # every family names the kernel its launcher resolves to (one resolver per family, shared with the launch path)
    assert g.kernel_name(g.make_config("uint8_t"), 512, 512, 512) == "mfma_i8_64x256x128_slab128"           # small problem: 64-row tile
    assert g.kernel_name(g.make_config("uint8_t"), 4096, 512, 4096) == "mfma_i8_256x256_pingpong_16x16x64"
    assert g.kernel_name(g.make_config("uint8_t"), 4096, 576, 4096) == "mfma_i8_256x256_pingpong_k64"
    assert g.kernel_name(g.make_config("uint8_t"), 4096, 96, 4096) == "mfma_i8_256x256x128_slab128"
    assert g.kernel_name(g.make_config("double"), 512, 512, 512) == "mfma_f64_64x64x16_w4x4"          # below a round of 128 x 128 tiles
    assert g.kernel_name(g.make_config("double"), 2048, 512, 2048).startswith("mfma_f64_128x128x16")
    assert g.kernel_name(g.make_config("double"), 16384, 16384, 16384) == "mfma_f64_256x128x16_w8"
    assert g.kernel_name(g.make_config("half"), 512, 512, 512) == "mfma_f16_64x256x64_slab64"      # small problem: 64-row tile
    assert g.kernel_name(g.make_config("half"), 2560, 512, 2560) == "mfma_f16_128x256x64_slab64"
    assert g.kernel_name(g.make_config("half"), 32768, 32768, 32768) == "mfma_f16_256x256_pingpong_16x16x32"
    assert g.kernel_name(g.make_config("half"), 4096, 4128, 4096) == "mfma_f16_256x256_pingpong_k32"   # K % 64 != 0
    assert g.kernel_name(g.make_config("half"), 4096, 4112, 4096) == "mfma_f16_256x256x64_slab64"      # K % 32 != 0
    assert g.kernel_name(g.make_config("half", transposed_a=True), 4096, 4096, 4096) == "mfma_f16_256x256_pingpong_k32_KxN"
    assert g.kernel_name(g.make_config("float", transposed_a=True), 512, 512, 512).startswith("mfma_f32")


@pytest.mark.skipif(_has_gpu(), reason="only meaningful without a GPU")
def test_compute_fails_loudly_without_gpu():
    import numpy as np
    with pytest.raises(g.MMError, match="no CPU fallback|gfx950"):
        g.device_count()
    a = np.ones((4, 16), np.float32)
    with pytest.raises(g.MMError):
        g.matmul_capi(a, np.ones((16, 16), np.float32))
    with pytest.raises(g.MMError):
        g.matmul_host(a, np.ones((16, 16), np.float32))


def test_bad_arguments_are_rejected_before_touching_a_device():
    cfg = g.Config(99, 0, 0, 0, 0)
    assert g.lib().mm_config_supported(ctypes.byref(cfg)) == 0
    assert g.lib().mm_set_default_config(ctypes.byref(cfg)) != 0
    assert b"invalid" in g.lib().mm_last_error()


def test_half_auto_path_never_falls_back_to_half_accumulation():
    """K % 16 != 0 (or M % 8 != 0) must not silently change half (x,+) semantics."""
    assert g.kernel_name(g.make_config("half"), 512, 4096, 512).startswith("mfma_f16_")
    assert g.kernel_name(g.make_config("half"), 512, 4104, 512) == "ordered_wide_f16"
    assert g.kernel_name(g.make_config("half"), 512, 4096, 516) == "ordered_wide_f16"
    assert g.kernel_name(g.make_config("half", path=g.PATH_ORDERED), 512, 4104, 512) == "ordered_tile"
    assert g.kernel_name(g.make_config("half", path=g.PATH_ORDERED), 512, 4102, 512) == "ordered"
    assert g.kernel_name(g.make_config("half", "Add", "Min"), 512, 4104, 512) == "valu_tile"


def test_half_contract_knob_routes_only_half_multiply_add_of_the_auto_path():
    """half_contract = 1 (MM_HALF_CONTRACT=reference): half (Multiply, Add) under MM_PATH_AUTO takes the k-ordered kernels
    (the reference's binary16-accumulating arithmetic); nothing else moves, and an explicit MM_PATH_SPLIT stays refused."""
    assert g.get_tuning("half_contract") == -1
    try:
        g.set_tuning("half_contract", 1)
        assert g.kernel_name(g.make_config("half"), 32768, 32768, 32768) == "ordered_tile"
        assert g.kernel_name(g.make_config("half"), 513, 542, 544) == "ordered"
        assert g.kernel_name(g.make_config("half", transposed_a=True), 4096, 4096, 4096) == "ordered_tile"
        assert g.kernel_name(g.make_config("half", "Add", "Min"), 512, 512, 512) == "valu_tile"
        assert g.kernel_name(g.make_config("float"), 16384, 16384, 16384).startswith("mfma_f32")
        assert g.kernel_name(g.make_config("uint8_t"), 4096, 512, 4096).startswith("mfma_i8")
        assert g.kernel_name(g.make_config("half", path=g.PATH_SPLIT), 512, 512, 512) == "unsupported"
        info = g.kernel_info(g.make_config("half"), 32768, 32768, 32768)
        assert (info.tile_n, info.tile_m, info.tile_k, info.inst_m) == (128, 128, 32, 64) and info.ops_per_clk_per_cu == 128.0
        g.set_tuning("half_contract", 0)
        assert g.kernel_name(g.make_config("half"), 32768, 32768, 32768).startswith("mfma_f16")
    finally:
        g.set_tuning("half_contract", -1)


@pytest.mark.parametrize("value,want", [("reference", 1), ("wide", 0), ("1", 1), ("0", 0)])
def test_half_contract_knob_is_read_from_the_environment_once(value, want):
    """MM_HALF_CONTRACT is the one knob with words for values; like the others it is read when the library initialises."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import gemm_hls_amd as g; "
            "print(g.get_tuning('half_contract'), g.kernel_name(g.make_config('half'), 4096, 4096, 4096))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MM_HALF_CONTRACT=value), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got, name = r.stdout.split()
    assert int(got) == want and (name == "ordered_tile") == (want == 1)


def test_tuning_knobs_are_library_state_not_environment(monkeypatch):
    assert g.get_tuning("f32_variant") == -1
    monkeypatch.setenv("MM_F32_VARIANT", "3")            # too late: the environment is read once
    assert g.get_tuning("f32_variant") == -1
    try:
        g.set_tuning("f32_variant", 3)
        assert g.get_tuning("f32_variant") == 3
        assert g.kernel_name(g.make_config("float"), 4096, 4096, 4096) == "mfma_f32_256x256x16_w8"
    finally:
        g.set_tuning("f32_variant", -1)
    assert g.kernel_name(g.make_config("float"), 16384, 16384, 16384) == "mfma_f32_256x256x16_w8_flush4096"
    with pytest.raises(g.MMError, match="unknown tuning knob"):
        g.set_tuning("no_such_knob", 1)


def test_tuning_environment_is_read_at_first_use():
    import subprocess
    import sys
    code = ("import gemm_hls_amd as g; print(g.get_tuning('f32_variant'), g.get_tuning('band_rows'), "
            "g.kernel_name(g.make_config('float'), 4096, 4096, 4096))")
    env = dict(os.environ, MM_F32_VARIANT="35", MM_BAND_ROWS="8", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["35", "8", "mfma_f32_128x128x32_w4x2_flush4096"]


def test_f32_variant_ids_of_the_product():
    """The library holds six fp32 geometries: the four MM_PATH_AUTO picks from and two cross-check ones; every other
    id (the lab's) names nothing here.  tests/test_gpu_parity.py iterates over exactly this list."""
    valid = []
    try:
        for v in range(0, 80):
            g.set_tuning("f32_variant", v)
            if g.kernel_name(g.make_config("float"), 4096, 4096, 4096) != "unsupported":
                valid.append(v)
    finally:
        g.set_tuning("f32_variant", -1)
    assert valid == [0, 3, 8, 33, 35, 64]


def test_kernel_info_binding():
    info = g.kernel_info(g.make_config("float"), 16384, 16384, 16384)
    assert (info.tile_n, info.tile_m, info.tile_k, info.wavefronts) == (256, 256, 16, 8)   # whole rounds of 256 x 256 tiles (round 4)
    mid = g.kernel_info(g.make_config("float"), 4096, 4096, 4096)
    assert (mid.tile_n, mid.tile_m, mid.tile_k, mid.wavefronts) == (128, 256, 16, 4)        # one round: two such workgroups per CU
    small = [g.kernel_info(g.make_config(t), 512, 512, 512) for t in ("float", "double", "half", "uint8_t")]   # the small-problem geometries (round 3)
    assert [(i.tile_n, i.tile_m, i.tile_k, i.wavefronts) for i in small] == [(64, 64, 32, 4), (64, 64, 16, 4), (64, 256, 64, 4), (64, 256, 128, 4)]
    assert info.compute_units == 256 and info.max_clock_mhz == 2400.0


DISPATCH_TABLE = [
    # the shape-adaptive rules as measured in round 3 (profiles/r03w_*, r03x_*, r03y_*, r03z_f32_sizes_back_to_back_vs_library.txt):
    # a change here is a change of a measured decision and wants a new measurement, not just a new expectation
    ("float", (512, 512, 512), False, "mfma_f32_64x64x32_w4x2_flush4096"),          # below a round of 128 x 128 tiles: the 64 x 64 geometry
    ("float", (1024, 1024, 1024), False, "mfma_f32_64x64x32_w4x2_flush4096"),
    ("float", (1280, 1280, 1280), False, "mfma_f32_64x64x32_w4x2_flush4096"),
    ("float", (1024, 512, 1024), False, "mfma_f32_64x64x32_w4x2_flush4096"),
    ("float", (512, 4096, 512), False, "mfma_f32_64x64x32_w4x2_splitk4"),           # few tiles, long K: it splits K
    ("float", (256, 8192, 256), False, "mfma_f32_64x64x32_w4x2_splitk8"),
    ("float", (1536, 1536, 1536), False, "mfma_f32_128x128x32_w4x2_splitk3"),       # 576 tiles of 64 x 64 would be a second round
    ("float", (1792, 1792, 1792), False, "mfma_f32_128x128x32_w4x2_flush4096"),
    ("float", (2048, 2048, 2048), False, "mfma_f32_128x128x32_w4x2_flush4096"),     # one workgroup per CU: whole tiles
    ("float", (2048, 256, 2048), False, "mfma_f32_128x128x32_w4x2_flush4096"),
    ("float", (2304, 2304, 2304), False, "mfma_f32_128x128x32_w4x2_streamk"),       # between whole rounds: stream-K in teams
    ("float", (2560, 2560, 2560), False, "mfma_f32_128x128x32_w4x2_streamk"),
    ("float", (3072, 3072, 3072), False, "mfma_f32_128x128x32_w4x2_streamk"),
    ("float", (5120, 5120, 5120), False, "mfma_f32_128x128x32_w4x2_streamk"),
    ("float", (7680, 7680, 7680), False, "mfma_f32_128x128x32_w4x2_streamk"),       # 60 x 60 tiles: 4 x 4 teams divide the grid
    ("float", (6912, 6912, 6912), False, "mfma_f32_128x256x16_w4x2_flush4096"),     # 54 x 54: only 2 x 2 teams, too big for them
    ("float", (2816, 2816, 2816), False, "mfma_f32_128x256x16_w4x2_flush4096"),     # whole tiles fit
    ("float", (4096, 4096, 4096), False, "mfma_f32_128x256x16_w4x2_flush4096"),
    ("float", (6144, 6144, 6144), False, "mfma_f32_128x128x32_w4x2_flush4096"),
    # round 4: >= 4 whole rounds of 256 x 256 tiles take that geometry -- within 1 % of the 128 x 256 one in steady state, 4.6 %
    # less board power, half the fabric traffic (profiles/r04b_f32_energy_33_vs_8.txt, r04c_f32_default_ab_steady_state.txt)
    ("float", (8192, 8192, 8192), False, "mfma_f32_256x256x16_w8_flush4096"),
    ("float", (16384, 16384, 16384), False, "mfma_f32_256x256x16_w8_flush4096"),    # BASELINE configs[1]
    ("float", (65536, 16384, 16384), False, "mfma_f32_256x256x16_w8_flush4096"),    # BASELINE configs[4], unsplit
    ("float", (8192, 16384, 16384), False, "mfma_f32_256x256x16_w8_flush4096"),     # ... and its row slab on 8 GPUs
    ("float", (12288, 4096, 8192), False, "mfma_f32_256x256x16_w8_flush4096"),      # 48 x 32 tiles = 6 rounds
    ("float", (1024, 1024, 1024), True, "mfma_f32_64x64x32_w4x2_flush4096"),        # K x N A outside whole rounds: transposed first,
    ("float", (2048, 2048, 2048), True, "mfma_f32_128x128x32_w4x2_flush4096"),      #   then the row-major rules
    ("float", (6144, 6144, 6144), True, "mfma_f32_128x128x32_w4x2_flush4096"),
    ("float", (4096, 4096, 4096), True, "mfma_f32_256x256x16_w8_flush4096"),        # whole rounds: the K x N kernel itself
    ("float", (16384, 16384, 16384), True, "mfma_f32_256x256x16_w8_flush4096"),
    # round 4: half / int8 with a K x N A and M >= 6144: transposition pre-pass, then the row-major default
    # (profiles/r04b_kxn_prepass.txt, r04c_kxn_prepass_forced_small_m.txt); below that the K x N ping-pong kernels
    ("half", (16384, 16384, 16384), True, "mfma_f16_256x256_pingpong_16x16x32"),
    ("half", (32768, 32768, 32768), True, "mfma_f16_256x256_pingpong_16x16x32"),
    ("half", (6144, 6144, 6144), True, "mfma_f16_256x256_pingpong_16x16x32"),
    ("half", (4096, 4096, 4096), True, "mfma_f16_256x256_pingpong_k32_KxN"),
    ("uint8_t", (32768, 32768, 32768), True, "mfma_i8_256x256_pingpong_16x16x64"),
    ("uint8_t", (8192, 8192, 8192), True, "mfma_i8_256x256_pingpong_16x16x64"),
    ("uint8_t", (4096, 4096, 4096), True, "mfma_i8_256x256_pingpong_k64_KxN"),
    ("double", (1024, 1024, 1024), False, "mfma_f64_64x64x16_w4x4"),
    ("double", (3072, 3072, 3072), False, "mfma_f64_64x64x16_w4x4"),
    ("double", (1792, 1792, 1792), False, "mfma_f64_128x128x16_w4x2"),
    ("double", (4096, 4096, 4096), False, "mfma_f64_256x128x16_w8"),
    ("double", (16384, 16384, 16384), False, "mfma_f64_256x128x16_w8"),             # BASELINE configs[3]
    ("half", (1024, 1024, 1024), False, "mfma_f16_64x256x64_slab64"),
    ("half", (2560, 2560, 2560), False, "mfma_f16_128x256x64_slab64"),
    ("half", (4096, 4096, 4096), False, "mfma_f16_256x256_pingpong_16x16x32"),
    ("half", (32768, 32768, 32768), False, "mfma_f16_256x256_pingpong_16x16x32"),   # BASELINE configs[4]
    ("uint8_t", (2048, 2048, 2048), False, "mfma_i8_64x256x128_slab128"),
    ("uint8_t", (2560, 2560, 2560), False, "mfma_i8_256x256_pingpong_16x16x64"),
    ("uint8_t", (32768, 32768, 32768), False, "mfma_i8_256x256_pingpong_16x16x64"),
]


@pytest.mark.parametrize("dtype,shape,transposed_a,expect", DISPATCH_TABLE,
                         ids=[f"{d}-{'x'.join(map(str, s))}{'-KxN' if t else ''}" for d, s, t, _ in DISPATCH_TABLE])
def test_shape_adaptive_dispatch_table(dtype, shape, transposed_a, expect):
    """mm_kernel_name answers from the resolver the launcher uses (no GPU needed): the measured decisions of the shape-adaptive
    rules, pinned."""
    assert g.kernel_name(g.make_config(dtype, transposed_a=transposed_a), *shape) == expect
