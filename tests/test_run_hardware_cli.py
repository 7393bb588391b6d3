"""The C++ host runner keeps the reference's command line (host/RunHardware.cpp:18-91): argument
count, size checks with the reference's messages, mode/verify keywords, exit codes."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "RunHardware.exe")
USAGE = "Usage: ./RunHardware.exe N K M [<mode [hw/hw_emu]>] [<verify [on/off]>]"


def run(exe, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, env=e, timeout=1200)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(EXE):
        import sys
        sys.path.insert(0, ROOT)
        from gemm_hls_amd import build
        build.build(verbose=False)


@pytest.mark.parametrize("args", [(), (64,), (64, 64), (64, 64, 64, "hw", "on", "extra"), (64, 64, 64, "fpga"),
                                  (64, 64, 64, "hw", "maybe")])
def test_usage_errors(args):
    r = run(EXE, *args)
    assert r.returncode == 1 and USAGE in r.stderr


def test_divisibility_messages():
    r = run(EXE, 64, 17, 64)
    assert r.returncode == 1
    assert "K (17) must be divisable by the memory width in K (16)." in r.stderr
    r = run(EXE, 64, 16, 24)
    assert r.returncode == 1
    assert "M (24) must be divisable by the memory width in M (16)." in r.stderr
    # double: 64-byte bus = 8 elements
    r = run(os.path.join(ROOT, "bin", "RunHardware_double_Multiply_Add.exe"), 64, 12, 64)
    assert "memory width in K (8)" in r.stderr


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful without a GPU")
def test_no_device_is_an_execution_failure_not_a_fallback():
    r = run(EXE, 64, 16, 64)
    assert r.returncode == 1
    assert r.stderr.startswith('Execution failed with error: "') and "no CPU fallback" in r.stderr
    assert "Successfully verified." not in r.stdout


PERF_LINE = re.compile(r"Kernel executed in ([\d\.e\-+]+) seconds, corresponding to a performance of ([\d\.e\-+]+) GOp/s\.")
# the regex the reference's benchmark driver applies to this output (scripts/build_manager.py:601-602)
BUILD_MANAGER_RE = re.compile(r"([\d\.]+) seconds[^\d]+([\d\.]+) GOp/s")


@pytest.mark.gpu
@pytest.mark.parametrize("exe,shape", [
    ("RunHardware.exe", (513, 528, 528)),                    # the reference's CTest shape
    ("RunHardware_double_Multiply_Add.exe", (257, 264, 264)),
    ("RunHardware_half_Multiply_Add.exe", (129, 160, 288)),
    ("RunHardware_float_Add_Min.exe", (257, 272, 272)),
    ("RunHardware_int_Multiply_Add.exe", (257, 272, 272)),
    ("RunHardware_uint8_t_Multiply_Add.exe", (129, 192, 192)),
])
@pytest.mark.parametrize("mode", ["hw", "hw_emu"])
def test_run_hardware_verifies_on_gpu(exe, shape, mode):
    r = run(os.path.join(ROOT, "bin", exe), *shape, mode, "on")
    assert r.returncode == 0, r.stdout + r.stderr
    assert PERF_LINE.search(r.stdout) and "Successfully verified." in r.stdout
    if "half" in exe:
        # hw: f32-accumulate contract vs the wide reference; hw_emu: the reference's own half-accumulating
        # Naive, compared EXACTLY as the reference compares half (test/TestSimulation.cpp:81-85)
        want = "half-accumulating reference" if mode == "hw_emu" else "wide-accumulate half reference"
        assert want in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("exe,shape,kernel", [
    ("RunHardware.exe", (16384, 16384, 16384), "mfma_f32_256x256x16_w8_flush4096"),                      # BASELINE configs[1]
    ("RunHardware_double_Multiply_Add.exe", (16384, 16384, 16384), "mfma_f64_256x128x16_w8"),           # configs[3]
    ("RunHardware_float_Add_Min.exe", (8192, 8192, 8192), "valu_tile"),                                 # configs[4], min-plus
])
def test_run_hardware_full_baseline_size_verifies_every_element(exe, shape, kernel):
    """The reference's whole flow at BASELINE's sizes (host/RunHardware.cpp:199-227): seeded host generation, copies, the
    shipped default kernel, the host reference on all cores (BLAS for (x,+), threaded Naive for min-plus), and the
    element-by-element comparison of the FULL matrix at the reference's 1e-3 rule tightened to 1e-5 -- `Successfully verified.`"""
    r = run(os.path.join(ROOT, "bin", exe), *shape, "hw", "on")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"Executing kernel ({kernel})" in r.stdout, r.stdout
    assert PERF_LINE.search(r.stdout) and "Successfully verified." in r.stdout
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):     # kept as evidence next to the GPU test log
        with open(os.path.join(out_dir, f"runhardware_{exe[:-4]}_{'x'.join(map(str, shape))}_hw_on.log"), "w") as f:
            f.write(r.stdout + r.stderr)


@pytest.mark.gpu
def test_half_hw_emu_is_exact_even_where_half_accumulation_overflows():
    """K = 4096 on [1,10) data: the reference's half semantics give inf everywhere; hw_emu must
    reproduce exactly that and the exact comparison must accept inf == inf."""
    r = run(os.path.join(ROOT, "bin", "RunHardware_half_Multiply_Add.exe"), 65, 4096, 64, "hw_emu", "on")
    assert r.returncode == 0 and "Successfully verified." in r.stdout, r.stdout + r.stderr


# ---- MM_TRANSPOSED_A builds (CMakeLists.txt:30,100-103; include/Utility.h:31-35) ------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("exe,shape", [("RunHardware_transposedA_float.exe", (516, 528, 528)),
                                       ("RunHardware_transposedA_float.exe", (300, 64, 272)),   # N != K: the lda the reference gets wrong
                                       ("RunHardware_transposedA_int.exe", (260, 272, 272))])
@pytest.mark.parametrize("mode", ["hw", "hw_emu"])
def test_transposed_a_build_verifies_on_gpu(exe, shape, mode):
    r = run(os.path.join(ROOT, "bin", exe), *shape, mode, "on")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Successfully verified." in r.stdout


def test_transposed_a_build_keeps_the_cli():
    exe = os.path.join(ROOT, "bin", "RunHardware_transposedA_float.exe")
    assert run(exe).returncode == 1 and USAGE in run(exe).stderr
    r = run(exe, 64, 17, 64)
    assert r.returncode == 1 and "K (17) must be divisable by the memory width in K (16)." in r.stderr


@pytest.mark.gpu
def test_test_simulation_transposed_a_and_half_on_gpu():
    r = run(os.path.join(ROOT, "bin", "TestSimulation_transposedA.exe"), 516, 528, 528)
    assert r.returncode == 0 and "successfully verified" in r.stdout, r.stdout + r.stderr
    r = run(os.path.join(ROOT, "bin", "TestSimulation_half.exe"), 129, 160, 288)
    assert r.returncode == 0 and "successfully verified" in r.stdout, r.stdout + r.stderr


# ---- MM_POWER_METER build (CMakeLists.txt:13,212-214; host/RunHardware.cpp:156-172,182-185) ------------------
# the second regex of the reference's benchmark parser (scripts/build_manager.py:603-604)
POWER_RE = re.compile(r"([\d\.]+) W")


@pytest.mark.gpu
def test_power_meter_build_samples_while_the_kernel_runs():
    r = run(os.path.join(ROOT, "bin", "RunHardware_power.exe"), 8192, 8192, 8192, "hw", "off",
            env={"MM_POWER_WINDOW_MS": "400"})
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("Measured an average power of")]
    assert line and PERF_LINE.search(r.stdout)
    m = POWER_RE.search(line[0])
    samples = int(re.search(r"\((\d+) samples", line[0]).group(1))
    assert m and samples >= 20, line
    watts = float(m.group(1))
    # an MI355X multiplying flat out draws several hundred watts; idle is ~150 W.  A sensor that is
    # missing reports 0 samples and fails above; this bound catches a sample taken on an idle chip
    assert 250.0 < watts < 1600.0, line


@pytest.mark.gpu
def test_run_hardware_timing_only_and_report_format():
    r = run(EXE, 4096, 4096, 4096, "hw", "off")
    assert r.returncode == 0, r.stdout + r.stderr
    m = PERF_LINE.search(r.stdout)
    assert m and "Verifying" not in r.stdout
    secs, gops = float(m.group(1)), float(m.group(2))
    assert abs(gops - 1e-9 * 2 * 4096.0 ** 3 / secs) / gops < 1e-3
    assert gops > 50e3  # an MI355X does far better than 50 TFLOP/s on this
    assert BUILD_MANAGER_RE.search(r.stdout)


@pytest.mark.gpu
def test_run_hardware_multi_gpu_env_single_device():
    r = run(EXE, 300, 64, 272, "hw", "on", env={"MM_GPUS": "1"})
    assert r.returncode == 0 and "Successfully verified." in r.stdout


@pytest.mark.gpu
def test_run_hardware_split_path_env():
    """MM_PATH=split: the float (Multiply, Add) runner verifies through MM_PATH_SPLIT; other builds refuse it loudly."""
    r = run(EXE, 513, 528, 528, "hw", "on", env={"MM_PATH": "split"})
    assert r.returncode == 0 and "Successfully verified." in r.stdout, r.stdout + r.stderr
    r = run(EXE, 4096, 4096, 4096, "hw", "off", env={"MM_PATH": "split"})
    assert r.returncode == 0 and PERF_LINE.search(r.stdout)
    r = run(os.path.join(ROOT, "bin", "RunHardware_int_Multiply_Add.exe"), 64, 64, 64, "hw", "on", env={"MM_PATH": "split"})
    assert r.returncode == 1 and "MM_PATH_SPLIT" in r.stderr


def test_run_hardware_rejects_unknown_path_env():
    r = run(EXE, 64, 64, 64, "hw", "off", env={"MM_PATH": "fast"})
    assert r.returncode == 1 and "MM_PATH must be" in r.stderr


# ---- the other two host binaries of the reference: TestSimulation, PrintSpecifications ---------
def test_print_specifications_cli_and_model():
    exe = os.path.join(ROOT, "bin", "PrintSpecifications.exe")
    r = run(exe, 16384, 16384)
    assert r.returncode == 1 and "N K M [<routed_frequency>]" in r.stderr
    r = run(exe, 16384, 16384, 16384)
    assert r.returncode == 0
    out = r.stdout
    assert "Number of operations: 8796093022208" in out            # 2*N*K*M (src/PrintSpecifications.cpp:40-41)
    assert "Ideal performance:    157286 GOp/s" in out              # 256 CU x 256 FLOP/clk x 2.4 GHz
    assert "Memory tile size: 256x256" in out and "Frequency:            2400 MHz" in out   # whole rounds of 256 x 256 tiles: that geometry (round 4, by energy)
    comm = int(re.search(r"Communication volume: (\d+)", out).group(1))
    assert comm == 16384 * 16384 * (1 + 16384 // 256 + 16384 // 256)  # N*M*(1 + K/TN + K/TM), :72-74
    small = run(exe, 4096, 4096, 4096).stdout                       # one round of tiles: 128 x 256, two such workgroups per CU
    assert "Memory tile size: 128x256" in small
    assert int(re.search(r"Communication volume: (\d+)", small).group(1)) == 4096 * 4096 * (1 + 4096 // 128 + 4096 // 256)
    r2 = run(exe, 16384, 16384, 16384, 1200)
    assert "Ideal performance:    78643" in r2.stdout               # scales with the routed frequency


def test_build_time_tile_knob_pins_the_geometry():
    """-DMM_MEMORY_TILE_SIZE_N=256 -DMM_MEMORY_TILE_SIZE_M=256 (reference CMakeLists.txt:18-20 -> Config.h.in:19-23): a
    binary built with the reference's tile knob runs the 256 x 256 geometry where the default build takes the
    library's per-problem pick; no GPU needed to see it (PrintSpecifications reads mm_kernel_info)."""
    default = run(os.path.join(ROOT, "bin", "PrintSpecifications.exe"), 4096, 4096, 4096).stdout
    pinned = run(os.path.join(ROOT, "bin", "PrintSpecifications_tile256x256.exe"), 4096, 4096, 4096).stdout
    assert "mfma_f32_128x256x16_w4x2_flush4096" in default and "Memory tile size: 128x256" in default
    assert "mfma_f32_256x256x16_w8_flush4096" in pinned and "Memory tile size: 256x256" in pinned


@pytest.mark.gpu
def test_build_time_tile_knob_runs_and_verifies_on_gpu():
    r = run(os.path.join(ROOT, "bin", "RunHardware_tile256x256.exe"), 4096, 512, 4096, "hw", "on")
    assert r.returncode == 0 and "Successfully verified." in r.stdout, r.stdout + r.stderr
    assert "Executing kernel (mfma_f32_256x256x16_w8_flush4096)" in r.stdout


def test_test_simulation_cli_errors():
    exe = os.path.join(ROOT, "bin", "TestSimulation.exe")
    assert run(exe).returncode == 1 and "Usage: ./TestSimulation N K M" in run(exe).stderr
    r = run(exe, 64, 17, 64)
    assert r.returncode == 1 and "K must be divisable by memory width." in r.stderr
    r = run(exe, 64, 16, 17)
    assert r.returncode == 1 and "M must be divisable by memory width." in r.stderr


def test_kernel_shims_export_the_reference_symbol_for_their_configuration():
    """The build-time configured kernel libraries: dynamic = 6 arguments, static = the 3-pointer
    form with the sizes baked in (include/MatrixMultiplication.h:155-169)."""
    import ctypes
    dyn = ctypes.CDLL(os.path.join(ROOT, "bin", "libmmkernel.so"))
    assert dyn.MatrixMultiplicationKernel is not None and not hasattr(dyn, "MatrixMultiplicationKernelSizes")
    st = ctypes.CDLL(os.path.join(ROOT, "bin", "libmmkernel_static_float_528x512x560.so"))
    n, k, m = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    st.MatrixMultiplicationKernelSizes(ctypes.byref(n), ctypes.byref(k), ctypes.byref(m))
    assert (n.value, k.value, m.value) == (528, 512, 560)
    # the symbol of the shim is its own definition, not the run-time configured one of libmm_gemm_amd.so
    import gemm_hls_amd as g
    addr = lambda lib: ctypes.cast(lib.MatrixMultiplicationKernel, ctypes.c_void_p).value
    assert len({addr(dyn), addr(st), addr(g.lib())}) == 3


@pytest.mark.gpu
def test_static_kernel_shim_three_pointer_call_on_gpu():
    import ctypes
    import numpy as np
    import _oracle
    st = ctypes.CDLL(os.path.join(ROOT, "bin", "libmmkernel_static_float_528x512x560.so"))
    st.MatrixMultiplicationKernel.argtypes = [ctypes.c_void_p] * 3
    st.MatrixMultiplicationKernel.restype = None
    a, b = _oracle.fill("float", 528, 512, 560)
    c = np.zeros((528, 560), np.float32)
    st.MatrixMultiplicationKernel(a.ctypes.data, b.ctypes.data, c.ctypes.data)
    assert _oracle.compare("float", c, a @ b, 1e-5)[0] == 0
    r = run(os.path.join(ROOT, "bin", "TestSimulation_static_float_528x512x560.exe"))
    assert r.returncode == 0 and "successfully verified" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_test_simulation_reference_ctest_shape_on_gpu():
    """The reference's CTest: TestSimulation 513 528 528 (CMakeLists.txt:155-159)."""
    r = run(os.path.join(ROOT, "bin", "TestSimulation.exe"), 513, 528, 528)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Running simulation..." in r.stdout and "Matrix-matrix multiplication successfully verified." in r.stdout


# ---- static-size build: MM_DYNAMIC_SIZES=OFF (CMakeLists.txt:21-24, host/RunHardware.cpp:18-22,62-70) -----
STATIC_EXE = os.path.join(ROOT, "bin", "RunHardware_static_float_528x512x560.exe")


def test_static_size_build_usage():
    r = run(STATIC_EXE, "hw", "on", "extra")
    assert r.returncode == 1 and "Usage: ./RunHardware.exe <mode [hw/hw_emu]> [<verify [on/off]>]" in r.stderr
    assert run(STATIC_EXE, "512").returncode == 1  # sizes are not arguments of a static build


@pytest.mark.gpu
@pytest.mark.parametrize("args", [(), ("hw",), ("hw_emu", "on"), ("hw", "off")])
def test_static_size_build_runs_on_gpu(args):
    r = run(STATIC_EXE, *args)
    assert r.returncode == 0, r.stdout + r.stderr
    m = PERF_LINE.search(r.stdout)
    assert m
    if "off" not in args:
        assert "Successfully verified." in r.stdout


def test_optimal_tile_size_tool_reproduces_the_shipped_tiles():
    """tools/optimal_tile_size.py = the reference's scripts/optimal_memory_tile_size.py with a CU's register file and
    LDS as the budgets: under the power-of-two restriction it lands on the tiles the kernels ship with, and without
    it on the 320 x 256 tile HISTORY.md lists as open."""
    import sys
    tool = os.path.join(ROOT, "tools", "optimal_tile_size.py")

    def first_lines(*args):
        r = subprocess.run([sys.executable, tool, *map(str, args)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r.stdout.splitlines()

    assert first_lines(32, 16384, 16384, "--pow2")[:2] == ["Tile sizes: 256x256", "Matrix sizes: 16384xKx16384"]
    assert first_lines(16, 513, 528, "--pow2")[:2] == ["Tile sizes: 256x256", "Matrix sizes: 768xKx768"]
    assert first_lines(64, 16384, 16384, "--pow2")[0] == "Tile sizes: 256x128"
    assert "128x64 per wavefront" in first_lines(32, 16384, 16384, "--pow2")[2] or "64x128 per wavefront" in first_lines(32, 16384, 16384, "--pow2")[2]
    assert first_lines(32, 16384, 16384)[0] == "Tile sizes: 320x256"
