"""The CMake configuration surface (CMakeLists.txt at the repo root): the reference's cache
variables (CMakeLists.txt:8-36 there) configure and build the host side here.  The device library
is taken pre-built (-DMM_PREBUILT_LIBRARY, what gemm_hls_amd/build.py produced) so that this stays
a seconds-long CPU test; the full path -- hipcc custom commands for every kernel -- is the same
file without that option."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gemm_hls_amd", "libmm_gemm_amd.so")

pytestmark = pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")


def _configure(tmp_path, *defs):
    build = tmp_path / "build"
    cmd = ["cmake", "-S", ROOT, "-B", str(build), f"-DMM_PREBUILT_LIBRARY={LIB}", *defs]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    return build, r


def _build(build):
    r = subprocess.run(["cmake", "--build", str(build), "-j", "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_default_configuration_builds_the_reference_targets(tmp_path):
    build, r = _configure(tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    _build(build)
    for name in ("RunHardware.exe", "TestSimulation", "PrintSpecifications", "libmmkernel.so"):
        assert (build / name).exists(), name
    u = subprocess.run([str(build / "RunHardware.exe")], capture_output=True, text=True)
    assert u.returncode == 1 and "Usage: ./RunHardware.exe N K M [<mode [hw/hw_emu]>] [<verify [on/off]>]" in u.stderr
    # CTest is registered with the reference's sizes: 2*256+1, 2*32*8+64/4, 2*256+16 (CMakeLists.txt:155-159)
    t = subprocess.run(["ctest", "-N", "-V"], cwd=build, capture_output=True, text=True)
    assert "TestSimulation" in t.stdout and "513" in t.stdout and "528" in t.stdout


def _compile_flags(build):
    out = ""
    for root, _, files in os.walk(build / "CMakeFiles"):
        for f in files:
            if f == "flags.make":
                out += open(os.path.join(root, f)).read()
    return out


def test_reconfiguring_does_not_pin_the_tile(tmp_path):
    """the tile knob's intent must not depend on how many times cmake ran.  Not given: never pinned, however
    often the tree is re-configured; given once: pinned on every later configure (the cache keeps the user's value)."""
    build, r = _configure(tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MM_MEMORY_TILE_SIZE_N" not in _compile_flags(build)
    for _ in range(2):
        r = subprocess.run(["cmake", str(build)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "MM_MEMORY_TILE_SIZE_N" not in _compile_flags(build)
    r = subprocess.run(["cmake", str(build), "-DMM_MEMORY_TILE_SIZE_N=128", "-DMM_MEMORY_TILE_SIZE_M=256"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for _ in range(2):
        flags = _compile_flags(build)
        assert "MM_MEMORY_TILE_SIZE_N=128" in flags and "MM_MEMORY_TILE_SIZE_M=256" in flags
        r = subprocess.run(["cmake", str(build)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0
    t = subprocess.run(["ctest", "-N", "-V"], cwd=build, capture_output=True, text=True)
    assert "257" in t.stdout and "528" in t.stdout       # CTest sizes follow the given tile: 2*128+1, 2*256+16


def test_static_half_transposed_power_configuration(tmp_path):
    build, r = _configure(tmp_path, "-DMM_DATA_TYPE=half", "-DMM_DYNAMIC_SIZES=OFF", "-DMM_SIZE_N=528", "-DMM_SIZE_K=512",
                          "-DMM_SIZE_M=576", "-DMM_TRANSPOSED_A=ON", "-DMM_POWER_METER=ON", "-DMM_MAP_OP=Multiply",
                          "-DMM_REDUCE_OP=Add", "-DMM_PLATFORM=xilinx_u250", "-DMM_PARALLELISM_N=16")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FPGA build option of the reference; ignored" in r.stdout
    _build(build)
    u = subprocess.run([str(build / "RunHardware.exe"), "hw", "on", "extra"], capture_output=True, text=True)
    assert u.returncode == 1 and "Usage: ./RunHardware.exe <mode [hw/hw_emu]> [<verify [on/off]>]" in u.stderr
    import ctypes
    shim = ctypes.CDLL(str(build / "libmmkernel.so"))
    n, k, m = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    shim.MatrixMultiplicationKernelSizes(ctypes.byref(n), ctypes.byref(k), ctypes.byref(m))
    assert (n.value, k.value, m.value) == (528, 512, 576)


@pytest.mark.parametrize("defs,msg", [(("-DMM_DATA_TYPE=quaternion",), "Could not get size of data type"),
                                      (("-DMM_MAP_OP=Divide",), "must be one of"),
                                      (("-DMM_DYNAMIC_SIZES=OFF", "-DMM_SIZE_K=100"), "divisible by the memory bus width")])
def test_configuration_errors_are_reported_at_configure_time(tmp_path, defs, msg):
    _, r = _configure(tmp_path, *defs)
    assert r.returncode != 0 and msg in (r.stdout + r.stderr)


def test_full_build_description_compiles_every_kernel_source():
    """Without MM_PREBUILT_LIBRARY every .hip under gemm_hls_amd/csrc gets a hipcc command (checked on
    the generated build system, not executed here: that is minutes of compilation)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["cmake", "-S", ROOT, "-B", d], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        mk = open(os.path.join(d, "Makefile")).read() + open(os.path.join(d, "CMakeFiles", "Makefile2")).read()
        rules = ""
        for root, _, files in os.walk(os.path.join(d, "CMakeFiles")):
            for f in files:
                if f == "build.make":
                    rules += open(os.path.join(root, f)).read()
        for src in os.listdir(os.path.join(ROOT, "gemm_hls_amd", "csrc")):
            if src.endswith(".hip"):
                assert src in rules, src
        assert "--offload-arch=gfx950" in rules and "-ffp-contract=off" in rules and "libmm_gemm_amd.so" in rules
        # the two units of the k-ordered contract (the reference's unfused Naive) are the ones compiled with contraction off
        for unit in ("mm_ordered.hip", "mm_valu_tile_fp_exact.hip"):
            cmds = [ln for ln in rules.split("\n") if "hipcc" in ln and f"-c {ROOT}/gemm_hls_amd/csrc/{unit}" in ln]
            assert cmds and all("-ffp-contract=off" in ln for ln in cmds), unit
        fast = [ln for ln in rules.split("\n") if "hipcc" in ln and f"-c {ROOT}/gemm_hls_amd/csrc/mm_valu_tile_fp.hip" in ln]
        assert fast and not any("-ffp-contract=off" in ln for ln in fast)


@pytest.mark.skipif(not os.path.isdir("/root/reference/host"), reason="needs the reference checkout (not present on the GPU box)")
def test_reference_dir_option_builds_the_references_own_hosts_unmodified(tmp_path):
    """-DMM_REFERENCE_DIR=<gemm_hls checkout>: the reference's host/RunHardware.cpp, test/TestSimulation.cpp and
    src/PrintSpecifications.cpp are compiled from where they lie against include/compat, with Config.h configured from the
    reference's own include/Config.h.in by CMake's configure_file (as the reference's CMakeLists.txt:136 does)."""
    build, r = _configure(tmp_path, "-DMM_REFERENCE_DIR=/root/reference")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Reference hosts:" in r.stdout
    _build(build)
    cfg = open(build / "reference" / "Config.h").read()
    assert "using Data_t = float;" in cfg and "hlslib::op::Multiply<Data_t>" in cfg and "${" not in cfg
    u = subprocess.run([str(build / "RunHardware_reference.exe")], capture_output=True, text=True)
    assert u.returncode == 1 and "Usage: ./RunHardware.exe N K M" in u.stderr
    u = subprocess.run([str(build / "RunHardware_reference.exe"), "513", "520", "528"], capture_output=True, text=True)
    assert u.returncode == 1 and "K (520) must be divisable by the memory width in K (16)." in u.stderr
    p = subprocess.run([str(build / "PrintSpecifications_reference"), "16384", "16384", "16384"], capture_output=True, text=True)
    assert p.returncode == 0 and "Number of operations: 8796093022208" in p.stdout
    assert (build / "TestSimulation_reference").exists()
    t = subprocess.run(["ctest", "-N"], cwd=build, capture_output=True, text=True)
    assert "ReferenceRunHardwareVerify" in t.stdout and "ReferenceTestSimulation" in t.stdout
    # a half, K x N build of the same sources
    build2, r = _configure(tmp_path / "h", "-DMM_REFERENCE_DIR=/root/reference", "-DMM_DATA_TYPE=half", "-DMM_TRANSPOSED_A=ON")
    assert r.returncode == 0, r.stdout + r.stderr
    _build(build2)
    assert "using Data_t = half;" in open(build2 / "reference" / "Config.h").read()
