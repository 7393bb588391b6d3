"""include/compat -- hlslib::ocl over include/mm_gemm.h -- exercised by a client written for this test (tests/compat/), so that the
adapter is covered wherever the reference checkout is not (the GPU box): its whole surface, the forms the reference's host does
not use included.  The reference's own sources against the same headers: tests/test_gpu_ref_hosts.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = "/opt/rocm/lib/llvm/bin/clang++"
EXE = os.path.join(ROOT, "bin", "compat_adapter_user.exe")


@pytest.fixture(scope="module")
def adapter_user():
    src = os.path.join(ROOT, "tests", "compat", "adapter_user.cpp")
    cmd = [CXX, "-std=c++17", "-O1", "-Wall", "-Werror", "-DMM_DYNAMIC_SIZES", "-I" + os.path.join(ROOT, "tests", "compat"),
           "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"), src, "-o", EXE,
           "-L" + os.path.join(ROOT, "gemm_hls_amd"), "-lmm_gemm_amd", "-Wl,-rpath," + os.path.join(ROOT, "gemm_hls_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return EXE


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_adapter_compiles_warning_free_and_fails_loudly_without_a_device(adapter_user):
    if _has_gpu():
        pytest.skip("GPU box: the run is test_adapter_client_runs_on_the_device")
    r = subprocess.run([adapter_user], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and 'Execution failed with error: "no usable device' in r.stderr and "no CPU fallback" in r.stderr


def test_compat_headers_stand_alone():
    """Each compat header compiles on its own (include order must not matter to a client)."""
    for header in ("hlslib/xilinx/DataPack.h", "hlslib/xilinx/Operators.h", "hlslib/xilinx/Stream.h", "hlslib/xilinx/Resource.h",
                   "hlslib/xilinx/Utility.h", "hls_half.h", "hlslib/xilinx/OpenCL.h"):
        code = f'#include "{header}"\nint main() {{ return 0; }}\n'
        r = subprocess.run([CXX, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DMM_DYNAMIC_SIZES", "-x", "c++", "-", "-I" + os.path.join(ROOT, "tests", "compat"),
                            "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include")],
                           input=code, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, header + "\n" + r.stderr[-2000:]


def test_half_class_follows_binary16_arithmetic():
    """hls_half.h: every operation rounds to binary16 once (what the reference's Naive relies on for Data_t = half), the class is
    not std::is_floating_point (so the reference's verification compares half exactly), and it is 2 bytes."""
    code = r'''
#include <cstdio>
#include <type_traits>
#include "hls_half.h"
int main() {
  static_assert(!std::is_floating_point<half>::value && sizeof(half) == 2, "");
  half a(2049.0), b(1.0);                  // 2049 is not representable: rounds to 2048; 2048 + 1 rounds to 2048 (ties to even)
  half s = a + b, p = half(3.0) * half(0.1), acc(0);
  for (int i = 0; i < 4096; ++i) acc += half(1.0);     // saturates at 2048: binary16 accumulation, not float
  std::printf("%.12g %.12g %.12g %d %d\n", (double)(float)s, (double)(float)p, (double)(float)acc, (int)(half(5) != 0), (int)(std::abs(half(-2.5)) == half(2.5)));
  return 0;
}
'''
    exe = os.path.join(ROOT, "bin", "compat_half_check.exe")
    r = subprocess.run([CXX, "-std=c++17", "-x", "c++", "-", "-o", exe, "-I" + os.path.join(ROOT, "include", "compat")], input=code,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout.split()
    import numpy as np
    assert float(out[0]) == 2048.0 and float(out[2]) == 2048.0 and out[3:] == ["1", "1"]
    assert float(out[1]) == float(np.float16(3.0) * np.float16(0.1))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["hw", "hw_emu"])
def test_adapter_client_runs_on_the_device(adapter_user, mode):
    env = {k: v for k, v in os.environ.items() if k != "XCL_EMULATION_MODE"}
    r = subprocess.run([adapter_user, mode], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "adapter ok:" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert ("ordered" in r.stdout) == (mode == "hw_emu"), r.stdout
