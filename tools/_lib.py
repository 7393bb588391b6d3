"""Which build of the C-ABI library the measurement tools drive.

    MM_LIB unset   the product, gemm_hls_amd/libmm_gemm_amd.so
    MM_LIB=lab     tools/lab/libmm_gemm_amd_lab.so: the same ABI with the matrix-core translation units in their lab
                   editions -- the retired schedules / ring depths and the work-skipping ablations that the sweeps under
                   profiles/ name by variant number.  Never loaded by the product, the tests of the product or bench.py.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gemm_hls_amd as g  # noqa: E402

if os.environ.get("MM_LIB") == "lab":
    g.LIB_PATH = os.path.join(ROOT, "tools", "lab", "libmm_gemm_amd_lab.so")
    g._lib = None
