"""ctypes binding of the CPU oracle (oracle/mm_oracle.c) and, when built, of the reference's
own kernel (oracle/_ref/*/libmmkernel_ref.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by gemm_hls_amd/."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# numerically identical to include/mm_gemm.h (asserted in tests/test_capi_symbols.py)
DTYPES = {"float": 0, "double": 1, "half": 2, "int8_t": 3, "uint8_t": 4, "int16_t": 5,
          "uint16_t": 6, "int": 7, "unsigned": 8, "long": 9, "unsigned long": 10}
OPS = {"Add": 0, "Multiply": 1, "And": 2, "Min": 3, "Max": 4}
NP_DTYPES = {"float": np.float32, "double": np.float64, "half": np.float16, "int8_t": np.int8,
             "uint8_t": np.uint8, "int16_t": np.int16, "uint16_t": np.uint16, "int": np.int32,
             "unsigned": np.uint32, "long": np.int64, "unsigned long": np.uint64}

_lib = None


def build():
    """(Re)build the oracle .so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", os.path.join(ORACLE_DIR, "libmm_oracle.so")],
                   check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libmm_oracle.so")
        src = os.path.join(ORACLE_DIR, "mm_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = ctypes.CDLL(path)
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.mm_oracle_fill.argtypes = [i, vp, sz, vp, sz]
        L.mm_oracle_fill.restype = i
        for fn in (L.mm_oracle_naive, L.mm_oracle_naive_wide):
            fn.argtypes = [i, i, i, i, vp, vp, vp, sz, sz, sz, i]
            fn.restype = i
        L.mm_oracle_gemm_f32_in_f64.argtypes = [vp, vp, vp, sz, sz, sz, i]
        L.mm_oracle_compare.argtypes = [i, vp, vp, sz, sz, ctypes.c_double,
                                        ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]
        L.mm_oracle_compare.restype = ctypes.c_long
        L.mm_oracle_draws_real.argtypes = [vp, sz]
        L.mm_oracle_draws_int.argtypes = [vp, sz]
        L.mm_oracle_dtype_size.argtypes = [i]
        L.mm_oracle_dtype_size.restype = sz
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fill(dtype, n, k, m, transposed_a=False):
    """Reference input generator: seed 5, dist(1,10), all of A then all of B."""
    npdt = NP_DTYPES[dtype]
    a = np.empty((k, n) if transposed_a else (n, k), dtype=npdt)
    b = np.empty((k, m), dtype=npdt)
    rc = lib().mm_oracle_fill(DTYPES[dtype], _ptr(a), a.size, _ptr(b), b.size)
    assert rc == 0
    return a, b


def naive(dtype, map_op, reduce_op, a, b, transposed_a=False, threads=None, wide_half=False):
    """Naive<Map,Reduce> (include/Utility.h:18-42)."""
    k, m = b.shape
    n = a.shape[1] if transposed_a else a.shape[0]
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    c = np.empty((n, m), dtype=NP_DTYPES[dtype])
    fn = lib().mm_oracle_naive_wide if wide_half else lib().mm_oracle_naive
    rc = fn(DTYPES[dtype], OPS[map_op], OPS[reduce_op], int(transposed_a), _ptr(a), _ptr(b), _ptr(c),
            n, k, m, threads or (os.cpu_count() or 1))
    assert rc == 0
    return c


def gemm_f32_in_f64(a, b):
    n, k = a.shape
    m = b.shape[1]
    c = np.empty((n, m), dtype=np.float64)
    lib().mm_oracle_gemm_f32_in_f64(_ptr(np.ascontiguousarray(a)), _ptr(np.ascontiguousarray(b)), _ptr(c), n, k, m, 0)
    return c


def compare(dtype, test, ref, tol):
    """Reference comparison rule. Returns (mismatches, first_index, max_rel)."""
    test = np.ascontiguousarray(test)
    ref = np.ascontiguousarray(ref)
    assert test.shape == ref.shape and test.dtype == ref.dtype == NP_DTYPES[dtype]
    first = ctypes.c_long(-1)
    worst = ctypes.c_double(0)
    bad = lib().mm_oracle_compare(DTYPES[dtype], _ptr(test), _ptr(ref), test.shape[0], test.shape[1],
                                  tol, ctypes.byref(first), ctypes.byref(worst))
    return bad, first.value, worst.value


def draws_real(n):
    out = np.empty(n, dtype=np.float64)
    lib().mm_oracle_draws_real(_ptr(out), n)
    return out


def draws_int(n):
    out = np.empty(n, dtype=np.uint64)
    lib().mm_oracle_draws_int(_ptr(out), n)
    return out


# ---- the reference's own kernel, compiled by oracle/build_ref.sh -----------------------------
def ref_dir(dtype="float", map_op="Multiply", reduce_op="Add", tiles="256x256_32x8", transposed_a=False):
    return os.path.join(ORACLE_DIR, "_ref", f"{dtype.replace(' ', '_')}_{map_op}_{reduce_op}_{tiles}"
                        + ("_transposedA" if transposed_a else ""))


def ref_available(dtype="float", map_op="Multiply", reduce_op="Add", transposed_a=False):
    return os.path.exists(os.path.join(ref_dir(dtype, map_op, reduce_op, transposed_a=transposed_a), "libmmkernel_ref.so"))


_ref_libs = {}


def ref_kernel(dtype, map_op, reduce_op, a, b, transposed_a=False):
    """Call the reference's extern "C" MatrixMultiplicationKernel (kernel/Top.cpp:6) compiled
    from /root/reference against the hlslib shim: the repo's own CPU simulation path.
    transposed_a: the reference's -DMM_TRANSPOSED_A build, `a` is K x N (N a multiple of the bus
    width in elements: SizeNMemory floors, include/MatrixMultiplication.h:61-64)."""
    key = (dtype, map_op, reduce_op, bool(transposed_a))
    if key not in _ref_libs:
        L = ctypes.CDLL(os.path.join(ref_dir(dtype, map_op, reduce_op, transposed_a=transposed_a), "libmmkernel_ref.so"))
        L.MatrixMultiplicationKernel.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint] * 3
        L.MatrixMultiplicationKernel.restype = None
        _ref_libs[key] = L
    k, m = b.shape
    n = a.shape[1] if transposed_a else a.shape[0]
    assert a.shape == ((k, n) if transposed_a else (n, k))
    npdt = NP_DTYPES[dtype]
    # The reference's readers index whole outer tiles and rely on WriteC's mask: on a ragged
    # last tile ReadA runs up to 255 rows past A (kernel/Memory.cpp:11-20,64), ReadB up to one tile width past B
    # (:36-44,282).  Back both operands with zero-filled slack so those reads stay inside memory this process
    # owns (without it the call segfaults now and then, depending on where numpy put the arrays).
    def backed(x, slack):
        buf = np.zeros(x.size + slack, dtype=npdt)
        buf[:x.size] = np.ascontiguousarray(x, dtype=npdt).reshape(-1)
        return buf
    a_buf = backed(a, 256 * max(n, k) + 4096)
    b_buf = backed(b, 256 * max(k, m) + 4096)
    c = np.zeros((n, m), dtype=npdt)
    # stdout of WriteC's per-tile progress line (kernel/Memory.cpp:384-389) is left alone
    _ref_libs[key].MatrixMultiplicationKernel(_ptr(a_buf), _ptr(b_buf), _ptr(c), n, k, m)
    return c
