"""gemm_hls_amd -- MI355X (gfx950) implementation of spcl/gemm_hls's hot path
C = A (map, reduce) B behind the reference's own boundary.

The product is the C-ABI shared library ``libmm_gemm_amd.so`` (``include/mm_gemm.h``) and the C++
host runner ``bin/RunHardware.exe``.  This module is the thin Python binding used by the tests
and ``bench.py``: ctypes onto the C ABI, with PyTorch only as plumbing for device memory, streams
and ``torch.distributed``.  There is no CPU fallback: importing works without a GPU (so the
symbol checks can run), every compute call raises ``MMError`` without one.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmm_gemm_amd.so")

# numerically identical to include/mm_gemm.h (asserted by tests/test_capi_symbols.py)
DTYPES = {"float": 0, "double": 1, "half": 2, "int8_t": 3, "uint8_t": 4, "int16_t": 5,
          "uint16_t": 6, "int": 7, "unsigned": 8, "long": 9, "unsigned long": 10}
OPS = {"Add": 0, "Multiply": 1, "And": 2, "Min": 3, "Max": 4}
PATH_AUTO, PATH_ORDERED, PATH_SPLIT = 0, 1, 2

EXPORTS = ["mm_init", "mm_alloc", "mm_free", "mm_copy_to_device", "mm_copy_to_host",
           "mm_fill_device", "mm_gemm_launch", "mm_gemm_enqueue", "mm_gemm_multi_device",
           "MatrixMultiplicationKernel", "mm_set_default_config", "mm_dtype_size",
           "mm_config_supported", "mm_kernel_name", "mm_kernel_info", "mm_last_error",
           "mm_gemm_host", "mm_tuning_set", "mm_tuning_get", "mm_release_workspace", "mm_device_pci_bus_id",
           "mm_row_slab", "mm_gemm_multi_device_timed"]


class MMError(RuntimeError):
    pass


class Config(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("map_op", ctypes.c_int), ("reduce_op", ctypes.c_int),
                ("path", ctypes.c_int), ("layout_a", ctypes.c_int)]


class KernelInfo(ctypes.Structure):  # mm_kernel_info_t
    _fields_ = [("tile_n", ctypes.c_uint), ("tile_m", ctypes.c_uint), ("tile_k", ctypes.c_uint),
                ("wavefronts", ctypes.c_uint), ("inst_n", ctypes.c_uint), ("inst_m", ctypes.c_uint),
                ("inst_k", ctypes.c_uint), ("ops_per_clk_per_cu", ctypes.c_double),
                ("compute_units", ctypes.c_uint), ("max_clock_mhz", ctypes.c_double),
                ("measured_issue_efficiency", ctypes.c_double)]


_lib = None


def lib():
    """The loaded C-ABI library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMError(f"{LIB_PATH} is missing: run `python gemm_hls_amd/build.py` "
                          "(or __graft_entry__.build()); there is no fallback implementation")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  Whichever is loaded first serves both, but if THIS library came first and
        # torch then brought a second runtime, the second HSA initialisation finds no device
        # ("no ROCm-capable device is detected").  So torch -- the plumbing for device memory and
        # streams in this binding anyway -- is imported first whenever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, u, i, sz = ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_size_t
        cfgp = ctypes.POINTER(Config)
        L.mm_init.argtypes = [ctypes.POINTER(i)]
        L.mm_alloc.argtypes = [i, sz, ctypes.POINTER(vp)]
        L.mm_free.argtypes = [i, vp]
        L.mm_copy_to_device.argtypes = [i, vp, vp, sz]
        L.mm_copy_to_host.argtypes = [i, vp, vp, sz]
        L.mm_fill_device.argtypes = [i, i, vp, sz, ctypes.c_ulonglong]
        L.mm_gemm_launch.argtypes = [i, cfgp, vp, vp, vp, u, u, u, ctypes.POINTER(ctypes.c_double)]
        L.mm_gemm_enqueue.argtypes = [vp, cfgp, vp, vp, vp, u, u, u]
        L.mm_gemm_multi_device.argtypes = [i, cfgp, vp, vp, vp, u, u, u, ctypes.POINTER(ctypes.c_double)]
        L.mm_gemm_multi_device_timed.argtypes = [i, cfgp, vp, vp, vp, u, u, u, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                                 ctypes.POINTER(ctypes.c_double)]
        L.MatrixMultiplicationKernel.argtypes = [vp, vp, vp, u, u, u]
        L.MatrixMultiplicationKernel.restype = None
        L.mm_set_default_config.argtypes = [cfgp]
        L.mm_dtype_size.argtypes = [i]
        L.mm_dtype_size.restype = sz
        L.mm_config_supported.argtypes = [cfgp]
        L.mm_kernel_name.argtypes = [cfgp, u, u, u]
        L.mm_kernel_name.restype = ctypes.c_char_p
        L.mm_kernel_info.argtypes = [cfgp, u, u, u, ctypes.POINTER(KernelInfo)]
        L.mm_last_error.restype = ctypes.c_char_p
        L.mm_gemm_host.argtypes = [cfgp, vp, vp, vp, u, u, u]
        L.mm_tuning_set.argtypes = [ctypes.c_char_p, i]
        L.mm_tuning_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(i)]
        L.mm_release_workspace.argtypes = [i]
        L.mm_device_pci_bus_id.argtypes = [i, ctypes.c_char_p, i]
        L.mm_row_slab.argtypes = [cfgp, u, u, u, i, i, ctypes.POINTER(u), ctypes.POINTER(u)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MMError(f"mm_gemm status {rc}: {lib().mm_last_error().decode()}")


def make_config(dtype="float", map_op="Multiply", reduce_op="Add", path=PATH_AUTO, transposed_a=False):
    return Config(DTYPES[dtype], OPS[map_op], OPS[reduce_op], path, int(transposed_a))


def device_count():
    n = ctypes.c_int(0)
    _check(lib().mm_init(ctypes.byref(n)))
    return n.value


def kernel_name(cfg, n, k, m):
    return lib().mm_kernel_name(ctypes.byref(cfg), n, k, m).decode()


def set_tuning(name, value):
    """Tuning knob of the library ("f32_variant", "f16_variant", "band_rows", ...; -1 = default)."""
    _check(lib().mm_tuning_set(name.encode(), int(value)))


def get_tuning(name):
    v = ctypes.c_int(0)
    _check(lib().mm_tuning_get(name.encode(), ctypes.byref(v)))
    return v.value


def kernel_info(cfg, n, k, m):
    info = KernelInfo()
    _check(lib().mm_kernel_info(ctypes.byref(cfg), n, k, m, ctypes.byref(info)))
    return info


_TORCH_DTYPES = None


def torch_dtype(dtype):
    global _TORCH_DTYPES
    import torch
    if _TORCH_DTYPES is None:
        _TORCH_DTYPES = {"float": torch.float32, "double": torch.float64, "half": torch.float16,
                         "int8_t": torch.int8, "uint8_t": torch.uint8, "int16_t": torch.int16,
                         "uint16_t": torch.uint16, "int": torch.int32, "unsigned": torch.uint32,
                         "long": torch.int64, "unsigned long": torch.uint64}
    return _TORCH_DTYPES[dtype]


def matmul(a, b, dtype="float", map_op="Multiply", reduce_op="Add", path=PATH_AUTO, transposed_a=False,
           out=None):
    """C = A (map, reduce) B on the current CUDA(HIP) device and torch's current stream.
    a: (N, K) -- or (K, N) with transposed_a -- b: (K, M); contiguous device tensors whose torch
    dtype matches `dtype`.  Asynchronous, like any torch op."""
    import torch
    if not (a.is_cuda and b.is_cuda):
        raise MMError("matmul needs device tensors: there is no CPU path")
    if a.device != b.device:
        raise MMError(f"operands live on different devices: {a.device} and {b.device}")
    tdt = torch_dtype(dtype)
    if not (a.is_contiguous() and b.is_contiguous()):
        raise MMError("matmul needs contiguous (row-major) operands")
    if a.dtype != tdt or b.dtype != tdt:
        raise MMError(f"operand dtypes {a.dtype}, {b.dtype} do not match Data_t={dtype} ({tdt})")
    if a.dim() != 2 or b.dim() != 2:
        raise MMError("matmul takes 2-D operands")
    k, m = b.shape
    n = a.shape[1] if transposed_a else a.shape[0]
    if (a.shape[0] if transposed_a else a.shape[1]) != k:
        raise MMError(f"inner dimensions differ: A {tuple(a.shape)}{' (K x N)' if transposed_a else ''}, B {tuple(b.shape)}")
    if out is None:
        out = torch.empty((n, m), dtype=tdt, device=a.device)
    elif (tuple(out.shape) != (n, m) or out.dtype != tdt or out.device != a.device or not out.is_contiguous()):
        raise MMError(f"out must be a contiguous {tdt} tensor of shape {(n, m)} on {a.device}; got "
                      f"{tuple(out.shape)}, {out.dtype}, {out.device}, contiguous={out.is_contiguous()}")
    cfg = make_config(dtype, map_op, reduce_op, path, transposed_a)
    # mm_gemm_enqueue launches on the CURRENT device: make that the operands' device
    with torch.cuda.device(a.device):
        stream = torch.cuda.current_stream(a.device).cuda_stream
        _check(lib().mm_gemm_enqueue(ctypes.c_void_p(stream), ctypes.byref(cfg), a.data_ptr(), b.data_ptr(),
                                     out.data_ptr(), n, k, m))
    return out


def row_slab(cfg, n, k, m, world_size, rank):
    """(row0, rows) of `rank`'s slab of the N split, as mm_gemm_multi_device deals the rows out (mm_row_slab)."""
    row0, rows = ctypes.c_uint(0), ctypes.c_uint(0)
    _check(lib().mm_row_slab(ctypes.byref(cfg), n, k, m, world_size, rank, ctypes.byref(row0), ctypes.byref(rows)))
    return row0.value, rows.value


def matmul_host(a, b, dtype="float", map_op="Multiply", reduce_op="Add", path=PATH_AUTO, devices=1, transposed_a=False, timing=False):
    """numpy in, numpy out, through mm_gemm_multi_device (rows of C split over `devices` GPUs).
    a: (N, K), or (K, N) with transposed_a.  Returns (C, kernel_seconds); with timing=True (C, kernel_seconds, per-device
    kernel seconds, host-clock seconds) from mm_gemm_multi_device_timed."""
    import numpy as np
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    k, m = b.shape
    n = a.shape[1] if transposed_a else a.shape[0]
    c = np.empty((n, m), dtype=a.dtype)
    cfg = make_config(dtype, map_op, reduce_op, path, transposed_a)
    t = ctypes.c_double(0)
    if timing:
        per_device = (ctypes.c_double * devices)()
        wall = ctypes.c_double(0)
        _check(lib().mm_gemm_multi_device_timed(devices, ctypes.byref(cfg), a.ctypes.data, b.ctypes.data, c.ctypes.data,
                                                n, k, m, ctypes.byref(t), per_device, ctypes.byref(wall)))
        return c, t.value, list(per_device), wall.value
    _check(lib().mm_gemm_multi_device(devices, ctypes.byref(cfg), a.ctypes.data, b.ctypes.data, c.ctypes.data,
                                      n, k, m, ctypes.byref(t)))
    return c, t.value


def matmul_capi(a, b, dtype="float", map_op="Multiply", reduce_op="Add", path=PATH_AUTO, transposed_a=False,
                device=0):
    """numpy in, numpy out, through the split-phase C ABI exactly as host/RunHardware.cpp drives the
    reference: mm_alloc x3, mm_copy_to_device x2, mm_gemm_launch (blocking, HIP-event timed),
    mm_copy_to_host.  Returns (C, kernel_seconds)."""
    import numpy as np
    L = lib()
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    k, m = b.shape
    n = a.shape[1] if transposed_a else a.shape[0]
    c = np.empty((n, m), dtype=a.dtype)
    cfg = make_config(dtype, map_op, reduce_op, path, transposed_a)
    ptrs = [ctypes.c_void_p() for _ in range(3)]
    t = ctypes.c_double(0)
    try:
        for p, arr in zip(ptrs, (a, b, c)):
            _check(L.mm_alloc(device, arr.nbytes, ctypes.byref(p)))
        _check(L.mm_copy_to_device(device, ptrs[0], a.ctypes.data, a.nbytes))
        _check(L.mm_copy_to_device(device, ptrs[1], b.ctypes.data, b.nbytes))
        _check(L.mm_gemm_launch(device, ctypes.byref(cfg), ptrs[0], ptrs[1], ptrs[2], n, k, m, ctypes.byref(t)))
        _check(L.mm_copy_to_host(device, c.ctypes.data, ptrs[2], c.nbytes))
    finally:
        for p in ptrs:
            if p.value:
                L.mm_free(device, p)
    return c, t.value
