#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer boundary: one call of the reference's entry point
MatrixMultiplicationKernel(a, b, c, N, K, M) with HOST arrays (alloc + H2D + kernel + D2H), float
16384^3.  Reported in DESIGN.md next to the kernel-only number; it is never bench.py's `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402  (MM_LIB=lab selects the lab build)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
a = rng.uniform(1, 10, size=(n, n)).astype(np.float32)
b = rng.uniform(1, 10, size=(n, n)).astype(np.float32)
c = np.empty((n, n), np.float32)
g.device_count()
for i in range(2):
    t0 = time.perf_counter()
    g.lib().MatrixMultiplicationKernel(a.ctypes.data, b.ctypes.data, c.ctypes.data, n, n, n)
    dt = time.perf_counter() - t0
    print(f"call {i}: {dt*1e3:.1f} ms  -> {2.0*n**3/dt/1e12:.2f} TFLOP/s PCIe-inclusive "
          f"({3*n*n*4/1e9:.2f} GB over the link)")
