#!/usr/bin/env python3
"""Is a kernel power/clock limited?  Same kernel, same shape, three operand fills: the reference's
uniform [1,10), all zeros, and small integers.  MI355X_MICROARCH.md (DVFS give-back): identical
instruction streams run at different clocks depending on operand toggling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g  # noqa: E402  (MM_LIB=lab selects the lab build)

for dtype, tdt, n in (("half", torch.float16, 16384), ("float", torch.float32, 16384)):
    dev = torch.device("cuda:0")
    a = torch.empty((n, n), dtype=tdt, device=dev)
    b = torch.empty((n, n), dtype=tdt, device=dev)
    c = torch.empty((n, n), dtype=tdt, device=dev)
    for name in ("uniform[1,10)", "zeros", "uniform[-1,1)"):
        if name == "zeros":
            a.zero_(); b.zero_()
        elif name == "uniform[-1,1)":
            a.copy_(torch.rand((n, n), device=dev) * 2 - 1); b.copy_(torch.rand((n, n), device=dev) * 2 - 1)
        else:
            g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], a.data_ptr(), a.numel(), 1))
            g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], b.data_ptr(), b.numel(), 2))
        for _ in range(2):
            g.matmul(a, b, dtype, out=c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.matmul(a, b, dtype, out=c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{dtype:6s} {n}^3 {name:14s} {2.0*n**3/dt/1e12:8.1f} TOp/s", flush=True)
