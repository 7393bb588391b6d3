#!/usr/bin/env python3
"""Race screen: the same launch repeated many times must give the same bits every time (the
kernels are deterministic by construction: fixed tile ownership, fixed summation order), while other
work runs concurrently on a second stream to perturb timing.  Also checks against the first result
of a different geometry where one exists."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gemm_hls_amd as g  # noqa: E402

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
noise = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for dtype, tdt, n, reps in (("float", torch.float32, 8192, 60), ("half", torch.float16, 8192, 60),
                            ("double", torch.float64, 4096, 40), ("uint8_t", torch.uint8, 8192, 60)):
    a = torch.empty((n + 37, n), dtype=tdt, device=dev)   # ragged N on purpose
    b = torch.empty((n, n), dtype=tdt, device=dev)
    g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], a.data_ptr(), a.numel(), 5))
    g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], b.data_ptr(), b.numel(), 6))
    ref = g.matmul(a, b, dtype).clone()
    torch.cuda.synchronize()
    bad = 0
    t0 = time.perf_counter()
    for i in range(reps):
        with torch.cuda.stream(side):          # uneven background load on the memory system
            if i % 3:
                noise.mul_(1.0001)
        c = g.matmul(a, b, dtype)
        if not torch.equal(c.view(torch.uint8), ref.view(torch.uint8)):
            bad += 1
    torch.cuda.synchronize()
    print(f"{dtype:8s} {n+37}x{n}x{n}: {reps} launches, {bad} differing results, {time.perf_counter()-t0:.1f} s", flush=True)
    assert bad == 0
print("soak ok")
