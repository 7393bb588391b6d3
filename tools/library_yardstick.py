"""torch.matmul (hipBLASLt / rocBLAS) on the BASELINE-sized operands of each data type, next to this library's kernels, queued
back to back on one stream.  A yardstick for how much of the roofline a tuned vendor kernel reaches on the same box under
the same power limit -- not a parity reference (different summation orders; int8 accumulates in int32 there and here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g

dev = torch.device("cuda:0")
def rate(fn, flop, reps):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return round(flop / best / 1e12, 1)
torch.backends.cuda.matmul.allow_tf32 = False
cases = [("float", torch.float32, 16384, 4), ("double", torch.float64, 16384, 2), ("half", torch.float16, 16384, 8), ("half", torch.float16, 32768, 2),
         ("uint8_t", torch.int8, 16384, 8)]
for dtype, tdt, s, reps in cases:
    if tdt == torch.int8:
        a = torch.randint(-100, 100, (s, s), device=dev, dtype=torch.int8); b = torch.randint(-100, 100, (s, s), device=dev, dtype=torch.int8)
        ours_a, ours_b = a.view(torch.uint8), b.view(torch.uint8)
    else:
        a = torch.empty((s, s), device=dev, dtype=tdt).uniform_(1, 10); b = torch.empty((s, s), device=dev, dtype=tdt).uniform_(1, 10)
        if tdt == torch.float16:
            a.mul_(2.0 ** -6); b.mul_(2.0 ** -6)
        ours_a, ours_b = a, b
    flop = 2.0 * s ** 3
    out = torch.empty((s, s), device=dev, dtype=ours_a.dtype)
    ours = rate(lambda: g.matmul(ours_a, ours_b, dtype, out=out), flop, reps)
    del out
    try:
        if tdt == torch.int8:
            lib = rate(lambda: torch._int_mm(a, b), flop, reps)
        else:
            o2 = torch.empty((s, s), device=dev, dtype=tdt)
            lib = rate(lambda: torch.matmul(a, b, out=o2), flop, reps)
            del o2
    except Exception as e:
        lib = f"n/a ({type(e).__name__}: {str(e)[:80]})"
    print(dtype, f"{s}^3", g.kernel_name(g.make_config(dtype), s, s, s), "this library", ours, "T(FL)OP/s, torch", lib, flush=True)
    del a, b, ours_a, ours_b
    torch.cuda.empty_cache()
