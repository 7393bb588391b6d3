"""K x N A (MM_TRANSPOSED_A) against row-major A for every MFMA family, back to back."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
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
for dtype, tdt in (("double", torch.float64), ("half", torch.float16), ("uint8_t", torch.uint8), ("float", torch.float32)):
    for s in (1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
        if tdt == torch.uint8:
            a = torch.randint(0, 255, (s, s), device=dev, dtype=torch.uint8); b = torch.randint(0, 255, (s, s), device=dev, dtype=torch.uint8)
        else:
            a = torch.empty((s, s), device=dev, dtype=tdt).uniform_(1, 10); b = torch.empty((s, s), device=dev, dtype=tdt).uniform_(1, 10)
            if tdt == torch.float16: a.mul_(2.0 ** -6); b.mul_(2.0 ** -6)
        flop = 2.0 * s ** 3
        reps = max(3, min(100, int(2e13 / flop)))
        out = torch.empty((s, s), device=dev, dtype=tdt)
        rm = rate(lambda: g.matmul(a, b, dtype, out=out), flop, reps)
        at = rate(lambda: g.matmul(a, b, dtype, out=out, transposed_a=True), flop, reps)
        print(dtype, s, "row-major", rm, g.kernel_name(g.make_config(dtype), s, s, s), "| K x N", at, g.kernel_name(g.make_config(dtype, transposed_a=True), s, s, s), flush=True)
        del a, b, out
        torch.cuda.empty_cache()
