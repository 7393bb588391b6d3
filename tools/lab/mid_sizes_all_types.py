"""Mid-size rates of every MFMA family next to torch (library yardstick), back to back."""
import os, sys
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
for dtype, tdt in (("double", torch.float64), ("half", torch.float16), ("uint8_t", torch.int8)):
    for s in (1024, 2048, 3072, 4096, 5120, 6144, 8192, 10240, 12288):
        if tdt == torch.int8:
            a = torch.randint(-100, 100, (s, s), device=dev, dtype=torch.int8); b = torch.randint(-100, 100, (s, s), device=dev, dtype=torch.int8)
            oa, ob = a.view(torch.uint8), b.view(torch.uint8)
        else:
            a = torch.empty((s, s), device=dev, dtype=tdt).uniform_(1, 10); b = torch.empty((s, s), device=dev, dtype=tdt).uniform_(1, 10)
            if tdt == torch.float16: a.mul_(2.0 ** -6); b.mul_(2.0 ** -6)
            oa, ob = a, b
        flop = 2.0 * s ** 3
        reps = max(4, min(100, int(4e13 / flop)))
        out = torch.empty((s, s), device=dev, dtype=oa.dtype)
        ours = rate(lambda: g.matmul(oa, ob, dtype, out=out), flop, reps)
        try:
            lib = rate((lambda: torch._int_mm(a, b)) if tdt == torch.int8 else (lambda: torch.matmul(a, b)), flop, reps)
        except Exception as e:
            lib = "n/a"
        print(dtype, s, g.kernel_name(g.make_config(dtype), s, s, s), "ours", ours, "torch", lib, flush=True)
        del a, b, oa, ob, out
        torch.cuda.empty_cache()
