import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from _lib import g
import torch
dev = torch.device("cuda:0")
def b2b(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best
for nm in (1024, 512):
    for v in (64, 35):
        prev = None
        for k in (32, 512, 1024, 2048, 4096):
            a = torch.empty((nm, k), device=dev).uniform_(1, 10); b = torch.empty((k, nm), device=dev).uniform_(1, 10); c = torch.empty((nm, nm), device=dev)
            g.set_tuning("f32_variant", v); g.set_tuning("f32_splitk", 1)
            t = b2b(lambda: g.matmul(a, b, out=c), 200) * 1e6
            g.set_tuning("f32_variant", -1); g.set_tuning("f32_splitk", -1)
            print(f"n=m={nm} variant {v} K={k}: {t:.1f} us" + (f"  (+{(t - prev[1]) / ((k - prev[0]) / 32):.3f} us per 32-k slab)" if prev else ""), flush=True)
            prev = (k, t)
