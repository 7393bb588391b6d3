"""The 64 x 256 half geometry (f16_variant 5) against the other kernels: equality and back-to-back rates."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from _lib import g
import torch
dev = torch.device("cuda:0")
for (n, k, m) in [(64, 64, 256), (1, 16, 8), (65, 80, 264), (300, 64, 272), (513, 1040, 528), (1024, 1024, 1024), (130, 8208, 136)]:
    a = torch.empty((n, k), device=dev, dtype=torch.float16).uniform_(-1, 2); b = torch.empty((k, m), device=dev, dtype=torch.float16).uniform_(-1, 2)
    out = {}
    for v in (5, 4, 0):
        g.set_tuning("f16_variant", v); out[v] = g.matmul(a, b, "half").clone()
    g.set_tuning("f16_variant", 5); name = g.kernel_name(g.make_config("half"), n, k, m); g.set_tuning("f16_variant", -1)
    exact = a.double() @ b.double()
    err = ((out[5].double() - exact).abs() / (a.double().abs() @ b.double().abs())).max().item()
    print((n, k, m), name, "== 128x256:", torch.equal(out[5], out[4]), "== 256x256 slab64:", torch.equal(out[5], out[0]), "err", err, flush=True)
def b2b(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best
for s in (512, 1024, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144):
    a = torch.empty((s, s), device=dev, dtype=torch.float16).uniform_(0.01, 0.15); b = torch.empty((s, s), device=dev, dtype=torch.float16).uniform_(0.01, 0.15); c = torch.empty((s, s), device=dev, dtype=torch.float16)
    fl = 2.0 * s ** 3 / 1e12
    reps = max(10, min(300, int(10.0 / fl)))
    row = {}
    for label, v in (("auto", -1), ("64x256", 5), ("128x256", 4), ("256x256 pp", 200)):
        g.set_tuning("f16_variant", v)
        if v == -1: row["auto kernel"] = g.kernel_name(g.make_config("half"), s, s, s)
        row[label] = round(fl / b2b(lambda: g.matmul(a, b, "half", out=c), reps), 1)
    g.set_tuning("f16_variant", -1)
    row["torch"] = round(fl / b2b(lambda: torch.matmul(a, b, out=c), reps), 1)
    print(s, row, flush=True)
