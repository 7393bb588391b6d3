"""The 64 x 256 uint8 tile (i8_variant 5) against the other kernels: exact equality (integers mod 2^8) and back-to-back rates."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from _lib import g
import torch
dev = torch.device("cuda:0")
for (n, k, m) in [(64, 128, 256), (1, 32, 16), (65, 160, 272), (300, 64, 272), (513, 1056, 528), (1024, 1024, 1024), (130, 8224, 144)]:
    gen = torch.Generator(device=dev).manual_seed(n + k)
    a = torch.randint(0, 256, (n, k), device=dev, dtype=torch.uint8, generator=gen); b = torch.randint(0, 256, (k, m), device=dev, dtype=torch.uint8, generator=gen)
    out = {}
    for v in (5, 0, -1):
        g.set_tuning("i8_variant", v); out[v] = g.matmul(a, b, "uint8_t").clone()
    g.set_tuning("i8_variant", 5); name = g.kernel_name(g.make_config("uint8_t"), n, k, m); g.set_tuning("i8_variant", -1)
    exact = (a.cpu().to(torch.int64) @ b.cpu().to(torch.int64)) % 256
    print((n, k, m), name, "== slab128 256x256:", torch.equal(out[5], out[0]), "== default:", torch.equal(out[5], out[-1]), "== exact mod 256:", torch.equal(out[5].cpu().to(torch.int64), exact), flush=True)
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
for s in (512, 1024, 1536, 2048, 2560, 3072, 4096):
    a = torch.randint(0, 256, (s, s), device=dev, dtype=torch.uint8); b = torch.randint(0, 256, (s, s), device=dev, dtype=torch.uint8); c = torch.empty((s, s), device=dev, dtype=torch.uint8)
    fl = 2.0 * s ** 3 / 1e12
    reps = max(10, min(300, int(20.0 / fl)))
    row = {}
    for label, v in (("auto", -1), ("64x256", 5), ("256x256 slab128", 0), ("256x256 pp", 200)):
        g.set_tuning("i8_variant", v)
        if v == -1: row["auto kernel"] = g.kernel_name(g.make_config("uint8_t"), s, s, s)
        row[label] = round(fl / b2b(lambda: g.matmul(a, b, "uint8_t", out=c), reps), 1)
    g.set_tuning("i8_variant", -1)
    row["torch"] = round(fl / b2b(lambda: torch._int_mm(a.view(torch.int8), b.view(torch.int8)), reps), 1)
    print(s, row, flush=True)
