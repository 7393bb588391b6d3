import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from _lib import g
dev = torch.device("cuda:0")
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
shapes = [(512,)*3, (768,)*3, (1024,)*3, (1280,)*3, (1536,)*3, (1792,)*3, (2048,)*3, (2304,)*3, (2560,)*3, (256, 8192, 256), (512, 4096, 512), (1024, 512, 1024), (2048, 256, 2048), (128, 32768, 128)]
for (n, k, m) in shapes:
    a = torch.empty((n, k), device=dev).uniform_(1, 10); b = torch.empty((k, m), device=dev).uniform_(1, 10); c = torch.empty((n, m), device=dev)
    fl = 2.0 * n * k * m / 1e12
    reps = max(20, min(400, int(1e12 / (fl * 1e12))))
    row = {}
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    row["auto:" + g.kernel_name(g.make_config("float"), n, k, m).replace("mfma_f32_", "")] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
    g.set_tuning("f32_variant", 35)
    g.set_tuning("f32_splitk", 1); row["plain35"] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
    for ms in (4, 8, 16):
        g.set_tuning("ablations", ms << 16); g.set_tuning("f32_splitk", 0)
        row[f"sk>= {ms}"] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
    g.set_tuning("ablations", 0)
    for sp in (2, 4, 8):
        g.set_tuning("f32_splitk", sp); row[f"splitk{sp}"] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    row["torch"] = round(fl / b2b(lambda: torch.matmul(a, b, out=c), reps), 1)
    print((n, k, m), row, flush=True)
