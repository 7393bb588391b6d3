import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from _lib import g
import torch
rng = np.random.default_rng(0)
for (n, k, m) in [(64, 32, 64), (64, 8, 64), (1, 8, 4), (65, 40, 68), (300, 64, 272), (513, 4112, 528), (257, 4128, 260), (130, 8216, 132), (1024, 1024, 1024), (100, 12320, 36), (1000, 96, 3000)]:
    a = rng.uniform(-3, 10, (n, k)).astype(np.float32); b = rng.uniform(-3, 10, (k, m)).astype(np.float32)
    g.set_tuning("f32_splitk", 1)
    g.set_tuning("f32_variant", 64); name = g.kernel_name(g.make_config("float"), n, k, m); c64, _ = g.matmul_capi(a, b)
    g.set_tuning("f32_variant", 35); c35, _ = g.matmul_capi(a, b)
    g.set_tuning("f32_variant", -1); g.set_tuning("f32_splitk", -1)
    exact = a.astype(np.float64) @ b.astype(np.float64); scale = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    print((n, k, m), name, "bit-identical to 35:", np.array_equal(c64, c35), "err", float(np.max(np.abs(c64 - exact) / scale)), flush=True)
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
for (n, k, m) in [(256,)*3, (512,)*3, (768,)*3, (1024,)*3, (1280,)*3, (1536,)*3, (1792,)*3, (2048,)*3, (2560,)*3, (4096,)*3, (256, 8192, 256), (512, 4096, 512), (1024, 512, 1024), (2048, 256, 2048), (1024, 4096, 1024)]:
    a = torch.empty((n, k), device=dev).uniform_(1, 10); b = torch.empty((k, m), device=dev).uniform_(1, 10); c = torch.empty((n, m), device=dev)
    fl = 2.0 * n * k * m / 1e12
    reps = max(20, min(400, int(1.0 / fl)))
    row = {}
    g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    row["auto:" + g.kernel_name(g.make_config("float"), n, k, m).replace("mfma_f32_", "")] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
    g.set_tuning("f32_variant", 64); row["64x64"] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
    g.set_tuning("f32_variant", -1)
    row["torch"] = round(fl / b2b(lambda: torch.matmul(a, b, out=c), reps), 1)
    print((n, k, m), row, flush=True)
