"""The 64 x 64 fp64 geometry (f64_variant 4) against the 128 x 128 / 256 x 128 ones: bit-identity and back-to-back rates."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from _lib import g
import torch
rng = np.random.default_rng(0)
for (n, k, m) in [(64, 16, 64), (1, 8, 2), (65, 40, 70), (300, 64, 272), (513, 1032, 528), (1024, 1024, 1024), (130, 8216, 132)]:
    a = rng.uniform(-3, 10, (n, k)); b = rng.uniform(-3, 10, (k, m))
    out = {}
    for v in (4, 1, 0):
        g.set_tuning("f64_variant", v); out[v], _ = g.matmul_capi(a, b, "double")
    at = np.ascontiguousarray(a.T) if n % 2 == 0 else None
    g.set_tuning("f64_variant", 4)
    name = g.kernel_name(g.make_config("double"), n, k, m)
    c_at = g.matmul_capi(at, b, "double", transposed_a=True)[0] if at is not None else out[4]
    g.set_tuning("f64_variant", -1)
    print((n, k, m), name, "== 128x128:", np.array_equal(out[4], out[1]), "== 256x128:", np.array_equal(out[4], out[0]), "K x N ==:", np.array_equal(c_at, out[4]),
          "err", float(np.max(np.abs(out[4] - a @ b) / (np.abs(a) @ np.abs(b)))), flush=True)
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
for s in (256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2560, 3072, 3584, 4096):
    a = torch.empty((s, s), device=dev, dtype=torch.float64).uniform_(1, 10); b = torch.empty((s, s), device=dev, dtype=torch.float64).uniform_(1, 10); c = torch.empty((s, s), device=dev, dtype=torch.float64)
    fl = 2.0 * s ** 3 / 1e12
    reps = max(10, min(300, int(1.0 / fl)))
    row = {}
    for label, v in (("auto", -1), ("64x64", 4), ("128x128", 1), ("256x128", 0)):
        g.set_tuning("f64_variant", v)
        if v == -1: row["auto kernel"] = g.kernel_name(g.make_config("double"), s, s, s)
        row[label] = round(fl / b2b(lambda: g.matmul(a, b, "double", out=c), reps), 1)
    g.set_tuning("f64_variant", -1)
    row["torch"] = round(fl / b2b(lambda: torch.matmul(a, b, out=c), reps), 1)
    print(s, row, flush=True)
