import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from _lib import g
import torch
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
for s in (512, 1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192):
    a = torch.empty((s, s), device=dev).uniform_(1, 10); b = torch.empty((s, s), device=dev).uniform_(1, 10); c = torch.empty((s, s), device=dev)
    fl = 2.0 * s ** 3 / 1e12
    reps = max(10, min(200, int(2.0 / fl)))
    rm = fl / b2b(lambda: g.matmul(a, b, out=c), reps)
    at = fl / b2b(lambda: g.matmul(a, b, out=c, transposed_a=True), reps)
    tr = b2b(lambda: a.t().contiguous(), reps) * 1e6
    print(s, "row-major %.1f TF" % rm, " K x N A %.1f TF" % at, g.kernel_name(g.make_config("float", transposed_a=True), s, s, s), " torch transpose %.1f us" % tr, flush=True)
