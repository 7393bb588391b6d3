"""Back-to-back throughput of mid-size fp32 problems: R launches queued on one stream between two events (what a caller
that keeps the stream busy sees), next to the one-launch-at-a-time figure of tools/sweep.py (an idle GPU before every
launch).  Also torch.matmul (hipBLASLt / rocBLAS) on the same operands as a library yardstick -- not a parity reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g
import sweep

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [512, 768, 1024, 1536, 2048, 2304, 2560, 3072, 3584, 4096, 5120, 6144, 8192]
dev = torch.device("cuda:0")
def b2b(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best
for s in sizes:
    a = torch.empty((s, s), device=dev).uniform_(1, 10); b = torch.empty((s, s), device=dev).uniform_(1, 10); c = torch.empty((s, s), device=dev)
    reps = max(10, min(400, int(2e12 / (2.0 * s ** 3))))
    fl = 2.0 * s ** 3 / 1e12
    row = {}
    for label, sk in (("auto", -1), ("whole tiles", 1)):
        g.set_tuning("f32_splitk", sk)
        row[label] = round(fl / b2b(lambda: g.matmul(a, b, out=c), reps), 1)
        med, _ = sweep.time_config("float", "Multiply", "Add", s, s, s, 7)
        row[label + " (one at a time)"] = round(fl / med, 1)
    g.set_tuning("f32_splitk", -1)
    torch.backends.cuda.matmul.allow_tf32 = False
    row["torch.matmul"] = round(fl / b2b(lambda: torch.matmul(a, b, out=c), reps), 1)
    print(s, g.kernel_name(g.make_config("float"), s, s, s), row, flush=True)
