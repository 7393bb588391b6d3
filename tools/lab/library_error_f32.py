"""Worst-element relative error of fp32 16384^3 on the reference's input distribution (uniform [1, 10)): this library's
default kernel (chain bounded: accumulators flushed into C every 4096 k) and torch.matmul (hipBLASLt), against fp64 on
256 sampled rows.  The reference's own comparison tolerance is 1e-5 relative."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from _lib import g
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for s in (4096, 8192, 16384):
    gen = torch.Generator(device=dev).manual_seed(s)
    a = torch.empty((s, s), device=dev).uniform_(1, 10, generator=gen); b = torch.empty((s, s), device=dev).uniform_(1, 10, generator=gen)
    rows = torch.randint(0, s, (256,), device=dev, generator=gen)
    exact = a[rows].double() @ b.double()
    ours = g.matmul(a, b)[rows].double()
    lib = torch.matmul(a, b)[rows].double()
    e_ours = ((ours - exact).abs() / exact).max().item(); e_lib = ((lib - exact).abs() / exact).max().item()
    print(f"{s}^3  max rel err: this library {e_ours:.2e} ({g.kernel_name(g.make_config('float'), s, s, s)})   torch.matmul {e_lib:.2e}", flush=True)
    del a, b, exact, ours, lib
