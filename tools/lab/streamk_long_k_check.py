import sys
sys.path.insert(0, "/root/repo")
import torch, gemm_hls_amd as g
dev = torch.device("cuda:0")
for (n, k, m) in [(2304, 16384, 2304), (2560, 12320, 2560)]:
    a = torch.empty((n, k), device=dev).uniform_(-3, 10); b = torch.empty((k, m), device=dev).uniform_(-3, 10)
    name = g.kernel_name(g.make_config("float"), n, k, m)
    c1 = g.matmul(a, b); c2 = g.matmul(a, b)
    rows = torch.arange(0, n, 37, device=dev)
    exact = a[rows].double() @ b.double(); scale = a[rows].double().abs() @ b.double().abs()
    err = ((c1[rows].double() - exact).abs() / scale).max().item()
    g.set_tuning("f32_splitk", 9); g.set_tuning("f32_variant", 35); cf = g.matmul(a, b); g.set_tuning("f32_splitk", -1); g.set_tuning("f32_variant", -1)
    print((n, k, m), name, "deterministic", torch.equal(c1, c2), "err %.2e" % err, "vs fix-up form %.2e" % ((c1 - cf).abs() / scale.new_tensor(1.0)).max().item() if False else "", "max |c - fixup| rel %.2e" % ((c1[rows].double() - cf[rows].double()).abs() / scale).max().item(), flush=True)
