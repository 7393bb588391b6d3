"""Random shapes through the fp32 shape-adaptive dispatch (64 x 64 geometry, split-K, stream-K in teams, whole tiles, the
transposition pre-pass for a K x N A): every result against fp64 on the device (normwise: |c - exact| <= 6e-6 * (|a| . |b|), inside the 1e-5 contract),
twice for run-to-run identity, and the K x N call against the row-major call bit for bit.  Prints the kernels met.
  python tools/fuzz_f32_dispatch.py [--shapes 300] [--seed 1]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _lib import g
ap = argparse.ArgumentParser(); ap.add_argument("--shapes", type=int, default=300); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(args.seed)
def rnd(lo, hi, mult):
    return max(mult, int(torch.randint(lo, hi + 1, (1,), generator=gen)) // mult * mult)
met = collections.Counter()
worst = 0.0
for i in range(args.shapes):
    style = i % 5
    if style == 0: n, k, m = rnd(1, 1500, 1), rnd(8, 2048, 8), rnd(4, 1500, 4)            # below a round of tiles
    elif style == 1: n, k, m = rnd(1500, 6000, 1), rnd(256, 4096, 32), rnd(1500, 6000, 4)  # between rounds
    elif style == 2: n, k, m = rnd(1, 700, 1), rnd(2048, 20000, 8), rnd(4, 700, 4)         # few tiles, long K
    elif style == 3: n, k, m = rnd(64, 4096, 64), rnd(8, 9000, 8), rnd(64, 4096, 64)       # multiples of 64
    else: n, k, m = rnd(1, 8000, 1), rnd(8, 1024, 8), rnd(4, 8000, 4)                      # flat
    dg = torch.Generator(device=dev).manual_seed(1000 + i)
    a = torch.empty((n, k), device=dev).uniform_(-3, 10, generator=dg); b = torch.empty((k, m), device=dev).uniform_(-3, 10, generator=dg)
    name = g.kernel_name(g.make_config("float"), n, k, m)
    c1 = g.matmul(a, b); c2 = g.matmul(a, b)
    assert torch.equal(c1, c2), ("not deterministic", (n, k, m), name)
    exact = a.double() @ b.double(); scale = a.double().abs() @ b.double().abs()
    err = ((c1.double() - exact).abs() / scale).max().item()
    assert err < 6e-6, ("error", err, (n, k, m), name)
    worst = max(worst, err)
    met[name.replace("mfma_f32_", "")] += 1
    if name.endswith("_streamk"):      # the two-kernel cross-check (f32_splitk 11) performs the same additions in the same order: same bits
        g.set_tuning("f32_variant", 35)
        try:
            for knob, tag in ((11, "_streamk_two_kernels"),):
                g.set_tuning("f32_splitk", knob)
                assert g.kernel_name(g.make_config("float"), n, k, m).endswith(tag)
                assert torch.equal(g.matmul(a, b), c1), (tag + " != the default (last-arriver) form", (n, k, m))
        finally:
            g.set_tuning("f32_variant", -1); g.set_tuning("f32_splitk", -1)
        met["(two-kernel form == default form)"] += 1
    if n % 4 == 0:
        at = a.t().contiguous()
        name_t = g.kernel_name(g.make_config("float", transposed_a=True), n, k, m)
        ct = g.matmul(at, b, transposed_a=True)
        if name_t == name:
            assert torch.equal(ct, c1), ("K x N != row-major", (n, k, m), name)
        else:   # the K x N kernel itself (whole rounds of its tiles): same whole-tile arithmetic unless the row-major call split K
            errt = ((ct.double() - exact).abs() / scale).max().item()
            assert errt < 6e-6, ("K x N error", errt, (n, k, m), name_t)
        met["KxN:" + name_t.replace("mfma_f32_", "")] += 1
    del a, b, c1, c2, exact, scale
print(args.shapes, "shapes ok; worst normwise error %.2e" % worst)
for k_, v in sorted(met.items(), key=lambda kv: -kv[1]): print("  %4d  %s" % (v, k_))
