#!/usr/bin/env python3
"""Stream-K combine forms, round-robin in one process (clock / thermal drift hits all alike): f32_splitk 0 = the shipped
last-arriver form (flags: sc1 stores, counted waits), 12 = the counter-TICKET form written in the language's memory model
(release fence + one acq_rel read-modify-write per part), 11 = teams + fix-up kernel.  Also checks that
the three give the same bits on every size (whole output, on the device), with debug_poison on for one pass.

  python tools/streamk_ticket_ab.py [sizes] [rounds]
"""
import ctypes
import sys

sys.path.insert(0, "tools")
from _lib import g  # noqa: E402
import torch  # noqa: E402

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2304, 2560, 2944, 3072, 3584, 4608, 5120, 5888, 7680]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 15
FORMS = [(0, "last_arriver"), (12, "ticket"), (11, "two_kernels")]
L = g.lib()
dev = torch.device("cuda", 0)
worst = 0.0
for s in sizes:
    a = torch.empty((s, s), dtype=torch.float32, device=dev)
    b = torch.empty((s, s), dtype=torch.float32, device=dev)
    g._check(L.mm_fill_device(0, 0, a.data_ptr(), a.numel(), 1))
    g._check(L.mm_fill_device(0, 0, b.data_ptr(), b.numel(), 2))
    cs = {f: torch.empty((s, s), dtype=torch.float32, device=dev) for f, _ in FORMS}
    cfg = g.make_config("float")
    t = ctypes.c_double(0)
    times = {f: [] for f, _ in FORMS}
    g.set_tuning("f32_variant", 35)
    for rnd in range(rounds + 1):
        g.set_tuning("debug_poison", 1 if rnd == 0 else -1)
        for f, _ in FORMS:
            g.set_tuning("f32_splitk", f)
            g._check(L.mm_gemm_launch(0, ctypes.byref(cfg), a.data_ptr(), b.data_ptr(), cs[f].data_ptr(), s, s, s, ctypes.byref(t)))
            if rnd:
                times[f].append(t.value)
        if rnd == 0:
            torch.cuda.synchronize()
            same = all(torch.equal(cs[0], cs[f]) for f, _ in FORMS[1:]) and bool(torch.isfinite(cs[0]).all())
            if not same:
                print(f"{s}: BITS DIFFER between the forms (or a tile was left unfinished)", flush=True)
                sys.exit(1)
    g.set_tuning("f32_splitk", -1)
    g.set_tuning("f32_variant", -1)
    med = {f: sorted(v)[len(v) // 2] for f, v in times.items()}
    tf = {f: 2.0 * s ** 3 / med[f] / 1e12 for f in med}
    delta = 100.0 * (tf[12] / tf[0] - 1.0)
    worst = min(worst, delta)
    print(f"{s:5d}^3  " + "  ".join(f"{name} {tf[f]:7.2f} TF" for f, name in FORMS) + f"   ticket vs last_arriver {delta:+.2f} %   same bits", flush=True)
print(f"worst ticket vs last_arriver: {worst:+.2f} %")
