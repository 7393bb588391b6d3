#!/usr/bin/env python3
"""K x N A (MM_TRANSPOSED_A) of half / uint8 at BASELINE sizes: the row-major default, the K x N call under the
shape-adaptive pick (round 4: transposition pre-pass + row-major default) and the K x N ping-pong kernel itself (pinned
variant), launched ROUND-ROBIN in one process so that clock / thermal drift hits all three alike.  Times are HIP events
around everything a call enqueues (pre-pass included).   python tools/kxn_prepass_check.py [--sizes 16384,32768]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gemm_hls_amd as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="16384,32768")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--types", default="half,uint8_t")
ap.add_argument("--min-m", type=int, default=-1, help="kxn_prepass_min_m knob: M from which a K x N A is transposed first (-1 = library default)")
args = ap.parse_args()
dev = torch.device("cuda:0")
g.set_tuning("kxn_prepass_min_m", args.min_m)
TYPES = {"half": ("half", torch.float16, "f16_variant", 11), "uint8_t": ("uint8_t", torch.uint8, "i8_variant", 10)}
for dtype, tdt, knob, pinned in (TYPES[t] for t in args.types.split(",")):
    for s in [int(x) for x in args.sizes.split(",")]:
        a = torch.empty((s, s), dtype=tdt, device=dev)
        b = torch.empty((s, s), dtype=tdt, device=dev)
        c = torch.empty((s, s), dtype=tdt, device=dev)
        g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], a.data_ptr(), a.numel(), 1))
        g._check(g.lib().mm_fill_device(0, g.DTYPES[dtype], b.data_ptr(), b.numel(), 2))
        if dtype == "half":
            a.mul_(2.0 ** -6)            # keep the sums finite
        cases = [("row-major", False, -1), ("KxN auto", True, -1), ("KxN kernel", True, pinned)]
        times = {c_[0]: [] for c_ in cases}
        names = {}
        for rnd in range(args.reps + 1):
            for label, kxn, v in cases:
                g.set_tuning(knob, v)
                names[label] = g.kernel_name(g.make_config(dtype, transposed_a=kxn), s, s, s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.matmul(a, b, dtype, transposed_a=kxn, out=c)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[label].append(e0.elapsed_time(e1))
        g.set_tuning(knob, -1)
        base = sorted(times["row-major"])[len(times["row-major"]) // 2]
        for label, _, _ in cases:
            ts = sorted(times[label])
            med = ts[len(ts) // 2]
            print(f"{dtype:8s} {s:6d}^3 {label:11s} {names[label]:40s} med {med:9.3f} ms {2.0 * s ** 3 / med / 1e9:9.1f} TOp/s  "
                  f"{100.0 * base / med:6.1f} % of row-major", flush=True)
        del a, b, c
        torch.cuda.empty_cache()
