#!/bin/bash
# Kernel durations and the gaps between them for small fp32 problems launched back to back (GPU box):
#   bash tools/lab/small_kernel_trace.sh [n]      MM_F32_SPLITK = -1 (auto: split-K), 0 (ordered stream-K), 1 (plain)
cd /tmp; export TMPDIR=/tmp
N=${1:-1024}
cat > /tmp/drv.py <<PY
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from _lib import g
n = $N
a = torch.empty((n, n), device="cuda").uniform_(1, 10); b = torch.empty((n, n), device="cuda").uniform_(1, 10); c = torch.empty((n, n), device="cuda")
g.set_tuning("f32_variant", 35)
import os
g.set_tuning("f32_splitk", int(os.environ["SK"]))
print(g.kernel_name(g.make_config("float"), n, n, n))
for _ in range(3):
    for _ in range(40): g.matmul(a, b, out=c)
    torch.cuda.synchronize()
PY
for sk in 4 0 1; do
  rm -rf /tmp/kt$sk
  SK=$sk rocprofv3 --kernel-trace -d /tmp/kt$sk -o kt --output-format csv -- python /tmp/drv.py > /tmp/kt$sk.log 2>&1
  echo "== f32_splitk=$sk  $(grep mfma_f32 /tmp/kt$sk.log)"
  python - <<PY
import csv, glob, collections
d = collections.OrderedDict()
for f in glob.glob("/tmp/kt$sk/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    prev_end = None
    for r in rows:
        full = r["Kernel_Name"]
        name = next((t for t in ("streamk_ordered", "streamk_fixup", "streamk_kernel", "splitk_reduce", "mfma_f32_kernel", "fill", "Fill") if t in full), full[:40])
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) if prev_end else 0
        prev_end = e
        d.setdefault(name, []).append(((e - s) / 1e3, gap / 1e3))
for k, v in d.items():
    v = v[len(v)//2:]
    print("  ", k, "launches", len(v), "median us %.1f" % sorted(x[0] for x in v)[len(v)//2], " median gap before it us %.1f" % sorted(x[1] for x in v)[len(v)//2])
PY
done
