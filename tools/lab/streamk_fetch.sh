#!/bin/bash
# L2-miss traffic (FETCH_SIZE, x2 for 16-B-per-lane streams as tools/pmc_traffic.py) of the three ways to run a mid-size fp32
# problem on the 128 x 128 geometry: MM_F32_SPLITK=0 stream-K in teams with the ordered hand-over, 9 stream-K with one range
# per workgroup and the fix-up kernel, 1 whole tiles.  Usage (GPU box): bash tools/lab/streamk_fetch.sh [sizes]
cd /tmp; export TMPDIR=/tmp
SIZES=${1:-2560,3584,3968,4096,5120}
for sk in 0 9 1; do
  rm -rf /tmp/fs$sk
  MM_F32_SPLITK=$sk rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fs$sk -o pmc --output-format csv -- python /root/repo/tools/sweep.py f32 --sizes $SIZES --reps 2 --variants 35 > /tmp/fs$sk.log 2>&1
  python - <<PY
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob("/tmp/fs$sk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "mfma_f32" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-48:], r.get("Grid_Size"))
            rows.setdefault(key, []).append(float(r["Counter_Value"]) * 2 * 1024 / 1e9)
for (name, grid), v in rows.items():
    print("MM_F32_SPLITK=$sk", name, "grid", grid, "fetch GB per launch, in launch order (3 per size, sizes $SIZES):", " ".join("%.2f" % x for x in v))
PY
done
