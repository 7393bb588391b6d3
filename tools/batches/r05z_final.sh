#!/usr/bin/env bash
# round 5, closing batch: the whole GPU suite under -x with the throughput floors ARMED (MM_PERF_FLOORS=1), the bench
# line (live counters for the headline and BASELINE's C3 / C4 / C5b), rocprofv3 kernel stats of the headline-only command,
# the 8-rank dry run's line, and the reference's own unmodified runner at BASELINE size.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
T=${1:-r05z}
git rev-parse --short HEAD > gpurun_out/${T}_pytest_gpu.log 2>/dev/null
MM_PERF_FLOORS=1 python -m pytest tests -x -q -m gpu >> gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$? (MM_PERF_FLOORS=1)" >> gpurun_out/${T}_pytest_gpu.log
tail -4 gpurun_out/${T}_pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; wc -c gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
python - "$T" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["frac"], {k: d["roofline"].get(k) for k in ("traffic", "mfma_util_pct", "profiled_clock_GHz", "counters_measured_in_this_run", "counter_passes_s")})
for w in d["workloads"]:
    r = w.get("roofline", {})
    print(w.get("key"), w.get("value"), r.get("frac"), r.get("traffic"), r.get("mfma_util_pct"), r.get("profiled_clock_GHz"), r.get("counters_measured_in_this_run"))
P
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $R/gpurun_out/${T}_bench_headline_under_rocprofv3.json 2> $R/gpurun_out/${T}_rocprof.err)
cat $(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1) | cut -c1-400
cat gpurun_out/${T}_bench_headline_under_rocprofv3.json | cut -c1-400
env MM_BENCH_DEVICE_MOD=1 MM_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/${T}_bench_8_ranks_on_one_device_dry_run.json 2> gpurun_out/${T}_bench8.err; cut -c1-700 gpurun_out/${T}_bench_8_ranks_on_one_device_dry_run.json
for c in float double; do echo "== bin/ref_hosts/$c/RunHardware.exe 16384 16384 16384 hw off"; bin/ref_hosts/$c/RunHardware.exe 16384 16384 16384 hw off 2>&1 | tail -1; done > gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log 2>&1
echo "== bin/ref_hosts/float/RunHardware.exe 513 528 528 hw on" >> gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log; bin/ref_hosts/float/RunHardware.exe 513 528 528 hw on >> gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log 2>&1
echo "== bin/ref_hosts/float/TestSimulation.exe 513 528 528" >> gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log; bin/ref_hosts/float/TestSimulation.exe 513 528 528 2>&1 | tail -2 >> gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log
cat gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log
