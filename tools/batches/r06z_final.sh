#!/usr/bin/env bash
# round 6, closing batch: the whole GPU suite under -x with the throughput floors ARMED (MM_PERF_FLOORS=1), smoke(), the bench line
# (live counters incl. the DRAM-destination pass for C2 / C3, the half reference-contract workload), rocprofv3 kernel stats of the
# headline-only command, the 8-rank dry run's line with its per-rank records, and the reference's own unmodified runner at
# BASELINE size -- timed (hw off), VERIFIED against its BLAS oracle (hw on, the -DMM_HAS_BLAS build), and the half build under the
# reference half contract.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
T=${1:-r06z}
echo "HEAD ${2:-unknown}" > gpurun_out/${T}_pytest_gpu.log
MM_PERF_FLOORS=1 python -m pytest tests -x -q -m gpu >> gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$? (MM_PERF_FLOORS=1)" >> gpurun_out/${T}_pytest_gpu.log
tail -12 gpurun_out/${T}_pytest_gpu.log
python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2 | tee gpurun_out/${T}_smoke.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; wc -c gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
python - "$T" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["frac"], {k: d["roofline"].get(k) for k in ("traffic", "achieved_fabric_GBps", "achieved_HBM_GBps", "hbm_busy_pct", "dram_destined_bytes", "mfma_util_pct", "profiled_clock_GHz", "counters_measured_in_this_run", "counter_passes_s")})
for w in d["workloads"]:
    r = w.get("roofline", {})
    print(w.get("key"), w.get("value"), r.get("frac"), r.get("traffic"), r.get("achieved_fabric_GBps"), r.get("achieved_HBM_GBps"), r.get("mfma_util_pct"), r.get("profiled_clock_GHz"), r.get("counters_measured_in_this_run"))
print(d.get("hbm_calibration"))
print({k: d["cpu_baseline"].get(k) for k in ("value", "seconds", "blas_sgemm_gflops_same_sample", "blas_threads", "naive_1thread_gflops_same_sample", "cpu_model")})
P
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $R/gpurun_out/${T}_bench_headline_under_rocprofv3.json 2> $R/gpurun_out/${T}_rocprof.err)
cp "$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1)" gpurun_out/${T}_bench_headline_kernel_stats.csv; cut -c1-400 gpurun_out/${T}_bench_headline_kernel_stats.csv
cut -c1-400 gpurun_out/${T}_bench_headline_under_rocprofv3.json
env MM_BENCH_DEVICE_MOD=1 MM_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 20 --warmup 5 --scale-base 150000 > gpurun_out/${T}_bench_8_ranks_on_one_device_dry_run.json 2> gpurun_out/${T}_bench8.err; cut -c1-900 gpurun_out/${T}_bench_8_ranks_on_one_device_dry_run.json
L=gpurun_out/${T}_reference_hosts_unmodified_baseline_size.log
for c in float double; do echo "== bin/ref_hosts/$c/RunHardware.exe 16384 16384 16384 hw off"; bin/ref_hosts/$c/RunHardware.exe 16384 16384 16384 hw off 2>&1 | tail -1; done > $L 2>&1
for c in float_blas double_blas; do echo "== bin/ref_hosts/$c/RunHardware.exe 16384 16384 16384 hw on   (the reference's runner with the reference's BLAS oracle)"; bin/ref_hosts/$c/RunHardware.exe 16384 16384 16384 hw on 2>&1 | tail -6; done >> $L 2>&1
echo "== MM_HALF_CONTRACT=reference bin/ref_hosts/half/RunHardware.exe 513 544 544 hw on" >> $L; MM_HALF_CONTRACT=reference bin/ref_hosts/half/RunHardware.exe 513 544 544 hw on 2>&1 | tail -4 >> $L
echo "== bin/ref_hosts/half_reference_contract/RunHardware.exe 32768 32768 32768 hw off" >> $L; bin/ref_hosts/half_reference_contract/RunHardware.exe 32768 32768 32768 hw off 2>&1 | tail -1 >> $L
echo "== bin/ref_hosts/float/RunHardware.exe 513 528 528 hw on" >> $L; bin/ref_hosts/float/RunHardware.exe 513 528 528 hw on >> $L 2>&1
echo "== bin/ref_hosts/float/TestSimulation.exe 513 528 528" >> $L; bin/ref_hosts/float/TestSimulation.exe 513 528 528 2>&1 | tail -2 >> $L
cat $L
