#!/usr/bin/env bash
# round 5, second batch: the whole GPU suite again (after the K x N test fix), the bench line with live counters, and the
# rocprofv3 kernel stats of the same bench command
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05b_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05b_pytest_gpu.log
tail -4 gpurun_out/r05b_pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r05b_bench.json; echo; wc -c gpurun_out/r05b_bench.json; tail -3 gpurun_out/r05b_bench.err
export TMPDIR=/tmp
R=$PWD
(cd /tmp && MM_BENCH_NO_PMC=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05b_prof -o bench --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r05b_bench_under_rocprof.json 2> $R/gpurun_out/r05b_rocprof.err)
find gpurun_out/r05b_prof -name "*kernel_stats.csv" | head -2
head -8 $(find gpurun_out/r05b_prof -name "*kernel_stats.csv" | head -1)
