mkdir -p gpurun_out
timeout 300 python tools/pmc_mfma.py f32 --size 16384 --out gpurun_out/r04d_pmc_f32_16384.json > /dev/null 2> gpurun_out/r04d_pmc_err.txt
timeout 300 python tools/pmc_traffic.py --what f32 --size 16384 --out gpurun_out/r04d_traffic_f32_16384.json > /dev/null 2>> gpurun_out/r04d_pmc_err.txt
cp gpurun_out/r04d_pmc_f32_16384.json gpurun_out/r04d_traffic_f32_16384.json profiles/
tail -3 gpurun_out/r04d_pmc_err.txt
echo "HEAD $1" > gpurun_out/r04d_pytest_gpu.log
MM_PERF_FLOORS=1 timeout 1200 python -m pytest tests -m gpu -x -q -rs --timeout 600 2>&1 | tail -40 >> gpurun_out/r04d_pytest_gpu.log
tail -8 gpurun_out/r04d_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r04d_bench.json 2> gpurun_out/r04d_bench.err; tail -2 gpurun_out/r04d_bench.err
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04d_prof -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04d_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04d_rocprof.err
cd $GRAFT_REPO_ROOT; ls gpurun_out/r04d_prof | head; find gpurun_out/r04d_prof -name "*kernel_stats.csv" | head -2
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('mfma_util_pct'))
for w in d['workloads']: print(w.get('workload','')[:60], w.get('value'), w.get('roofline',{}).get('frac'), w.get('kernel'), w.get('error'))
print(d.get('scale_base'))
PY
