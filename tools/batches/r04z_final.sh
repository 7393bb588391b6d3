# the round's closing run: full GPU suite under -x on the commit given as $1, smoke(), the bench line, and the headline-only
# bench under rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out
echo "HEAD $1" > gpurun_out/r04z_pytest_gpu.log
MM_PERF_FLOORS=1 timeout 1500 python -m pytest tests -m gpu -x -q -rs --timeout 600 2>&1 | tail -40 >> gpurun_out/r04z_pytest_gpu.log
tail -6 gpurun_out/r04z_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r04z_smoke.txt
timeout 600 python bench.py > gpurun_out/r04z_bench.json 2> gpurun_out/r04z_bench.err; tail -2 gpurun_out/r04z_bench.err
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04z_prof -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04z_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04z_rocprof.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04z_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('mfma_util_pct'))
for w in d['workloads']: print(w.get('workload','')[:70], w.get('value'), w.get('roofline',{}).get('frac'), w.get('kernel'), w.get('error'))
PY
head -3 gpurun_out/r04z_prof/bench_kernel_stats.csv | cut -c1-80,300-420
