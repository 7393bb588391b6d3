mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "wide_kxn or valu_tile or fuzz_valu or benchmark_driver or minplus_8192 or max_reduce or auto_path_exact" 2>&1 | tail -15 > gpurun_out/r04b_pytest_subset.log
tail -5 gpurun_out/r04b_pytest_subset.log
timeout 300 python tools/kxn_prepass_check.py > gpurun_out/r04b_kxn_prepass.txt 2>&1; cat gpurun_out/r04b_kxn_prepass.txt
timeout 200 python tools/sweep.py minplus_f64 --sizes 8192 --variants 2,-1,2,-1 --reps 5 > gpurun_out/r04b_minplus_f64_rows_per_thread.txt 2>&1; cat gpurun_out/r04b_minplus_f64_rows_per_thread.txt
timeout 300 python tools/f32_energy.py > gpurun_out/r04b_f32_energy_33_vs_8.txt 2>&1; tail -4 gpurun_out/r04b_f32_energy_33_vs_8.txt
timeout 300 python tools/pmc_mfma.py f16 --size 32768 --out gpurun_out/r04b_pmc_f16_32768.json > /dev/null 2> gpurun_out/r04b_pmc_err.txt
timeout 300 python tools/pmc_traffic.py --what f16 --size 32768 --out gpurun_out/r04b_traffic_f16_32768.json > /dev/null 2>> gpurun_out/r04b_pmc_err.txt
timeout 300 python tools/pmc_mfma.py uint8 --size 32768 --out gpurun_out/r04b_pmc_uint8_32768.json > /dev/null 2>> gpurun_out/r04b_pmc_err.txt
timeout 300 python tools/pmc_traffic.py --what uint8 --size 32768 --out gpurun_out/r04b_traffic_uint8_32768.json > /dev/null 2>> gpurun_out/r04b_pmc_err.txt
timeout 300 python tools/pmc_mfma.py minplus_f64 --size 8192 --out gpurun_out/r04b_pmc_minplus_f64_8192.json > /dev/null 2>> gpurun_out/r04b_pmc_err.txt
timeout 300 python tools/pmc_traffic.py --what minplus_f64 --size 8192 --out gpurun_out/r04b_traffic_minplus_f64_8192.json > /dev/null 2>> gpurun_out/r04b_pmc_err.txt
tail -5 gpurun_out/r04b_pmc_err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04b_*.json')):
    d=json.load(open(f)); print(f, d.get('kernel_name'), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('MfmaUtil_pct','effective_clock_GHz_profiled','L2_hit_rate','hbm_bytes_per_launch','SQ_WAIT_INST_LDS_over_WAVE_CYCLES','SQ_WAIT_ANY_over_WAVE_CYCLES')})
PY
