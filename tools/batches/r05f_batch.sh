#!/usr/bin/env bash
# round 5: the last-arriver stream-K form -- parity / stress tests, fuzz with poisoned slots, the forms side by side
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streamk_stress.py tests/test_gpu_capi.py -x -q -m gpu -k "stream_k or streamk or poison or two_host or two_processes or cu_masked or graph" > gpurun_out/r05f_streamk_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05f_streamk_tests.log
MM_DEBUG_POISON=1 timeout 900 python tools/fuzz_f32_dispatch.py --shapes 1200 --seed 7 > gpurun_out/r05f_fuzz_f32_dispatch_poisoned.txt 2>&1; echo "fuzz rc=$?"; head -8 gpurun_out/r05f_fuzz_f32_dispatch_poisoned.txt
timeout 900 python tools/streamk_sweep.py 2304,2560,2944,3072,3584,3840,4608,5120,5888,6656,7168,7680 > gpurun_out/r05f_f32_streamk_forms.txt 2>&1
cat gpurun_out/r05f_f32_streamk_forms.txt
MM_DEBUG_POISON=1 timeout 600 python tools/soak.py 2>&1 | grep -i "stream\|soak" 
