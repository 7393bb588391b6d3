#!/usr/bin/env bash
# round 5: fuzz and soak of what changed -- fp32 dispatch (stream-K now two kernels; hand-over form bit-compared on every
# stream-K shape) with poisoned scratch, the race screen, and the multi-device split over virtual devices
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
MM_DEBUG_POISON=1 timeout 900 python tools/fuzz_f32_dispatch.py --shapes 1200 --seed 5 > gpurun_out/r05e_fuzz_f32_dispatch_poisoned.txt 2>&1; echo "fuzz rc=$?"; tail -14 gpurun_out/r05e_fuzz_f32_dispatch_poisoned.txt
timeout 900 python tools/fuzz_multi_device.py --cases 300 --seed 5 > gpurun_out/r05e_fuzz_multi_device_virtual.txt 2>&1; echo "md fuzz rc=$?"; tail -25 gpurun_out/r05e_fuzz_multi_device_virtual.txt
MM_DEBUG_POISON=1 timeout 900 python tools/soak.py > gpurun_out/r05e_soak_poisoned.txt 2>&1; echo "soak rc=$?"; tail -12 gpurun_out/r05e_soak_poisoned.txt
python -m pytest tests/test_gpu_multi_device.py -q -k transposed 2>&1 | tail -3
