#!/usr/bin/env bash
# round 5: the reference's benchmark loop with power on this round's kernels, and the mid-size fp32 table back to back
# next to the vendor library (stream-K is now one wait-free kernel)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
timeout 900 python tools/benchmark.py --repetitions 3 --out gpurun_out/r05g_benchmark_driver_with_power.csv > gpurun_out/r05g_benchmark.log 2>&1; echo "benchmark rc=$?"; cat gpurun_out/r05g_benchmark_driver_with_power.csv | cut -c1-200
timeout 900 python tools/throughput_b2b.py 1024,1536,2048,2304,2560,3072,3584,4096,5120,6144,7680,8192 > gpurun_out/r05g_f32_sizes_back_to_back_vs_library.txt 2>&1; echo "b2b rc=$?"; cat gpurun_out/r05g_f32_sizes_back_to_back_vs_library.txt | cut -c1-260
