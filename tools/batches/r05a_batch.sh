#!/usr/bin/env bash
# round 5, first full batch: the whole GPU suite (new: virtual-device multi-GPU, the reference's own hosts, stream-K forms),
# the stream-K forms side by side, one bench line
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05a_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a_pytest_gpu.log
tail -5 gpurun_out/r05a_pytest_gpu.log
python tools/streamk_sweep.py 2304,2560,2944,3072,3584,3840,4608,5120,5888,6656,7168,7680 > gpurun_out/r05a_f32_streamk_forms.txt 2>&1
cat gpurun_out/r05a_f32_streamk_forms.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r05a_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["kernel"])
for w in d.get("workloads", []):
    print(w.get("workload", "?")[:60], w.get("value"), w.get("roofline", {}).get("frac"))
P
