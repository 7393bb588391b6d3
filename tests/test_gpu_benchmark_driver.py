"""the benchmark loop the reference ships as `scripts/build_manager.py benchmark` +
`extract_benchmarks` (:578-669) -- run `RunHardware N K M hw off` per configuration, find
`([\\d\\.]+) seconds[^\\d]+([\\d\\.]+) GOp/s` (:601-602) in its output, write a CSV row.  tools/benchmark.py is this
repo's driver with that contract; here it runs on the GPU over its quick configurations (the BASELINE configs' types and
operators at 8192^3) and every row is checked against the reference's own regex and arithmetic."""
import csv
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_PERF = re.compile(r"([\d\.]+) seconds[^\d]+([\d\.]+) GOp/s")       # scripts/build_manager.py:601-602


def test_benchmark_driver_rows_follow_the_reference_contract(tmp_path):
    out = tmp_path / "benchmark.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "benchmark.py"), "--repetitions", "1", "--configs", "quick",
                        "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "iteration 1 / 1" in r.stdout                      # the reference's progress line (:627-628)
    rows = list(csv.DictReader(open(out)))
    assert len(rows) >= 4
    seen = set()
    for row in rows:
        n, k, m = int(row["size_n"]), int(row["size_k"]), int(row["size_m"])
        t, perf = float(row["time"]), float(row["performance"])
        assert t > 0 and perf > 0 and row["kernel"], row
        # the runner's metric (host/RunHardware.cpp:174-180): 1e-9 * 2 N K M / t, printed with the time it came from
        assert abs(perf - 2e-9 * n * k * m / t) / perf < 0.02, row
        if row["power"]:
            assert float(row["power"]) > 0 and abs(float(row["power_efficiency"]) - perf / float(row["power"])) < 1e-6 * perf
        seen.add((row["data_type"], row["map_op"], row["reduce_op"]))
    assert {("float", "Multiply", "Add"), ("half", "Multiply", "Add"), ("double", "Multiply", "Add"), ("float", "Add", "Min")} <= seen


def test_runner_output_matches_the_reference_regex_directly():
    """The line itself, not the driver's parse of it: what `extract_benchmarks` would find in a benchmark_*.out."""
    r = subprocess.run([os.path.join(ROOT, "bin", "RunHardware.exe"), "1024", "1024", "1024", "hw", "off"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    mt = REFERENCE_PERF.search(r.stdout)
    assert mt and float(mt.group(1)) > 0 and float(mt.group(2)) > 0, r.stdout
