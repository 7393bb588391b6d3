"""CPU checks of bench.py's side of the measurement contract: the CPU-baseline leg (the
reference's own simulation path through oracle/_ref) produces the required object, the traffic
figure comes from a committed PMC profile, and the module imports without a GPU."""
import importlib.util
import json
import os

import pytest

import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_object_shape():
    b = _bench()
    info = b.cpu_baseline(sample_n=256)  # small sample here; bench.py uses 1024 (BASELINE C1)
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(info)
    assert info["kind"] == "reference" and info["unit"] == "GFLOP/s" and info["cores"] >= 1
    if _oracle.ref_available():
        assert info["value"] and info["value"] > 0


def test_traffic_comes_from_committed_profile():
    b = _bench()
    t = b.hbm_traffic_per_launch()
    assert t is None or t > 3.2e9  # never below the compulsory bytes of fp32 16384^3
    files = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")]
    assert files, "profiles/*_traffic.json missing"
    r = json.load(open(os.path.join(ROOT, "profiles", sorted(files)[-1])))
    assert abs(r["write_bytes"] / r["c_bytes_exact"] - round(r["write_bytes"] / r["c_bytes_exact"])) < 1e-6


def test_constants_match_baseline():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert str(b.SIZE) in base["metric"] and b.PEAK_TFLOPS_F32_MFMA == 157.3
