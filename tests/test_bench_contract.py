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


def test_replayed_counters_only_come_from_files_that_name_the_dispatched_kernel(tmp_path, monkeypatch):
    """the traffic figure used to be picked by file-name sort, whatever kernel the file had profiled.
    Now a committed PMC file is replayed only if its `kernel_name` and shape are the run's."""
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump({"kernel_name": "mfma_f32_256x256x16_w8_flush4096", "shape": [16384] * 3, "hbm_bytes_per_launch": 37.6e9},
              open(prof / "r09_traffic_other_kernel.json", "w"))
    json.dump({"kernel_name": "mfma_f32_128x256x16_w4x2_flush4096", "shape": [16384] * 3, "hbm_bytes_per_launch": 51.8e9},
              open(prof / "r03_traffic.json", "w"))
    json.dump({"kernel_name": "mfma_f32_128x256x16_w4x2_flush4096", "size": 16384, "MfmaUtil_pct": 97.8,
               "effective_clock_GHz_profiled": 2.37, "L2_hit_rate": 0.5}, open(prof / "r03_pmc.json", "w"))
    json.dump({"kernel_name": "mfma_f32_128x256x16_w4x2_flush4096", "shape": [8192] * 3, "hbm_bytes_per_launch": 1.0},
              open(prof / "r10_traffic_other_shape.json", "w"))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    rc = b.replayed_counters("mfma_f32_128x256x16_w4x2_flush4096", (16384, 16384, 16384))
    assert rc["traffic"] == 51.8e9 and rc["traffic_source"] == "profiles/r03_traffic.json"
    assert rc["mfma_util_pct"] == 97.8 and rc["mfma_util_source"] == "profiles/r03_pmc.json"
    assert b.replayed_counters("mfma_f32_128x128x32_w4x2_flush4096", (16384, 16384, 16384)) == {}
    rl = b.attach_replayed({"avg_launch_ms": 58.0, "traffic": None}, "some_other_kernel", (16384, 16384, 16384))
    assert rl["traffic"] is None and "not replayed from another kernel" in rl["traffic_note"]
    rl = b.attach_replayed({"avg_launch_ms": 58.0, "traffic": None}, "mfma_f32_128x256x16_w4x2_flush4096", (16384,) * 3)
    assert abs(rl["achieved_fabric_GBps"] - 51.8e9 / 58.0e-3 / 1e9) < 0.1 and rl["counters_measured_in_this_run"] is False


def test_replayed_counters_of_the_generic_semiring_kernel_are_matched_by_data_type(tmp_path, monkeypatch):
    """`valu_tile` names the kernel of EVERY Data_t; the double (Add, Min) workload used to replay the
    float file.  A file that says which type it profiled speaks for that type only; older files (no field) were float."""
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump({"kernel_name": "valu_tile", "shape": [8192] * 3, "hbm_bytes_per_launch": 5.0e9}, open(prof / "r03_traffic_minplus.json", "w"))
    json.dump({"kernel_name": "valu_tile", "dtype": "double", "shape": [8192] * 3, "hbm_bytes_per_launch": 13.5e9},
              open(prof / "r04_traffic_minplus_f64.json", "w"))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    assert b.replayed_counters("valu_tile", (8192,) * 3, "float")["traffic"] == 5.0e9
    assert b.replayed_counters("valu_tile", (8192,) * 3, "double")["traffic"] == 13.5e9
    assert b.replayed_counters("valu_tile", (8192,) * 3, "int") == {}


def test_committed_traffic_files_are_self_consistent():
    files = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(".json") and "traffic" in f]
    assert files, "profiles/*traffic*.json missing"
    for f in files:
        r = json.load(open(os.path.join(ROOT, "profiles", f)))
        assert r["hbm_bytes_per_launch"] > r["algorithmic_bytes_compulsory"]   # never below the compulsory bytes
        if r.get("workload", "float") == "float" and "split" not in f:         # C written a whole number of times (flushes)
            assert abs(r["write_bytes"] / r["c_bytes_exact"] - round(r["write_bytes"] / r["c_bytes_exact"])) < 1e-6, f


def test_constants_match_baseline():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert str(b.SIZE) in base["metric"] and b.WORKLOADS["float"][5] == 157.3
    # the strong-scaling job is BASELINE configs[4]'s: float 65536 x 16384 x 16384 split along N
    assert f"{b.C5A_ROWS}" in base["configs"][4] and b.WORKLOADS["float"][3] == b.SIZE
    # every other single-GPU BASELINE config has a workloads[] entry at its size
    assert b.WORKLOADS["half"][3] == 32768 and "32768" in base["configs"][2]
    assert b.WORKLOADS["double"][3] == 16384 and "double" in base["configs"][3]
    assert b.WORKLOADS["minplus"][3] == 8192 and "8192" in base["configs"][4]


def test_default_mode_for_more_than_one_gpu_is_baselines_fixed_job():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="strong"' in src and "C5A_ROWS if headline" in src


def test_strong_split_of_the_baseline_job_is_tile_aligned_and_balanced():
    import gemm_hls_amd as g
    from gemm_hls_amd.partition import row_slab_for
    b = _bench()
    for world in (1, 2, 4, 8):
        slabs = [row_slab_for(g.make_config("float"), b.C5A_ROWS, b.SIZE, b.SIZE, world, r) for r in range(world)]
        assert sum(rows for _, rows in slabs) == b.C5A_ROWS
        assert all(rows == b.C5A_ROWS // world and row0 % 256 == 0 for row0, rows in slabs)


def test_plain_launch_with_more_than_one_gpu_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus 8` outside torch.distributed.run must not stop with a usage message: it
    re-runs itself under the launcher (one rank per GPU, 127.0.0.1, a free port) and returns its status."""
    import subprocess
    import sys
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert b.main() == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_plain_launch_reaches_the_ranks_without_a_gpu():
    """End to end on this CPU box: the self-launched ranks start and each one stops at the product's
    `needs an MI355X` (there is no CPU path) -- not at a launcher usage message."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_gpu_capi.py")
    assert r.returncode != 0 and "needs an MI355X" in r.stderr and "torch.distributed.run" in r.stderr


def test_baseline_summary_is_compact_and_last():
    """VERDICT r4 missing 4: the driver's record keeps the TAIL of the line; the BASELINE configs must be there, short."""
    b = _bench()
    rl = {"frac": 0.95}
    out = {"value": 150000.0, "ms_per_step": 58.5, "roofline": rl, "config": {"kernel": "mfma_f32_256x256x16_w8_flush4096"},
           "workloads": [{"key": k, "workload": k, "value": 1.0e6, "ms_per_step": 47.123456, "roofline": {"frac": 0.6}, "kernel": "mfma_f16_256x256_pingpong_16x16x32"}
                         for k in ("half", "double", "minplus", "minplus_f64", "uint8", "float", "float_split", "half_kxn")] + [{"workload": "uint8_kxn", "error": "x" * 500}],
           "cpu_baseline": {"value": 0.28, "unit": "GFLOP/s", "cores": 39, "host_cores": 256, "seconds": 7.6}}
    summary = b.baseline_summary(out)
    assert {"C2_float_16384", "C3_half_32768", "C4_double_16384", "C5b_minplus_8192", "C5a_float_65536_rows_1gpu",
            "C1_float_1024_ref_cpu_sim"} == set(summary) - {"unit"}
    assert summary["C3_half_32768"]["frac"] == 0.6 and summary["C2_float_16384"]["frac"] == 0.95
    assert len(json.dumps(summary)) < 900
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('out["baseline_summary"] = baseline_summary(out)') > src.index('out["context"]') > src.index('out["cpu_baseline"] = cpu_baseline()')


def test_live_counters_are_skipped_quietly_where_they_cannot_run(monkeypatch):
    b = _bench()
    monkeypatch.setenv("MM_BENCH_NO_PMC", "1")
    assert b.live_counters(b.LIVE_KERNEL, (16384,) * 3, 58.0) is None
    monkeypatch.delenv("MM_BENCH_NO_PMC")
    assert b.live_counters("mfma_f32_128x256x16_w4x2_flush4096", (16384,) * 3, 58.0) is None      # not the kernel the passes pin
    assert b.live_counters(b.LIVE_KERNEL, (8192,) * 3, 58.0) is None
    assert {"float", "half", "double", "minplus"} <= set(b.LIVE)      # the headline + BASELINE C3 / C4 / C5b (+ the extras)
    assert b.live_counters("mfma_f16_256x256_pingpong_16x16x32", (16384,) * 3, 47.0, "half") is None   # not the BASELINE size


def test_live_counter_passes_share_one_budget_and_stop_at_the_first_failure(monkeypatch):
    """A hung or unsupported rocprofv3 must not hold the bench line up: every pass is bounded, all passes of a run share one
    wall-clock budget, and the first failure ends them (the replayed figures stay)."""
    import shutil
    import subprocess
    b = _bench()
    monkeypatch.delenv("MM_BENCH_NO_PMC", raising=False)
    monkeypatch.setattr(shutil, "which", lambda name: "/usr/bin/" + name)
    calls = []

    def fake_run(cmd, **kw):
        calls.append(kw.get("timeout"))
        raise subprocess.TimeoutExpired(cmd, kw.get("timeout"))
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert b.live_counters(b.LIVE_KERNEL, (16384,) * 3, 58.0) is None
    assert len(calls) == 1 and 0 < calls[0] <= 75
    assert b.live_counters("mfma_f16_256x256_pingpong_16x16x32", (32768,) * 3, 47.0, "half") is None
    assert len(calls) == 1                              # no further passes after the first failure
    assert b.LIVE_BUDGET_S <= 180


def test_hbm_busy_probe_samples_sysfs_while_launching(tmp_path):
    """bench.py's DRAM-side probe (HbmBusy): the mean of amdgpu's mem_busy_percent, sampled by a thread while the launch callable is
    called back to back, the first 0.3 s discarded; attach() turns it into achieved_HBM_GBps with the run's calibration factor and
    leaves the roofline alone where sysfs has no such file."""
    b = _bench()

    class _Cuda:
        @staticmethod
        def synchronize():
            pass

    class _Torch:
        cuda = _Cuda()

    sysfs = tmp_path / "mem_busy_percent"
    sysfs.write_text("19\n")
    probe = object.__new__(b.HbmBusy)
    probe.torch, probe.dev, probe.path, probe.k, probe.cal = _Torch(), None, str(sysfs), 82.0, {"copy_4GiB": {}, "fill_4GiB": {}}
    calls = []
    pct, n, wall = probe.busy(lambda: calls.append(1), seconds=0.5, depth=4)
    assert pct == 19 and n == len(calls) and n % 4 == 0 and wall >= 0.5
    rl = {"avg_launch_ms": 50.0}
    probe.busy = lambda launch, seconds=1.0, depth=4: (19.0, 4, 1.0)   # noqa: E731
    probe.attach(rl, lambda: None)
    assert rl == {"avg_launch_ms": 50.0, "hbm_busy_pct": 19.0, "achieved_HBM_GBps": 1558.0, "achieved_HBM_frac_of_8TBps": round(1558.0 / 8000.0, 4),
                  "hbm_bytes_per_launch": 77900000000}
    probe.path = None
    rl2 = {}
    probe.attach(rl2, lambda: None)
    assert rl2 == {}
    assert b.HbmBusy._read(str(tmp_path / "absent")) is None
