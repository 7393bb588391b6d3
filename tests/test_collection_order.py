"""The driver runs `pytest -m gpu -x`: one failure blanks everything collected after it (round 3 lost 495 tests that way).
tests/conftest.py therefore orders the collection -- SURVEY.md section-8 rows first, self-imposed guards and the one test
whose failure mode is a hung GPU last.  This (CPU) test pins that order, so that a new test file or a renamed test
cannot silently move a guard in front of the parity rows."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _collected():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q"], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if "::" in ln]


def test_section_8_rows_first_guards_and_the_hang_risk_last():
    ids = _collected()
    assert len(ids) > 600

    def first(pattern):
        return next(i for i, t in enumerate(ids) if pattern in t)

    def last(pattern):
        return max(i for i, t in enumerate(ids) if pattern in t)

    # the rows, in the order the verdict asked for
    order = ["test_against_reference_kernel_golden_outputs", "test_ordered_path_is_bit_identical_to_naive",
             "test_reference_entry_point_symbol", "test_c1_float_1024_device_next_to_the_reference_kernel_itself",
             "test_f32_mfma_default_vs_blas_and_exact", "test_double_16384_sampled_rows_and_properties",
             "test_run_hardware_verifies_on_gpu", "test_benchmark_driver_rows_follow_the_reference_contract",
             "tests/test_gpu_capi.py::test_host_pointer_entry_with_explicit_config"]
    pos = [first(p) for p in order]
    assert pos == sorted(pos) and pos[0] == 0, list(zip(order, pos))
    # every row test is collected before any stress / fuzz / soak test, those before the floors, the floors before the masked-stream test
    rows_end = max(last(p) for p in order)
    stress_begin = min(first(p) for p in ("test_f32_stream_k_", "tests/test_gpu_fuzz.py", "test_race_screen_", "tests/test_gpu_streamk_stress.py::test_poisoned"))
    assert rows_end < stress_begin
    assert last("tests/test_gpu_fuzz.py") < first("test_throughput_floor_") and last("test_f32_stream_k_") < first("test_throughput_floor_")
    assert last("test_throughput_floor_") < first("test_cu_masked_stream_") == len(ids) - 1
