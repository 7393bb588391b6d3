import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "guard: a self-imposed regression bound, tighter than the north_star bar")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    _order(items)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """Tests never build implicitly EXCEPT when the product library is absent altogether (fresh
    checkout): then build it once, in-tree, exactly as __graft_entry__.build() would."""
    lib = os.path.join(ROOT, "gemm_hls_amd", "libmm_gemm_amd.so")
    exe = os.path.join(ROOT, "bin", "RunHardware.exe")
    if not (os.path.exists(lib) and os.path.exists(exe)):
        from gemm_hls_amd import build
        build.build(verbose=False)


# ---- collection order ---------------------------------------------------------------------------------------------
# The driver runs `pytest -m gpu -x`: one failure blanks everything collected after it.  So the tests that ARE the
# SURVEY.md section-8 rows run first (golden fixtures of the reference's own kernel, ordered path == Naive, the
# MatrixMultiplicationKernel symbol, parity at BASELINE.json's sizes, RunHardware / TestSimulation /
# PrintSpecifications), then the C-ABI contract tests, then the rest of the parity matrix; shape-dispatch stress, fuzz
# and soak after those; self-imposed regression guards (throughput floors) last of all.
_TIERS = [
    # tier 0: section-8 rows
    (0, ("test_against_reference_kernel_golden_outputs", "test_ordered_path_", "test_ordered_transposed_a_layout", "test_ordered_tile_",
         "test_half_reference_contract_",
         "test_reference_entry_point_", "test_kernel_shims_export_the_reference_symbol", "test_static_kernel_shim_",
         "test_c1_float_1024_device_next_to_the_reference_kernel_itself", "test_f32_mfma_default_vs_blas_and_exact",
         "test_f64_mfma_vs_blas", "test_f16_mfma_wide_accumulate_contract", "test_i8_mfma_is_bit_exact_mod_256",
         "test_auto_path_exact_semirings", "test_f32_mfma_transposed_a_layout", "test_f64_f16_i8_mfma_transposed_a_layout")),
    (1, ("test_minplus_8192_", "test_double_16384_", "test_half_32768_", "test_uint8_32768_", "test_f32_full_size_properties",
         "test_f32_baseline_c5a_shape_", "test_f32_mixed_sign_full_size_sampled", "test_split_full_size_sampled_rows",
         "test_split_c5a_shape_")),
    (2, ("tests/test_gpu_ref_hosts.py", "tests/test_run_hardware_cli.py", "tests/test_gpu_benchmark_driver.py")),
    (3, ("test_multi_device_", "tests/test_gpu_capi.py")),
    # tier 5 (default): the rest of the parity matrix
    # tier 7: shape-dispatch stress, stream-K stress, fuzz, soak
    (7, ("test_f32_small_problems_take_the_64x64", "test_f32_64x64_geometry_", "test_f32_split_k_for_small_problems",
         "test_f32_stream_k_", "tests/test_gpu_fuzz.py", "test_race_screen_", "tests/test_gpu_streamk_stress.py")),
    # tier 9: self-imposed guards
    (9, ("test_throughput_floor_",)),
    # tier 10: the one test whose failure mode is a hung GPU (run in a child process under a watchdog; still: nothing after it)
    (10, ("test_cu_masked_stream_",)),
]
# a tier-7/9 pattern wins over a file-level pattern of an earlier tier (the soak lives in test_gpu_capi.py)
_LATE_FIRST = sorted(_TIERS, key=lambda t: -t[0])


def _tier(nodeid):
    """(tier, position of the matching pattern inside its tier): within a tier the patterns' own order is kept."""
    for tier, pats in _LATE_FIRST:
        if tier >= 7:
            for i, p in enumerate(pats):
                if p in nodeid:
                    return (tier, i)
    for tier, pats in _TIERS:
        for i, p in enumerate(pats):
            if p in nodeid:
                return (tier, i)
    return (5, 0)


def _order(items):
    keyed = sorted(enumerate(items), key=lambda iv: (_tier(iv[1].nodeid), iv[0]))
    items[:] = [it for _, it in keyed]
