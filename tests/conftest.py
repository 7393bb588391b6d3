import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
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
