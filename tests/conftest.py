import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # a fresh checkout has no built library yet (it is git-ignored): build it once, where hipcc exists
    lib = os.path.join(ROOT, "detr_amd", "libdetr_match.so")
    helper = os.path.join(ROOT, "detr_amd", "_index_views.so")
    if (not os.path.exists(lib) or not os.path.exists(helper)) and not os.environ.get("DETR_AMD_LIB"):
        import shutil

        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            import build_ext

            build_ext.build_library()


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
