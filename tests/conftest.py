import importlib
import os
import sys

import pytest

try:                      # torch BEFORE the HIP library: torch brings its own copy of the HIP runtime, and a process that has initialised the
    import torch          # system's copy first (tdv_ctx_create) makes torch's report "No HIP GPUs are available" - whatever the test order
except ImportError:       # (a test file that only imported torch inside a function failed when it was run on its own)
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "study: exercises an A/B variant that only the -DTDV_STUDY library holds (lib3dvision_hip_study.so, loaded with "
                                       "TDV_LIB_VARIANT=study); skipped on the product library, run by tests/test_gpu_study_build.py in a process of its own")


def pytest_collection_modifyitems(config, items):
    """Tests marked `study` need the study library: skip them unless this process loaded it."""
    if os.environ.get("TDV_LIB_VARIANT") == "study":
        return
    skip = pytest.mark.skip(reason="needs lib3dvision_hip_study.so (TDV_LIB_VARIANT=study): run by tests/test_gpu_study_build.py")
    for item in items:
        if "study" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def tdv():
    """The product package (directory name starts with a digit -> importlib)."""
    return importlib.import_module("3dvision_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("3dvision_amd.synth")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx(tdv):
    """One backend context on GPU 0.  Fails loudly (no fallback) when the library or GPU is missing."""
    assert os.path.exists(tdv.LIB_PATH), "lib3dvision_hip.so not built; run __graft_entry__.build()"
    assert tdv.device_count() > 0, "no HIP device visible"
    c = tdv.Context(0)
    yield c
    c.close()
