import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree libraries normally arrive prebuilt (__graft_entry__.build()); if they did not, build them once
    lib = os.path.join(ROOT, "zebra_amd", "lib", "libzebra_hip.so")
    ora = os.path.join(ROOT, "oracle", "libzebra_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        try:
            sys.path.insert(0, ROOT)
            import __graft_entry__
            __graft_entry__.build()
        except Exception as e:  # noqa: BLE001 -- the tests that need the libraries will say what is missing
            print("conftest: could not build the libraries:", e, file=sys.stderr)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
