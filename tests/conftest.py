import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = os.environ.get("JSMPEG_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "src", "wasm"))


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def libs():
    """Builds (if needed) and returns the paths of the CPU-side libraries."""
    from jsmpeg_amd import build
    out = {"synth": build.build_synth(), "oracle": build.build_oracle(), "ref": build.build_ref()}
    return out


@pytest.fixture(scope="session")
def hip_lib():
    from jsmpeg_amd import build
    if not os.path.exists(build.LIB_HIP):
        build.build_hip()
    return build.LIB_HIP
