import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ORACLE_LIB = os.environ.get("GJX_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libgjx_oracle.so")  # (override: the sanitizer build)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_ops():
    """Ops bound to the CPU oracle (test infrastructure; host pointers, CPU tensors)."""
    from genjax._amd.abi import GjxLib
    from genjax._amd.ops import Ops

    if not os.path.exists(ORACLE_LIB):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return Ops(GjxLib(ORACLE_LIB, "cpu"))


@pytest.fixture(scope="session")
def hip_ops():
    """Ops bound to libgjx_hip.so on cuda:0 — the product path.  Fails (does not skip) if the
    library is missing, so a GPU run can never silently pass on a fallback."""
    from genjax._amd.runtime import load_hip_ops

    return load_hip_ops()
