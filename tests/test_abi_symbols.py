"""The C-ABI library loads and exports every symbol include/gjx.h declares (no compute calls, so
this runs without a GPU), the ctypes table covers the header, and the product has no CPU fallback."""

import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "gjx.h")
HIP_LIB = os.path.join(ROOT, "genjax-chi_amd", "lib", "libgjx_hip.so")


def header_symbols():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(gjx_[a-z0-9_]+)\s*\(", txt))


def test_ctypes_table_matches_header():
    from genjax._amd import abi

    assert header_symbols() == set(abi.PROTOTYPES)


def test_hip_library_exports_every_symbol():
    if not os.path.exists(HIP_LIB):
        import __graft_entry__ as g

        g.build()
    dll = ctypes.CDLL(HIP_LIB)
    for name in header_symbols():
        assert hasattr(dll, name), name
    dll.gjx_backend_name.restype = ctypes.c_char_p
    assert dll.gjx_backend_name() == b"hip-gfx950"
    out = subprocess.run(["nm", "-D", "--defined-only", HIP_LIB], capture_output=True, text=True).stdout
    assert "gjo_" not in out, "oracle-only probes must not leak into the product library"


def test_oracle_exports_every_symbol(oracle_ops):
    for name in header_symbols():
        assert hasattr(oracle_ops.lib._dll, name), name
    assert oracle_ops.lib.name == "oracle-cpu"


def test_no_cpu_fallback_without_gpu():
    import torch

    from genjax._amd import runtime

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(runtime.BackendUnavailable):
        runtime.load_hip_ops()
    import genjax

    @genjax.gen
    def m():
        return genjax.normal(0.0, 1.0) @ "x"

    with pytest.raises(runtime.BackendUnavailable):
        m.simulate(genjax.random.key(0), ())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "genjax-chi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "libgjx_oracle" not in src and "oracle/" not in src.replace("oracle/libgjx", "X"), f
