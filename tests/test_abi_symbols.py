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
ORACLE_LIB = os.path.join(ROOT, "oracle", "libgjx_oracle.so")


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


def test_a_library_of_another_version_is_refused(monkeypatch):
    """struct layouts and the sampling spec change from minor to minor behind unchanged entry points, so
    the bindings refuse a library whose gjx_version differs from the one they were written for — the product library and
    the oracle alike (tools/ab_lib.sh loads other builds through GJX_HIP_LIB)."""
    import re

    from genjax._amd import abi

    hdr = open(os.path.join(ROOT, "include", "gjx.h")).read()
    major = int(re.search(r"#define GJX_VERSION_MAJOR (\d+)", hdr).group(1))
    minor = int(re.search(r"#define GJX_VERSION_MINOR (\d+)", hdr).group(1))
    assert abi.ABI_VERSION == (major, minor)
    for path in (HIP_LIB, ORACLE_LIB):
        assert abi.GjxLib(path, "cpu").version == abi.ABI_VERSION
    monkeypatch.setattr(abi, "ABI_VERSION", (major, minor - 1))
    for path in (HIP_LIB, ORACLE_LIB):
        with pytest.raises(abi.AbiVersionMismatch):
            abi.GjxLib(path, "cpu")
