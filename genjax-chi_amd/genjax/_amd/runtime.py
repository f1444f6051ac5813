"""Backend selection: the product has exactly one compute backend, libgjx_hip.so on a gfx950 GPU.

There is NO CPU fallback: if the HIP library is missing, fails to load, or no GPU is visible, the
first compute call raises `BackendUnavailable`.  (Tests may install another `Ops` — e.g. one bound
to the CPU oracle — with `use_ops()`; nothing in the package does.)
"""

from __future__ import annotations

import contextlib
import os

from .abi import GjxLib
from .ops import Ops

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("GJX_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libgjx_hip.so"))

_current: Ops | None = None


class BackendUnavailable(RuntimeError):
    pass


def load_hip_ops() -> Ops:
    import torch

    if not os.path.exists(HIP_LIB_PATH):
        raise BackendUnavailable(
            f"{HIP_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    if not torch.cuda.is_available():
        raise BackendUnavailable("no ROCm GPU visible to torch; the genjax AMD backend needs an MI355X (gfx950)")
    return Ops(GjxLib(HIP_LIB_PATH, "cuda"))


def get_ops() -> Ops:
    global _current
    if _current is None:
        _current = load_hip_ops()
    return _current


@contextlib.contextmanager
def use_ops(ops: Ops):
    """Temporarily install `ops` as the backend (test hook)."""
    global _current
    prev, _current = _current, ops
    try:
        yield ops
    finally:
        _current = prev


def compiler_switched_off(err) -> bool:
    """A fused plan that the library refuses with GJX_ERR_UNSUPPORTED while the plan compiler is switched off (GJX_PLAN_JIT=0):
    bodies with arithmetic between their sites (GJX_ARG_EXPR programs) or nested `@gen` calls run as generated kernels only —
    the table interpreters do not evaluate programs or keep a key stack — so with the compiler off such a body takes the
    PER-SITE path (one kernel per `@` site: the same bits, several launches), the documented route without a compiler.  A
    compiler that is ON and fails (GJX_ERR_JIT) stays an error: never a silent slower route."""
    return getattr(err, "code", None) == -2 and os.environ.get("GJX_PLAN_JIT") == "0"


# ---- opt-in fast math for fused importance / scan plans (gjx.h GJX_PLAN_FAST_MATH) -------------------------------------
_fast_math = False


def fast_math_enabled() -> bool:
    return _fast_math


@contextlib.contextmanager
def fast_math(enabled: bool = True):
    """`with genjax.fast_math(): ...` — fused `@gen` / `Scan` plans created inside use the hardware transcendentals for the
    CONTINUOUS parts of the arithmetic (Box-Muller, log / exp inside log-densities): log-weights within 1e-5 relative of
    the bit-exact mode, ~1.4x faster on the 10-latent benchmark model.  Bit-exact parity with the oracle holds only
    outside this context; resampling / SMC kernels are never affected."""
    global _fast_math
    prev, _fast_math = _fast_math, bool(enabled)
    try:
        yield
    finally:
        _fast_math = prev
