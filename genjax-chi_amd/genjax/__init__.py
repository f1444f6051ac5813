"""MI355X-native backend for the GenJAX SMC / ImportanceK hot path.

Public names mirror the `genjax` package of genjax-dev/genjax-chi for that path
(src/genjax/__init__.py:35-41 star-exports): `gen`, the distributions, `ChoiceMap` /
`ChoiceMapBuilder` / `Selection` / `SelectionBuilder`, `Target`, and `genjax.inference.smc`.
Compute happens in hand-written HIP kernels behind the C-ABI of include/gjx.h; there is no CPU
fallback (genjax._amd.runtime)."""

from ._amd.choicemap import (ChoiceMap, ChoiceMapBuilder as _CMB, ChoiceMapNoValueAtAddress, Mask, Selection,
                             SelectionBuilder, C as _C)
from ._amd.lang import (AddressReuse, Distribution, GenerativeFunction, GenerativeFunctionClosure, MissingAddress,
                        StaticGenerativeFunction, Trace, bernoulli, beta, categorical, exact_density, flip, gamma,
                        gen, normal, uniform)
from ._amd.combinators import Scan, Vmap, scan, vmap
from ._amd.edit import (Diff, EditRequest, EmptyRequest, IndexRequest, NoChange, NotSupportedEditRequest, Regenerate, Rejuvenate, StaticRequest,
                        UnknownChange, Update)
from ._amd.inference import Algorithm, Marginal, SampleDistribution, Target, marginal
from ._amd import prng as _prng
from ._amd.lang import split as _split, fold_in as _fold_in
from ._amd import jaxlike
from ._amd.runtime import fast_math
from . import inference

ChoiceMapBuilder = _C  # `from genjax import ChoiceMapBuilder as C`; C["x"].set(v)
Pytree = object


class _Random:
    """`genjax.random`: the counter-based PRNG keys this backend consumes (jax.random-compatible
    derivation for impl="threefry"; native Philox4x32-10 for impl="philox")."""

    key = staticmethod(_prng.key)
    PRNGKey = staticmethod(_prng.key)
    split = staticmethod(_split)
    fold_in = staticmethod(_fold_in)
    set_default_impl = staticmethod(_prng.set_default_impl)


random = _Random()

__all__ = [
    "AddressReuse", "Algorithm", "ChoiceMap", "ChoiceMapBuilder", "ChoiceMapNoValueAtAddress", "Diff", "Distribution",
    "EditRequest", "EmptyRequest", "IndexRequest", "NoChange", "NotSupportedEditRequest", "Regenerate", "Rejuvenate", "StaticRequest", "UnknownChange", "Update", "GenerativeFunction", "GenerativeFunctionClosure", "Marginal", "Mask", "MissingAddress", "SampleDistribution", "Scan",
    "Selection", "SelectionBuilder", "StaticGenerativeFunction", "Target", "Trace", "bernoulli", "beta",
    "categorical", "exact_density", "fast_math", "flip", "gamma", "gen", "inference", "jaxlike", "marginal", "normal", "random",
    "scan", "uniform", "Vmap", "vmap",
]
__version__ = "0.1.0"
