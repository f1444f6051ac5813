"""The generative-function interface on vectorised traces.

Mirror of the reference interface for the importance / SMC path (same names, argument meaning and
error behaviour):
  GenerativeFunction / Trace ........ core/generative/generative_function.py:72-230, 238-689
  GenerativeFunctionClosure (`@`) ... core/generative/generative_function.py:1557-1684
  `@gen`, handlers, StaticTrace ..... generative_functions/static.py:80-119, 209-399, 725-810, 1044-1049
  Distribution / ExactDensity ....... generative_functions/distributions/distribution.py:59-147, 359-476

MI355X-first design instead of a tracing compiler: a model body runs ONCE over the whole particle
population.  Keys are per-particle key batches (lazy `split`), every `dist(args) @ addr` is one fused
sample+log-density kernel over a [n] column, values flow between sites as device columns, and the
trace is the struct-of-arrays of those columns.  When the body's structure is static and its
inter-site arithmetic is affine, the whole walk is lowered to ONE kernel (`gjx_importance_run`)
through `plan.py`; both routes implement the same arithmetic spec and are bit-identical.
"""

from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Any, Callable

import torch

from . import abi, prng
from .choicemap import ChoiceMap, Mask, Selection
from .ops import KeyBatch
from .runtime import get_ops

# =================================================================================================
# keys
# =================================================================================================


@dataclass(frozen=True)
class ParticleKeys:
    """n per-particle keys (the result of `split(key, n)`), possibly lazy."""

    kb: KeyBatch
    n: int

    @property
    def impl(self):
        return self.kb.impl

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            start, stop, step = i.indices(self.n)
            if step != 1:
                raise IndexError("ParticleKeys supports contiguous slices only")
            if self.kb.mode == 1:
                return ParticleKeys(KeyBatch(self.kb.impl, 1, parent=self.kb.parent, first=self.kb.first + start,
                                             fold=self.kb.fold, parent_lane=self.kb.parent_lane), max(0, stop - start))
            if self.kb.mode == 0:
                return ParticleKeys(KeyBatch(self.kb.impl, 0, tensor=self.kb.tensor[start:stop].contiguous(),
                                             fold=self.kb.fold), max(0, stop - start))
            return ParticleKeys(self.kb, max(0, stop - start))
        i = int(i)
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        if self.kb.fold is not None:
            raise ValueError("cannot index a folded key batch")
        parent = prng.PRNGKey(*self.kb.parent, self.kb.impl, self.kb.parent_lane)
        if self.kb.mode == 1:
            return prng.split_at(parent, self.kb.first + i)
        if self.kb.mode == 2:
            return parent
        w = [x & 0xFFFFFFFF for x in self.kb.tensor[i].cpu().tolist()]
        lane = (w[2] | (w[3] << 32)) if len(w) == 4 else 0
        return prng.PRNGKey(w[0], w[1], self.kb.impl, lane)

    def __iter__(self):
        return (self[i] for i in range(self.n))


def as_particle_keys(key) -> tuple[ParticleKeys, bool]:
    """-> (keys, batched).  A scalar key runs as a population of one and results are squeezed."""
    if isinstance(key, ParticleKeys):
        return key, True
    if isinstance(key, prng.PRNGKey):
        return ParticleKeys(key.literal(), 1), False
    raise TypeError(f"expected a PRNG key, got {type(key).__name__}")


def split(key, num: int = 2):
    """`jax.random.split`: scalar key -> `num` keys (lazy batch; iterable / indexable)."""
    if isinstance(key, prng.PRNGKey):
        return ParticleKeys(prng.split_lazy(key, num), num)
    if isinstance(key, ParticleKeys):
        # vmap(split): every particle key is split `num` ways (one nested-split kernel)
        t = get_ops().rng_split_each(key.kb, key.n, num).view(key.n, num, -1)
        return tuple(ParticleKeys(KeyBatch(key.impl, 0, tensor=t[:, j].contiguous()), key.n) for j in range(num))
    raise TypeError(f"expected a PRNG key, got {type(key).__name__}")


def fold_in(key, data: int):
    if isinstance(key, prng.PRNGKey):
        return prng.fold_in(key, data)
    if isinstance(key, ParticleKeys):
        ops = get_ops()
        t = ops.rng_keys(key.kb.with_fold(data), key.n)
        return ParticleKeys(KeyBatch(key.impl, 0, tensor=t), key.n)
    raise TypeError(type(key))


def scan_step_keys(key: ParticleKeys, t: int) -> ParticleKeys:
    """PHILOX keys of step t of a Scan (gjx.h gjx_scan_run; gjx_device.hpp scan_step_key_philox): every particle keeps its
    cipher key and moves to lane L + (t + 1) 2^40 — no cipher block, and for the lazy children of a lane-0 key no kernel
    either (the batch's first lane moves)."""
    kb, bump = key.kb, (t + 1) << 40
    if kb.fold is not None:
        raise ValueError("cannot step a folded key batch")
    if t + 1 >= (1 << 24) - 1:
        raise ValueError("a PHILOX scan has fewer than 2^24 - 1 steps")
    if kb.mode == 1 and kb.parent_lane == 0:
        if kb.first + key.n >= 1 << 40:
            raise ValueError("PHILOX scan keys need lanes below 2^40")
        return ParticleKeys(KeyBatch(kb.impl, 1, parent=kb.parent, first=kb.first + bump), key.n)
    if kb.mode == 2:
        return ParticleKeys(KeyBatch(kb.impl, 2, parent=kb.parent, parent_lane=kb.parent_lane + bump), key.n)
    # explicit keys, or the hashed children of a laned parent: materialise, then move every key's lane (word 3 += (t+1) << 8)
    keys = (kb.tensor if kb.mode == 0 else get_ops().rng_keys(kb, key.n)).clone()
    w3 = keys[:, 3].to(torch.int64) + ((t + 1) << 8)
    keys[:, 3] = (((w3 + (1 << 31)) % (1 << 32)) - (1 << 31)).to(keys.dtype)
    return ParticleKeys(KeyBatch(kb.impl, 0, tensor=keys), key.n)


def site_keys(key: ParticleKeys, fold: int, leaf: bool) -> ParticleKeys:
    """Per-`@`-site key: fold_in(key, fold) (static.py:349-352).  Leaf distributions take the
    fold lazily (fused into their kernel); nested generative functions get materialised keys."""
    if leaf:
        return ParticleKeys(key.kb.with_fold(fold), key.n)
    return fold_in(key, fold)


class _SiteCounter:
    """The per-body site numbering.  threefry: the reference's counter, from 1, advanced by every `@`
    site (static.py:349-352, 374-375).  philox: the 0-based index among the sites that consume
    randomness (unconstrained leaves and nested calls), so four consecutive single-word draws share
    one cipher block whatever is observed in between."""

    def __init__(self, impl: int):
        self.impl, self.counter, self.draws = impl, 1, 0

    def next(self, consumes: bool) -> int:
        if self.impl == prng.THREEFRY:
            f = self.counter
        else:
            f = self.draws
            self.draws += 1 if consumes else 0
        self.counter += 1
        return f


# =================================================================================================
# values
# =================================================================================================
def squeeze_leaf(v):
    if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == 1:
        return v[0]
    return v


def batch_size_of(v) -> int | None:
    if isinstance(v, torch.Tensor) and v.dim() >= 1:
        return int(v.shape[0])
    return None


# =================================================================================================
# traces
# =================================================================================================
class Trace:
    def get_args(self):
        raise NotImplementedError

    def get_retval(self):
        raise NotImplementedError

    def get_gen_fn(self):
        raise NotImplementedError

    def get_score(self):
        raise NotImplementedError

    def get_choices(self) -> ChoiceMap:
        raise NotImplementedError

    def get_sample(self) -> ChoiceMap:
        return self.get_choices()

    def project(self, key, selection: Selection):
        return self.get_gen_fn().project(key, self, selection)

    def edit(self, key, request, argdiffs=None):
        """generative_function.py:153-166: answer an edit request; arguments unchanged unless given."""
        from .edit import Diff

        return request.edit(key, self, Diff.no_change(self.get_args()) if argdiffs is None else argdiffs)

    def update(self, key, constraint: ChoiceMap, argdiffs=None):
        """generative_function.py:168-183 -> (new trace, weight, retdiff, discard)."""
        from .edit import Diff

        return self.get_gen_fn().update(key, self, constraint,
                                        Diff.no_change(self.get_args()) if argdiffs is None else argdiffs)

    def map_leaves(self, fn) -> "Trace":
        raise NotImplementedError


class MaterialTrace(Trace):
    """A trace reduced to what the inference layer reads: choices, score, retval, args."""

    def __init__(self, gen_fn, args, retval, choices: ChoiceMap, score):
        self.gen_fn, self.args, self.retval, self.choices, self.score = gen_fn, args, retval, choices, score

    @staticmethod
    def of(tr: Trace) -> "MaterialTrace":
        return MaterialTrace(tr.get_gen_fn(), tr.get_args(), tr.get_retval(), tr.get_choices(), tr.get_score())

    def get_args(self):
        return self.args

    def get_retval(self):
        return self.retval

    def get_gen_fn(self):
        return self.gen_fn

    def get_score(self):
        return self.score

    def get_choices(self):
        return self.choices

    def map_leaves(self, fn):
        return MaterialTrace(self.gen_fn, _map_any(fn, self.args), _map_any(fn, self.retval),
                             self.choices.map_leaves(lambda v: _map_any(fn, v) if not isinstance(v, torch.Tensor) else fn(v)),
                             _map_any(fn, self.score))


class EmptyTrace(Trace):
    def __init__(self, gen_fn):
        self.gen_fn = gen_fn

    def get_args(self):
        return ()

    def get_retval(self):
        return None

    def get_gen_fn(self):
        return self.gen_fn

    def get_score(self):
        return 0.0

    def get_choices(self):
        return ChoiceMap.empty()


class DistributionTrace(Trace):
    """(gen_fn, args, value, score) — distribution.py:59-82."""

    def __init__(self, gen_fn, args, value, score):
        self.gen_fn, self.args, self.value, self.score = gen_fn, args, value, score

    def get_args(self):
        return self.args

    def get_retval(self):
        return self.value

    def get_gen_fn(self):
        return self.gen_fn

    def get_score(self):
        return self.score

    def get_choices(self):
        return ChoiceMap.choice(self.value)

    def map_leaves(self, fn):
        return DistributionTrace(self.gen_fn, tuple(_map_any(fn, a) for a in self.args), fn(self.value), fn(self.score))


def _map_any(fn, v):
    if isinstance(v, torch.Tensor):
        return fn(v)
    if isinstance(v, (tuple, list)):
        return type(v)(_map_any(fn, x) for x in v)
    if isinstance(v, dict):
        return {k: _map_any(fn, x) for k, x in v.items()}
    if isinstance(v, Trace):
        return v.map_leaves(fn)
    if isinstance(v, ChoiceMap):
        return v.map_leaves(lambda x: _map_any(fn, x))
    return v


class StaticTrace(Trace):
    """static.py:80-119: score = sum of sub-trace scores; choices = {address: sub-choices}."""

    def __init__(self, gen_fn, args, retval, subtraces: dict, score=None):
        self.gen_fn, self.args, self.retval, self.subtraces = gen_fn, args, retval, subtraces
        self._score = score  # the fused path produces the total directly

    def get_args(self):
        return self.args

    def get_retval(self):
        return self.retval

    def get_gen_fn(self):
        return self.gen_fn

    def get_choices(self):
        return ChoiceMap.d({addr: tr.get_choices() for addr, tr in self.subtraces.items()})

    def get_score(self):
        if self._score is not None:
            return self._score
        total = None
        for tr in self.subtraces.values():
            s = tr.get_score()
            total = s if total is None else total + s
        return 0.0 if total is None else total

    def get_inner_trace(self, address):
        return self.subtraces[address]

    get_subtrace = get_inner_trace

    def map_leaves(self, fn):
        return StaticTrace(self.gen_fn, _map_any(fn, self.args), _map_any(fn, self.retval),
                           {a: t.map_leaves(fn) for a, t in self.subtraces.items()},
                           None if self._score is None else _map_any(fn, self._score))


class ValueTrace(DistributionTrace):
    """Sub-trace of a fused run: the site's column is materialised, its individual score is not
    (only the particle's total is); asking for it computes it with the log-density kernel."""

    def __init__(self, gen_fn, args_thunk, value):
        self.gen_fn, self._args_thunk, self.value = gen_fn, args_thunk, value
        self._score = None

    @property
    def args(self):
        return self._args_thunk()

    @property
    def score(self):
        if self._score is None:
            self._score = self.gen_fn.estimate_logpdf(None, self.value, *self.args)
        return self._score

    def map_leaves(self, fn):
        t = ValueTrace(self.gen_fn, lambda: _map_any(fn, self._args_thunk()), fn(self.value))
        return t


# =================================================================================================
# generative functions
# =================================================================================================
class AddressReuse(Exception):
    """Attempt to re-use an address within one generative function call (static.py:139-143)."""


class MissingAddress(Exception):
    """`assess` was not given a value for a visited address (static.py:145-148)."""


class GenerativeFunctionClosure:
    """`gen_fn(*args)`; `closure @ "addr"` traces it at an address (generative_function.py:1568-1583)."""

    def __init__(self, gen_fn, args: tuple, kwargs: dict):
        self.gen_fn, self.args, self.kwargs = gen_fn, args, kwargs

    def __matmul__(self, addr):
        return trace(addr, self.gen_fn, self._call_args())

    def _call_args(self):
        return self.gen_fn.canonical_args(self.args, self.kwargs)

    def __call__(self, key, *more):
        return self.gen_fn.simulate(key, self._call_args()).get_retval()

    def simulate(self, key, args=()):
        return self.gen_fn.simulate(key, self._call_args() + tuple(args))


class GenerativeFunction:
    def canonical_args(self, args, kwargs):
        if kwargs:
            raise TypeError(f"{type(self).__name__} does not take keyword arguments")
        return tuple(args)

    def __call__(self, *args, **kwargs) -> GenerativeFunctionClosure:
        return GenerativeFunctionClosure(self, args, kwargs)

    # -- GFI (generative_function.py:378-675) ------------------------------------------------------
    def simulate(self, key, args) -> Trace:
        raise NotImplementedError

    def assess(self, sample: ChoiceMap, args):
        raise NotImplementedError

    def generate(self, key, constraint: ChoiceMap, args):
        raise NotImplementedError

    def importance(self, key, constraint: ChoiceMap, args):
        """Same as `generate` (generative_function.py:629-675)."""
        return self.generate(key, constraint, args)

    def propose(self, key, args):
        tr = self.simulate(key, args)
        return tr.get_choices(), tr.get_score(), tr.get_retval()

    def project(self, key, trace: Trace, selection: Selection):
        raise NotImplementedError

    def edit(self, key, trace: Trace, edit_request, argdiffs):
        """generative_function.py:496-610.  `Update` is answered by re-generation unless a subclass
        walks its own structure; other requests are not supported on this path."""
        from .edit import NotSupportedEditRequest, Regenerate, Update, generic_regenerate, generic_update

        if isinstance(edit_request, Update):
            return generic_update(self, key, trace, edit_request.constraint, argdiffs)
        if isinstance(edit_request, Regenerate):
            return generic_regenerate(self, key, trace, edit_request.selection, argdiffs)
        raise NotSupportedEditRequest(edit_request)

    def update(self, key, trace: Trace, constraint: ChoiceMap, argdiffs):
        """generative_function.py:611-627."""
        from .edit import Update

        tr, w, rd, bwd = Update(constraint).edit(key, trace, argdiffs)
        assert isinstance(bwd, Update), type(bwd)
        return tr, w, rd, bwd.constraint

    # -- combinator sugar ----------------------------------------------------------------------------
    def marginal(self, /, *, selection: Selection = Selection.all(), algorithm=None):
        from .inference import Marginal

        return Marginal(self, selection, algorithm)

    def scan(self, /, *, n: int | None = None):
        from .combinators import Scan

        return Scan(self, length=n)

    def vmap(self, /, *, in_axes=0):
        from .combinators import Vmap

        return Vmap(self, in_axes=in_axes)

    def repeat(self, /, *, n: int):
        """`repeat` = vmap over a dummy axis (generative_functions/combinators/repeat.py:28-40)."""
        from .combinators import Vmap

        inner = self

        class _Repeat(Vmap):
            def _axes(self, args):
                return (None,) * len(args)

        return _Repeat(inner, in_axes=None, axis_size=n)

    def partial_apply(self, *first_args):
        outer = self

        def body(*rest):
            return outer(*first_args, *rest) @ "_partial"

        return StaticGenerativeFunction(body)


# -------------------------------------------------------------------------------------------------
# handlers (static.py:209-399)
# -------------------------------------------------------------------------------------------------
_tls = threading.local()


def _stack() -> list:
    if not hasattr(_tls, "stack"):
        _tls.stack = []
    return _tls.stack


class SpecTensor(torch.Tensor):
    """A site value on the per-site (column) path.  A model body may put `exp`, `log` or a division by a number between
    its sites (the reference's bodies use `jnp` freely, e.g. tests/inference/test_smc.py:63); inside a fused plan those are
    GJX_EXPR_EXP / _LOG / _DIV — the spec's f32 functions and an IEEE division.  torch's own kernels compute other bits
    (`x / c` multiplies by a reciprocal, `exp` / `log` are the device library's), so on this path the same three
    operations go through `gjx_map_f32`: a body gives the same trace whichever route runs it.  Everything else is
    torch's, and results stay `SpecTensor`s."""

    @staticmethod
    def __new__(cls, x):
        return x.as_subclass(cls)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        h = _SPEC_FUNCS.get(func)
        if h is not None:
            r = h(*args, **(kwargs or {}))
            if r is not NotImplemented:
                return r
        return super().__torch_function__(func, types, args, kwargs or {})


def _spec_plain(x):
    return x.as_subclass(torch.Tensor) if isinstance(x, SpecTensor) else x


def _spec_unary(op):
    def f(x, *a, **k):
        if a or k or not isinstance(x, torch.Tensor):
            return NotImplemented
        return get_ops().map_f32(op, _spec_plain(x)).as_subclass(SpecTensor)

    return f


def _spec_scalar(c):
    if isinstance(c, (bool, int, float)):
        return float(c)
    if isinstance(c, torch.Tensor) and c.dim() == 0 and not isinstance(c, SpecTensor):
        return float(c)
    return None


def _spec_div(a, b, *extra, rounding_mode=None, **k):
    if extra or k or rounding_mode is not None:
        return NotImplemented
    cb, ca = _spec_scalar(b), _spec_scalar(a)
    if cb is not None and isinstance(a, torch.Tensor):  # tensor / number: an IEEE division, not a reciprocal multiply
        return get_ops().map_f32(abi.MAP_DIV, _spec_plain(a), cb).as_subclass(SpecTensor)
    if ca is not None and isinstance(b, torch.Tensor):
        return get_ops().map_f32(abi.MAP_RDIV, _spec_plain(b), ca).as_subclass(SpecTensor)
    return NotImplemented  # (tensor / tensor is a true division on every backend)


def _spec_rdiv(b, a):
    return _spec_div(a, b)


def _spec_sigmoid(x, *a, **k):
    """1 / (1 + exp(-x)): the steps the fused program takes (plan._sym_sigmoid), each through the spec's function."""
    if a or k or not isinstance(x, torch.Tensor):
        return NotImplemented
    ops = get_ops()
    e = ops.map_f32(abi.MAP_EXP, -_spec_plain(x).to(torch.float32))
    return ops.map_f32(abi.MAP_RDIV, e + 1.0, 1.0).as_subclass(SpecTensor)


def _spec_reciprocal(x, *a, **k):
    if a or k or not isinstance(x, torch.Tensor):
        return NotImplemented
    return get_ops().map_f32(abi.MAP_RDIV, _spec_plain(x), 1.0).as_subclass(SpecTensor)


def _spec_softplus(x, *a, **k):
    """max(x, 0) + log(1 + exp(-|x|)) in the fused program's steps (plan._sym_torch_function)."""
    if a or k or not isinstance(x, torch.Tensor):
        return NotImplemented
    ops = get_ops()
    xp = _spec_plain(x).to(torch.float32)
    t = ops.map_f32(abi.MAP_LOG, ops.map_f32(abi.MAP_EXP, -xp.abs()) + 1.0)
    return (torch.maximum(xp, torch.zeros((), dtype=torch.float32, device=xp.device)) + t.to(xp.device)).as_subclass(SpecTensor)


_SPEC_FUNCS = {
    torch.nn.functional.softplus: _spec_softplus,
    torch.exp: _spec_unary(abi.MAP_EXP), torch.Tensor.exp: _spec_unary(abi.MAP_EXP),
    torch.log: _spec_unary(abi.MAP_LOG), torch.Tensor.log: _spec_unary(abi.MAP_LOG),
    torch.sqrt: _spec_unary(abi.MAP_SQRT), torch.Tensor.sqrt: _spec_unary(abi.MAP_SQRT),
    torch.div: _spec_div, torch.true_divide: _spec_div, torch.Tensor.div: _spec_div, torch.Tensor.true_divide: _spec_div,
    torch.Tensor.__truediv__: _spec_div, torch.Tensor.__rtruediv__: _spec_rdiv,
    torch.sigmoid: _spec_sigmoid, torch.Tensor.sigmoid: _spec_sigmoid, torch.nn.functional.sigmoid: _spec_sigmoid,
    torch.reciprocal: _spec_reciprocal, torch.Tensor.reciprocal: _spec_reciprocal,
}  # (abs, square, negation: torch's own results are exact; sqrt: the device's may not be correctly rounded)


def _spec_wrap(v):
    """Site values handed to a model body on the per-site path (see SpecTensor); containers of them (a callee's retval)."""
    if isinstance(v, torch.Tensor):
        return v if isinstance(v, SpecTensor) or v.is_complex() else v.as_subclass(SpecTensor)
    if isinstance(v, tuple):
        return tuple(_spec_wrap(x) for x in v)
    if isinstance(v, list):
        return [_spec_wrap(x) for x in v]
    return v


def _spec_unwrap(v):
    if isinstance(v, SpecTensor):
        return v.as_subclass(torch.Tensor)
    if isinstance(v, tuple):
        return tuple(_spec_unwrap(x) for x in v)
    if isinstance(v, list):
        return [_spec_unwrap(x) for x in v]
    if isinstance(v, dict):
        return {k: _spec_unwrap(x) for k, x in v.items()}
    return v


def trace(addr, gen_fn: GenerativeFunction, args: tuple):
    st = _stack()
    if not st:
        raise RuntimeError("`@` used outside of a generative function (`@gen`) body")
    addr = addr if isinstance(addr, tuple) else (addr,)
    for seg in addr:
        if not isinstance(seg, (str, int)):
            raise TypeError(f"static addresses must be strings (or ints), got {seg!r}")
    if isinstance(gen_fn, Distribution):
        args = _spec_unwrap(args)  # (the sampler / log-density wrappers get plain tensors: no subclass dispatch inside them)
    return _spec_wrap(st[-1].handle_trace(addr if len(addr) > 1 else addr[0], gen_fn, args))


class _Handler:
    def __init__(self):
        self.traces: dict = {}

    def record(self, addr, tr):
        if addr in self.traces:
            raise AddressReuse(addr)
        self.traces[addr] = tr

    def run(self, source, args):
        st = _stack()
        st.append(self)
        try:
            # (tensor arguments — a scan's carry — enter as site values do: SpecTensor; what the body returns leaves plain)
            return _spec_unwrap(source(*_spec_wrap(tuple(args))))
        finally:
            st.pop()


class SimulateHandler(_Handler):
    def __init__(self, key: ParticleKeys):
        super().__init__()
        self.key, self.sites = key, _SiteCounter(key.impl)

    def handle_trace(self, addr, gen_fn, args):
        sub_key = site_keys(self.key, self.sites.next(True), isinstance(gen_fn, Distribution))
        tr = gen_fn.simulate(sub_key, args)
        self.record(addr, tr)
        return tr.get_retval()


class GenerateHandler(_Handler):
    def __init__(self, key: ParticleKeys, constraint: ChoiceMap):
        super().__init__()
        self.key, self.constraint, self.sites = key, constraint, _SiteCounter(key.impl)
        self.weight = 0.0

    def handle_trace(self, addr, gen_fn, args):
        sub = self.constraint.get_submap(*(addr if isinstance(addr, tuple) else (addr,)))
        leaf = isinstance(gen_fn, Distribution)
        draws = not leaf or sub.static_is_empty() or isinstance(sub.get_value(), Mask)  # masked sites sample too
        sub_key = site_keys(self.key, self.sites.next(draws), leaf)
        tr, w = gen_fn.generate(sub_key, sub, args)
        self.weight = self.weight + w
        self.record(addr, tr)
        return tr.get_retval()


class UpdateHandler(_Handler):
    """static.py:405-461: every visited site edits its previous sub-trace with the sub-constraint;
    weights add up, the backward constraints are collected per address."""

    def __init__(self, key: ParticleKeys, previous_trace: "StaticTrace", constraint: ChoiceMap):
        super().__init__()
        self.key, self.previous_trace, self.constraint = key, previous_trace, constraint
        self.sites = _SiteCounter(key.impl)
        self.weight = 0.0
        self.bwd_constraints: dict = {}

    def handle_trace(self, addr, gen_fn, args):
        from .edit import Diff, Update

        leaf = isinstance(gen_fn, Distribution)
        sub = self.constraint.get_submap(*(addr if isinstance(addr, tuple) else (addr,)))
        try:
            subtrace = self.previous_trace.get_inner_trace(addr)
        except KeyError:
            raise MissingAddress(addr) from None
        # exact densities draw nothing during an update; nested functions get their folded key
        sub_key = site_keys(self.key, self.sites.next(not leaf), leaf)
        tr, w, retdiff, bwd = Update(sub).edit(sub_key, subtrace, Diff.unknown_change(tuple(args)))
        assert isinstance(bwd, Update) and isinstance(bwd.constraint, ChoiceMap)
        self.bwd_constraints[addr] = bwd.constraint
        self.weight = self.weight + w
        self.record(addr, tr)
        return Diff.tree_primal(retdiff)


class RequestHandler(_Handler):
    """static.py:510-566 (StaticEditRequestHandler) and 616-676 (RegenerateRequestHandler): each visited site edits its
    previous sub-trace with its own request; the site key is `fold_in(key, counter)`, counter from 1 over ALL sites."""

    def __init__(self, key, previous_trace: "StaticTrace", request):
        super().__init__()
        self.key, self.previous_trace, self.request = key, previous_trace, request
        self.counter = 1
        self.weight = 0.0
        self.bwd_requests: list = []

    def handle_trace(self, addr, gen_fn, args):
        from .edit import Diff, EmptyRequest, Regenerate

        try:
            subtrace = self.previous_trace.get_inner_trace(addr)
        except KeyError:
            raise MissingAddress(addr) from None
        a = addr if isinstance(addr, tuple) else (addr,)
        if isinstance(self.request, Regenerate):
            sub = Regenerate(self.request.selection(*a))
        else:
            sub = self.request.addressed.get(addr, EmptyRequest())
        sub_key = fold_in(self.key, self.counter)
        self.counter += 1
        tr, w, retdiff, bwd = sub.edit(sub_key, subtrace, Diff.unknown_change(tuple(args)))
        self.bwd_requests.append((addr, bwd))
        self.weight = self.weight + w
        self.record(addr, tr)
        return Diff.tree_primal(retdiff)


_ASSESS_BATCH: list = [None]  # stack: the column length of an enclosing Vmap.assess (assess carries no key to tell)


class AssessHandler(_Handler):
    def __init__(self, sample: ChoiceMap):
        super().__init__()
        self.sample, self.score = sample, 0.0

    def handle_trace(self, addr, gen_fn, args):
        sub = self.sample.get_submap(*(addr if isinstance(addr, tuple) else (addr,)))
        if sub.static_is_empty():
            raise MissingAddress(addr)
        score, v = gen_fn.assess(sub, args)
        self.score = self.score + score
        self.record(addr, None)
        return v


class AddressHandler(_Handler):
    """Records which addresses a body visits (for `ChoiceMap.invalid_subset`)."""

    def __init__(self, key):
        super().__init__()
        self.key, self.shape = key, ChoiceMap.empty()

    def handle_trace(self, addr, gen_fn, args):
        tr = gen_fn.simulate(site_keys(self.key, 1, isinstance(gen_fn, Distribution)), args)
        a = addr if isinstance(addr, tuple) else (addr,)
        self.shape = self.shape | ChoiceMap.entry(tr.get_choices() if not isinstance(gen_fn, Distribution) else True, *a)
        return tr.get_retval()


def visited_addresses(gen_fn, args) -> ChoiceMap:
    if isinstance(gen_fn, StaticGenerativeFunction):
        h = AddressHandler(as_particle_keys(prng.key(0))[0])
        h.run(gen_fn.source, args)
        return h.shape
    return gen_fn.simulate(prng.key(0), args).get_choices().map_leaves(lambda v: True)


class StaticGenerativeFunction(GenerativeFunction):
    """A Python function whose body traces other generative functions with `@` (static.py:725-1036)."""

    def __init__(self, source: Callable):
        self.source = source
        self.__name__ = getattr(source, "__name__", "gen_fn")
        self._plan_cache: dict = {}

    def __get__(self, instance, owner=None):  # `@gen` on methods (static.py:757-763)
        if instance is None:
            return self
        return StaticGenerativeFunction(self.source.__get__(instance, owner))

    def simulate(self, key, args):
        pk, batched = as_particle_keys(key)
        h = SimulateHandler(pk)
        retval = h.run(self.source, args)
        tr = StaticTrace(self, args, retval, h.traces)
        return tr if batched else tr.map_leaves(squeeze_leaf)

    def generate(self, key, constraint: ChoiceMap, args):
        pk, batched = as_particle_keys(key)
        if batched:
            from .plan import try_fused_generate

            fused = try_fused_generate(self, pk, constraint, args)
            if fused is not None:
                return fused
        h = GenerateHandler(pk, constraint)
        retval = h.run(self.source, args)
        tr = StaticTrace(self, args, retval, h.traces)
        w = h.weight
        if not batched:
            return tr.map_leaves(squeeze_leaf), squeeze_leaf(w)
        if not isinstance(w, torch.Tensor):  # no constrained site: a column of zeros
            w = torch.zeros(pk.n, dtype=torch.float32, device=get_ops().device()) + w
        return tr, w

    def assess(self, sample: ChoiceMap, args):
        h = AssessHandler(sample)
        retval = h.run(self.source, args)
        return h.score, retval

    def edit(self, key, trace: StaticTrace, edit_request, argdiffs):
        """static.py:827-865 (`edit_update`): re-run the body at the new arguments, editing each site's
        previous sub-trace; the backward request restores the discarded values."""
        from .edit import Diff, NotSupportedEditRequest, Regenerate, StaticRequest, Update

        if isinstance(edit_request, (Regenerate, StaticRequest)):
            # static.py:505-715, 867-960: every visited site edits its previous sub-trace with its sub-request
            # (`Regenerate(selection(addr))`, or the addressed request / `EmptyRequest`), site keys fold_in(key, counter)
            args = Diff.tree_primal(argdiffs)
            pk, batched = as_particle_keys(key)
            if not batched and batch_size_of(trace.get_score()) is not None:
                raise TypeError("a population trace needs per-particle keys (split(key, n)) for an edit")
            h = RequestHandler(key, trace, edit_request)  # (a scalar key stays scalar: its sites draw scalars)
            retval = h.run(self.source, args)
            new_trace = StaticTrace(self, args, retval, h.traces)
            return new_trace, h.weight, Diff.unknown_change(retval), StaticRequest(dict(h.bwd_requests))
        if not isinstance(edit_request, Update):
            raise NotSupportedEditRequest(edit_request)
        constraint = edit_request.constraint
        args = Diff.tree_primal(argdiffs)
        pk, batched = as_particle_keys(key)
        if not batched and batch_size_of(trace.get_score()) is not None:
            raise TypeError("a population trace needs per-particle keys (split(key, n)) for update")
        h = UpdateHandler(pk, trace, constraint)
        retval = h.run(self.source, args)
        new_trace = StaticTrace(self, args, retval, h.traces)
        discard = ChoiceMap.d({a: c for a, c in h.bwd_constraints.items() if not c.static_is_empty()})
        unchanged = constraint.static_is_empty() and Diff.static_check_no_change(argdiffs)
        weight = h.weight
        if not batched:  # nested combinators ran as a population of one
            new_trace, weight, discard = new_trace.map_leaves(squeeze_leaf), squeeze_leaf(weight), discard.map_leaves(squeeze_leaf)
            retval = new_trace.get_retval()
        retdiff = Diff.no_change(retval) if unchanged else Diff.unknown_change(retval)
        return new_trace, weight, retdiff, Update(discard)

    def project(self, key, trace: StaticTrace, selection: Selection):
        total = 0.0
        for addr, sub in trace.subtraces.items():
            a = addr if isinstance(addr, tuple) else (addr,)
            total = total + sub.get_gen_fn().project(key, sub, selection(*a))
        return total


def gen(f: Callable) -> StaticGenerativeFunction:
    """`@gen` (static.py:1044-1049)."""
    if isinstance(f, GenerativeFunction):
        return f
    return StaticGenerativeFunction(f)


# =================================================================================================
# distributions
# =================================================================================================
def _to_device_col(v, n: int, dtype=torch.float32):
    """Operand for a kernel: Python scalar / 0-d tensor -> float; [n] tensor -> contiguous column."""
    ops = get_ops()
    if isinstance(v, torch.Tensor):
        if v.dim() == 0:
            return float(v)
        if v.dim() == 1 and v.shape[0] == n:
            return v.to(device=ops.device(), dtype=dtype).contiguous()
        if v.dim() == 1 and v.shape[0] == 1:
            return float(v[0])
        raise NotImplementedError(
            f"distribution argument of shape {tuple(v.shape)} with {n} particles: only scalar events are supported")
    if isinstance(v, (bool, int, float)):
        return float(v)
    try:
        import numpy as np

        a = np.asarray(v)
        if a.ndim == 0:
            return float(a)
    except Exception:
        pass
    raise TypeError(f"unsupported distribution argument {type(v).__name__}")


class Distribution(GenerativeFunction):
    """Exact-density distribution backed by one fused sample+log-density kernel
    (mirrors ExactDensity, distribution.py:359-419)."""

    name = "distribution"
    dist_id: int = -1
    value_dtype = torch.float32

    # subclasses: _sample(keys: ParticleKeys, args) -> (value, score); _logpdf(n, value, args) -> score
    def __repr__(self):
        return f"genjax.{self.name}"

    # -- ExactDensity surface ----------------------------------------------------------------------
    def sample(self, key, *args, **kwargs):
        return self.random_weighted(key, *self.canonical_args(args, kwargs))[1]

    def logpdf(self, v, *args, **kwargs):
        return self.estimate_logpdf(None, v, *self.canonical_args(args, kwargs))

    def random_weighted(self, key, *args):
        pk, batched = as_particle_keys(key)
        value, score = self._sample(pk, args)
        if not batched:
            value, score = squeeze_leaf(value), squeeze_leaf(score)
        return score, value

    def estimate_logpdf(self, key, v, *args):
        n = batch_size_of(v)
        for a in args:
            n = n or batch_size_of(a)
        batched = n is not None
        score = self._logpdf(n or 1, v, args)
        return score if batched else squeeze_leaf(score)

    # -- vector-valued sites ---------------------------------------------------------------------------
    def _event_vmap(self, key, args, value=None):
        """A site whose arguments (or constrained value) carry an EVENT axis — `normal(mu_vec, 1.0) @ "x"` — is the
        distribution mapped over that axis: values `[d]` (`[n, d]` over a population), score summed over it
        (distribution.py:392-396 `jnp.sum(w)`).  -> the `Vmap` answering it, or None for a scalar event.  Element e draws
        from `split(site_key, d)[e]` (the mapped-site rule of `Vmap`), not from counter e of the site key as jax does:
        the same distribution, another stream (the samplers are not bit-pinned against jax either way)."""
        if not isinstance(self, (_RealDist, Flip)):
            return None
        n = key.n if isinstance(key, ParticleKeys) else 1
        batched = isinstance(key, ParticleKeys)

        def axis(a):
            if isinstance(a, Mask):
                a = a.value
            if not isinstance(a, torch.Tensor):
                return None
            if a.dim() >= 2:
                return 0
            if a.dim() == 1 and a.shape[0] > 1 and not (batched and a.shape[0] == n and n > 1):
                return 0
            return None

        axes = tuple(axis(a) for a in args)
        if all(x is None for x in axes) and (value is None or axis(value) is None):
            return None
        from .combinators import Vmap

        size = None
        if all(x is None for x in axes):  # only the constrained value is a vector: its length is the mapped axis
            v = value.value if isinstance(value, Mask) else value
            size = int(v.shape[-1] if v.dim() >= 2 else v.shape[0])
        return Vmap(self, in_axes=axes, axis_size=size)

    # -- GFI -----------------------------------------------------------------------------------------
    def simulate(self, key, args):
        ev = self._event_vmap(key, args) if key is not None else None
        if ev is not None:
            return ev.simulate(key, args)
        w, v = self.random_weighted(key, *args)
        return DistributionTrace(self, args, v, w)

    def generate(self, key, constraint: ChoiceMap, args):
        """distribution.py:117-147: unconstrained -> simulate, weight 0; constrained -> weight =
        score = logpdf(value)."""
        v = constraint.get_value()
        ev = self._event_vmap(key, args, v)
        if ev is not None:
            return ev.generate(key, constraint, args)
        if v is None:
            if not constraint.static_is_empty():
                raise ValueError("constraint for a distribution must be a value (ChoiceMap.choice)")
            return self.simulate(key, args), 0.0
        pk, batched = as_particle_keys(key)
        n = pk.n
        if isinstance(v, Mask):
            # distribution.py:129-142: cond(flag, importance, simulate), here over a whole column — every
            # element is sampled AND scored at the constrained value, the flag selects per element
            if isinstance(v.flag, bool):
                return self.generate(key, ChoiceMap.choice(v.value) if v.flag else ChoiceMap.empty(), args)
            flag = v.flag.to(get_ops().device()).reshape(-1).bool()
            if flag.numel() != n:
                raise ValueError(f"mask flag has {flag.numel()} elements for a population of {n}")
            sim_v, sim_s = self._sample(pk, args)
            con_v = self._canonical_value(v.value, n)
            if not isinstance(con_v, torch.Tensor) or con_v.dim() == 0:
                con_v = torch.zeros_like(sim_v) + con_v
            con_v = con_v.to(device=sim_v.device, dtype=sim_v.dtype).reshape(-1)
            con_s = self._logpdf(n, torch.where(flag, con_v, sim_v), args)  # unconstrained elements: any in-support value
            value, score = torch.where(flag, con_v, sim_v), torch.where(flag, con_s, sim_s)
            w = torch.where(flag, con_s, torch.zeros_like(con_s))
            return DistributionTrace(self, args, value, score), w
        w = self._logpdf(n, v, args)
        if not batched:
            w = squeeze_leaf(w)
        return DistributionTrace(self, args, self._canonical_value(v, n if batched else None), w), w

    def assess(self, sample: ChoiceMap, args):
        v = sample.get_value()
        if v is None:
            raise MissingAddress(())
        # `assess` carries no key: a 1-D column is a POPULATION (one value per particle, as everywhere on this path), so a
        # vector-valued site is recognised here only by a 2-D value / argument ([n, d]); for a single trace write the
        # site as `dist.vmap()(...)`, which carries its structure
        if _ASSESS_BATCH[-1] is None and any(isinstance(a, torch.Tensor) and a.dim() >= 2 for a in (v, *args)):
            ev = self._event_vmap(None, args, v)
            if ev is not None:  # the sum of the elements' log-densities
                return ev.assess(sample, args)
        w = self.estimate_logpdf(None, v, *args)
        return w, v

    def project(self, key, trace: DistributionTrace, selection: Selection):
        return trace.get_score() if selection.check() else 0.0

    def edit(self, key, trace: DistributionTrace, edit_request, argdiffs):
        """distribution.py:302-340 (dispatch) and 179-258 (`Update` with a ChoiceMap constraint):
        no value -> the old value is kept and re-scored at the new arguments (weight fwd − bwd, retval
        unchanged, nothing discarded); a value -> it replaces the old one (retval changed, the old choice
        is the discard).  A `Mask(value, flag)` constraint (distribution.py:214-243: `cond(flag, ...)`) replaces the
        value where the flag holds and keeps the old one elsewhere — over a population: one select and ONE
        log-density kernel on the merged column; the discard is the old value under the same flag."""
        from .edit import Diff, NotSupportedEditRequest, Regenerate, Update

        if isinstance(edit_request, Regenerate):
            # distribution.py:258-300: selected -> a fresh draw at the new arguments, weight = its score − the old score,
            # the old value is the backward Update; not selected -> keep the value (re-score it if the arguments moved)
            primals = Diff.tree_primal(argdiffs)
            if edit_request.selection.check():
                w, new_v = self.random_weighted(key, *primals)
                return (DistributionTrace(self, primals, new_v, w), w - trace.get_score(), Diff.unknown_change(new_v),
                        Update(ChoiceMap.choice(trace.get_retval())))
            if Diff.static_check_no_change(argdiffs):
                return trace, 0.0, Diff.no_change(trace.get_retval()), Update(ChoiceMap.empty())
            old = trace.get_retval()
            fwd = self.estimate_logpdf(key, old, *primals)
            return DistributionTrace(self, primals, old, fwd), fwd - trace.get_score(), Diff.no_change(old), Update(ChoiceMap.empty())
        if not isinstance(edit_request, Update):
            raise NotSupportedEditRequest(edit_request)
        constraint = edit_request.constraint
        primals = Diff.tree_primal(argdiffs)
        bwd = trace.get_score()
        v = constraint.get_value()
        if isinstance(v, Mask):
            if isinstance(v.flag, bool):
                sub = ChoiceMap.choice(v.value) if v.flag else ChoiceMap.empty()
                return self.edit(key, trace, Update(sub), argdiffs)
            old = trace.get_retval()
            if not isinstance(old, torch.Tensor) or old.dim() == 0:
                raise ValueError("a per-element mask needs a population trace")
            n = old.shape[0]
            flag = v.flag.to(old.device).reshape(-1).bool()
            if flag.numel() != n:
                raise ValueError(f"mask flag has {flag.numel()} elements for a population of {n}")
            new = self._canonical_value(v.value, n)
            if not isinstance(new, torch.Tensor) or new.dim() == 0:
                new = torch.zeros_like(old) + new
            merged = torch.where(flag, new.to(device=old.device, dtype=old.dtype).reshape(-1), old)
            fwd = self.estimate_logpdf(key, merged, *primals)
            if batch_size_of(fwd) is None:
                fwd = torch.zeros_like(bwd) + fwd
            return (DistributionTrace(self, primals, merged, fwd), fwd - bwd, Diff.unknown_change(merged),
                    Update(ChoiceMap.choice(Mask(old, flag))))
        if v is None:
            if not constraint.static_is_empty():
                raise ValueError("constraint for a distribution must be a value (ChoiceMap.choice)")
            old = trace.get_retval()
            fwd = self.estimate_logpdf(key, old, *primals)
            return DistributionTrace(self, primals, old, fwd), fwd - bwd, Diff.no_change(old), Update(ChoiceMap.empty())
        n = batch_size_of(bwd)
        fwd = self.estimate_logpdf(key, v, *primals)
        if n is not None and batch_size_of(fwd) is None:  # scalar value and arguments on a population
            fwd = torch.zeros_like(bwd) + fwd
        new_v = self._canonical_value(v, n)
        return DistributionTrace(self, primals, new_v, fwd), fwd - bwd, Diff.unknown_change(new_v), Update(trace.get_choices())

    def _canonical_value(self, v, n):
        return v


def _bool_col(t: torch.Tensor) -> torch.Tensor:
    return t.view(torch.bool) if t.dtype == torch.uint8 else t


class _RealDist(Distribution):
    abi_name = ""
    arg_names: tuple = ()

    def canonical_args(self, args, kwargs):
        args = list(args)
        for nm in self.arg_names[len(args):]:
            if nm not in kwargs:
                raise TypeError(f"genjax.{self.name}: missing argument {nm!r}")
            args.append(kwargs[nm])
        if len(args) != len(self.arg_names):
            raise TypeError(f"genjax.{self.name} takes {len(self.arg_names)} arguments")
        return tuple(args)

    def _sample(self, pk: ParticleKeys, args):
        ops = get_ops()
        a, b = (_to_device_col(x, pk.n) for x in args)
        return ops.sample_logpdf(self.abi_name, pk.kb, pk.n, a, b)

    def _logpdf(self, n, v, args):
        ops = get_ops()
        a, b = (_to_device_col(x, n) for x in args)
        return ops.logpdf(self.abi_name, n, _to_device_col(v, n), a, b)


class Normal(_RealDist):
    """tfd.Normal(loc, scale) — tensorflow_probability/__init__.py:259."""
    name, abi_name, dist_id, arg_names = "normal", "normal", 0, ("loc", "scale")


class Gamma(_RealDist):
    """tfd.Gamma(concentration, rate) — tensorflow_probability/__init__.py:164."""
    name, abi_name, dist_id, arg_names = "gamma", "gamma", 1, ("concentration", "rate")


class Beta(_RealDist):
    """tfd.Beta(concentration1, concentration0) — tensorflow_probability/__init__.py:82."""
    name, abi_name, dist_id, arg_names = "beta", "beta", 2, ("concentration1", "concentration0")


class Flip(Distribution):
    """tfd.Bernoulli(probs=p, dtype=bool) — tensorflow_probability/__init__.py:155."""
    name, dist_id, value_dtype = "flip", 3, torch.bool

    def canonical_args(self, args, kwargs):
        if kwargs:
            if set(kwargs) != {"probs"} or args:
                raise TypeError("genjax.flip takes a single probability")
            return (kwargs["probs"],)
        if len(args) != 1:
            raise TypeError("genjax.flip takes a single probability")
        return tuple(args)

    def _probs(self, args, n):
        return _to_device_col(args[0], n)

    def _sample(self, pk, args):
        v, s = get_ops().sample_logpdf("bernoulli", pk.kb, pk.n, self._probs(args, pk.n))
        return _bool_col(v), s

    def _logpdf(self, n, v, args):
        ops = get_ops()
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n and n > 1:
            v = v.to(ops.device()).to(torch.uint8).contiguous()
        else:
            v = bool(v)
        return ops.logpdf("bernoulli", n, v, self._probs(args, n))


class Bernoulli(Flip):
    """tfd.Bernoulli(logits=... | probs=...) — tensorflow_probability/__init__.py:72; a bare
    positional argument means logits and warns (distribution.py:479-500)."""
    name = "bernoulli"

    def canonical_args(self, args, kwargs):
        import warnings

        if args and not kwargs:
            warnings.warn("The use of a bare argument to genjax.bernoulli is deprecated. Please specify "
                          "`logits=` or `probs=` for the parameters. The default, which will be used in this "
                          "case, is logits.", DeprecationWarning)
            return (("logits", args[0]),)
        if len(kwargs) == 1 and not args and next(iter(kwargs)) in ("logits", "probs"):
            return (next(iter(kwargs.items())),)
        raise TypeError("genjax.bernoulli takes logits= or probs=")

    def _probs(self, args, n):
        kind, v = args[0]
        if kind == "probs":
            return _to_device_col(v, n)
        t = torch.as_tensor(v, dtype=torch.float32)
        return _to_device_col(torch.sigmoid(t), n)


class Categorical(Distribution):
    """tfd.Categorical(logits=... | probs=...) — tensorflow_probability/__init__.py:102-104.
    `sampling` selects how a category is drawn: "gumbel" = jax.random.categorical's Gumbel-max
    (n_cat uniforms per draw, the reference semantics), "inverse_cdf" = one uniform on the
    fixed-point CDF (the HBM-friendly default for large category counts)."""
    name, dist_id, value_dtype = "categorical", 4, torch.int32
    sampling = "gumbel"

    def canonical_args(self, args, kwargs):
        import warnings

        if args and not kwargs:
            warnings.warn("The use of a bare argument to genjax.categorical is deprecated. Please specify "
                          "`logits=` or `probs=` for the parameters. The default, which will be used in this "
                          "case, is logits.", DeprecationWarning)
            return (("logits", args[0]),)
        if len(kwargs) == 1 and not args and next(iter(kwargs)) in ("logits", "probs"):
            return (next(iter(kwargs.items())),)
        raise TypeError("genjax.categorical takes logits= or probs=")

    @staticmethod
    def _logits(arg, n):
        """-> f32 logits [rows, K] on the device, rows in {1, n}."""
        ops = get_ops()
        kind, v = arg if (isinstance(arg, tuple) and len(arg) == 2 and arg[0] in ("logits", "probs")) else ("logits", arg)
        t = torch.as_tensor(v, dtype=torch.float32).to(ops.device())
        if kind == "probs":
            t = torch.log(t)
        if t.dim() == 1:
            t = t[None, :]
        if t.dim() != 2 or t.shape[0] not in (1, n):
            raise NotImplementedError(f"categorical parameters of shape {tuple(t.shape)} with {n} particles")
        return t.contiguous()

    def _sample(self, pk, args):
        mode = 0 if self.sampling == "gumbel" else 1
        return get_ops().sample_logpdf_categorical(pk.kb, pk.n, self._logits(args[0], pk.n), None, mode)

    def _logpdf(self, n, v, args):
        ops = get_ops()
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n and n > 1:
            v = v.to(ops.device()).to(torch.int32).contiguous()
        else:
            v = int(v)
        return ops.logpdf_categorical(n, v, self._logits(args[0], n))


class Uniform(_RealDist):
    """tfd.Uniform(low, high) — tensorflow_probability/__init__.py (`uniform`): `low + (high − low)·u`, u the 24-bit uniform
    of the site's one-word draw (the bits every one-word sampler takes); log-density `−log(high − low)` inside `[low, high]`.
    Not a plan site (no `gjx_dist` entry): it runs as a bits kernel plus element-wise torch ops — enough for what the
    reference uses it for on this path (Metropolis–Hastings acceptance draws)."""
    name, abi_name, arg_names = "uniform", "", ("low", "high")

    def _sample(self, pk: ParticleKeys, args):
        ops = get_ops()
        bits = ops.rng_bits(pk.kb, pk.n)
        u = ((bits >> 9) & 0x7FFFFF | 0x3F800000).view(torch.float32) - 1.0
        lo, hi = (_to_device_col(x, pk.n) for x in args)
        v = lo + (hi - lo) * u
        return v, self._logpdf(pk.n, v, args)

    def _logpdf(self, n, v, args):
        ops = get_ops()
        lo, hi = (torch.as_tensor(_to_device_col(x, n), dtype=torch.float32, device=ops.device()) for x in args)
        v = torch.as_tensor(_to_device_col(v, n), dtype=torch.float32, device=ops.device())
        inside = (v >= lo) & (v <= hi)
        out = torch.where(inside, -torch.log(hi - lo), torch.full_like(v + lo, float("-inf")))
        return out if out.dim() else out.reshape(1).expand(n).clone()


normal = Normal()
uniform = Uniform()
gamma = Gamma()
beta = Beta()
flip = Flip()
bernoulli = Bernoulli()
categorical = Categorical()


class ExactDensityFromCallables(Distribution):
    """`exact_density(sample, logpdf, name)` (distribution.py:436-476): a user-defined distribution
    from two Python callables operating on columns."""

    def __init__(self, sample, logpdf, name):
        self._s, self._l, self.name = sample, logpdf, name

    def canonical_args(self, args, kwargs):
        return tuple(args) + ((kwargs,) if kwargs else ())

    def _split_kwargs(self, args):
        if args and isinstance(args[-1], dict):
            return args[:-1], args[-1]
        return args, {}

    def _sample(self, pk, args):
        a, kw = self._split_kwargs(args)
        v = self._s(pk, *a, **kw)
        return v, self._logpdf(pk.n, v, args)

    def _logpdf(self, n, v, args):
        a, kw = self._split_kwargs(args)
        w = self._l(v, *a, **kw)
        if isinstance(w, torch.Tensor) and w.dim() > 1:
            w = w.reshape(w.shape[0], -1).sum(1)  # non-scalar logpdf is summed (distribution.py:392-396)
        return w


def exact_density(sample, logpdf, name: str | None = None):
    import warnings

    if name is None:
        warnings.warn("You should supply a name argument to exact_density")
        name = "unknown"
    return ExactDensityFromCallables(sample, logpdf, name)
