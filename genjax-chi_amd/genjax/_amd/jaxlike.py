"""A deliberately small stand-in for the `jax` names GenJAX model code touches, over torch, so
existing model bodies (tests/inference/test_smc.py:59-66, README.md:89-116) run unmodified in
spirit:  `jax.random.key/split/fold_in`, `jax.lax.cond`, `jax.vmap` over key batches, and a few
`jnp` functions.  Supported-op scope is explicit (SURVEY §7 "front-end fidelity"): affine
arithmetic, `where`/`cond` selection, constant-table indexing, elementwise log/exp/abs/sqrt."""

from __future__ import annotations

import types

import torch

from . import prng
from .choicemap import ChoiceMap
from .lang import fold_in as _fold_in, split as _split
from .runtime import get_ops

# ---- jax.random ---------------------------------------------------------------------------------
random = types.SimpleNamespace(
    key=prng.key,
    PRNGKey=prng.key,
    split=_split,
    fold_in=_fold_in,
)


# ---- jax.lax --------------------------------------------------------------------------------------
def _traced(x) -> bool:
    """A symbolic value of the plan tracer (plan.Sym / SymExpr): torch functions of it lower to the fused kernel's program."""
    return hasattr(x, "tracer") and not isinstance(x, torch.Tensor)


def _cond(pred, true_fun, false_fun, *operands):
    """`jax.lax.cond` on a column: both branches are evaluated, `where` selects."""
    t, f = true_fun(*operands), false_fun(*operands)
    if _traced(pred) or _traced(t) or _traced(f):
        return torch.where(pred, t, f)  # (GJX_EXPR_SELECT in the fused kernel: the same selection)
    if isinstance(pred, torch.Tensor):
        tt = torch.as_tensor(t, device=pred.device)
        ff = torch.as_tensor(f, device=pred.device)
        return torch.where(pred.bool(), tt, ff)
    return t if pred else f


lax = types.SimpleNamespace(cond=_cond)


# ---- jax.numpy --------------------------------------------------------------------------------------
def _array(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x
    t = torch.as_tensor(x)
    if t.dtype == torch.float64:
        t = t.to(torch.float32)
    try:
        return t.to(get_ops().device())
    except Exception:
        return t


jnp = types.SimpleNamespace(
    array=_array,
    asarray=_array,
    where=lambda c, a, b: torch.where(c, a, b) if (_traced(c) or _traced(a) or _traced(b))
    else torch.where(torch.as_tensor(c).bool(), torch.as_tensor(a), torch.as_tensor(b)),
    log=lambda x: torch.log(x if (_traced(x) or isinstance(x, torch.Tensor)) else torch.as_tensor(x, dtype=torch.float32)),
    exp=lambda x: torch.exp(x if (_traced(x) or isinstance(x, torch.Tensor)) else torch.as_tensor(x, dtype=torch.float32)),
    sqrt=lambda x: torch.sqrt(x if (_traced(x) or isinstance(x, torch.Tensor)) else torch.as_tensor(x, dtype=torch.float32)),
    abs=torch.abs,
    maximum=lambda a, b: torch.maximum(a, b) if (_traced(a) or _traced(b)) else torch.maximum(torch.as_tensor(a), torch.as_tensor(b)),
    minimum=lambda a, b: torch.minimum(a, b) if (_traced(a) or _traced(b)) else torch.minimum(torch.as_tensor(a), torch.as_tensor(b)),
    clip=lambda x, lo=None, hi=None: torch.clamp(x, min=lo, max=hi),
    square=torch.square,
    mean=lambda x, axis=None: torch.mean(x.float()) if axis is None else torch.mean(x.float(), dim=axis),
    sum=lambda x, axis=None: torch.sum(x) if axis is None else torch.sum(x, dim=axis),
    ones=lambda *s: torch.ones(*s),
    zeros=lambda *s: torch.zeros(*s),
    arange=torch.arange,
    float32=torch.float32,
    int32=torch.int32,
)


# ---- jax.vmap over keys -----------------------------------------------------------------------------
def _stack(items):
    first = items[0]
    if isinstance(first, torch.Tensor):
        return torch.stack(items, 0)
    if isinstance(first, ChoiceMap):
        addrs = [a for a, _ in first.leaves()]
        dicts = [dict(it.leaves()) for it in items]
        return ChoiceMap.from_mapping([(a, _stack([d[a] for d in dicts])) for a in addrs])
    if isinstance(first, (tuple, list)):
        return type(first)(_stack([it[i] for it in items]) for i in range(len(first)))
    if isinstance(first, (bool, int, float)):
        return torch.as_tensor(items)
    return items


def vmap(fn, in_axes=0):
    """`jax.vmap(fn, in_axes=(0, None, ...))` where the mapped argument is a key batch: runs `fn`
    once per key and stacks every leaf of the results along a new leading axis.  (Independent
    *trials* are mapped this way; the particle axis inside each trial is the vectorised one.)"""

    def mapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        n = None
        for a, ax in zip(args, axes):
            if ax is not None:
                n = len(a)
        # r04: a trial axis over `alg.random_weighted` / `alg.log_marginal_likelihood_estimate` runs BATCHED when the algorithm
        # can (a bounded number of launches instead of a run per key); element b equals the scalar call bit for bit
        owner, name = getattr(fn, "__self__", None), getattr(fn, "__name__", "")
        if owner is not None and axes and axes[0] is not None and all(ax is None for ax in axes[1:]):
            if name == "random_weighted" and hasattr(owner, "random_weighted_batch"):
                got = owner.random_weighted_batch(args[0], *args[1:])
                if got is not None:
                    return got
            if name == "log_marginal_likelihood_estimate" and len(args) == 1 and hasattr(owner, "log_marginal_likelihood_estimates"):
                return owner.log_marginal_likelihood_estimates(args[0])
        outs = []
        for i in range(n):
            call = [a[i] if ax is not None else a for a, ax in zip(args, axes)]
            outs.append(fn(*call))
        return _stack(outs)

    return mapped


def jit(fn=None, **_):
    """No tracing compiler here: kernels are ahead-of-time HIP; `jit` is the identity."""
    return fn if fn is not None else (lambda f: f)
