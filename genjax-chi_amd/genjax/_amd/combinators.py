"""`Scan` — the state-space combinator (reference: generative_functions/combinators/scan.py:55-96
ScanTrace, 200-294 simulate/generate, 638-664 assess).

General (any kernel) path: a host loop over the T steps; step t runs the kernel generative
function once over the whole particle population with the chained key
`key_t = fold_in(key_{t-1}, t)` (scan.py:212-213, 267-268: the folded key is carried).  Every
choice is stored time-major `[T, n]` (particle axis contiguous) and presented particle-major
`[n, T]` like the reference's vmapped ScanTrace.  Bootstrap SMC on the benchmark state-space
models does not go through this loop: it uses the fused `gjx_smc_run_*` kernels (smc_fused.py).
"""

from __future__ import annotations

import torch

from .choicemap import ChoiceMap, Selection
from .lang import GenerativeFunction, ParticleKeys, Trace, _map_any, as_particle_keys, fold_in, squeeze_leaf


def _index_xs(xs, t):
    return _map_any(lambda v: v[t], xs) if xs is not None else None


def _stack_time(items: list, batched: bool):
    """list over t of pytrees with [n] leaves -> pytree with [n, T] leaves (views of [T, n]).
    Constrained steps contribute Python scalars; they are broadcast to the column shape."""
    first = items[0]
    ref = next((it for it in items if isinstance(it, torch.Tensor)), None)
    if ref is not None or isinstance(first, (bool, int, float)):
        if ref is None:
            return torch.as_tensor(items)
        cols = [it if isinstance(it, torch.Tensor) else torch.as_tensor(it, dtype=ref.dtype, device=ref.device).expand(ref.shape)
                for it in items]
        st = torch.stack(cols, 0)
        return st.movedim(0, 1) if (batched and st.dim() >= 2) else st
    if isinstance(first, (tuple, list)):
        return type(first)(_stack_time([it[i] for it in items], batched) for i in range(len(first)))
    if isinstance(first, dict):
        return {k: _stack_time([it[k] for it in items], batched) for k in first}
    return None


class ScanTrace(Trace):
    def __init__(self, gen_fn, step_traces: list, args, retval, score, batched: bool):
        self.gen_fn, self.step_traces, self.args, self.retval, self.score = gen_fn, step_traces, args, retval, score
        self.batched = batched

    def get_args(self):
        return self.args

    def get_retval(self):
        return self.retval

    def get_gen_fn(self):
        return self.gen_fn

    def get_score(self):
        return self.score

    def get_choices(self) -> ChoiceMap:
        per_step = [dict(tr.get_choices().leaves()) for tr in self.step_traces]
        pairs = [(addr, _stack_time([d[addr] for d in per_step], self.batched)) for addr in per_step[0]]
        return ChoiceMap.from_mapping(pairs)

    def map_leaves(self, fn):
        return ScanTrace(self.gen_fn, [t.map_leaves(fn) for t in self.step_traces], _map_any(fn, self.args),
                         _map_any(fn, self.retval), _map_any(fn, self.score), self.batched)


class Scan(GenerativeFunction):
    """`kernel.scan(n=T)`: kernel(carry, x_t) -> (carry', y_t)."""

    def __init__(self, kernel_gen_fn: GenerativeFunction, length: int | None = None):
        self.kernel_gen_fn, self.length = kernel_gen_fn, length

    def _length(self, xs):
        if self.length is not None:
            return int(self.length)
        leaves = []
        _map_any(lambda v: (leaves.append(v), v)[1], xs)
        if not leaves:
            raise ValueError("scan needs either n= or scanned inputs")
        return int(leaves[0].shape[0])

    def _run(self, key, args, step):
        carry, xs = args
        pk, batched = as_particle_keys(key)
        T = self._length(xs)
        traces, ys, score, weight = [], [], 0.0, 0.0
        for t in range(T):
            pk = fold_in(pk, t)  # chained: the folded key becomes the carry (scan.py:267-268,276)
            tr, w = step(pk, t, (carry, _index_xs(xs, t)))
            carry, y = tr.get_retval()
            traces.append(tr)
            ys.append(y)
            score = score + tr.get_score()
            weight = weight + w
        retval = (carry, _stack_time(ys, True) if ys and ys[0] is not None else None)
        return ScanTrace(self, traces, args, retval, score, True), weight, batched

    def simulate(self, key, args):
        tr, _, batched = self._run(key, args, lambda pk, t, a: (self.kernel_gen_fn.simulate(pk, a), 0.0))
        return tr if batched else tr.map_leaves(squeeze_leaf)

    def generate(self, key, constraint: ChoiceMap, args):
        tr, w, batched = self._run(
            key, args, lambda pk, t, a: self.kernel_gen_fn.generate(pk, constraint.get_submap(t), a))
        if batched:
            return tr, w
        return tr.map_leaves(squeeze_leaf), squeeze_leaf(w)

    def assess(self, sample: ChoiceMap, args):
        carry, xs = args
        T = self._length(xs)
        score, ys = 0.0, []
        for t in range(T):
            s, (carry, y) = self.kernel_gen_fn.assess(sample.get_submap(t), (carry, _index_xs(xs, t)))
            score = score + s
            ys.append(y)
        return score, (carry, _stack_time(ys, False) if ys and ys[0] is not None else None)

    def project(self, key, trace: ScanTrace, selection: Selection):
        total = 0.0
        for tr in trace.step_traces:
            total = total + tr.project(key, selection)
        return total


def scan(*, n: int | None = None):
    def decorator(f) -> Scan:
        from .lang import gen

        return Scan(gen(f), length=n)

    return decorator
