"""`Scan` — the state-space combinator (reference: generative_functions/combinators/scan.py:55-96
ScanTrace, 200-294 simulate/generate, 638-664 assess).

Two routes, the same numbers.  A plan-able kernel (scan_plan.py: supported distributions, affine arguments,
the same scalar constraints at every step) runs the WHOLE scan of a particle population as one launch
(`gjx_scan_run`): key chain, carry and running weight in registers, every sampled value streamed into a
time-major `[T, n]` column.  Anything else takes the general path: a host loop over the T steps; step t runs
the kernel generative function once over the whole population with the chained key
`key_t = fold_in(key_{t-1}, t)` (scan.py:212-213, 267-268: the folded key is carried).  Either way every
choice is stored time-major `[T, n]` (particle axis contiguous) and presented particle-major
`[n, T]` like the reference's vmapped ScanTrace.  Bootstrap SMC on the benchmark state-space
models does not go through either: it uses the fused `gjx_smc_run_*` kernels (smc_fused.py).
"""

from __future__ import annotations

import torch

from .choicemap import ChoiceMap, Mask, Selection
from . import prng
from .lang import GenerativeFunction, ParticleKeys, Trace, _map_any, as_particle_keys, fold_in, scan_step_keys, squeeze_leaf
from .ops import KeyBatch
from .runtime import get_ops


def _index_xs(xs, t):
    return _map_any(lambda v: v[t], xs) if xs is not None else None


def _stack_time(items: list, batched: bool):
    """list over t of pytrees with [n] leaves -> pytree with [n, T] leaves (views of [T, n]).
    Constrained steps contribute Python scalars; they are broadcast to the column shape."""
    first = items[0]
    tens = [it for it in items if isinstance(it, torch.Tensor)]
    # (a step may contribute a 0-d value — the carried scalar, which starts on the host: the widest tensor, on the device if
    # any step's value lives there, sets shape, dtype and device of the stack)
    ref = max(tens, key=lambda t: (t.dim(), t.device.type != "cpu")) if tens else None
    if ref is not None or isinstance(first, (bool, int, float)):
        if ref is None:
            return torch.as_tensor(items)
        cols = [(it.to(ref.device) if it.dim() == ref.dim() else it.to(device=ref.device, dtype=ref.dtype).expand(ref.shape))
                if isinstance(it, torch.Tensor) else torch.as_tensor(it, dtype=ref.dtype, device=ref.device).expand(ref.shape) for it in items]
        st = torch.stack(cols, 0)
        return st.movedim(0, 1) if (batched and st.dim() >= 2) else st
    if isinstance(first, (tuple, list)):
        return type(first)(_stack_time([it[i] for it in items], batched) for i in range(len(first)))
    if isinstance(first, dict):
        return {k: _stack_time([it[k] for it in items], batched) for k in first}
    return None


def _float_leaves_to_f32(v):
    if isinstance(v, float):
        return torch.tensor(v, dtype=torch.float32)
    if isinstance(v, tuple):
        return tuple(_float_leaves_to_f32(x) for x in v)
    if isinstance(v, list):
        return [_float_leaves_to_f32(x) for x in v]
    if isinstance(v, dict):
        return {k: _float_leaves_to_f32(x) for k, x in v.items()}
    return v


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
        if not per_step:  # a zero-length scan has no choices (test_scan_combinator.py:402-420)
            return ChoiceMap.empty()
        pairs = [(addr, _stack_time([d[addr] for d in per_step], self.batched)) for addr in per_step[0]]
        return ChoiceMap.from_mapping(pairs)

    def map_leaves(self, fn):
        return ScanTrace(self.gen_fn, [t.map_leaves(fn) for t in self.step_traces], _map_any(fn, self.args),
                         _map_any(fn, self.retval), _map_any(fn, self.score), self.batched)


FUSED_SCAN = True  # tests switch the one-launch route off to compare it with the host loop


class FusedScanTrace(ScanTrace):
    """Trace of a one-launch scan: the `[T, n]` columns the kernel wrote.  The per-step traces the general
    `ScanTrace` methods walk (project, per-step scores) are rebuilt on demand by re-running the kernel generative
    function with every choice constrained — deterministic, nothing is sampled."""

    def __init__(self, gen_fn, args, retval, score, leaves: list, pk, constraint_free_choices=None):
        self.gen_fn, self.args, self.retval, self.score = gen_fn, args, retval, score
        self.batched = True
        self.leaves = leaves  # [(addr, [n, T] tensor | [T] observed vector)]
        self._pk = pk
        self._steps = None

    @property
    def step_traces(self):
        if self._steps is None:
            carry, xs = self.args
            steps = []
            for t in range(self.gen_fn._length(xs)):
                chm_t = ChoiceMap.from_mapping([(a, v[:, t] if v.dim() == 2 else v[t]) for a, v in self.leaves])
                tr, _ = self.gen_fn.kernel_gen_fn.generate(self._pk, chm_t, (carry, _index_xs(xs, t)))
                carry = tr.get_retval()[0]
                steps.append(tr)
            self._steps = steps
        return self._steps

    def get_choices(self) -> ChoiceMap:
        return ChoiceMap.from_mapping(self.leaves)

    def map_leaves(self, fn):
        out = FusedScanTrace(self.gen_fn, _map_any(fn, self.args), _map_any(fn, self.retval), _map_any(fn, self.score),
                             [(a, _map_any(fn, v)) for a, v in self.leaves], self._pk)
        return out


class Scan(GenerativeFunction):
    """`kernel.scan(n=T)`: kernel(carry, x_t) -> (carry', y_t)."""

    def __init__(self, kernel_gen_fn: GenerativeFunction, length: int | None = None):
        self.kernel_gen_fn, self.length = kernel_gen_fn, length

    def _length(self, xs):
        leaves = []
        _map_any(lambda v: (leaves.append(v), v)[1], xs)
        sizes = [int(v.shape[0]) for v in leaves if hasattr(v, "shape") and len(v.shape) >= 1]
        if len(set(sizes)) > 1:  # scan.py:178-196
            raise ValueError("scan got values with different leading axis sizes: " + ", ".join(str(z) for z in sizes) + ".")
        if self.length is not None:
            if sizes and sizes[0] != int(self.length):
                raise ValueError(f"scan got `length` argument of {int(self.length)} which disagrees with leading axis sizes {sizes[0]}.")
            return int(self.length)
        if not sizes:
            raise ValueError("scan needs either n= or scanned inputs")
        return sizes[0]

    def _run(self, key, args, step):
        carry, xs = args
        # `lax.scan` makes every carry leaf an array: a Python float enters as float32, so the arithmetic a kernel does on a
        # carried scalar (`v * 1.45`, step after step) is f32 arithmetic — as in the one-launch scan, where it is a state
        # column — not Python's float64
        carry = _float_leaves_to_f32(carry)
        pk, batched = as_particle_keys(key)
        T = self._length(xs)
        traces, ys, score, weight = [], [], 0.0, 0.0
        pk0 = pk
        for t in range(T):
            # THREEFRY, chained: the folded key becomes the carry (scan.py:267-268, 276).  PHILOX: no cipher block for a
            # key — step t draws under (the particle's cipher key, its lane + (t + 1) 2^40), as gjx_scan_run does
            pk = fold_in(pk, t) if pk0.impl == prng.THREEFRY else scan_step_keys(pk0, t)
            tr, w = step(pk, t, (carry, _index_xs(xs, t)))
            carry, y = tr.get_retval()
            traces.append(tr)
            ys.append(y)
            score = score + tr.get_score()
            weight = weight + w
        retval = (carry, _stack_time(ys, True) if ys and ys[0] is not None else None)
        return ScanTrace(self, traces, args, retval, score, True), weight, batched

    def _fused(self, key, constraint: ChoiceMap, args):
        """-> (FusedScanTrace, weight) through one `gjx_scan_run` launch, or None (not a population / not plan-able)."""
        if not FUSED_SCAN or not isinstance(key, ParticleKeys) or key.n < 2 or key.kb.fold is not None:
            return None
        from . import scan_plan as SP
        from .abi import GjxError
        from .plan import PlanUnsupported

        carry0, xs = args
        try:  # lowering: the kernel body is traced again on every call (it may close over values that changed)
            T = self._length(xs)
            if T < 1:
                return None
            obs_addrs, obs_values = SP.step_constraints(constraint, T)
            from .runtime import fast_math_enabled

            low = SP.lower_scan(self.kernel_gen_fn, carry0, xs, obs_addrs, fast_math=fast_math_enabled())
            table = SP.observation_table(obs_values, xs, T)
        except GjxError:
            raise  # the library refused or failed to create the plan (ABI mismatch, allocation, ...): no silent slow route
        except Exception:
            # PlanUnsupported, or symbolic values fed to code that needs tensors: the host loop runs the model and
            # raises genuine model errors itself
            return None
        try:
            out = SP.run_scan(low, key, T, carry0, table)  # launch / hiprtc failures propagate: no silent slow route
        except PlanUnsupported:
            return None
        except GjxError as e:
            from .runtime import compiler_switched_off

            if compiler_switched_off(e):  # GJX_PLAN_JIT=0: one-launch scans are generated kernels; the host loop runs the scan
                return None
            raise
        dev = out["logw"].device
        values_nt = []
        for m, v in zip(low.value_meta, out["values"]):
            v = v.t()  # [n, T] view of the time-major column
            values_nt.append(v != 0 if m["dtype"] == torch.bool else v)
        leaves = []
        k_obs = {a: i for i, a in enumerate(obs_addrs)}
        for m in low.tracer.meta:
            a = m["path"]  # (the full address: the enclosing calls' addresses, then the site's)
            addr = a[0] if len(a) == 1 else a
            if m["out_col"] >= 0:
                leaves.append((addr, values_nt[m["out_col"]]))
            else:
                leaves.append((addr, _stack_time(obs_values[k_obs[a]], True)))
        final = [c[0] if u else c for c, u in zip(out["carry"], low.uniform_carry)]
        final = [(c != 0 if dt == torch.bool else c.to(dt)) if dt is not None else c for c, dt in zip(final, low.carry_dtypes)]
        retval = (low.rebuild_carry(final), SP.resolve(low, low.ret_y, values_nt, table, dev, carry0))
        tr = FusedScanTrace(self, args, retval, out["score"], leaves, key)
        tr.max_partials, tr.row_stats = out["max_partials"], out["rows"]
        return tr, out["logw"]

    def simulate(self, key, args):
        fused = self._fused(key, ChoiceMap.empty(), args)
        if fused is not None:
            return fused[0]
        tr, _, batched = self._run(key, args, lambda pk, t, a: (self.kernel_gen_fn.simulate(pk, a), 0.0))
        return tr if batched else tr.map_leaves(squeeze_leaf)

    def generate(self, key, constraint: ChoiceMap, args):
        if isinstance(key, ParticleKeys):
            # per-particle constraints arrive particle-major, [n, T], like the trace's own choices (the reference's vmap
            # batches the leading axis): make them time-major so that `get_submap(t)` takes step t's column
            n, T = key.n, self._length(args[1])
            constraint = constraint.map_leaves(
                lambda v: v.t() if isinstance(v, torch.Tensor) and v.dim() == 2 and tuple(v.shape) == (n, T) else v)
        fused = self._fused(key, constraint, args)
        if fused is not None:
            return fused
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

    def edit(self, key, trace, edit_request, argdiffs):
        from .edit import Diff, IndexRequest, generic_index_request

        if isinstance(edit_request, IndexRequest):
            T = self._length(Diff.tree_primal(argdiffs)[1])
            return generic_index_request(self, key, trace, edit_request.idx, edit_request.request, argdiffs, T)
        return super().edit(key, trace, edit_request, argdiffs)


def scan(*, n: int | None = None):
    def decorator(f) -> Scan:
        from .lang import gen

        return Scan(gen(f), length=n)

    return decorator


# =================================================================================================
# Vmap (reference: generative_functions/combinators/vmap.py:55-94 VmapTrace, 180-218 simulate/generate)
# =================================================================================================
class VmapTrace(Trace):
    """The inner trace runs over n*m "virtual particles" (particle i, element j at i*m + j); leaves
    are presented as [n, m] ([m] for a scalar key), score = sum over the mapped axis."""

    def __init__(self, gen_fn, inner: Trace, args, n: int, m: int, batched: bool):
        self.gen_fn, self.inner, self.args, self.n, self.m, self.batched = gen_fn, inner, args, n, m, batched

    def _shape(self, v):
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == self.n * self.m:
            v = v.reshape((self.n, self.m) + tuple(v.shape[1:]))
            return v if self.batched else v[0]
        return v

    def get_args(self):
        return self.args

    def get_gen_fn(self):
        return self.gen_fn

    def get_retval(self):
        return _map_any(self._shape, self.inner.get_retval())

    def get_choices(self) -> ChoiceMap:
        return self.inner.get_choices().map_leaves(self._shape)

    def get_score(self):
        s = self.inner.get_score()
        if isinstance(s, torch.Tensor) and s.dim() >= 1:
            s = s.reshape(self.n, self.m).sum(1)
            return s if self.batched else s[0]
        return s * self.m

    def map_leaves(self, fn):
        # a population of one being squeezed (a scalar key's run inside a static function): present [m] leaves and a
        # scalar score from now on (the inner [1 * m] columns themselves have nothing to squeeze)
        batched = self.batched and not (fn is squeeze_leaf and self.n == 1)
        return VmapTrace(self.gen_fn, self.inner.map_leaves(fn), _map_any(fn, self.args), self.n, self.m, batched)


class EmptyVmapTrace(Trace):
    """A vmap over a zero-length axis: no choices, score 0 (test_vmap_combinator.py:230-243)."""

    def __init__(self, gen_fn, args, n: int, batched: bool):
        self.gen_fn, self.args, self.n, self.batched = gen_fn, args, n, batched

    def get_args(self):
        return self.args

    def get_gen_fn(self):
        return self.gen_fn

    def get_retval(self):
        return None

    def get_choices(self) -> ChoiceMap:
        return ChoiceMap.empty()

    def get_score(self):
        return torch.zeros(self.n, device=get_ops().device()) if self.batched else 0.0

    def map_leaves(self, fn):
        return self


class Vmap(GenerativeFunction):
    """`gen_fn.vmap(in_axes=(0, None, ...))`: independent copies over a mapped axis of the arguments;
    element j of particle i uses key split(key_i, m)[j] (vmap.py:186, 201)."""

    def __init__(self, gen_fn: GenerativeFunction, in_axes=0, axis_size: int | None = None):
        self.gen_fn, self.in_axes, self.axis_size = gen_fn, in_axes, axis_size

    def _axes(self, args):
        ax = self.in_axes
        if not isinstance(ax, (tuple, list)):
            ax = (ax,) * len(args)
        if len(ax) != len(args):
            raise ValueError("in_axes must match the arguments")
        for a in ax:
            if a not in (0, None):
                raise NotImplementedError("Vmap supports in_axes entries 0 and None")
        return ax

    def _length(self, args, axes, n: int, batched: bool) -> int:
        sizes = []
        for a, ax in zip(args, axes):
            if ax == 0:
                t = torch.as_tensor(a)
                if t.dim() == 0:  # (jax.vmap's message, test_vmap_combinator.py:180-184)
                    raise ValueError("vmap was requested to map its argument along axis 0, which implies that its rank "
                                     "should be at least 1, but is only 0 (its shape is ())")
                # a mapped argument is [m] (shared by all particles) or [n, m] (per particle)
                sizes.append(int(t.shape[1] if (batched and t.dim() >= 2 and t.shape[0] == n) else t.shape[0]))
        if len(set(sizes)) > 1:
            raise IndexError("vmap got inconsistent sizes for the mapped axis: " + ", ".join(str(z) for z in sizes))
        if self.axis_size is not None:
            return int(self.axis_size)
        if not sizes:
            raise ValueError("Vmap needs a mapped argument or axis_size")
        return sizes[0]

    def _expand_args(self, args, axes, n, m, batched):
        ops = get_ops()
        out = []
        for a, ax in zip(args, axes):
            if ax == 0:
                t = torch.as_tensor(a).to(ops.device())
                if t.dtype == torch.float64:
                    t = t.to(torch.float32)
                if batched and t.dim() >= 2 and t.shape[0] == n and t.shape[1] == m:
                    out.append(t.reshape((n * m,) + tuple(t.shape[2:])).contiguous())
                else:
                    out.append(t.repeat((n,) + (1,) * (t.dim() - 1)).contiguous())  # index i*m+j -> a[j]
            elif isinstance(a, torch.Tensor) and a.dim() >= 1 and a.shape[0] == n and n > 1:
                out.append(a.repeat_interleave(m, dim=0))  # per-particle value shared by its m elements
            else:
                out.append(a)
        return tuple(out)

    def _inner_keys(self, pk: ParticleKeys, m: int) -> ParticleKeys:
        t = get_ops().rng_split_each(pk.kb, pk.n, m)
        return ParticleKeys(KeyBatch(pk.impl, 0, tensor=t), pk.n * m)

    def _expand_constraint(self, constraint: ChoiceMap, n: int, m: int) -> ChoiceMap:
        """Constraints on a vmapped site.  Vectors over the whole mapped axis (`C[:, "x"].set(xs)`,
        test_choice_maps.py:1033): leaf [m] (all particles) or [n, m].  Constraints on SOME indices
        (`C[0, "x"].set(v)`, `C[jnp.array([0, 2]), "x"].set(vs)`; vmap.py:193-218 `get_submap(idx)` over an
        indexed choice map, test_vmap_combinator.py:84-106): the site receives a `Mask` — the constrained
        values scattered into a column plus the flags of the constrained elements — and samples the rest."""
        ops = get_ops()
        dev = ops.device()

        def expand(v):
            if isinstance(v, Mask):  # masked by an enclosing vmap: every element of a masked particle is masked
                return Mask(expand(v.value), v.flag.reshape(n, 1).expand(n, m).reshape(-1))
            t = torch.as_tensor(v).to(dev)
            if t.dtype == torch.float64:
                t = t.to(torch.float32)
            if t.dim() >= 2 and t.shape[0] == n and t.shape[1] == m:
                return t.reshape((n * m,) + tuple(t.shape[2:])).contiguous()
            if t.dim() >= 1 and t.shape[0] == m:
                return t.repeat((n,) + (1,) * (t.dim() - 1)).contiguous()
            raise ValueError(f"a constraint on a vmapped site needs a leading axis of {m} (got shape {tuple(t.shape)})")

        whole = ChoiceMap(constraint._value, {s: c for s, c in constraint._children.items() if not isinstance(s, int)})
        out = whole.map_leaves(expand)
        by_addr: dict = {}
        for j, sub in constraint._children.items():
            if not isinstance(j, int):
                continue
            if not 0 <= j < m:
                raise IndexError(f"constraint index {j} outside the mapped axis of length {m}")
            for addr, v in sub.leaves():
                by_addr.setdefault(addr, []).append((j, v))
        for addr, items in by_addr.items():
            v0 = items[0][1]
            first = torch.as_tensor(v0.value if isinstance(v0, Mask) else v0)
            vals = torch.zeros((n, m), dtype=torch.float32 if first.dtype.is_floating_point else first.dtype, device=dev)
            flag = torch.zeros((n, m), dtype=torch.bool, device=dev)
            for j, v in items:
                outer = v.flag.to(dev).reshape(-1) if isinstance(v, Mask) else None  # set by an enclosing vmap
                t = torch.as_tensor(v.value if isinstance(v, Mask) else v).to(dev).to(vals.dtype)
                vals[:, j] = t.reshape(-1) if t.dim() >= 1 and t.numel() == n else t.reshape(())
                flag[:, j] = True if outer is None else outer
            out = ChoiceMap.entry(Mask(vals.reshape(-1), flag.reshape(-1)), *addr) | out
        return out

    def simulate(self, key, args):
        pk, batched = as_particle_keys(key)
        axes = self._axes(args)
        m = self._length(args, axes, pk.n, batched)
        if m == 0:
            return EmptyVmapTrace(self, args, pk.n, batched)
        inner = self.gen_fn.simulate(self._inner_keys(pk, m), self._expand_args(args, axes, pk.n, m, batched))
        return VmapTrace(self, inner, args, pk.n, m, batched)

    def generate(self, key, constraint: ChoiceMap, args):
        pk, batched = as_particle_keys(key)
        axes = self._axes(args)
        m = self._length(args, axes, pk.n, batched)
        if m == 0:
            tr = EmptyVmapTrace(self, args, pk.n, batched)
            return tr, tr.get_score()
        inner, w = self.gen_fn.generate(self._inner_keys(pk, m), self._expand_constraint(constraint, pk.n, m),
                                        self._expand_args(args, axes, pk.n, m, batched))
        if isinstance(w, torch.Tensor) and w.dim() >= 1:
            w = w.reshape(pk.n, m).sum(1)
            w = w if batched else w[0]
        else:
            w = w * m
        return VmapTrace(self, inner, args, pk.n, m, batched), w

    def assess(self, sample: ChoiceMap, args):
        axes = self._axes(args)
        m = self._length(args, axes, 1, False)
        from . import lang as _lang

        _lang._ASSESS_BATCH.append(m)
        try:
            score, ret = self.gen_fn.assess(self._expand_constraint(sample, 1, m), self._expand_args(args, axes, 1, m, False))
        finally:
            _lang._ASSESS_BATCH.pop()
        if isinstance(score, torch.Tensor) and score.dim() >= 1:
            score = score.sum()
        return score, ret

    def project(self, key, trace: VmapTrace, selection: Selection):
        p = trace.inner.project(key, selection)
        if isinstance(p, torch.Tensor) and p.dim() >= 1:
            p = p.reshape(trace.n, trace.m).sum(1)
            return p if trace.batched else p[0]
        return p

    def edit(self, key, trace, edit_request, argdiffs):
        from .edit import IndexRequest, generic_index_request

        if isinstance(edit_request, IndexRequest):
            return generic_index_request(self, key, trace, edit_request.idx, edit_request.request, argdiffs, trace.m)
        return super().edit(key, trace, edit_request, argdiffs)


def vmap(*, in_axes=0):
    def decorator(f):
        from .lang import gen

        return Vmap(gen(f), in_axes=in_axes)

    return decorator
