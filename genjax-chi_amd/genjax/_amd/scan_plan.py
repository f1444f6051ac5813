"""Lowering `kernel.scan(n=T)` to ONE launch per importance pass (`gjx_scan_run`).

The reference's `Scan.generate` (generative_functions/combinators/scan.py:237-294) is a `lax.scan` over T steps:
`key_t = fold_in(key_{t-1}, t)`, `kernel.generate(key_t, constraint(t), (carry, x_t))`, weights and scores summed.
The general path here (combinators.py) is a host loop of T x per-site launches.  When the kernel generative function
is plan-able — supported distributions, affine arguments (plan.py) — the whole loop becomes one kernel: the carry is
`GJX_ARG_STATE`, the per-step observations and scanned inputs are one `[T, n_obs]` table (`GJX_ARG_OBS`), the returned
carry gives the next-state expressions, and every sampled value streams into a time-major `[T, n]` column.  Same key
chain, same device functions in the same order: the trace, the weights and the score equal the host loop's bit for
bit (tests/host_api_cases.py::case_scan_fused_equals_loop).

Not lowered (the host loop runs instead): constraints that differ in shape between steps or are per-particle, nested
generative functions in the kernel, a `y_t` output that reads a carry component which is not itself a stored float site
value, division by numbers and transcendental functions of traced values (plan.py).
"""

from __future__ import annotations

import numpy as np
import torch

from . import abi
from .choicemap import ChoiceMap
from .lang import StaticGenerativeFunction, ParticleKeys
from .plan import PlanTracer, PlanUnsupported, Sym, SymExpr, _Table
from .runtime import get_ops


class _ScanTracer(PlanTracer):
    """PlanTracer of one scan step: observed sites read this step's row of the observation table."""

    def __init__(self, obs_index: dict):
        super().__init__(ChoiceMap.empty(), 1, use_params=False)
        self.obs_index = obs_index  # full address (the calls' addresses, then the site's) -> column of the observation table

    def _arg(self, v) -> abi.Arg:
        if isinstance(v, Sym) and v.src[0] == "state":
            return abi.Arg(abi.ARG_STATE, v.src[1], v.scale, v.offset, None)
        if isinstance(v, Sym) and v.src[0] == "obs":
            return abi.Arg(abi.ARG_OBS, v.src[1], v.scale, v.offset, None)
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.numel() > 1:
            raise PlanUnsupported("per-particle tensors cannot enter a scan plan")
        return super()._arg(v)

    def _callee_constraint(self, a: tuple) -> ChoiceMap:
        return ChoiceMap.empty()  # (observed sites are recognised by their full address, see handle_trace)

    def handle_trace(self, addr, gen_fn, args):
        from .lang import Distribution

        local = addr if isinstance(addr, tuple) else (addr,)
        key = self.prefix + local
        if isinstance(gen_fn, Distribution) and key in self.obs_index:
            self.constraint = ChoiceMap.entry(0.0, *local)  # a placeholder; the value is this step's table entry
            super().handle_trace(addr, gen_fn, args)
            k = self.obs_index[key]
            self.sites[-1].obs = abi.Arg(abi.ARG_OBS, k, 1.0, 0.0, None)
            self.constraint = ChoiceMap.empty()
            return Sym(self, ("obs", k), is_int=self.meta[-1]["is_int"])
        self.constraint = ChoiceMap.empty()
        return super().handle_trace(addr, gen_fn, args)


def _flatten(v, out: list):
    """Leaves of a carry / xs pytree (tuples, lists, dicts) in a fixed order; returns a rebuild function."""
    if isinstance(v, (tuple, list)):
        fs = [_flatten(x, out) for x in v]
        return lambda leaves, fs=fs, ty=type(v): ty(f(leaves) for f in fs)
    if isinstance(v, dict):
        fs = {k: _flatten(x, out) for k, x in v.items()}
        return lambda leaves, fs=fs: {k: f(leaves) for k, f in fs.items()}
    idx = len(out)
    out.append(v)
    return lambda leaves, idx=idx: leaves[idx]


def _is_scalar(v) -> bool:
    return isinstance(v, (bool, int, float)) or (isinstance(v, torch.Tensor) and v.dim() == 0)


class ScanLowering:
    """The traced form of one `Scan`'s kernel for one constraint shape."""

    def __init__(self, tracer, plan, obs_addrs, n_xs, rebuild_carry, ret_carry, ret_y, value_meta):
        self.tracer, self.plan, self.obs_addrs, self.n_xs = tracer, plan, obs_addrs, n_xs
        self.rebuild_carry, self.ret_carry, self.ret_y, self.value_meta = rebuild_carry, ret_carry, ret_y, value_meta


def lower_scan(kernel_gen_fn, carry0, xs, obs_addrs: list[tuple], fast_math: bool = False) -> ScanLowering:
    """Trace `kernel(carry, x_t)` once with symbolic carry / inputs / observed values.  Raises PlanUnsupported."""
    if not isinstance(kernel_gen_fn, StaticGenerativeFunction):
        raise PlanUnsupported("scan kernel is not a @gen function")
    carry_leaves: list = []
    rebuild_carry = _flatten(carry0, carry_leaves)
    if not 1 <= len(carry_leaves) <= abi.SMC_MAX_STATE:
        raise PlanUnsupported(f"the carry must have 1..{abi.SMC_MAX_STATE} scalar components")
    xs_leaves: list = []
    rebuild_xs = _flatten(xs, xs_leaves) if xs is not None else None
    n_obs_total = len(obs_addrs) + len(xs_leaves)
    if n_obs_total > abi.SMC_MAX_OBS:
        raise PlanUnsupported(f"at most {abi.SMC_MAX_OBS} observed addresses + scanned inputs per step")
    obs_index = {a: k for k, a in enumerate(obs_addrs)}
    tr = _ScanTracer(obs_index)
    carry_sym = rebuild_carry([Sym(tr, ("state", k)) for k in range(len(carry_leaves))])
    x_sym = rebuild_xs([Sym(tr, ("obs", len(obs_addrs) + k)) for k in range(len(xs_leaves))]) if xs is not None else None
    ret = tr.run(kernel_gen_fn.source, (carry_sym, x_sym))
    if not (isinstance(ret, tuple) and len(ret) == 2):
        raise PlanUnsupported("a scan kernel returns (carry, y)")
    new_carry, y = ret
    nc_leaves: list = []
    _flatten(new_carry, nc_leaves)
    if len(nc_leaves) != len(carry_leaves):
        raise PlanUnsupported("the kernel must return a carry of the shape it received")
    next_state = []
    for v in nc_leaves:
        if isinstance(v, _Table):
            raise PlanUnsupported("a table lookup cannot be a carry component")
        next_state.append(tr._arg(v))  # (an expression over sites / the carry / the input is a postfix program)
    # a carry component that IS a stored float site value can be read back step by step (the carry before step t is the
    # site's value at step t - 1, the initial carry at t = 0): y_t may then read the previous carry
    state_col = []
    for v in nc_leaves:
        m = tr.meta[v.src[1]] if isinstance(v, Sym) and v.src[0] == "site" and not (v.has_mul or v.has_add) else None
        state_col.append(m["out_col"] if m is not None and m["out_col"] >= 0 and not m["is_int"] else None)
    y_leaves: list = []
    _flatten(y, y_leaves)
    for v in y_leaves:  # y_t must be recoverable from what the launch stores
        if isinstance(v, Sym) and v.src[0] == "state" and state_col[v.src[1]] is None:
            raise PlanUnsupported("y_t reads a carry component that is not a stored site value")
        if isinstance(v, _Table):
            raise PlanUnsupported("y_t is a table lookup")
        if isinstance(v, Sym) and v.src[0] == "site" and tr.meta[v.src[1]]["out_col"] < 0:
            raise PlanUnsupported("y_t reads a site that is not stored")
        if isinstance(v, SymExpr):
            for op, ref, _ in v.prog:
                if (op == abi.EXPR_STATE and state_col[ref] is None) or (op == abi.EXPR_SITE and tr.meta[ref]["out_col"] < 0):
                    raise PlanUnsupported("y_t reads a carry component / a site that is not stored")
    seen = {m["path"] for m in tr.meta}
    if any(a not in seen for a in obs_addrs):
        raise PlanUnsupported("a constrained address is not visited by the kernel")
    if not tr.sites:
        raise PlanUnsupported("no sites")
    plan = get_ops().scan_plan_create(tr.sites, next_state, n_obs_total, fast_math=fast_math, scopes=[tuple(k) for k in tr.scopes])
    plan._keep = tr.keep
    value_meta = [m for m in tr.meta if m["out_col"] >= 0]
    low = ScanLowering(tr, plan, obs_addrs, len(xs_leaves), rebuild_carry, new_carry, y, value_meta)
    low.state_col = state_col
    # carry components that no particle-dependent value ever reaches (a step counter, a schedule): presented as
    # scalars, like the host loop's Python values
    uniform = [_is_scalar(v) for v in carry_leaves]
    for _ in range(len(uniform)):
        for k, a in enumerate(next_state):
            if a.kind == abi.ARG_SITE or a.kind == abi.ARG_TABLE or (a.kind == abi.ARG_STATE and not uniform[a.ref]):
                uniform[k] = False
            if a.kind == abi.ARG_EXPR and any(op == abi.EXPR_SITE or (op == abi.EXPR_STATE and not uniform[ref])
                                              for op, ref, _ in tr.expr_progs[a.table]):
                uniform[k] = False
    low.uniform_carry = uniform
    # a carry component that IS an integer-valued site (an HMM's state) is presented in the site's dtype (the kernel's
    # state columns are f32)
    low.carry_dtypes = []
    for v in nc_leaves:
        m = tr.meta[v.src[1]] if isinstance(v, Sym) and v.src[0] == "site" and not (v.has_mul or v.has_add) else None
        low.carry_dtypes.append(m["dtype"] if m is not None and m["is_int"] else None)
    return low


def step_constraints(constraint: ChoiceMap, T: int):
    """-> (obs_addrs, per-address list of the T per-step values) when every step constrains the same scalar leaves;
    raises PlanUnsupported otherwise."""
    per_step = [dict(constraint.get_submap(t).leaves()) for t in range(T)]
    addrs = list(per_step[0].keys())
    for d in per_step:
        if list(d.keys()) != addrs:
            raise PlanUnsupported("the constrained addresses differ between steps")
        for v in d.values():
            if not _is_scalar(v):
                raise PlanUnsupported("per-particle or structured constraint")
    return [a if isinstance(a, tuple) else (a,) for a in addrs], [[d[a] for d in per_step] for a in addrs]


def _to_f32_host(vals: list) -> np.ndarray:
    if any(isinstance(v, torch.Tensor) for v in vals):
        return torch.stack([torch.as_tensor(v, dtype=torch.float32).cpu() if not isinstance(v, torch.Tensor)
                            else v.detach().to(torch.float32).cpu() for v in vals]).numpy()
    return np.asarray([float(v) for v in vals], dtype=np.float32)


def observation_table(obs_values: list[list], xs, T: int) -> np.ndarray:
    """[T, n_obs] float32: the constrained values (one column per address), then the scanned inputs' leaves."""
    cols = [_to_f32_host(v) for v in obs_values]
    if xs is not None:
        leaves: list = []
        _flatten(xs, leaves)
        for v in leaves:
            v = torch.as_tensor(v)
            if v.dim() != 1 or v.shape[0] != T:
                raise PlanUnsupported("scanned inputs must be length-T vectors")
            cols.append(v.detach().to(torch.float32).cpu().numpy())
    if not cols:
        return np.zeros((T, 0), dtype=np.float32)
    return np.ascontiguousarray(np.stack(cols, axis=1))


def run_scan(low: ScanLowering, pk: ParticleKeys, T: int, carry0, table: np.ndarray, want_score=True, out=None):
    """One `gjx_scan_run` launch -> the raw outputs dict of `Ops.scan_run`."""
    leaves: list = []
    _flatten(carry0, leaves)
    for v in leaves:
        if not (_is_scalar(v) or (isinstance(v, torch.Tensor) and v.dim() == 1 and v.shape[0] == pk.n)):
            raise PlanUnsupported("carry components must be scalars or per-particle columns")
    dtypes = [torch.int32 if m["is_int"] else torch.float32 for m in low.value_meta]
    return get_ops().scan_run(low.plan, pk.kb, pk.n, T, table, leaves, dtypes, want_score=want_score, out=out)


def resolve(low: ScanLowering, x, values_nt: list, table: np.ndarray, device, carry0=None):
    """A symbolic per-step output -> its [n, T] (or [T]) tensor, with the f32 operation order of the kernel.
    `carry0`: the initial carry (outputs that read the previous carry, `state_col` of lower_scan)."""
    def previous_carry(k):  # [n, T]: the carry before every step
        col = values_nt[low.state_col[k]]
        leaves: list = []
        _flatten(carry0, leaves)
        c0 = torch.as_tensor(leaves[k], dtype=torch.float32).to(col.device)
        c0 = c0.reshape(-1, 1).expand(col.shape[0], 1)
        return torch.cat([c0, col[:, :-1]], dim=1)

    if isinstance(x, SymExpr):
        def leaf(kind, ref):
            if kind == abi.EXPR_SITE:
                return values_nt[low.tracer.meta[ref]["out_col"]]
            if kind == abi.EXPR_OBS:
                return torch.from_numpy(table[:, ref].copy()).to(device)
            if kind == abi.EXPR_STATE and carry0 is not None:
                return previous_carry(ref)
            raise PlanUnsupported("unresolvable output")
        return x.evaluate(leaf)
    if isinstance(x, Sym):
        if x.src[0] == "state" and carry0 is not None:
            base = previous_carry(x.src[1])
            if x.has_mul:
                base = base * x.scale
            if x.has_add:
                base = base + x.offset
            return base
        if x.src[0] == "site":
            m = low.tracer.meta[x.src[1]]
            if not (x.has_mul or x.has_add):
                return values_nt[m["out_col"]]  # the site's own value, in its presented dtype
            base = values_nt[m["out_col"]].to(torch.float32)
        elif x.src[0] == "obs":
            base = torch.from_numpy(table[:, x.src[1]].copy()).to(device)
        else:
            raise PlanUnsupported("unresolvable output")
        out = base
        if x.has_mul:
            out = out * x.scale
        if x.has_add:
            out = out + x.offset
        return out
    if isinstance(x, (tuple, list)):
        return type(x)(resolve(low, y, values_nt, table, device, carry0) for y in x)
    if isinstance(x, dict):
        return {k: resolve(low, y, values_nt, table, device, carry0) for k, y in x.items()}
    if x is None:
        return None
    if _is_scalar(x):
        return torch.as_tensor(x).expand(table.shape[0]) if not isinstance(x, torch.Tensor) else x.expand(table.shape[0])
    raise PlanUnsupported("unresolvable output")
