"""Lowering a user-written state-space model to the fused bootstrap-SMC kernels.

    @gen
    def init():                       # x_0 and its observation
        x = normal(0.0, 1.0) @ "x"
        normal(x, 0.5) @ "y"
        return x

    @gen
    def step(x):                      # one transition: kernel(carry) -> carry'
        x2 = normal(0.9 * x, 1.0) @ "x"
        normal(x2, 0.5) @ "y"
        return x2

    smc = BootstrapSMC(StateSpaceModel(init, step), C["y"].set(ys), n_particles=1_000_000)

Both bodies are run once with symbolic values (the same tracer as `plan.py`): the carry becomes
`GJX_ARG_STATE` references to the resampled ancestor's state columns, addresses present in the
observations become observed sites whose values are this step's observation constants
(`GJX_ARG_OBS`), and the returned carry becomes the next state expressions.  `gjx_smc_plan_create`
+ `gjx_smc_run_plan` then generate and run one fused resample+propagate+weight kernel per step.
The same restrictions as for importance plans apply (supported distributions, affine arguments)."""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import abi
from .choicemap import ChoiceMap
from .lang import GenerativeFunction, StaticGenerativeFunction
from .plan import PlanTracer, PlanUnsupported, Sym, _Table
from .runtime import get_ops


@dataclass(frozen=True)
class StateSpaceModel:
    """x_0 ~ init();  x_t ~ step(x_{t-1}).  `init` takes no arguments and returns the first carry;
    `step` takes the carry (a scalar or a tuple of up to 4 scalars) and returns the next one."""

    init: GenerativeFunction
    step: GenerativeFunction


class _SmcTracer(PlanTracer):
    """PlanTracer whose observed sites read per-step observation constants."""

    def __init__(self, obs_index: dict):
        super().__init__(ChoiceMap.empty(), 1, use_params=False)
        self.obs_index = obs_index  # full address (the calls' addresses, then the site's) -> observation column

    def _arg(self, v) -> abi.Arg:
        if isinstance(v, Sym) and v.src[0] == "state":
            return abi.Arg(abi.ARG_STATE, v.src[1], v.scale, v.offset, None)
        if isinstance(v, Sym) and v.src[0] == "obs":
            return abi.Arg(abi.ARG_OBS, v.src[1], v.scale, v.offset, None)
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.numel() > 1:
            raise PlanUnsupported("per-particle tensors cannot enter an SMC plan")
        return super()._arg(v)

    def _callee_constraint(self, a: tuple) -> ChoiceMap:
        return ChoiceMap.empty()  # (observed sites are recognised by their full address, see handle_trace)

    def handle_trace(self, addr, gen_fn, args):
        from .lang import Distribution

        local = addr if isinstance(addr, tuple) else (addr,)
        key = self.prefix + local
        if isinstance(gen_fn, Distribution) and key in self.obs_index:
            # constrain with a placeholder, then point the site's observed value at the obs vector
            self.constraint = ChoiceMap.entry(0.0, *local)
            out = super().handle_trace(addr, gen_fn, args)
            k = self.obs_index[key]
            self.sites[-1].obs = abi.Arg(abi.ARG_OBS, k, 1.0, 0.0, None)
            self.constraint = ChoiceMap.empty()
            is_int = self.meta[-1]["is_int"]
            return Sym(self, ("obs", k), is_int=is_int)
        self.constraint = ChoiceMap.empty()
        out = super().handle_trace(addr, gen_fn, args)
        if isinstance(gen_fn, Distribution):  # (a nested call's own sites have passed through here already)
            self.sites[-1].out_col = -1  # SMC plans keep state columns, not per-site columns
        return out


def _state_args(tracer: _SmcTracer, ret, n_expected: int | None):
    vals = ret if isinstance(ret, (tuple, list)) else (ret,)
    if n_expected is not None and len(vals) != n_expected:
        raise PlanUnsupported("init and step must return carries of the same length")
    if not 1 <= len(vals) <= abi.SMC_MAX_STATE:
        raise PlanUnsupported(f"the carry must have 1..{abi.SMC_MAX_STATE} components")
    out = []
    for v in vals:
        if isinstance(v, _Table):
            raise PlanUnsupported("a table lookup cannot be a carry component")
        out.append(tracer._arg(v))  # (an expression over sites / the carry / observations is a postfix program)
    return out


def build_smc_plan(model: StateSpaceModel, obs_addrs: list[tuple]):
    """-> (SmcPlan, n_state).  obs_addrs: addresses (tuples) observed at every step, in the column
    order of the observation matrix."""
    if not isinstance(model.init, StaticGenerativeFunction) or not isinstance(model.step, StaticGenerativeFunction):
        raise TypeError("StateSpaceModel needs `@gen` functions")
    if len(obs_addrs) > abi.SMC_MAX_OBS:
        raise PlanUnsupported(f"at most {abi.SMC_MAX_OBS} observed addresses per step")
    obs_index = {a: k for k, a in enumerate(obs_addrs)}
    ti = _SmcTracer(obs_index)
    init_ret = ti.run(model.init.source, ())
    init_state = _state_args(ti, init_ret, None)
    ts = _SmcTracer(obs_index)
    carry = tuple(Sym(ts, ("state", k)) for k in range(len(init_state)))
    step_ret = ts.run(model.step.source, (carry[0],) if len(carry) == 1 else (carry,))
    next_state = _state_args(ts, step_ret, len(init_state))
    seen = {m["path"] for m in ti.meta + ts.meta}
    missing = [a for a in obs_addrs if a not in seen]
    if missing:
        raise ValueError(f"observed addresses not visited by the model: {missing}")
    plan = get_ops().smc_plan_create(ti.sites, ts.sites, init_state, next_state, len(obs_addrs),
                                     init_scopes=[tuple(k) for k in ti.scopes], step_scopes=[tuple(k) for k in ts.scopes])
    plan._keep = (ti.keep, ts.keep)  # constant tables the site tables point into
    return plan, len(init_state)


def observation_matrix(observations, obs_addrs: list[tuple]) -> np.ndarray:
    """[T, n_obs] float32 from a ChoiceMap whose observed leaves are length-T vectors."""
    cols = []
    for a in obs_addrs:
        v = observations[a if len(a) > 1 else a[0]]
        cols.append(np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, dtype=np.float32).reshape(-1))
    T = cols[0].shape[0]
    if any(c.shape[0] != T for c in cols):
        raise ValueError("all observed sequences must have the same length")
    return np.stack(cols, axis=1)
