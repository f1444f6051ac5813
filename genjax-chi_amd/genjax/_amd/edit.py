"""Trace edits: `Update` requests, change-tagged values (`Diff`) and the per-site update walk.

Mirror of the reference's edit interface for the part of it an SMC driver uses to move a population of
particles to a new target:
  Diff / NoChange / UnknownChange .... core/compiler/interpreters/incremental.py:60-260
  EditRequest, Update ................ core/generative/concepts.py:95-167, generative_function.py:1688-1700
  Trace.edit / Trace.update .......... core/generative/generative_function.py:153-183
  GenerativeFunction.update .......... core/generative/generative_function.py:611-627
  Distribution edit_update ........... generative_functions/distributions/distribution.py:163-258, 302-340
  Static UpdateHandler ............... generative_functions/static.py:405-504, 827-865

Semantics (same as the reference): `update(key, trace, constraint, argdiffs)` returns
`(new_trace, w, retdiff, discard)` with `w = log p'(new choices; new args) − log p(old choices; old args)`
restricted to what changed (sites that appear for the first time are sampled from their prior and
contribute nothing), and `discard` the old values the constraint replaced.

Design difference, stated: the reference pushes `Diff` tags through the body with an incremental
interpreter so that untouched sites can be skipped; here a body runs once over the whole particle
population and every visited site re-evaluates its log-density column on the GPU (one `gjx_logpdf_*`
kernel per site) — an unchanged site re-evaluates to exactly its old score, so its weight increment is
exactly 0.  Inner sites therefore always receive `UnknownChange` arguments; return-value diffs are
`no_change` only when nothing was constrained and no argument changed.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import torch

from .choicemap import ChoiceMap


# ---- change tags ----------------------------------------------------------------------------------
class ChangeTangent:
    def __repr__(self):
        return type(self).__name__.lstrip("_")


class _UnknownChange(ChangeTangent):
    pass


class _NoChange(ChangeTangent):
    pass


UnknownChange = _UnknownChange()
NoChange = _NoChange()


def _tree_map(fn, tree):
    if isinstance(tree, Diff):
        return fn(tree)
    if isinstance(tree, tuple):
        return tuple(_tree_map(fn, x) for x in tree)
    if isinstance(tree, list):
        return [_tree_map(fn, x) for x in tree]
    if isinstance(tree, dict):
        return {k: _tree_map(fn, x) for k, x in tree.items()}
    return fn(tree)


def _tree_leaves(tree, out):
    if isinstance(tree, (tuple, list)):
        for x in tree:
            _tree_leaves(x, out)
    elif isinstance(tree, dict):
        for x in tree.values():
            _tree_leaves(x, out)
    else:
        out.append(tree)
    return out


@dataclass(frozen=True)
class Diff:
    """A value paired with a change tag (incremental.py:89-260).  Leaves of an argument tuple only."""

    primal: Any
    tangent: ChangeTangent

    def get_primal(self):
        return self.primal

    def get_tangent(self):
        return self.tangent

    @staticmethod
    def tree_primal(tree):
        return _tree_map(lambda v: v.primal if isinstance(v, Diff) else v, tree)

    @staticmethod
    def tree_tangent(tree):
        return _tree_map(lambda v: v.tangent if isinstance(v, Diff) else NoChange, tree)

    @staticmethod
    def tree_diff(tree, tangent_tree):
        flat_t = _tree_leaves(tangent_tree, [])
        it = iter(flat_t)
        return _tree_map(lambda p: Diff(p, next(it)), tree)

    @staticmethod
    def no_change(tree):
        return _tree_map(lambda p: Diff(p, NoChange), Diff.tree_primal(tree))

    @staticmethod
    def unknown_change(tree):
        return _tree_map(lambda p: Diff(p, UnknownChange), Diff.tree_primal(tree))

    @staticmethod
    def static_check_no_change(tree) -> bool:
        return all(not isinstance(v, Diff) or v.tangent is NoChange for v in _tree_leaves(tree, []))

    @staticmethod
    def static_check_tree_diff(tree) -> bool:
        return all(isinstance(v, Diff) for v in _tree_leaves(tree, []))


# ---- requests -------------------------------------------------------------------------------------
class NotSupportedEditRequest(Exception):
    """concepts.py:167."""

    def __init__(self, request):
        super().__init__(f"edit request not supported: {request!r}")
        self.request = request


class EditRequest:
    """concepts.py:95-140."""

    def edit(self, key, tr, argdiffs):
        raise NotImplementedError

    def dimap(self, *, pre=lambda v: v, post=lambda v: v):
        raise NotSupportedEditRequest(self)


class PrimitiveEditRequest(EditRequest):
    """A request a generative function answers itself (concepts.py:143-164)."""

    def edit(self, key, tr, argdiffs):
        return tr.get_gen_fn().edit(key, tr, self, argdiffs)


@dataclass(frozen=True)
class Update(PrimitiveEditRequest):
    """Move a trace to new arguments and / or new values at the constrained addresses
    (generative_function.py:1688-1700)."""

    constraint: ChoiceMap


@dataclass(frozen=True)
class Regenerate(PrimitiveEditRequest):
    """requests.py:64-66: re-draw the selected choices from their distributions at the (new) arguments, keep and
    re-score the others.  Answered by distributions (distribution.py:258-300: a selected site takes a fresh value, its
    weight is `new score − old score`, the old value is the backward `Update`), by the static language site by site
    (static.py:616-715: site keys `fold_in(key, counter)`, counter from 1) and by `Scan` / `Vmap` element-wise."""

    selection: Any


@dataclass(frozen=True)
class StaticRequest(PrimitiveEditRequest):
    """static.py:130-131: one edit request per address of a static generative function (addresses without an entry get
    `EmptyRequest`).  It is what a static `Regenerate` hands back as its backward request."""

    addressed: dict

    def __hash__(self):
        return id(self)


@dataclass(frozen=True)
class IndexRequest(PrimitiveEditRequest):
    """concepts.py:154-165: apply `request` to ONE element (`idx`) of a vector combinator's trace (`Scan` step,
    `Vmap` element); the other elements keep their choices (a later `Scan` step is re-scored where its carry moved).
    `request`: `Update(constraint)` or `Regenerate(selection)`.  Answered by re-generation (`generic_index_request`)."""

    idx: Any
    request: Any

    def __hash__(self):
        return id(self)


@dataclass(frozen=True)
class EmptyRequest(EditRequest):
    """requests.py:46-60: nothing changes unless the arguments did (then an `Update` with an empty constraint)."""

    def edit(self, key, tr, argdiffs):
        if Diff.static_check_no_change(argdiffs):
            return tr, 0.0, Diff.no_change(tr.get_retval()), EmptyRequest()
        return Update(ChoiceMap.empty()).edit(key, tr, argdiffs)


@dataclass(frozen=True)
class Rejuvenate(EditRequest):
    """inference/requests/rejuvenate.py:45-94: a Metropolis-Hastings move with a custom proposal, without the
    accept / reject step — the ratio is returned as the (SMCP3) weight.  `proposal.propose(key, argument_mapping(choices))`
    proposes new values, the trace is `Update`d to them, and the proposal is assessed at the discarded values:
    `w = update weight + log q(old | new) − log q(new | old)`.  Over a population every step is one kernel per site."""

    proposal: Any
    argument_mapping: Any

    def edit(self, key, tr, argdiffs):
        from .lang import split

        chm = tr.get_choices()
        fwd_args = self.argument_mapping(chm)
        key, sub_key = split(key)
        proposed, fwd_score, _ = self.proposal.propose(sub_key, fwd_args)
        new_tr, w, retdiff, bwd = Update(proposed).edit(key, tr, argdiffs)
        assert isinstance(bwd, Update)
        bwd_chm = bwd.constraint
        bwd_score, _ = self.proposal.assess(bwd_chm, self.argument_mapping(bwd_chm))
        return new_tr, w + bwd_score - fwd_score, retdiff, Rejuvenate(self.proposal, self.argument_mapping)

    def __hash__(self):
        return id(self)


# ---- the generic answer to Update ------------------------------------------------------------------
def generic_update(gen_fn, key, trace, constraint: ChoiceMap, argdiffs):
    """Update by re-generation: every old choice the constraint does not replace is constrained to its
    old value, so `generate` re-evaluates all log-densities at the new arguments and samples only
    addresses that did not exist before.  `generate`'s weight is then the log-density of everything but
    the fresh samples, and `w = that − old score` is the update weight.  Used by the combinators; the
    static language walks site by site (`UpdateHandler`)."""
    args = Diff.tree_primal(argdiffs)
    old = trace.get_choices()
    merged = constraint | old
    new_trace, gw = gen_fn.generate(key, merged, args)
    w = gw - trace.get_score()
    discard = ChoiceMap.empty()  # address by address: combinator choices are stacked leaves behind an index
    for addr, _ in constraint.leaves():
        if addr in old:
            discard = discard | ChoiceMap.entry(old[addr], *addr)
    unchanged = constraint.static_is_empty() and Diff.static_check_no_change(argdiffs)
    retval = new_trace.get_retval()
    return new_trace, w, (Diff.no_change(retval) if unchanged else Diff.unknown_change(retval)), Update(discard)


def generic_regenerate(gen_fn, key, trace, selection, argdiffs):
    """`Regenerate` by re-generation, for the combinators (scan.py:430-470, 613: the request is handed down to every
    element's kernel trace): the choices outside the selection are constrained to their old values, the selected ones
    are drawn afresh.  Under the reference's rule every site contributes `its new score − its old score` (a selected
    site's weight is the score of its fresh value minus the old one, distribution.py:268-280), so the weight of any
    structure is `new total score − old total score`; the backward request restores the discarded values."""
    args = Diff.tree_primal(argdiffs)
    old = trace.get_choices()
    kept = old.filter(~selection)
    new_trace, _ = gen_fn.generate(key, kept, args)
    w = new_trace.get_score() - trace.get_score()
    retval = new_trace.get_retval()
    return new_trace, w, Diff.unknown_change(retval), Update(old.filter(selection))


def generic_index_request(gen_fn, key, trace, idx, request, argdiffs, length: int):
    """`IndexRequest(idx, Update | Regenerate)` on a `Scan` / `Vmap` trace (scan.py:340-420, 627; vmap.py:260-332): every
    element is constrained to its old choices except element `idx`, whose selected addresses are drawn afresh
    (`Regenerate`) or take the given values (`Update`).  Weight = new total score − old total score — the reference's
    per-site rule (`new site score − old site score`, summed; sites after `idx` in a `Scan` are re-scored at the moved
    carry); the backward request puts the old values of element `idx` back."""
    idx = int(idx)
    if not 0 <= idx < length:
        raise IndexError(f"IndexRequest index {idx} outside the combinator's axis of length {length}")
    args = Diff.tree_primal(argdiffs)
    old = trace.get_choices()
    if isinstance(request, Regenerate):
        replaced = {a for a, _ in old.leaves() if request.selection[a]}
        new_vals = {}
    elif isinstance(request, Update):
        new_vals = dict(request.constraint.leaves())
        replaced = set(new_vals)
    else:
        raise NotSupportedEditRequest(request)
    # one explicit entry per element and address: old[..., j] for every element, except (idx, replaced addresses)
    constraint, discard = ChoiceMap.empty(), ChoiceMap.empty()
    for addr, v in old.leaves():
        for j in range(length):
            elem = v[..., j] if isinstance(v, torch.Tensor) and v.dim() >= 1 else v
            if j == idx and addr in replaced:
                discard = discard | ChoiceMap.entry(elem, *addr)
                if addr in new_vals:
                    constraint = constraint | ChoiceMap.entry(new_vals[addr], j, *addr)
                continue
            constraint = constraint | ChoiceMap.entry(elem, j, *addr)
    new_trace, _ = gen_fn.generate(key, constraint, args)
    w = new_trace.get_score() - trace.get_score()
    retval = new_trace.get_retval()
    return new_trace, w, Diff.unknown_change(retval), IndexRequest(idx, Update(discard))


def as_weight(w, like=None):
    """Weights may be Python floats (no constrained site touched) or [n] columns."""
    if isinstance(w, torch.Tensor) or like is None or not isinstance(like, torch.Tensor):
        return w
    return torch.zeros_like(like) + w
