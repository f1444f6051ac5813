"""`ChoiceMap`, `ChoiceMapBuilder` and `Selection` — the static-address subset the importance /
SMC path uses (reference: core/generative/choice_map.py:124-663 Selection, 847-1393 ChoiceMap,
752-844 builder; behaviours pinned by tests/core/test_choice_maps.py).

Design: addresses resolve on the host at model-run time (they are Python strings / ints), so a
choice map is a plain trie — `dict[segment -> ChoiceMap]` plus an optional leaf value.  Leaves hold
whatever the model produced: Python scalars, or torch columns with the particle axis leading.
Nothing here touches the device; only the leaves are device tensors.

Out of scope: dynamic (traced-array) indices producing `Mask` values and
`ChoiceMap.switch`.  Static integer indices (`C[3, "z"]`) and full slices (`C[:, "x"]`) are
supported: a full slice adds no node (choice_map.py:1483-1485) and an integer index on a leaf with
a leading axis selects `leaf[idx]` (choice_map.py:1444-1450).
"""

from __future__ import annotations

import warnings
from typing import Any, Iterable

_FULL = slice(None, None, None)


class ChoiceMapNoValueAtAddress(Exception):
    """Raised by `chm[addr]` when there is no value at `addr` (choice_map.py:672-682)."""

    def __init__(self, addr):
        super().__init__(f"No value at address {addr!r}")
        self.subaddr = addr


def _as_addr(addr) -> tuple:
    if isinstance(addr, tuple):
        return addr
    return (addr,)


def _index_list(seg):
    """An array-valued address segment (`C[jnp.array([0, 2]), "x"]`, choice_map.py:1454-1531) as a list of
    ints, else None."""
    if isinstance(seg, (str, bool)) or seg is Ellipsis:
        return None
    if isinstance(seg, (list, range)):
        return [int(i) for i in seg]
    if hasattr(seg, "ndim") and hasattr(seg, "tolist") and getattr(seg, "ndim", 0) == 1:
        return [int(i) for i in seg.tolist()]
    return None


def _check_segment(seg):
    if _index_list(seg) is not None:
        return
    if seg is Ellipsis or isinstance(seg, (str, int)) or seg == _FULL:
        return
    if isinstance(seg, slice):
        raise ValueError(f"Partial slices not supported: {seg}")  # choice_map.py:745-747
    raise TypeError(f"Unsupported address segment {seg!r}")


# =================================================================================================
# Selection
# =================================================================================================
class Selection:
    """A set of addresses with prefix semantics: `S["x"]` selects everything under "x"."""

    # -- construction ----------------------------------------------------------------------------
    @staticmethod
    def all() -> "Selection":
        return _ALL

    @staticmethod
    def none() -> "Selection":
        return _NONE

    @staticmethod
    def leaf() -> "Selection":
        return _LeafSel()

    def extend(self, *addr) -> "Selection":
        sel = self
        for seg in reversed(addr):
            _check_segment(seg)
            if seg == _FULL:
                continue
            sel = _StaticSel(seg, sel) if not isinstance(sel, _NoneSel) else sel
        return sel

    # -- algebra ---------------------------------------------------------------------------------
    def __invert__(self) -> "Selection":
        if isinstance(self, _AllSel):
            return _NONE
        if isinstance(self, _NoneSel):
            return _ALL
        if isinstance(self, _ComplementSel):
            return self.inner
        return _ComplementSel(self)

    def __and__(self, other: "Selection") -> "Selection":
        if isinstance(self, _NoneSel) or isinstance(other, _NoneSel):
            return _NONE
        if isinstance(self, _AllSel):
            return other
        if isinstance(other, _AllSel) or self == other:
            return self
        return _AndSel(self, other)

    def __or__(self, other: "Selection") -> "Selection":
        if isinstance(self, _AllSel) or isinstance(other, _AllSel):
            return _ALL
        if isinstance(self, _NoneSel):
            return other
        if isinstance(other, _NoneSel) or self == other:
            return self
        return _OrSel(self, other)

    # -- queries ---------------------------------------------------------------------------------
    def __call__(self, *addr) -> "Selection":
        """Sub-selection under `addr`."""
        addr = addr[0] if len(addr) == 1 and isinstance(addr[0], tuple) else addr
        sel = self
        for seg in addr:
            sel = sel._sub(seg)
        return sel

    def __getitem__(self, addr) -> bool:
        return self(*_as_addr(addr))._check()

    def __contains__(self, addr) -> bool:
        return self[addr]

    def check(self) -> bool:
        return self._check()

    # implemented by subclasses
    def _sub(self, seg) -> "Selection":
        raise NotImplementedError

    def _check(self) -> bool:
        raise NotImplementedError

    def __eq__(self, other):
        return type(self) is type(other) and self.__dict__ == other.__dict__

    def __hash__(self):
        return hash((type(self).__name__, tuple(sorted((k, repr(v)) for k, v in self.__dict__.items()))))


class _AllSel(Selection):
    def _sub(self, seg):
        return self

    def _check(self):
        return True

    def __repr__(self):
        return "Selection.all()"


class _NoneSel(Selection):
    def _sub(self, seg):
        return self

    def _check(self):
        return False

    def __repr__(self):
        return "Selection.none()"


class _LeafSel(Selection):
    """Matches exactly the empty remaining address (`Selection.leaf()`)."""

    def _sub(self, seg):
        return _NONE

    def _check(self):
        return True

    def __repr__(self):
        return "Selection.leaf()"


class _StaticSel(Selection):
    def __init__(self, seg, inner: Selection):
        self.seg, self.inner = seg, inner

    def _sub(self, seg):
        if self.seg is Ellipsis or seg is Ellipsis or seg == self.seg:
            return self.inner
        return _NONE

    def _check(self):
        return False

    def __repr__(self):
        return f"S[{self.seg!r}]->{self.inner!r}"


class _ComplementSel(Selection):
    def __init__(self, inner: Selection):
        self.inner = inner

    def _sub(self, seg):
        return ~self.inner._sub(seg)

    def _check(self):
        return not self.inner._check()

    def __repr__(self):
        return f"~({self.inner!r})"


class _AndSel(Selection):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def _sub(self, seg):
        return self.a._sub(seg) & self.b._sub(seg)

    def _check(self):
        return self.a._check() and self.b._check()

    def __repr__(self):
        return f"({self.a!r} & {self.b!r})"


class _OrSel(Selection):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def _sub(self, seg):
        return self.a._sub(seg) | self.b._sub(seg)

    def _check(self):
        return self.a._check() or self.b._check()

    def __repr__(self):
        return f"({self.a!r} | {self.b!r})"


class _ChmSel(Selection):
    """Addresses that have a value in a choice map (choice_map.py:627-663)."""

    def __init__(self, chm: "ChoiceMap"):
        self.chm = chm

    def _sub(self, seg):
        sub = self.chm.get_submap(seg)
        return _NONE if sub.static_is_empty() else _ChmSel(sub)

    def _check(self):
        return self.chm.has_value()

    def __eq__(self, other):
        return isinstance(other, _ChmSel) and self.chm is other.chm

    def __hash__(self):
        return id(self.chm)

    def __repr__(self):
        return f"ChmSel({self.chm!r})"


_ALL, _NONE = _AllSel(), _NoneSel()


class _SelectionBuilder:
    """`S["x"]`, `S["x", "y"]`, `S[...]`; `S.all`, `S.none`, `S.leaf` (choice_map.py:78-114: properties of the builder)."""

    @property
    def all(self) -> Selection:
        return Selection.all()

    @property
    def none(self) -> Selection:
        return Selection.none()

    @property
    def leaf(self) -> Selection:
        return Selection.leaf()

    def __getitem__(self, addr) -> Selection:
        return _ALL.extend(*_as_addr(addr))


SelectionBuilder = _SelectionBuilder()


# =================================================================================================
# ChoiceMap
# =================================================================================================
class Mask:
    """A value that is only present where `flag` holds (core/generative/functional_types.py Mask): the
    constraint a vectorised site receives when only some of its elements are constrained.  `flag` is a
    Python bool or a bool tensor broadcastable against `value`'s leading axis."""

    __slots__ = ("value", "flag")

    def __init__(self, value, flag):
        self.value, self.flag = value, flag

    def primal_flag(self):
        return self.flag

    def __repr__(self):
        return f"Mask({_short(self.value)}, {_short(self.flag)})"


class ChoiceMap:
    """Immutable trie of random choices.  `_value` is the leaf payload (or None); `_children`
    maps one address segment to a sub-map."""

    __slots__ = ("_value", "_children")

    def __init__(self, value=None, children: dict | None = None):
        self._value = value
        self._children = children or {}

    # -- constructors ------------------------------------------------------------------------------
    @staticmethod
    def empty() -> "ChoiceMap":
        return _EMPTY

    @staticmethod
    def choice(v) -> "ChoiceMap":
        if isinstance(v, ChoiceMap):
            return v
        return ChoiceMap(value=v)

    @staticmethod
    def entry(v, *addr) -> "ChoiceMap":
        if isinstance(v, dict):
            v = ChoiceMap.d(v)
        chm = ChoiceMap.choice(v)
        return chm.extend(*addr)

    @staticmethod
    def from_mapping(pairs: Iterable[tuple[Any, Any]]) -> "ChoiceMap":
        acc = _EMPTY
        for addr, v in pairs:
            acc = acc | ChoiceMap.entry(v, *_as_addr(addr))
        return acc

    @staticmethod
    def d(d: dict) -> "ChoiceMap":
        return ChoiceMap.from_mapping(d.items())

    @staticmethod
    def kw(**kwargs) -> "ChoiceMap":
        return ChoiceMap.d(kwargs)

    # -- structure ---------------------------------------------------------------------------------
    def extend(self, *addr) -> "ChoiceMap":
        chm = self
        if chm.static_is_empty():
            return chm
        for seg in reversed(addr):
            _check_segment(seg)
            if seg == _FULL:
                continue  # a full slice adds no node: the leading axis is positional
            chm = ChoiceMap(children={seg: chm})
        return chm

    def static_is_empty(self) -> bool:
        return self._value is None and not self._children

    def has_value(self) -> bool:
        return self._value is not None

    def get_value(self):
        return self._value

    def get_submap(self, *addr) -> "ChoiceMap":
        addr = addr[0] if len(addr) == 1 and isinstance(addr[0], tuple) else addr
        chm = self
        for seg in addr:
            chm = chm._step(seg)
        return chm

    def _step(self, seg) -> "ChoiceMap":
        if seg == _FULL:
            return self
        if seg in self._children:
            return self._children[seg]
        if isinstance(seg, int) and not isinstance(seg, bool):
            # integer index into vector-valued leaves below this node (Scan / Vmap get_submap(idx))
            return self._index(seg)
        return _EMPTY

    def _index(self, idx: int) -> "ChoiceMap":
        if self.static_is_empty():
            return self
        value = None
        if self._value is not None:
            v = self._value
            value = v[idx] if hasattr(v, "__getitem__") and getattr(v, "ndim", 0) >= 1 else v
        kids = {}
        for seg, sub in self._children.items():
            if isinstance(seg, int):
                continue  # other explicit indices do not apply to idx
            s = sub._index(idx)
            if not s.static_is_empty():
                kids[seg] = s
        return ChoiceMap(value, kids) if (value is not None or kids) else _EMPTY

    def __call__(self, *addr) -> "ChoiceMap":
        return self.get_submap(*addr)

    def __getitem__(self, addr):
        sub = self.get_submap(*_as_addr(addr))
        if sub._value is None:
            raise ChoiceMapNoValueAtAddress(addr)
        return sub._value

    def __contains__(self, addr) -> bool:
        return self.get_submap(*_as_addr(addr))._value is not None

    def get_selection(self) -> Selection:
        return _NONE if self.static_is_empty() else _ChmSel(self)

    # -- combination -------------------------------------------------------------------------------
    def merge(self, other: "ChoiceMap") -> "ChoiceMap":
        return self | other

    def __or__(self, other: "ChoiceMap") -> "ChoiceMap":
        """Left-biased union (`Or`, choice_map.py:1683-1693)."""
        if other.static_is_empty():
            return self
        if self.static_is_empty():
            return other
        if (self._value is not None) != (other._value is not None) and (
            (self._value is not None and other._children) or (other._value is not None and self._children)
        ):
            raise Exception("Cannot merge a Choice with a non-Choice at the same address")
        value = self._value if self._value is not None else other._value
        kids = dict(self._children)
        for seg, sub in other._children.items():
            kids[seg] = (kids[seg] | sub) if seg in kids else sub
        return ChoiceMap(value, kids)

    def __add__(self, other):
        return self | other

    def __xor__(self, other):
        warnings.warn("^ is deprecated, please use | or _.merge(...) instead.", DeprecationWarning)
        return self | other

    def __and__(self, other: "ChoiceMap") -> "ChoiceMap":
        """Common addresses, right-hand values (test_choice_maps.py:764-793)."""
        if self.static_is_empty() or other.static_is_empty():
            return _EMPTY
        value = other._value if (self._value is not None and other._value is not None) else None
        kids = {}
        for seg, sub in self._children.items():
            if seg in other._children:
                s = sub & other._children[seg]
                if not s.static_is_empty():
                    kids[seg] = s
        return ChoiceMap(value, kids) if (value is not None or kids) else _EMPTY

    def filter(self, selection: Selection) -> "ChoiceMap":
        if self.static_is_empty() or isinstance(selection, _NoneSel):
            return _EMPTY
        if isinstance(selection, _AllSel):
            return self
        value = self._value if (self._value is not None and selection._check()) else None
        kids = {}
        for seg, sub in self._children.items():
            s = sub.filter(selection._sub(seg))
            if not s.static_is_empty():
                kids[seg] = s
        return ChoiceMap(value, kids) if (value is not None or kids) else _EMPTY

    def mask(self, flag) -> "ChoiceMap":
        """choice_map.py `mask`: static flags resolve now, tensor flags wrap every leaf in a `Mask`."""
        if isinstance(flag, bool):
            return self if flag else _EMPTY
        return self.map_leaves(lambda v: Mask(v.value, v.flag & flag) if isinstance(v, Mask) else Mask(v, flag))

    # -- builders ----------------------------------------------------------------------------------
    @property
    def at(self) -> "ChoiceMapBuilder":
        return ChoiceMapBuilder(self, ())

    def invalid_subset(self, gen_fn, args):
        """Addresses present here that `gen_fn(*args)` does not visit (extra addresses are
        reported, missing ones are fine: test_choice_maps.py:875-1052)."""
        from .lang import visited_addresses

        shape = visited_addresses(gen_fn, args)
        bad = _subtract(self, shape)
        return None if bad.static_is_empty() else bad

    # -- misc --------------------------------------------------------------------------------------
    def leaves(self, prefix=()):
        if self._value is not None:
            yield prefix, self._value
        for seg, sub in self._children.items():
            yield from sub.leaves(prefix + (seg,))

    def map_leaves(self, fn) -> "ChoiceMap":
        value = fn(self._value) if self._value is not None else None
        return ChoiceMap(value, {s: c.map_leaves(fn) for s, c in self._children.items()})

    def __eq__(self, other):
        if not isinstance(other, ChoiceMap):
            return NotImplemented
        a, b = dict(self.leaves()), dict(other.leaves())
        if a.keys() != b.keys():
            return False
        return all(_leaf_equal(a[k], b[k]) for k in a)

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return "ChoiceMap(" + ", ".join(f"{k!r}: {_short(v)}" for k, v in self.leaves()) + ")"


def _leaf_equal(x, y) -> bool:
    try:
        import torch

        if isinstance(x, torch.Tensor) or isinstance(y, torch.Tensor):
            return bool(torch.equal(torch.as_tensor(x).cpu(), torch.as_tensor(y).cpu()))
    except Exception:
        pass
    try:
        r = x == y
        return bool(r.all()) if hasattr(r, "all") else bool(r)
    except Exception:
        return False


def _short(v):
    shape = getattr(v, "shape", None)
    return f"<{type(v).__name__}{tuple(shape)}>" if shape is not None and len(shape) else repr(v)


def _subtract(chm: ChoiceMap, shape: ChoiceMap) -> ChoiceMap:
    """Part of `chm` whose addresses are not in `shape`."""
    if chm.static_is_empty():
        return _EMPTY
    if shape.has_value():
        return _EMPTY if not chm._children else ChoiceMap(children=chm._children)
    value = chm._value
    kids = {}
    for seg, sub in chm._children.items():
        s = _subtract(sub, shape._step(seg)) if not shape._step(seg).static_is_empty() else sub
        if not s.static_is_empty():
            kids[seg] = s
    return ChoiceMap(value, kids) if (value is not None or kids) else _EMPTY


_EMPTY = ChoiceMap()


class ChoiceMapBuilder:
    """`C["a", "b"].set(v)`, `C.v(v)`, `C.n()`, `C.d({...})`, `C.kw(...)`, and
    `chm.at["a"].set(v)` which extends AND overwrites (choice_map.py:752-844)."""

    def __init__(self, chm: ChoiceMap | None = None, addr: tuple = ()):
        self._chm, self._addr = chm, addr

    def __getitem__(self, addr) -> "ChoiceMapBuilder":
        addr = _as_addr(addr)
        for seg in addr:
            _check_segment(seg)
        return ChoiceMapBuilder(self._chm, addr)

    def set(self, v) -> ChoiceMap:
        for k, seg in enumerate(self._addr):
            idx = _index_list(seg)
            if idx is not None:  # C[array, ...].set(values): one entry per index, values along their leading axis
                new = _EMPTY
                for pos, i in enumerate(idx):
                    vi = v[pos] if hasattr(v, "__getitem__") and getattr(v, "ndim", 1) >= 1 and not isinstance(v, (str, dict)) else v
                    new = new | ChoiceMapBuilder(None, self._addr[:k] + (i,) + self._addr[k + 1:]).set(vi)
                if self._chm is None:
                    return new
                return new | self._chm
        new = ChoiceMap.entry(v, *self._addr)
        if self._chm is None:
            return new
        return new | _remove(self._chm, tuple(s for s in self._addr if s != _FULL))  # new value wins

    def update(self, fn, *args, **kwargs) -> ChoiceMap:
        assert self._chm is not None
        sub = self._chm.get_submap(*self._addr)
        cur = sub.get_value() if sub.has_value() else (None if sub.static_is_empty() else sub)
        return self.set(fn(cur, *args, **kwargs))

    def n(self) -> ChoiceMap:
        return _EMPTY

    def v(self, v) -> ChoiceMap:
        return self.set(v)

    def d(self, d: dict) -> ChoiceMap:
        return self.set(ChoiceMap.d(d))

    def kw(self, **kwargs) -> ChoiceMap:
        return self.set(ChoiceMap.kw(**kwargs))

    def from_mapping(self, pairs) -> ChoiceMap:
        return self.set(ChoiceMap.from_mapping(pairs))


def _remove(chm: ChoiceMap, addr: tuple) -> ChoiceMap:
    """`chm` without the subtree at `addr`."""
    if not addr:
        return _EMPTY
    seg = addr[0]
    if seg not in chm._children:
        return chm
    kids = dict(chm._children)
    rest = _remove(kids[seg], addr[1:])
    if rest.static_is_empty():
        del kids[seg]
    else:
        kids[seg] = rest
    return ChoiceMap(chm._value, kids) if (chm._value is not None or kids) else _EMPTY


C = ChoiceMapBuilder()
S = SelectionBuilder
