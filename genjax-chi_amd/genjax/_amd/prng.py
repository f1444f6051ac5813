"""Host side of the counter-based PRNG: scalar keys are plain Python integers and are derived on
the host (a few cipher blocks); batches of per-particle keys stay *lazy* — `split(key, n)` with a
large n returns a `KeyBatch` describing `split(key, *)[first + i]`, and the kernels derive each
particle's key in registers instead of reading keys from HBM.

Derivation semantics:
  impl="threefry"  jax.random's Threefry2x32 key tree with jax_threefry_partitionable
                   (reference call sites: inference/smc.py:154,171,299-300; static.py:349-352;
                   scan.py:267-268).  Keys are 2 words.
  impl="philox"    native Philox4x32-10 counter scheme.  A key is a 64-bit cipher key plus a 64-bit
                   *lane* that fills counter words 0,1 of every block: the children of a lane-0 key
                   are (same cipher key, lane i+1) — free, and a whole population shares one cipher
                   key — while the children of a laned key are hashed to fresh lane-0 keys.
The scalar ciphers below restate Salmon et al. (SC'11) and are checked against the Random123
known-answer vectors in tests/test_prng_host.py.
"""

from __future__ import annotations

from dataclasses import dataclass

from .ops import KeyBatch

M32 = 0xFFFFFFFF
THREEFRY, PHILOX = 0, 1
_IMPL = {"threefry": THREEFRY, "philox": PHILOX, THREEFRY: THREEFRY, PHILOX: PHILOX}
TAG_SPLIT, TAG_FOLD, TAG_DRAW, TAG_STREAM = 0x53, 0x46, 0x44, 0x52  # low byte of counter word 3

_default_impl = THREEFRY


def set_default_impl(impl) -> None:
    global _default_impl
    _default_impl = _IMPL[impl]


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M32


def _threefry_source() -> str:
    """The 20 rounds written out (no loop, no call): the host derives a handful of keys per inference call."""
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    lines = ["def threefry2x32(k0, k1, c0, c1):", "    k2 = 0x1BD11BDA ^ k0 ^ k1", "    x0 = (c0 + k0) & 0xFFFFFFFF",
             "    x1 = (c1 + k1) & 0xFFFFFFFF"]
    ks = ("k0", "k1", "k2")
    for blk in range(5):
        for r in rot[blk & 1]:
            lines.append("    x0 = (x0 + x1) & 0xFFFFFFFF")
            lines.append(f"    x1 = (((x1 << {r}) | (x1 >> {32 - r})) & 0xFFFFFFFF) ^ x0")
        lines.append(f"    x0 = (x0 + {ks[(blk + 1) % 3]}) & 0xFFFFFFFF")
        lines.append(f"    x1 = (x1 + {ks[(blk + 2) % 3]} + {blk + 1}) & 0xFFFFFFFF")
    lines.append("    return x0, x1")
    return "\n".join(lines)


_ns: dict = {}
exec(_threefry_source(), _ns)  # noqa: S102 - generated from the constants above
threefry2x32 = _ns["threefry2x32"]


def philox4x32(k0, k1, c0, c1, c2, c3):
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & M32, (p0 >> 32) ^ c3 ^ k1, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


@dataclass(frozen=True)
class PRNGKey:
    """A scalar key: two 32-bit cipher-key words, the derivation scheme and (philox) the lane."""

    k0: int
    k1: int
    impl: int = THREEFRY
    lane: int = 0

    def words(self) -> tuple[int, int]:
        return (self.k0, self.k1)

    def literal(self) -> KeyBatch:
        """This key as a 1-element key batch (by value)."""
        return KeyBatch(self.impl, 2, parent=(self.k0, self.k1), parent_lane=self.lane)

    def __repr__(self):
        lane = f", lane={self.lane}" if self.lane else ""
        return f"PRNGKey({self.k0:#010x}, {self.k1:#010x}, {'threefry' if self.impl == 0 else 'philox'}{lane})"


def key(seed: int, impl=None) -> PRNGKey:
    """jax.random.key(seed): key words (seed >> 32, seed & 0xffffffff)."""
    impl = _default_impl if impl is None else _IMPL[impl]
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return PRNGKey((seed >> 32) & M32, seed & M32, impl)


PRNGKeyLike = PRNGKey


def _philox_lane(k: PRNGKey, b: int, a_tag: int):
    return philox4x32(k.k0, k.k1, k.lane & M32, (k.lane >> 32) & M32, b & M32, a_tag & M32)


def split_at_words(k0: int, k1: int, impl: int, lane: int, i: int):
    """split_at on bare words: (k0, k1, lane) of split(key)[i] (what split_at wraps in a PRNGKey)."""
    if impl == THREEFRY:
        o0, o1 = threefry2x32(k0, k1, (i >> 32) & M32, i & M32)
        return o0, o1, 0
    if lane == 0:
        return k0, k1, i + 1
    o0, o1, _, _ = philox4x32(k0, k1, lane & M32, (lane >> 32) & M32, i & M32, (((i >> 32) << 8) | TAG_SPLIT) & M32)
    return o0, o1, 0


def split_at(k: PRNGKey, i: int) -> PRNGKey:
    if k.impl == THREEFRY:
        o0, o1 = threefry2x32(k.k0, k.k1, (i >> 32) & M32, i & M32)
        return PRNGKey(o0, o1, k.impl)
    if k.lane == 0:
        return PRNGKey(k.k0, k.k1, k.impl, lane=i + 1)
    o0, o1, _, _ = _philox_lane(k, i & M32, ((i >> 32) << 8) | TAG_SPLIT)
    return PRNGKey(o0, o1, k.impl)


def split(k: PRNGKey, num: int = 2):
    """jax.random.split for a scalar key -> tuple of scalar keys (host-derived)."""
    return tuple(split_at(k, i) for i in range(num))


def split_lazy(k: PRNGKey, n: int, first: int = 0) -> KeyBatch:
    """split(k, n_total)[first : first + n] as a lazy per-particle key batch."""
    return KeyBatch(k.impl, 1, parent=(k.k0, k.k1), first=first, parent_lane=k.lane)


def fold_in(k: PRNGKey, data: int) -> PRNGKey:
    d = int(data) & M32
    if k.impl == THREEFRY:
        o0, o1 = threefry2x32(k.k0, k.k1, 0, d)
    else:
        o0, o1, _, _ = _philox_lane(k, d, TAG_FOLD)
    return PRNGKey(o0, o1, k.impl)


# ---- vectorised host derivation (numpy): the 2T step / resample keys of an SMC run ------------------
def fold_words(k: PRNGKey, n: int):
    """[fold_in(k, d) for d in range(n)] as a uint32 array [n, 2] (fresh lane-0 keys) computed with numpy
    (same ciphers, vectorised over d).  For threefry this equals split(k, n) (TF(k, (0, d)) either way)."""
    import numpy as np

    idx = np.arange(n, dtype=np.uint64)
    m = np.uint64(M32)

    def rotl(x, r):
        return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & m

    if k.impl == THREEFRY:
        ks = (np.uint64(k.k0), np.uint64(k.k1), np.uint64(0x1BD11BDA ^ k.k0 ^ k.k1))
        x0 = np.full(n, ks[0], dtype=np.uint64)
        x1 = ((idx & m) + ks[1]) & m
        rot = ((13, 15, 26, 6), (17, 29, 16, 24))
        for blk in range(5):
            for r in rot[blk & 1]:
                x0 = (x0 + x1) & m
                x1 = rotl(x1, r) ^ x0
            x0 = (x0 + ks[(blk + 1) % 3]) & m
            x1 = (x1 + ks[(blk + 2) % 3] + np.uint64(blk + 1)) & m
        return np.stack([x0, x1], axis=1).astype(np.uint32)
    c0 = np.full(n, k.lane & M32, dtype=np.uint64)
    c1 = np.full(n, (k.lane >> 32) & M32, dtype=np.uint64)
    c2 = idx & m
    c3 = np.full(n, TAG_FOLD, dtype=np.uint64)
    k0, k1 = np.uint64(k.k0), np.uint64(k.k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & m, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & m
        k0 = (k0 + np.uint64(0x9E3779B9)) & m
        k1 = (k1 + np.uint64(0xBB67AE85)) & m
    return np.stack([c0, c1], axis=1).astype(np.uint32)
