"""Host-side PRNG (pure Python scalars + numpy-vectorised split): Random123 known answers and
agreement with the oracle's key derivation."""

import json
import os

import numpy as np
import pytest

from genjax._amd import prng
from genjax._amd.ops import KeyBatch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rng_kat.json")


def test_host_ciphers_known_answers():
    kat = json.load(open(GOLD))
    for v in kat["threefry2x32_20"]:
        k, c = [int(x, 16) for x in v["key"]], [int(x, 16) for x in v["ctr"]]
        assert [f"{x:08x}" for x in prng.threefry2x32(*k, *c)] == v["out"]
    for v in kat["philox4x32_10"]:
        k, c = [int(x, 16) for x in v["key"]], [int(x, 16) for x in v["ctr"]]
        assert [f"{x:08x}" for x in prng.philox4x32(*k, *c)] == v["out"]


def _words(k: prng.PRNGKey):
    """The materialised form of a scalar key (gjx.h GJX_KEY_WORDS): threefry 2 words, philox + lane."""
    return (k.k0, k.k1) if k.impl == prng.THREEFRY else (k.k0, k.k1, k.lane & 0xFFFFFFFF, k.lane >> 32)


@pytest.mark.parametrize("impl", [0, 1])
def test_host_derivation_matches_oracle(oracle_ops, impl):
    k = prng.key(0x1234567890ABCDEF, impl)
    assert (k.k0, k.k1, k.lane) == (0x12345678, 0x90ABCDEF, 0)
    big = (1 << 33) + 5  # indices beyond 32 bits
    for parent in (k, prng.split_at(k, 3)):  # a lane-0 parent and (philox) a laned one
        kb = KeyBatch(impl, 1, parent=parent.words(), first=big, parent_lane=parent.lane)
        dev = oracle_ops.rng_keys(kb, 6).numpy().view(np.uint32)
        host = np.array([_words(prng.split_at(parent, big + i)) for i in range(6)], dtype=np.uint32)
        assert (dev == host).all()
        lit = KeyBatch(impl, 2, parent=parent.words(), parent_lane=parent.lane)
        f = oracle_ops.rng_keys(lit.with_fold(77), 1).numpy().view(np.uint32)[0]
        assert tuple(f) == _words(prng.fold_in(parent, 77)) and prng.fold_in(parent, 77).lane == 0
        assert (prng.fold_words(parent, 9) == np.array([prng.fold_in(parent, d).words() for d in range(9)])).all()
        # nested split: every element of the batch split 3 ways == scalar split of the scalar child
        se = oracle_ops.rng_split_each(kb, 2, 3).numpy().view(np.uint32).reshape(2, 3, -1)
        for i in range(2):
            for j in range(3):
                assert tuple(se[i, j]) == _words(prng.split_at(prng.split_at(parent, big + i), j))
    a, b = prng.split(k)
    assert a == prng.split_at(k, 0) and b == prng.split_at(k, 1)
    if impl == prng.PHILOX:
        # children of a lane-0 key keep its cipher key (lane i+1); children of a laned key are fresh lane-0 keys
        assert (b.k0, b.k1, b.lane) == (k.k0, k.k1, 2)
        c = prng.split_at(b, 1)
        assert c.lane == 0 and (c.k0, c.k1) != (k.k0, k.k1)
        assert len({_words(x) for x in (k, a, b, c, prng.split_at(a, 0), prng.split_at(a, 1), prng.fold_in(k, 1),
                                         prng.fold_in(a, 1), prng.fold_in(k, 2))}) == 9
    else:
        assert (prng.fold_words(k, 7) == np.array([prng.split_at(k, d).words() for d in range(7)])).all()
