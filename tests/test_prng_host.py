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


@pytest.mark.parametrize("impl", [0, 1])
def test_host_derivation_matches_oracle(oracle_ops, impl):
    k = prng.key(0x1234567890ABCDEF, impl)
    assert (k.k0, k.k1) == (0x12345678, 0x90ABCDEF)
    dev = oracle_ops.rng_keys(KeyBatch(impl, 1, parent=k.words(), first=5), 6).numpy().view(np.uint32)
    host = np.array([prng.split_at(k, 5 + i).words() for i in range(6)], dtype=np.uint32)
    assert (dev == host).all()
    assert (prng.split_words(k, 11)[5:] == host).all()
    f = oracle_ops.rng_keys(KeyBatch(impl, 2, parent=k.words()).with_fold(77), 1).numpy().view(np.uint32)[0]
    assert tuple(f) == prng.fold_in(k, 77).words()
    a, b = prng.split(k)
    assert a == prng.split_at(k, 0) and b == prng.split_at(k, 1)
