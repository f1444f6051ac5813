"""The run-time specialisation cache is bounded: with GJX_JIT_CACHE_MAX=3 a process that keeps building NEW model
structures holds at most 3 unreferenced code objects, modules in use are never unloaded, and a structure seen before
is not compiled again while it is cached.  Runs in a child process (the cap is read when the library starts)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import torch
from genjax._amd import abi, prng, workloads as W
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
n = 2048
kb = W.importance_particle_keys(prng.key(1, 1), n)

def plan_for(scale):
    sites = W.gaussian10_sites(W.gaussian10_data())[:4]
    sites[1].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, scale, None)  # a different constant = a different kernel
    return ops.plan_create(sites)

def run(plan):
    vals, score, logw, mp = ops.importance_run(plan, kb, n, [], [torch.float32, torch.float32])
    return logw.clone()

first = plan_for(0.5)
ref = run(first)
s0 = ops.jit_stats()
for i in range(8):                       # eight more structures, each dropped after one run
    p = plan_for(0.6 + 0.05 * i)
    run(p)
    del p
s1 = ops.jit_stats()
assert s1["compiles"] - s0["compiles"] == 8, (s0, s1)
assert s1["cached_modules"] <= 3 + 1, s1       # the cap, plus the module `first` still references
assert s1["evictions"] >= 5, s1
assert torch.equal(run(first), ref)            # a referenced module survives every eviction
again = plan_for(0.5)                          # same structure as `first`: cached, no compilation
run(again)
assert ops.jit_stats()["compiles"] == s1["compiles"]
print("ok", s1)
"""


def test_bounded_module_cache():
    env = dict(os.environ, GJX_JIT_CACHE_MAX="3")
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "genjax-chi_amd")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
