"""The C-ABI is re-entrant (gjx.h: "thread-safe given distinct output buffers"): four host threads, each on its own HIP
stream, build their own plans (concurrent hiprtc specialisation under a module cache capped at 3), run importance passes
and a fused bootstrap filter, and must get bit for bit what one thread gets afterwards.  Child process: the cache cap is
read when the library starts."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, threading
sys.path.insert(0, sys.argv[1])
import torch
from genjax._amd import abi, prng, workloads as W
from genjax._amd.ops import KeyBatch
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
n = 50_000

def job(k):
    sites = W.gaussian10_sites(W.gaussian10_data())[: 4 + 2 * (k % 3)]
    sites[1].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.5 + 0.125 * k, None)   # a different kernel per job
    plan = ops.plan_create(sites)
    kb = W.importance_particle_keys(prng.key(100 + k, 1), n)
    nlat = sum(1 for s in sites if not s.observed)
    out = []
    for rep in range(3):
        vals, score, logw, mp = ops.importance_run(plan, kb, n, [], [torch.float32] * nlat)
        out.append((torch.stack(vals).clone(), score.clone(), logw.clone()))
    smc = W.lgssm_smc(ops, 1, seed=k, n=20_000 + 1000 * k, T=12, want_ancestors=True)
    hmm = W.hmm_smc(ops, 1, seed=k, n=10_000, T=6, n_states=32)
    # the generic weight operations: the SAME sizes in every thread (whatever scratch they use must not be shared)
    g = torch.Generator().manual_seed(k)
    lw = (torch.randn(300_000, generator=g) * 3).cuda()
    gen = []
    for rep in range(4):
        key = KeyBatch(1, 2, parent=(k, rep))
        a, m, q = ops.resample("systematic", key, lw, 300_000)
        a2, _, _ = ops.resample("multinomial", key, lw, 50_000)
        gen.append((a.clone(), m.clone(), q.clone(), a2.clone(), ops.categorical_index(key, lw, 1).clone(),
                    *[x.clone() for x in ops.logsumexp(lw)]))
    torch.cuda.current_stream().synchronize()
    return out, smc, hmm, gen

results, errors = {}, []
def work(k):
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            results[k] = job(k)
    except BaseException as ex:  # reported by the parent
        errors.append((k, repr(ex)))

threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
[t.start() for t in threads]
[t.join() for t in threads]
assert not errors, errors
torch.cuda.synchronize()
for k in range(4):
    out, smc, hmm, gen = job(k)      # one thread, default stream
    tout, tsmc, thmm, tgen = results[k]
    for a, b in zip(gen, tgen):
        for x, y in zip(a, b):
            assert torch.equal(x, y), (k, "generic weight operations")
    for (a, b, c), (ta, tb, tc) in zip(out, tout):
        assert torch.equal(a, ta) and torch.equal(b, tb) and torch.equal(c, tc), k
    for name in ("out_e", "out_q", "state", "logw", "ancestors"):
        assert torch.equal(smc[name], tsmc[name]), (k, name)
    if hmm is not None:
        for name in ("out_e", "out_q", "state", "logw"):
            assert torch.equal(hmm[name], thmm[name]), (k, name)
print("ok", ops.jit_stats())
"""


def test_four_threads_four_streams():
    env = dict(os.environ, GJX_JIT_CACHE_MAX="3")
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "genjax-chi_amd")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
