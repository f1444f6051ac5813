"""Host-API latency of one ImportanceK estimate (bench.bench_host_api_call) with a breakdown: the call without the scalar
read-back, the read-back, the key construction."""
import argparse
import statistics
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--rng", default="threefry")
ap.add_argument("--calls", type=int, default=200)
args = ap.parse_args()
print(bench.bench_host_api_call(args, calls=args.calls))

import torch  # noqa: E402

import genjax  # noqa: E402
from genjax import ChoiceMapBuilder as C, Target, gen, normal  # noqa: E402
from genjax._amd import workloads as W  # noqa: E402
from genjax.inference.smc import ImportanceK  # noqa: E402

y = W.gaussian10_data()


@gen
def model():
    for i in range(10):
        z = normal(0.0, 1.0) @ f"z{i}"
        _ = normal(z, 0.5) @ f"y{i}"


chm = C.n()
for i in range(10):
    chm = chm | C[f"y{i}"].set(float(y[i]))
alg = ImportanceK(Target(model, (), chm), k_particles=args.particles)
keys = [genjax.random.key(rep, args.rng) for rep in range(args.calls)]
for k in keys[:10]:
    float(alg.log_marginal_likelihood_estimate(k))
torch.cuda.synchronize()
t_key, t_call, t_read = [], [], []
for rep in range(args.calls):
    t0 = time.perf_counter()
    k = genjax.random.key(5000 + rep, args.rng)
    t1 = time.perf_counter()
    z = alg.log_marginal_likelihood_estimate(k)
    t2 = time.perf_counter()
    float(z)
    t3 = time.perf_counter()
    t_key.append(t1 - t0); t_call.append(t2 - t1); t_read.append(t3 - t2)
med = lambda v: statistics.median(v) * 1e6
print(f"key {med(t_key):.1f} us  call(async) {med(t_call):.1f} us  read-back {med(t_read):.1f} us")
# back-to-back calls without a read-back: the sustained host cost per call
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in keys:
    z = alg.log_marginal_likelihood_estimate(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"pipelined: host {1e6 * (t1 - t0) / len(keys):.1f} us/call, device-complete {1e6 * (t2 - t0) / len(keys):.1f} us/call")
