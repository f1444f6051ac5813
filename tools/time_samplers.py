"""The rejection samplers under ImportanceK (bench.py: bench_site_model): python tools/time_samplers.py [--no-cpu]"""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
nocpu = "--no-cpu" in sys.argv
sys.argv_orig = list(sys.argv)
sys.argv = ["bench.py"] + (["--no-cpu-baseline"] if nocpu else [])
import bench
args = bench.parse()
from genjax._amd.runtime import load_hip_ops
ops = load_hip_ops()
only = [a for a in sys.argv_orig if a in ("beta_bernoulli", "gamma_normal")] if hasattr(sys, "argv_orig") else []
for nm in (only or ("beta_bernoulli", "gamma_normal")):
    r = bench.bench_site_model(args, ops, nm)
    print(nm, json.dumps({k: r[k] for k in r if k not in ("roofline", "config")}), "frac", r["roofline"]["frac"], flush=True)
