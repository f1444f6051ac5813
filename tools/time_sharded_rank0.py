import json, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import bench
args = bench.parse()
from genjax._amd.runtime import load_hip_ops
ops = load_hip_ops()
r = bench.bench_sharded_rank0_virtual(args, ops)
print(json.dumps(r, indent=1))
