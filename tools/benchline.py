"""Print a few fields of bench.py's JSON line (stdin): tag value ms_per_step kernel_ms log_z."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["log_z"])
