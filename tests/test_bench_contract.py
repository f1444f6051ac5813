"""bench.py's output contract, without a GPU: exactly one JSON line on stdout whatever libraries print there, and the
headline line still comes out when the multi-GPU extra never returns."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = f"""
import importlib.util, os, sys, time
spec = importlib.util.spec_from_file_location("bench", {os.path.join(ROOT, "bench.py")!r})
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
"""


def run(body: str):
    p = subprocess.run([sys.executable, "-c", PRELUDE + body], capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout, p.stderr


def test_stdout_is_reserved_for_the_line():
    rc, out, err = run("""
bench.reserve_stdout()
os.write(1, b"banner written by a C library\\n")   # what RCCL does at communicator creation
print("and a stray Python print")
bench.emit_line({"metric": "m", "value": 1.0})
""")
    assert rc == 0
    lines = out.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.0}
    assert "banner written by a C library" in err and "stray Python print" in err


def test_deadline_prints_the_headline_and_leaves():
    rc, out, err = run("""
bench.reserve_stdout()
args = type("A", (), {"workload": "importance"})()
g = bench.ExtrasDeadline(0, {"metric": "m", "value": 2.0, "n_gpus": 2}, 0.2, "smc_lgssm_sharded", args)
time.sleep(30)          # a rank stuck in a collective
bench.emit_line({"never": "reached"})
""")
    assert rc == 3  # a hang is a failed run: the headline is printed, the status is non-zero
    lines = out.splitlines()
    assert len(lines) == 1
    o = json.loads(lines[0])
    assert o["value"] == 2.0 and "error" in o["extra"]["smc_lgssm_sharded"]
    assert "deadline" in err


def test_deadline_not_rank0_prints_nothing_and_finish_disarms():
    rc, out, err = run("""
g = bench.ExtrasDeadline(1, None, 0.2, "x")
time.sleep(30)
""")
    assert rc == 3 and out == ""
    rc, out, err = run("""
g = bench.ExtrasDeadline(0, {"value": 3.0}, 0.3, "x")
assert g.claim_line() and not g.claim_line()
bench.emit_line({"value": 3.0, "extra": {"x": 1}})
g.finish()
time.sleep(0.8)
print("alive", file=sys.stderr)
""")
    assert rc == 0 and "alive" in err
    assert [json.loads(l) for l in out.splitlines()] == [{"value": 3.0, "extra": {"x": 1}}]


def test_gpus_n_without_n_gpus_fails_loudly():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs: exit status 2 and a message, never a line that says
    n_gpus 1 (the parent process counts devices without initialising the GPU runtime and starts nothing)."""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("this box has the GPUs: the command would run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2 and p.stdout.strip() == "" and "refusing" in p.stderr
