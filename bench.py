#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload importance|smc_lgssm|smc_hmm]
                  [--rng philox|threefry] [--no-cpu-baseline] [--no-extra]

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
  importance (default, BASELINE configs[1]): ImportanceK on the 10-latent Gaussian model, 1e6 particles per
      GPU: fused `@gen`-body kernel (RNG -> samplers -> SoA trace columns -> score, log-weights, row sums) +
      the fold of the row sums into the log-marginal.  value = particles/s.
  smc_lgssm (configs[2]): bootstrap SMC, T=100, ONE filter of 1e6 particles.  value = particle-steps/s.
  smc_hmm   (configs[4]): 256-state HMM, T=500, ONE filter of 1e6 particles.

Protocol: W untimed warm-up steps, then a block of EXACTLY K steps bracketed by barrier + synchronize on both
sides (max over ranks).  A K-step block of a 18 us step is far too short to carry a number, so the block is
repeated (each repetition bracketed the same way) until at least 50 ms and 5 blocks have been timed; `value`
and `ms_per_step` are the MEDIAN block, `timed_blocks` / `block_ms_{min,median,max}` report the rest.  The
dominant kernel is sampled with HIP events on the launch stream over all timed blocks (>= 20 samples; median).

`--gpus N` with N > 1: launched by `torch.distributed.run` (RANK / WORLD_SIZE in the environment) this process
is one rank; launched plainly it is the PARENT, which starts N rank processes before any GPU call, never touches
the GPU itself, and exits with their worst status.  Rank 0 prints ONE JSON line.  The N > 1 line carries the
sharded ImportanceK headline and `extra.smc_lgssm_sharded` (BASELINE configs[3]: 1e6 particles per GPU).
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# Untimed passes before the W warm-up steps.  Default 0: the driver's W is the only warm-up; the device's clock settling
# (tens of ms under load) is absorbed by timing MORE blocks instead (MIN_TIMED_S) and reporting the median.
RAMP_PASSES = int(os.environ.get("GJX_BENCH_RAMP", "0"))
N_PER_GPU = 1_000_000
MIN_TIMED_S = float(os.environ.get("GJX_BENCH_MIN_S", "0.15"))
# Algorithmic bytes per unit (SURVEY §8d / DESIGN.md §5).  SURVEY counts 52 B/particle for the pass (the last 4 are
# the log-sum-exp's re-read of logw, which the fused row-anchored partial sums made unnecessary).
BYTES_IMPORTANCE_KERNEL_PER_PARTICLE = 48  # 10 latent columns + score + logw written
BYTES_SMC_PER_PARTICLE_STEP = 44  # SURVEY §8d contract figure of the UNFUSED pipeline (the fused step moves ~20)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1024)
    p.add_argument("--warmup", type=int, default=64)
    p.add_argument("--workload", default="importance", choices=["importance", "smc_lgssm", "smc_hmm", "scan_lgssm", "scan_hmm"])
    p.add_argument("--fast-math", action="store_true", help="importance / scan: the opt-in hardware-transcendental plan (profiling passes)")
    p.add_argument("--rng", default="philox", choices=["philox", "threefry"])
    p.add_argument("--particles", type=int, default=N_PER_GPU, help="particles per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true")
    return p.parse_args()


FORCE_DIST = bool(os.environ.get("GJX_BENCH_FORCE_DIST"))  # exercise the N>1 code path with one rank
# REHEARSE: `--gpus N --workload smc_lgssm` with all N rank processes on the box's ONE device and a gloo process group — the
# N > 1 line (brackets, max over ranks, the peer transport through hipIpc, its validation) end to end where RCCL cannot run
# (it refuses two ranks on one GPU).  The reference log Z is then the single-rank filter of the same n_total.  Not a
# measurement: the ranks share the device.  tests/test_gpu_sharded.py runs it.
REHEARSE = bool(os.environ.get("GJX_BENCH_REHEARSE"))


# ------------------------------------------------------------------------------------------------------------
# process layout
# ------------------------------------------------------------------------------------------------------------
def parent_launch(args) -> int:
    """`python bench.py --gpus N` started plainly: start N rank processes (one per GPU) and wait.  The parent
    makes no GPU call (counting devices does not initialise the runtime on this image)."""
    import socket

    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not (REHEARSE and have >= 1):
        print(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to print a "
              f"line for fewer ranks than asked", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % have if REHEARSE else r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def init_dist(n_gpus):
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        print(f"bench.py: --gpus {n_gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if REHEARSE:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return rank, world


def barrier_sync(world):
    import torch

    if world > 1 or FORCE_DIST:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(seconds, world):
    if world == 1 and not FORCE_DIST:
        return seconds
    import torch
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if REHEARSE else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


BRACKET_MIN_S = float(os.environ.get("GJX_BENCH_BRACKET_MIN_S", "0.005"))
_EMPTY_BRACKET = {}


def empty_bracket_s(world):
    """The cost of the bracket itself (barrier + synchronize on both sides + the max-over-ranks all-reduce) with NOTHING
    inside: median of 20.  Reported in the line (`bracket.empty_ms`) so that a reader can see how much of a short block
    it would be; at N > 1 the blocks are made long enough (BRACKET_MIN_S) for it not to matter."""
    if world not in _EMPTY_BRACKET:
        ts = []
        for _ in range(20):
            barrier_sync(world)
            t0 = time.perf_counter()
            barrier_sync(world)
            ts.append(max_over_ranks(time.perf_counter() - t0, world))
        _EMPTY_BRACKET[world] = statistics.median(ts)
    return _EMPTY_BRACKET[world]


def timed_blocks(run_block, world, min_s=MIN_TIMED_S, min_blocks=5, max_blocks=2000):
    """Repeat `run_block()` (EXACTLY the K steps of the protocol), bracketed by barrier + synchronize on both sides; the
    duration of a bracket is the max over ranks (the same number on every rank, so all ranks stop together).  One rank:
    one K-step block per bracket.  Several ranks (or a forced one-rank group): a barrier costs tens of microseconds over
    RCCL — more than a short block — so a bracket holds M back-to-back K-step blocks, M chosen once (from the first
    bracket, agreed through the max over ranks) such that a bracket lasts at least BRACKET_MIN_S; the block time reported
    is bracket / M.  Returns the list of per-block durations in seconds and the last block's result."""
    blocks, out = [], None
    reps = 1
    multi = world > 1 or FORCE_DIST
    while True:
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = run_block()
        barrier_sync(world)
        dt = max_over_ranks(time.perf_counter() - t0, world)
        if multi and reps == 1 and dt < BRACKET_MIN_S and not blocks:
            reps = max(2, min(4096, int(BRACKET_MIN_S / max(dt, 1e-6)) + 1))  # (the first, short bracket is dropped)
            continue
        blocks.append(dt / reps)
        if (sum(blocks) * reps >= min_s and len(blocks) >= min_blocks) or len(blocks) >= max_blocks:
            timed_blocks.last_reps = reps
            return blocks, out


timed_blocks.last_reps = 1


def block_stats(blocks):
    return {"timed_blocks": len(blocks), "block_ms_min": min(blocks) * 1e3, "block_ms_median": statistics.median(blocks) * 1e3,
            "block_ms_max": max(blocks) * 1e3, "blocks_per_bracket": timed_blocks.last_reps}


def pmc_traffic(pattern: str, pick):
    """HBM bytes from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, collected by
    profiles/collect.sh on this same command) — newest round first; (None, None) if nothing matches."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            v = pick(json.load(open(f)))
            if v:
                return v, os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


# ------------------------------------------------------------------------------------------------------------
# ImportanceK (BASELINE configs[1])
# ------------------------------------------------------------------------------------------------------------
def bench_importance(args, ops, rank, world, launch_passes=None, rng=None, steps=None, warmup=None, ramp=True,
                     min_s=MIN_TIMED_S, fast_math=False):
    from genjax._amd import workloads as W
    from genjax._amd.ops import HipEvent

    rng = rng or args.rng
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    impl = 1 if rng == "philox" else 0
    sharded = world > 1 or FORCE_DIST
    total_particles = args.particles * world
    if sharded:
        from genjax._amd import dist as gdist

        # weak scaling: world x 1e6 particles split at 256-particle row boundaries (per-rank counts differ
        # by at most one row), which keeps the exact log-normaliser independent of the number of ranks
        first, n = gdist.shard_rows(total_particles, rank, world)
    else:
        first, n = 0, args.particles
    wl = W.Gaussian10(ops, impl, seed=0, n_local=n, first=first, n_total=total_particles, fast_math=fast_math)
    # BATCH passes share one fold launch (and, sharded, one exchanged block of records): 32 amortises a ~35 us
    # per-batch cost to ~1 us per pass.  LAUNCH independent passes (seeds 0, 1, ...) share one importance launch: a
    # single 1e6-particle pass is under two rounds of the machine, eight keep it full.  Persistent output buffers +
    # pre-marshalled C calls: no host allocation per step.
    # (as many passes per launch as the timed block has, up to the 32 a launch takes: fewer gaps between launches)
    default_launch = max(1, min(32, steps))
    LAUNCH = launch_passes if launch_passes else int(os.environ.get("GJX_BENCH_LAUNCH", str(default_launch)))
    BATCH = int(os.environ.get("GJX_BENCH_BATCH", "32")) // LAUNCH * LAUNCH or LAUNCH
    prep = wl.prepare(fold_batch=BATCH, passes=LAUNCH)
    if sharded:
        # each pass leaves a 65-word record of its shard's weights; one asynchronous all-gather per BATCH
        # passes overlaps with the next batch's kernels (genjax/_amd/dist.py)
        pipe = gdist.BatchedImportance(ops, wl, batch=BATCH, world=world, always_exchange=True, passes=LAUNCH)
    last_buf = [0]
    # HIP events (created before anything is timed) bracket every SAMPLE_EVERY-th timed launch on the launch stream
    ev_pool = [(HipEvent(), HipEvent()) for _ in range(256)]
    samples = []  # (start, stop, passes in the launch)
    launch_no = [0]
    SAMPLE_EVERY = 3

    def on_launch(phase, count, evs, timed):
        if phase == 0:
            launch_no[0] += 1 if timed else 0
            if timed and ev_pool and launch_no[0] % SAMPLE_EVERY == 1:
                evs.append(ev_pool.pop() + (count,))
                evs[-1][0].record(ops.stream())
            else:
                evs.append(None)
        elif evs[-1] is not None:
            evs[-1][1].record(ops.stream())
            samples.append(evs[-1])

    def run_batch(count, timed):
        """`count` (<= BATCH) passes: ceil(count / LAUNCH) importance launches + one fold launch."""
        evs = []
        if sharded:
            last_buf[0] = pipe.run(count, on_launch=lambda ph, c: on_launch(ph, c, evs, timed))
            return
        st, done = ops.stream(), 0
        while done < count:
            c = min(LAUNCH, count - done)
            on_launch(0, c, evs, timed)
            prep.launch_passes(done, c, st)  # trace columns, score, log-weights, row sums of c passes (slots done..)
            on_launch(1, c, evs, timed)
            done += c
        prep.launch_fold(count, st)  # one launch folds the 3907 (anchor, sum) pairs of each pass of the batch

    def run_steps(count, timed):
        if count <= 0:
            return None
        done = 0
        while done < count:
            c = min(BATCH, count - done)
            run_batch(c, timed)
            done += c
        if not sharded:
            return prep.e_all[:1], prep.q_all[:1]
        pipe.wait()  # every exchange has landed (stream-ordered; the host does not block)
        _, e_all, q_all = pipe.results(last_buf[0])
        return e_all[:1], q_all[:1]

    # Clock ramp: the device reaches its sustained clocks only after tens of milliseconds of load, so a fixed untimed
    # run of the same launches precedes the W warm-up steps (GJX_BENCH_RAMP=0 turns it off).
    if ramp:
        run_steps(max(0, RAMP_PASSES - warmup), False)
    run_steps(warmup, False)
    # An event record is a barrier packet in the HIP queue; a back-to-back pair with nothing in between measures
    # that fixed cost, which is subtracted from the kernel intervals.
    cal = [(HipEvent(), HipEvent()) for _ in range(16)]
    for a, b in cal:
        a.record(ops.stream())
        b.record(ops.stream())
    barrier_sync(world)
    ev_overhead_ms = statistics.median(a.elapsed_ms(b) for a, b in cal)
    blocks, (e, q) = timed_blocks(lambda: run_steps(steps, True), world, min_s=min_s)  # each block: EXACTLY `steps` passes
    dt = statistics.median(blocks)
    # the dominant kernel: launches of `c` passes each; full launches (c == LAUNCH) define the quoted duration
    full = [(a, b, c) for a, b, c in samples if c == min(LAUNCH, steps)] or samples
    raw = sorted(a.elapsed_ms(b) for a, b, _ in full)
    k_ms_raw = statistics.median(raw)
    k_ms = max(k_ms_raw - ev_overhead_ms, 1e-6)
    passes_per_launch = full[0][2]
    log_z = ops.log_z_from_rows(e, q, total_particles)  # exact (anchor, fixed-point sum) pair of pass 0
    bytes_per_launch = BYTES_IMPORTANCE_KERNEL_PER_PARTICLE * n * passes_per_launch
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    traffic, traffic_src, pmc_insts = (None, None, None)
    if n == N_PER_GPU:
        sel = (lambda pj: pj.get("fast_math", {})) if fast_math else (lambda pj: pj)
        per_pass, traffic_src = pmc_traffic(
            "r*_pmc.json", lambda pj: sel(pj).get("hbm_bytes_per_pass") if f"gjx_plan_kernel_{rng}" in sel(pj).get("kernel", "") else None)
        traffic = per_pass * passes_per_launch if per_pass else None
        # instruction counts of the same kernel from the committed SQ passes (per 1e6-particle pass)
        pmc_insts, _ = pmc_traffic(
            "r*_pmc.json", lambda pj: sel(pj).get("sq_per_pass") if f"gjx_plan_kernel_{rng}" in sel(pj).get("kernel", "") else None)
    res = {
        "metric": "particles/sec, ImportanceK log-marginal-likelihood estimate (1e6 particles per GPU)",
        "value": total_particles * steps / dt,
        "unit": "particles/s",
        "ms_per_step": dt / steps * 1e3,
        "config": {"workload": "ImportanceK k_particles=1e6/GPU on a 10-latent Gaussian model (BASELINE configs[1])",
                   "particles_per_gpu": args.particles, "latent_sites": 10, "observed_sites": 10, "rng": rng,
                   "math": "fast (hardware log/exp/sqrt/sin/cos, 1e-5 rel)" if fast_math else "bit-exact spec",
                   "passes_per_launch": LAUNCH, "passes_per_fold": BATCH,
                   "clock_ramp_passes_before_warmup": max(0, RAMP_PASSES - warmup) if ramp else 0,
                   "parallelism": (f"particle-sharded x{world}, row-aligned; one 520 B all-gather per pass, "
                                   f"bucketed x{BATCH} and overlapped with the next batch") if sharded
                   else "single device"},
        "roofline": {"bound": "hbm", "kernel": f"gjx_plan_kernel_{rng}", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel_ms": k_ms, "kernel_ms_raw_event_interval": k_ms_raw, "kernel_ms_min": max(raw[0] - ev_overhead_ms, 0.0),
                     "event_pair_overhead_ms": ev_overhead_ms, "kernel_launches_timed": len(full), "statistic": "median",
                     "passes_per_launch": passes_per_launch, "kernel_ms_per_pass": k_ms / passes_per_launch,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "limiter": "VALU issue / dependency latency (samplers), not HBM: PMC traffic == algorithmic bytes"},
        "log_z": log_z,
        "log_z_exact": W.gaussian10_exact_log_z(wl.y),
    }
    if pmc_insts:
        res["roofline"]["pmc_instructions_per_pass"] = {
            k: pmc_insts[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_WAVES") if k in pmc_insts}
    res.update(block_stats(blocks))
    return res, wl


# ------------------------------------------------------------------------------------------------------------
# bootstrap SMC (BASELINE configs[2], [4]; sharded: configs[3])
# ------------------------------------------------------------------------------------------------------------
def bench_smc_sharded(args, ops, rank, world, kind):
    """BASELINE configs[3] (and its HMM sibling): the filter sharded over the ranks, 1e6 particles per GPU.  Driver: the
    library's own (`gjx_smc_sharded_run_*` over a `gjx_comm` RCCL communicator: per step ONE launch, the all-gather of
    the tile records, the grouped send/recv of the ancestor shuffle — no interpreter between the launches);
    GJX_BENCH_PY_COMM=1 selects the Python `torch.distributed` loop instead (the same protocol, 2-3x the host time)."""
    from genjax._amd import dist as gdist

    impl = 1 if args.rng == "philox" else 0
    T = 100 if kind == "smc_lgssm" else 500
    n = -(-args.particles // ops.tile) * ops.tile  # the sharded filter exchanges whole 1024-particle tiles
    n_total = n * world
    exchange = os.environ.get("GJX_BENCH_SHUFFLE", "ranges")
    # (a forced one-rank group still issues every collective: the RCCL calls of the N > 1 path run on a one-GPU box)
    if REHEARSE:  # (no RCCL with several ranks on one device: the reference is the single-rank filter of the same population)
        native, smc, run = True, None, None
        ref = gdist.ShardedSMC(ops, kind[4:], impl, 1 if kind == "smc_lgssm" else 2, n_total, T, 0, 1, exchange=exchange,
                               comm=gdist.TorchComm(0, 1)).run()
    else:
        smc = gdist.ShardedSMC(ops, kind[4:], impl, 1 if kind == "smc_lgssm" else 2, n_total, T, rank, world, exchange=exchange,
                               comm=gdist.TorchComm(rank, world, always=FORCE_DIST))
        native = os.environ.get("GJX_BENCH_PY_COMM") != "1"
        comm = gdist.NativeComm.rccl(ops, rank, world) if native else None
        run = (lambda: smc.run_native(comm)) if native else smc.run
        ref = run()  # warm-up (also builds any generated kernels)
    # r04: the PEER transport (no collective, nothing decided on the host: DESIGN.md 6) is the N > 1 default when it proves
    # itself HERE: every rank maps its peers' arenas through hipIpc, runs the same filter through it, and every rank's log Z
    # must equal the RCCL run's bit for bit (the transports only move data).  Anything else — a mapping that fails, a wait
    # that times out, a different bit — and the line is measured through RCCL and says why.  GJX_BENCH_TRANSPORT=rccl|peers.
    transport, why = "rccl", None
    want = os.environ.get("GJX_BENCH_TRANSPORT", "auto")
    if native and world > 1 and want in ("auto", "peers"):
        import torch
        import torch.distributed as dist

        sdt = torch.float32 if kind == "smc_lgssm" else torch.int32
        reasons = []
        for fine in (False, True):  # ordinary device memory first; fine-grained arenas as the second chance
            ok, smc_p, why1 = 1, None, None
            try:
                arena = gdist.PeerArena.ipc(ops, rank, world, n_total, [sdt], False, fine_grained=fine)
                pcomm = gdist.NativeComm.peers(ops, arena, None, False, timeout_ms=5000)  # (an arrival word takes microseconds)
                smc_p = gdist.ShardedSMC(ops, kind[4:], impl, 1 if kind == "smc_lgssm" else 2, n_total, T, rank, world, arena=arena)
                got = smc_p.run_native(pcomm)
                if got["log_z"] != ref["log_z"]:
                    ok, why1 = 0, f"peer transport log Z {got['log_z']!r} != RCCL {ref['log_z']!r}"
            except Exception as ex:  # noqa: BLE001 - reported in the line
                ok, why1 = 0, f"{type(ex).__name__}: {ex}"[:160]
            votes = [None] * world
            dist.all_gather_object(votes, (ok, why1))
            if all(v[0] for v in votes):
                transport = "peers" + ("(fine-grained arenas)" if fine else "")
                run = (lambda sp=smc_p, pc=pcomm: sp.run_native(pc))
                break
            reasons.append(("fine-grained: " if fine else "") + "; ".join(f"rank {r}: {v[1]}" for r, v in enumerate(votes) if not v[0]))
        why = " | ".join(reasons) if transport == "rccl" else None
        if transport == "rccl" and (want == "peers" or REHEARSE):
            raise RuntimeError("the peer transport did not validate: " + str(why))
    if smc is not None:
        smc.received = 0
    runs = [0]

    def one_run():
        runs[0] += 1
        return run()

    blocks, r = timed_blocks(one_run, world, min_s=0.08, min_blocks=20, max_blocks=60)
    dt = statistics.median(blocks)
    per_step_ms = dt * 1e3 / T
    achieved = BYTES_SMC_PER_PARTICLE_STEP * n / (per_step_ms * 1e-3) / 1e9
    res = {
        "metric": "particle-steps/sec, bootstrap SMC (1e6 particles per GPU)",
        "value": n_total * T / dt, "unit": "particle-steps/s", "ms_per_step": per_step_ms,
        "config": {"workload": f"bootstrap SMC {kind} T={T} N={n_total} sharded x{world} (BASELINE configs[3] at world=8)",
                   "rng": args.rng, "shuffle": exchange,
                   "parallelism": (f"particle-sharded x{world}, peer transport: per step 1 step launch reading remote windows in place + 1 signal launch; no collective"
                                   if transport.startswith("peers") else
                                   f"particle-sharded x{world}: 1 launch + ONE all-gather(tile records) + ancestor shuffle per step"),
                   "driver": "native" if native else "python", "transport": transport,
                   **({"peer_transport_not_used_because": why} if why else {}),
                   "particles_received_per_rank_step": r["received"] / (1 if native else runs[0]) / max(1, T - 1)},
        "roofline": {"bound": "hbm", "kernel": "one sharded SMC step incl. exchange (per GPU)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "step_ms": per_step_ms, "algorithmic_bytes_per_launch": BYTES_SMC_PER_PARTICLE_STEP * n},
        "log_z": r["log_z"], "log_z_exact": r["log_z_exact"],
        "bracket": {"empty_ms": empty_bracket_s(world) * 1e3, "min_ms": BRACKET_MIN_S * 1e3},
    }
    res.update(block_stats(blocks))
    return res


def bench_site_model(args, ops, name, passes=8, min_s=0.05):
    """the rejection samplers under ImportanceK at 1e6 particles — `beta_bernoulli` (the README model:
    p ~ Beta(2, 2), flip(p) observed: two Marsaglia-Tsang gammas per particle) and `gamma_normal` (s ~ Gamma(3, 2) as the scale
    of five observed normals).  `passes` independent passes per launch; kernel time by HIP events; 12 algorithmic bytes per
    particle (the latent, score, log-weight) — these kernels are bound by the samplers' vector instructions, the roofline
    fraction says how far.  CPU baseline: the oracle on the host cores, a bounded sample; log Z: GPU == oracle bit for bit."""
    import torch

    from genjax._amd import workloads as W
    from genjax._amd.ops import HipEvent

    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    mk = W.beta_bernoulli_model if name == "beta_bernoulli" else W.gamma_normal_model
    wl = mk(ops, impl, 0, n)
    out = {}
    for L in (passes, 1):
        prep = wl.prepare(fold_batch=L, passes=L)
        st = ops.stream()

        def run():
            prep.launch_passes(0, L, st)
            prep.launch_fold(L, st)

        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:
            run()
            torch.cuda.synchronize()
        evs = []

        def one():
            e0, e1 = HipEvent(), HipEvent()
            e0.record(st)
            prep.launch_passes(0, L, st)
            e1.record(st)
            prep.launch_fold(L, st)
            evs.append((e0, e1))
            return None

        blocks, _ = timed_blocks(one, 1, min_s=min_s, min_blocks=20, max_blocks=200)
        k_ms = statistics.median(a.elapsed_ms(b) for a, b in evs)
        dt = statistics.median(blocks)
        bytes_per_launch = 12.0 * n * L
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        log_z = ops.log_z_from_rows(prep.e_all[:1], prep.q_all[:1], n)
        e = {"value": n * L / dt, "unit": "particles/s", "kernel_ms": k_ms, "kernel_us_per_pass": k_ms * 1e3 / L, "passes_per_launch": L,
             "roofline": {"bound": "hbm", "kernel": f"gjx_plan_kernel_{args.rng}", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                          "limiter": "vector instructions of the rejection sampler (cipher block + normal + two logs per attempt), not HBM"},
             "log_z": log_z, "log_z_exact": wl.log_z_exact}
        if L == passes:
            out.update(e)
        else:
            out["one_pass_per_launch"] = {k: e[k] for k in ("value", "unit", "kernel_ms", "kernel_us_per_pass")}
    out["config"] = {"workload": f"ImportanceK k_particles={n} on {name}", "rng": args.rng}
    if not args.no_cpu_baseline:
        ora = _oracle()
        if ora is not None:
            ow = mk(ora, impl, 0, n)
            o = ow.step()
            hc = host_cores()
            cores, _ = best_thread_count(ow.step, sorted({hc, max(1, hc // 2), min(hc, 16)}, reverse=True))
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 3.0:
                o = ow.step()
                reps += 1
            cdt = time.perf_counter() - t0
            clz = ora.log_z_from_rows(o["row_e"], o["row_q"], n)
            out["cpu_baseline"] = {"value": n * reps / cdt, "unit": "particles/s", "cores": cores, "kind": "port",
                                   "sample": f"{reps} full passes of the same {n}-particle workload (OpenMP, {cores} threads)",
                                   "log_z_abs_err_gpu_vs_cpu": abs(clz - out["log_z"])}
    return out


def bench_sharded_rank0_virtual(args, ops, world=8, steps=40):
    """what ONE rank of BASELINE configs[3] costs per step, measured on the one GPU.  A world-8 filter of
    8 x 1e6 particles (7 816 tile records: beyond the 1024 a workgroup merges itself) is stepped by 8 VIRTUAL ranks — threads
    of this process whose launches share the device and one stream, so they execute back to back and the device time of the
    whole run divided by (8 x steps) is what one rank's launches of one step take on a GPU of its own, less the xGMI latency
    of its remote windows (one device cannot show that).  Two transports: the peer transport (per step and rank: group-record
    launch that first waits for the peers + step launch reading remote windows where they live + signal launch writing the
    records into all 8 arenas; no collective) and the collective protocol over the virtual-rank copy transport (the kernels an
    RCCL rank launches: group records + step + range kernel, with device copies standing in for the all-gather and the
    send/recv).  Host time per step: rank 0's enqueue loop run alone (what the CPU spends per step and rank)."""
    import ctypes as C
    import threading

    import torch

    from genjax._amd import dist as gdist, prng, workloads as W
    from genjax._amd.ops import HipEvent

    impl = 1 if args.rng == "philox" else 0
    n = -(-args.particles // ops.tile) * ops.tile
    n_total = n * world
    out = {}

    def run_ranks(make, T):
        res, err = [None] * world, []

        def work(r):
            try:
                res[r] = make(r, T)()
            except BaseException as e:  # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        if err:
            raise err[0]
        return res

    arenas = gdist.PeerArena.virtual(ops, world, n_total, [torch.float32], False)
    pcomms = gdist.NativeComm.peers_virtual(ops, arenas, timeout_ms=20000)
    lcomms = gdist.NativeComm.local_group(ops, world)
    sh = gdist.ThreadComm.Shared(world)

    def make_peers(r, T):
        smc = gdist.ShardedSMC(ops, "lgssm", impl, 1, n_total, T, r, world, arena=arenas[r])
        return lambda: smc.run_native(pcomms[r])

    lsmc = {}

    def make_local(r, T):
        if (r, T) not in lsmc:
            lsmc[(r, T)] = gdist.ShardedSMC(ops, "lgssm", impl, 1, n_total, T, r, world, comm=gdist.ThreadComm(sh, r))
        return lambda: lsmc[(r, T)].run_native(lcomms[r])

    # both transports step the same filter: equal results
    log_z = {}
    for name, make in (("peers", make_peers), ("collective_protocol_virtual", make_local)):
        log_z[name] = run_ranks(make, 3)[0]["log_z"]
    torch.cuda.synchronize()
    # ---- device and host time of ONE rank's launches: rank 0 alone repeats step t = 3 of the peer run above (it reads
    # population 0 — step 2's, complete and consistent in every arena — and writes its own block of population 1), its
    # peers' arrival words raised once so that nothing waits; the same loop without the peer descriptor runs the kernels of
    # a collective rank on the local arrays (the tiles a shuffle would have delivered copied in first)
    for a in arenas:
        a.flags.fill_(1 << 62)
    T0 = 3
    sk, rk = W.smc_key_schedule(prng.key(1, impl), T0 + 1)
    y = W.lgssm_data(T0 + 1)
    model = W.lgssm_model()
    tl = n // ops.tile
    pops = arenas[0]._pops
    for col0, col1 in zip(pops[0].columns() + [pops[0].subs], arenas[1]._pops[0].columns() + [arenas[1]._pops[0].subs]):
        k = 16 * (ops.tile if col0.shape[0] == n_total else 1)  # rank 1's first 16 tiles: what rank 0's last windows read
        a0 = n if col0.shape[0] == n_total else tl
        col0[a0:a0 + k].copy_(col1[a0:a0 + k])
    pst = arenas[0].peers_struct(20000)
    pst.wait_value = 1
    ranges = torch.zeros(2 * world + 1, dtype=torch.int64).pin_memory()
    for name in ("peers", "collective_kernels"):
        cfg = ops.smc_config(impl, n_total, 0, n, sk, rk)
        if name == "peers":
            cfg.peers = C.pointer(pst)

        def block(kk):
            for _ in range(kk):
                if name == "peers":  # the step's group-record launch carries the previous step's (deferred) signal, as the driver has it
                    pst.signal_recs, pst.signal_ess = pops[1].recs.data_ptr(), None
                    pst.signal_first_tile, pst.signal_n_tiles, pst.signal_value = 0, tl, 1 << 62
                ops.smc_lgssm_step(cfg, model, T0, float(y[T0]), pops[0].struct(), pops[1].struct(0, with_logw=False), None, None, None)
                if name != "peers":
                    ops.smc_source_ranges(cfg, pops[1].recs, None, world, ranges, 1)

        block(4)
        torch.cuda.synchronize()
        dev, host = [], []
        for _ in range(12):
            e0, e1 = HipEvent(), HipEvent()
            e0.record(ops.stream())
            t0 = time.perf_counter()
            block(steps)
            host.append((time.perf_counter() - t0) / steps)
            e1.record(ops.stream())
            torch.cuda.synchronize()
            dev.append(e0.elapsed_ms(e1) / steps)
        out[name] = {"step_us_device": statistics.median(dev) * 1e3, "host_us_per_step": statistics.median(host) * 1e6}
    arenas[0].check(ops)
    out["peers"].update(launches_per_step=2, collectives_per_step=0, host_decisions_per_step=0,
                        what="group-record launch (first the previous step's records into the 8 arenas + arrival word, then the wait for the peers, then 31 workgroups of group records) + step launch (remote windows read in place)")
    out["collective_kernels"].update(
        launches_per_step=3, collectives_per_step=2, host_decisions_per_step=1,
        what="group records + step + range kernel: what an RCCL rank launches; on top come 1 all-gather + 1 grouped send/recv (20-50 us each on xGMI) and the host's poll of the range ticket")
    out["rccl_collectives_per_step"] = {"all_gather": 1, "grouped_send_recv": 1, "host_polls": 1,
                                        "r03": {"all_gather": 2, "grouped_send_recv": 1, "host_polls": 1}}
    out["log_z_equal_across_transports"] = log_z["peers"] == log_z["collective_protocol_virtual"]
    out["config"] = {"world": world, "n_total": n_total, "particles_per_rank": n, "tiles": n_total // ops.tile, "steps_timed": steps,
                     "virtual_ranks": "threads of this process on ONE device, one stream", "log_z_3_steps": log_z["peers"]}
    out["value"] = n / (out["peers"]["step_us_device"] * 1e-6)
    out["unit"] = "particle-steps/s per rank (peer transport; one device, no xGMI latency)"
    return out


def bench_smc(args, ops, kind, filters=1, min_s=0.08, variant=None, T=None):
    """Single device: `filters` independent filters (seeds s, s+1, ...) step in the same launches; filters=1 is
    the literal BASELINE config.  A "step" of the protocol here is one whole T-step filter run (one enqueue).
    `variant`: extra keywords of the workload (ess_threshold, another model / observation sequence)."""
    import torch

    from genjax._amd import workloads as W
    from genjax._amd.ops import HipEvent

    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    T = T if T else (100 if kind == "smc_lgssm" else 500)
    kw = dict(filters=filters)
    if variant:
        kw.update(variant)
    wl = W.LgssmSMC(ops, impl, 1, n, T, **kw) if kind == "smc_lgssm" else W.HmmSMC(ops, impl, 2, n, T, **kw)
    g0 = ops.smc_run_graph_stats()  # (r04: from the second run of a shape on, a one-filter run is one replayed hipGraph)
    # warm-up long enough for the clock ramp (see bench_importance): ~50 ms of the same launches
    t0 = time.perf_counter()
    out = wl.run()
    torch.cuda.synchronize()
    while time.perf_counter() - t0 < 0.05:
        out = wl.run()
        torch.cuda.synchronize()
    evs = []

    def one_run():
        e0, e1 = HipEvent(), HipEvent()
        e0.record(ops.stream())
        o = wl.run()  # enqueue only: data, keys and tables were prepared above
        e1.record(ops.stream())
        evs.append((e0, e1))
        return o

    blocks, out = timed_blocks(one_run, 1, min_s=min_s, min_blocks=20, max_blocks=60)
    r = wl.result(out)
    dev_ms = statistics.median(a.elapsed_ms(b) for a, b in evs)
    dt = statistics.median(blocks)
    per_step_ms = dev_ms / T
    achieved = BYTES_SMC_PER_PARTICLE_STEP * n * filters / (per_step_ms * 1e-3) / 1e9
    log_z = r["log_z"][0] if filters > 1 else r["log_z"]
    # HBM bytes of one step from the PMC passes committed under profiles/ (bytes per particle-step; FETCH_SIZE x2 + WRITE_SIZE)
    per, traffic_src = pmc_traffic(
        "r*_smc_pmc.json",
        lambda pj: sum((v["hbm_read_bytes"] + v["hbm_write_bytes"]) / v.get("particles_per_launch", 1e6)
                       for k, v in pj.get(kind, {}).items() if "k_resample" in k))
    traffic = per * n * filters if per else None
    res = {
        "metric": "particle-steps/sec, bootstrap SMC (1e6 particles)",
        "value": n * T * filters / dt, "unit": "particle-steps/s", "ms_per_step": dt * 1e3,
        "config": {"workload": f"bootstrap SMC {kind} T={T} N={n}, systematic resampling every step"
                               + (" (BASELINE configs[2])" if kind == "smc_lgssm" else " (BASELINE configs[4])"),
                   "rng": args.rng, "filters_per_launch": filters, **({"variant": variant} if variant else {})},
        "roofline": {"bound": "hbm", "kernel": "k_resample (ONE launch per SMC step" + (" of every filter)" if filters > 1 else ")"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_basis": "44 B/particle-step contract figure of the unfused pipeline (SURVEY 8d)",
                     "traffic": traffic, "traffic_source": traffic_src,
                     "frac_of_peak_on_pmc_traffic": (traffic / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "step_ms": per_step_ms, "step_ms_per_filter": per_step_ms / filters, "runs_timed": len(evs),
                     "statistic": "median", "algorithmic_bytes_per_launch": BYTES_SMC_PER_PARTICLE_STEP * n * filters,
                     },
        "log_z": log_z, "log_z_exact": r["log_z_exact"], "log_z_abs_err_vs_exact": abs(log_z - r["log_z_exact"]),
    }
    if r.get("resampled") is not None:
        fl = r["resampled"] if filters == 1 else r["resampled"][0]
        res["resampling_steps"] = int(fl.sum())
        res["steps"] = T
    res.update(block_stats(blocks))
    g1 = ops.smc_run_graph_stats()
    res["config"]["launch_form"] = (
        f"whole run replayed as ONE hipGraph ({g1['replays'] - g0['replays']} of this entry's runs; keys, observations and model "
        "scalars from a device block; GJX_SMC_GRAPH=0: a stream of T launches)" if g1["replays"] > g0["replays"]
        else "a stream of T launches per run")
    return res, r


def bench_scan(args, ops, min_s=0.08, T=100, fast_math=False, with_host_loop=True):
    """ImportanceK over `step.scan(n=T)` — the reference's literal semantics for a state-space model without
    resampling (scan.py:237-294, [N, T] leaves): N = 1e6 particles x T steps in ONE launch (gjx_scan_run).  Reported
    in particle-steps/s; the HBM floor is the 4-byte store of x_t per particle-step (+ 8 B / T for logw and score)."""
    import torch

    from genjax._amd import workloads as W
    from genjax._amd.ops import HipEvent

    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    wl = W.LgssmScan(ops, impl, 3, n, T, fast_math=fast_math)
    t0 = time.perf_counter()
    wl.run()
    torch.cuda.synchronize()
    while time.perf_counter() - t0 < 0.05:
        wl.run()
        torch.cuda.synchronize()
    evs = []

    def one_run():
        e0, e1 = HipEvent(), HipEvent()
        e0.record(ops.stream())
        o = wl.run()
        e1.record(ops.stream())
        evs.append((e0, e1))
        return o

    blocks, _ = timed_blocks(one_run, 1, min_s=min_s, min_blocks=5, max_blocks=40)
    r = wl.result()
    dev_ms = statistics.median(a.elapsed_ms(b) for a, b in evs)
    dt = statistics.median(blocks)
    bytes_per_launch = (4.0 * T + 8.0) * n
    achieved = bytes_per_launch / (dev_ms * 1e-3) / 1e9
    res = {
        "value": n * T / dt, "unit": "particle-steps/s", "ms_per_run": dt * 1e3, "kernel_ms": dev_ms, "runs_timed": len(evs),
        "config": {"workload": f"ImportanceK over step.scan(n={T}), LGSSM, N={n}, no resampling, [T, N] trajectories stored",
                   "rng": args.rng, "math": "fast" if fast_math else "exact"},
        "roofline": {"bound": "hbm", "kernel": "gjx_scan_kernel (one launch per pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "traffic": None,
                     "limiter": "VALU issue: half a pair block + half a table-driven Box-Muller transform + two log-densities per particle-step (quad form; r02: three blocks and a polynomial transform)"},
        "log_z": r["log_z"],
    }
    if with_host_loop:
        # the general route (host loop of per-site launches, combinators.py) on the same model, once, for scale
        import genjax
        from genjax import ChoiceMapBuilder as C, gen, normal
        from genjax._amd import combinators as CB
        from genjax._amd.runtime import use_ops

        m = W.LGSSM

        @gen
        def step(x, _):
            x2 = normal(m["a"] * x, m["q"]) @ "x"
            _ = normal(x2, m["r"]) @ "y"
            return x2, x2

        with use_ops(ops):
            keys = genjax.random.split(genjax.random.key(3, args.rng), n)
            chm = C["y"].set(torch.from_numpy(W.lgssm_data(T)))
            times = {}
            for fused in (True, False):
                CB.FUSED_SCAN = fused
                try:
                    step.scan(n=T).generate(keys, chm, (0.0, None))  # warm (hiprtc, allocator)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tr, w = step.scan(n=T).generate(keys, chm, (0.0, None))
                    torch.cuda.synchronize()
                    times[fused] = time.perf_counter() - t0
                finally:
                    CB.FUSED_SCAN = True
                del tr, w
        res["host_api"] = {"scan_generate_fused_ms": times[True] * 1e3, "scan_generate_host_loop_ms": times[False] * 1e3,
                           "speedup": times[False] / times[True],
                           "note": "Scan.generate through the host API (tracing, lowering, launch, trace assembly) vs the "
                                   "T x per-site-launch host loop it replaces"}
    return res


def bench_scan_hmm(args, ops, min_s=0.08, T=500):
    """ImportanceK over the 256-state HMM written as a scan (the reference's literal configs[4] semantics without
    resampling): `z' ~ categorical(trans[z]); y ~ categorical(obs[z'])`, N = 1e6, T = 500, one launch per pass.  The
    categorical rows are chosen by the carried state: the specialised kernel draws by binary search on the row's prepared
    fixed-point CDF and scores with the row's prepared log-sum-exp (same bits as the two-pass walk of the row)."""
    import torch

    from genjax._amd import workloads as W
    from genjax._amd.ops import HipEvent

    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    wl = W.HmmScan(ops, impl, 4, n, T, cat_mode=1)
    wl.run()
    torch.cuda.synchronize()
    evs = []

    def one_run():
        e0, e1 = HipEvent(), HipEvent()
        e0.record(ops.stream())
        o = wl.run()
        e1.record(ops.stream())
        evs.append((e0, e1))
        return o

    blocks, _ = timed_blocks(one_run, 1, min_s=min_s, min_blocks=3, max_blocks=12)
    r = wl.result()
    dev_ms = statistics.median(a.elapsed_ms(b) for a, b in evs)
    dt = statistics.median(blocks)
    bytes_per_launch = (4.0 * T + 8.0) * n
    achieved = bytes_per_launch / (dev_ms * 1e-3) / 1e9
    return {
        "value": n * T / dt, "unit": "particle-steps/s", "ms_per_run": dt * 1e3, "kernel_ms": dev_ms, "runs_timed": len(evs),
        "config": {"workload": f"ImportanceK over hmm_step.scan(n={T}), {wl.k} states, N={n}, no resampling, [T, N] state trajectories stored",
                   "rng": args.rng, "categorical": "inverse CDF (one uniform per draw)"},
        "roofline": {"bound": "hbm", "kernel": "gjx_scan_kernel (one launch per pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": None,
                     "limiter": "L2 request rate: one scattered 16-byte guide bucket per categorical draw (a buffer load: 1.08 L2 requests per particle-step, profiles/r04_scan_pmc.json) from a 2 MB table, and the cipher"},
        "log_z": r["log_z"],
    }


# ------------------------------------------------------------------------------------------------------------
# CPU baselines (the oracle: a port) — bounded samples of the same workloads, timed on this box's host cores
# ------------------------------------------------------------------------------------------------------------
def bench_host_api_call(args, calls=60):
    """`ImportanceK(Target(model, (), obs), k_particles=N).log_marginal_likelihood_estimate(key)` through the host API that
    mirrors the reference's (genjax.inference.smc), a fresh key per call, the scalar brought back to the host: what one
    literal call of BASELINE configs[1] costs end to end."""
    import torch

    import genjax
    from genjax import ChoiceMapBuilder as C, Target, gen, normal
    from genjax._amd import workloads as W
    from genjax.inference.smc import ImportanceK

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
    for rep in range(10):
        float(alg.log_marginal_likelihood_estimate(genjax.random.key(rep, args.rng)))
    torch.cuda.synchronize()
    ts, zs = [], []
    for rep in range(calls):
        t0 = time.perf_counter()
        z = float(alg.log_marginal_likelihood_estimate(genjax.random.key(1000 + rep, args.rng)))
        ts.append(time.perf_counter() - t0)
        zs.append(z)
    med = statistics.median(ts)
    return {"value": args.particles / med, "unit": "particles/s", "us_per_call": med * 1e6, "us_per_call_min": min(ts) * 1e6,
            "calls": calls, "rng": args.rng, "log_z_mean": statistics.fmean(zs), "log_z_exact": W.gaussian10_exact_log_z(y),
            "note": "host-inclusive latency of ONE eager call = ONE launch (~18 us of kernel: the walk without value "
                    "columns, its row sums folded by the last workgroup, lse - log K): key derivation on the host, output "
                    "allocation, launch, device-to-host scalar"}


def host_cores() -> int:
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands
    one GPU's share of a 256-thread host to the job; 256 OpenMP threads on a 16-core share run 5x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    env = os.environ.get("GJX_BENCH_CPU_THREADS")
    return int(env) if env else n


def best_thread_count(run_once, candidates):
    """Time `run_once()` at a few OpenMP thread counts (the quota may be invisible from inside the box) and return
    (threads, seconds) of the fastest."""
    best = None
    for c in candidates:
        if not _omp_threads(c):
            return candidates[0], None
        t0 = time.perf_counter()
        run_once()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (c, dt)
    _omp_threads(best[0])
    return best


def _omp_threads(nthreads: int):
    import ctypes

    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(nthreads))
        return True
    except OSError:
        return False


def _oracle():
    from genjax._amd.abi import GjxLib
    from genjax._amd.ops import Ops

    lib = os.path.join(ROOT, "oracle", "libgjx_oracle.so")
    return Ops(GjxLib(lib, "cpu")) if os.path.exists(lib) else None


def cpu_baseline_importance(args, gpu_log_z=None):
    from genjax._amd import workloads as W

    ora = _oracle()
    if ora is None:
        return None
    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    wl = W.Gaussian10(ora, impl, seed=0, n_local=n)
    out = wl.step()
    hc = host_cores()
    cores, _ = best_thread_count(wl.step, sorted({hc, max(1, hc // 2), min(hc, 16), min(hc, 32), min(hc, 64)}, reverse=True))
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 8.0:
        out = wl.step()
        reps += 1
    dt = time.perf_counter() - t0
    log_z = ora.log_z_from_rows(out["row_e"], out["row_q"], n)
    res = {"value": n * reps / dt, "unit": "particles/s", "cores": cores, "kind": "port",
           "sample": f"{reps} full passes of the same {n}-particle ImportanceK workload (OpenMP, {cores} threads)",
           "log_z": log_z}
    if gpu_log_z is not None:
        res["log_z_abs_err_gpu_vs_cpu"] = abs(gpu_log_z - log_z)
    if _omp_threads(1):
        t0 = time.perf_counter()
        wl.step()
        res["value_1_thread"] = n / (time.perf_counter() - t0)
        _omp_threads(cores)
    res["host_threads_visible"] = os.cpu_count()
    return res


def cpu_baseline_smc(args, kind, gpu_result=None):
    """The oracle filter on all host cores: LGSSM the full T=100 run (its log Z is compared with the GPU's); HMM
    the first 100 of 500 steps (compared with the GPU's log Z over the same prefix)."""
    import math

    from genjax._amd import workloads as W

    ora = _oracle()
    if ora is None:
        return None
    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    T_full = 100 if kind == "smc_lgssm" else 500
    T = 100
    # the key schedule of a T-step prefix is the prefix of the full schedule (fold_in(key, 2t), fold_in(key, 2t + 1))
    mk = (lambda t: W.LgssmSMC(ora, impl, 1, n, t)) if kind == "smc_lgssm" else (lambda t: W.HmmSMC(ora, impl, 2, n, t))
    wl = mk(T_full)
    if T < T_full:
        wl.T, wl.y, wl.sk, wl.rk = T, wl.y[:T], wl.sk[:T], wl.rk[:T]
    probe = mk(T_full)
    probe.T, probe.y, probe.sk, probe.rk = 4, probe.y[:4], probe.sk[:4], probe.rk[:4]
    probe.run()
    hc = host_cores()
    cores, _ = best_thread_count(probe.run, sorted({hc, max(1, hc // 2), min(hc, 16), min(hc, 32), min(hc, 64)}, reverse=True))
    t0 = time.perf_counter()
    out = wl.run()
    dt = time.perf_counter() - t0
    out_max, out_q = out[0], out[1]
    log_z = ora.log_z_from_pairs(out_max, out_q, n)
    res = {"value": n * T / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
           "sample": f"{'all' if T == T_full else 'first'} {T} steps of the same {n}-particle filter (OpenMP, {cores} threads)",
           "log_z_steps": T, "log_z": log_z}
    if gpu_result is not None:
        g = ora.log_z_from_pairs(gpu_result["out_e"][:T].cpu(), gpu_result["out_q"][:T].cpu(), n)
        res["log_z_gpu_same_steps"] = g
        res["log_z_abs_err_gpu_vs_cpu"] = abs(g - log_z)
    if _omp_threads(1):
        w1 = mk(T_full)
        t1 = 3
        w1.T, w1.y, w1.sk, w1.rk = t1, w1.y[:t1], w1.sk[:t1], w1.rk[:t1]
        t0 = time.perf_counter()
        w1.run()
        res["value_1_thread"] = n * t1 / (time.perf_counter() - t0)
        _omp_threads(cores)
    _ = math
    res["host_threads_visible"] = os.cpu_count()
    return res


def cpu_baseline_scan(args, T=100):
    """The oracle's gjx_scan_run (all host cores) on a bounded sample: the first 1/8 of the particles, all T steps."""
    from genjax._amd import workloads as W

    ora = _oracle()
    if ora is None:
        return None
    impl = 1 if args.rng == "philox" else 0
    n = max(1024, args.particles // 8)
    wl = W.LgssmScan(ora, impl, 3, n, T)
    probe = W.LgssmScan(ora, impl, 3, max(1024, n // 16), T)
    probe.run()
    hc = host_cores()
    cores, _ = best_thread_count(probe.run, sorted({hc, max(1, hc // 2), min(hc, 16), min(hc, 32)}, reverse=True))
    t0 = time.perf_counter()
    wl.run()
    dt = time.perf_counter() - t0
    res = {"value": n * T / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
           "sample": f"{n} of {args.particles} particles x {T} steps (OpenMP, {cores} threads)"}
    if _omp_threads(1):
        t0 = time.perf_counter()
        probe.run()
        res["value_1_thread"] = probe.n * T / (time.perf_counter() - t0)
        _omp_threads(cores)
    return res


def jax_cpu_plain(args):
    """BASELINE.md §2 secondary baseline: a plain `jax.jit(jax.vmap(...))` restatement of the 10-latent model on
    CPU, timed only if jax happens to be importable on this box (it is not part of the image)."""
    try:
        import jax  # noqa: F401
        import jax.numpy as jnp
    except Exception as e:  # ModuleNotFoundError on the stock image
        return {"available": False, "note": f"jax not importable on this box ({type(e).__name__})"}
    import numpy as np

    from genjax._amd import workloads as W

    y = jnp.asarray(W.gaussian10_data())
    n = args.particles

    def one(key):
        z = jax.random.normal(key, (10,))
        lw = jnp.sum(-0.5 * ((y - z) / 0.5) ** 2 - jnp.log(0.5) - 0.5 * jnp.log(2 * jnp.pi))
        return z, lw

    @jax.jit
    def run(key):
        z, lw = jax.vmap(one)(jax.random.split(key, n))
        return z, lw, jax.scipy.special.logsumexp(lw) - jnp.log(n)

    with jax.default_device(jax.devices("cpu")[0]):
        k = jax.random.key(0)
        run(k)[2].block_until_ready()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 5.0:
            run(jax.random.fold_in(k, reps))[2].block_until_ready()
            reps += 1
        dt = time.perf_counter() - t0
    _ = np
    return {"available": True, "value": n * reps / dt, "unit": "particles/s", "cores": os.cpu_count(),
            "sample": f"{reps} passes, jax.jit(jax.vmap) on CPU"}


# ------------------------------------------------------------------------------------------------------------
def entry(r, keys=("value", "unit", "ms_per_step", "config", "roofline", "log_z", "log_z_exact", "timed_blocks",
                   "block_ms_min", "block_ms_median", "block_ms_max")):
    e = {k: r[k] for k in keys if k in r}
    if "log_z" in e and "log_z_exact" in e:
        e["log_z_abs_err_vs_exact"] = abs(e["log_z"] - e["log_z_exact"])
    return e


_LINE_OUT = None


def reserve_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner from C at
    communicator creation): keep the real stdout for the line and point descriptor 1 at stderr for everything else."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_line(obj):
    out = _LINE_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


class ExtrasDeadline:
    """At N > 1 the extra workload runs collectives after the headline has been measured.  If it (or the closing barrier)
    has not come back after `seconds`, rank 0 prints the line it already has, with the extra marked as missing, and every
    rank leaves with `os._exit(0)` — a rank stuck in a collective cannot be unwound any other way."""

    def __init__(self, rank, out, seconds, name, args=None):
        import threading

        self.rank, self.out, self.seconds, self.name, self.args = rank, out, seconds, name, args
        self.lock, self.line_done, self.done = threading.Lock(), False, threading.Event()
        self.t0 = time.perf_counter()
        threading.Thread(target=self._watch, daemon=True).start()

    def claim_line(self) -> bool:
        with self.lock:
            first, self.line_done = not self.line_done, True
            return first

    def finish(self):
        self.done.set()

    def _watch(self):
        if self.done.wait(self.seconds):
            return
        if self.rank == 0 and self.claim_line():
            emit_line(compact_line(self.out, {self.name: {"error": f"no result within {self.seconds:g} s; the line carries the headline only"}}, self.args))
        print(f"bench.py: rank {self.rank}: '{self.name}' exceeded its {self.seconds:g} s deadline "
              f"({time.perf_counter() - self.t0:.1f} s after the headline)", file=sys.stderr, flush=True)
        os._exit(3)  # a hang is a FAILURE of the run (the headline line, already measured, has been printed)


def _rf(r):
    """The roofline object of the line: the fields the contract names plus the kernel's name and measured duration."""
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "step_ms", "algorithmic_bytes_per_launch")
    return {k: r[k] for k in keep if k in r and r[k] is not None or k == "traffic"}


def _cb(c):
    if not c:
        return None
    keep = ("value", "unit", "cores", "kind", "sample", "value_1_thread", "log_z_abs_err_gpu_vs_cpu")
    return {k: c[k] for k in keep if k in c}


def _short(e, per="step_ms"):
    """One `extra` entry of the line: value, time per unit, roofline fraction — the full entry goes to the detail file."""
    if not isinstance(e, dict):
        return e
    if "error" in e:
        return {"error": str(e["error"])[:120]}
    o = {}
    for k in ("value", "unit", "us_per_call", "kernel_ms", "resampling_steps", "ratio_to_normal_step", "step_ms", "normal_step_ms",
              "step_us_device", "host_us_per_step", "launches_per_step", "collectives_per_step", "host_decisions_per_step",
              "transport", "log_z_equal_across_transports"):
        if k in e:
            o[k] = e[k]
    rf = e.get("roofline") or {}
    if "step_ms" in rf:
        o["step_us"] = rf["step_ms"] * 1e3
    if "kernel_ms" in rf:
        o["kernel_us"] = rf["kernel_ms"] * 1e3
    if "frac" in rf:
        o["roofline_frac"] = rf["frac"]
    return o


def config_entry(e):
    """A BASELINE config in the line's closing `configs` object: throughput, time per unit, roofline, CPU baseline, log Z
    against the CPU on the same steps — short enough that the driver's record (the TAIL of stdout) keeps all of them."""
    rf, cb = e.get("roofline") or {}, e.get("cpu_baseline") or {}
    o = {"value": e.get("value"), "unit": e.get("unit")}
    if "step_ms" in rf:
        o["us_per_step"] = rf["step_ms"] * 1e3
    elif "kernel_ms" in rf:
        o["kernel_us"] = rf["kernel_ms"] * 1e3
    o["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if cb:
        o["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "value_1_thread")}
        o["log_z_abs_err_gpu_vs_cpu"] = cb.get("log_z_abs_err_gpu_vs_cpu")
    for k in ("log_z", "log_z_exact", "timed_blocks"):
        if k in e:
            o[k] = e[k]
    lf = (e.get("config") or {}).get("launch_form")
    if lf:
        o["launch_form"] = "replayed hipGraph" if lf.startswith("whole run replayed") else "stream of launches"
    return o


def write_detail(out, extra):
    """Everything measured, unabridged, beside the line: $GJX_BENCH_DETAIL or gpurun_out/bench_detail.json."""
    path = os.environ.get("GJX_BENCH_DETAIL") or (os.path.join(ROOT, "gpurun_out", "bench_detail.json")
                                                  if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    if not path:
        return
    try:
        with open(path, "w") as f:
            json.dump(dict(out, extra=extra), f, indent=1, default=str)
    except OSError:
        pass


def compact_line(out, extra, args):
    """The ONE JSON line: the contract's fields, `roofline`, `cpu_baseline`, short `extra` entries, and LAST a `configs`
    object with every BASELINE config measured in this run (the driver keeps the tail of stdout)."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(out.get("config") or {})
    for k in list(cfg):
        if isinstance(cfg[k], str) and len(cfg[k]) > 160:
            cfg[k] = cfg[k][:157] + "..."
    line["config"] = cfg
    line["roofline"] = _rf(out.get("roofline"))
    line["cpu_baseline"] = _cb(out.get("cpu_baseline"))
    for k in ("log_z", "log_z_exact", "log_z_abs_err_vs_exact", "timed_blocks", "blocks_per_bracket", "block_ms_median", "bracket"):
        if k in out:
            line[k] = out[k]
    if out.get("cpu_baseline") and "log_z_abs_err_gpu_vs_cpu" in out["cpu_baseline"]:
        line["log_z_abs_err_gpu_vs_cpu"] = out["cpu_baseline"]["log_z_abs_err_gpu_vs_cpu"]
    jp = out.get("jax_cpu_plain")
    if jp is not None:
        line["jax_cpu_plain"] = {"available": bool(jp.get("available"))} if isinstance(jp, dict) else jp
    configs = {}
    head = dict(out)
    if args.workload == "importance" and out.get("n_gpus", 1) == 1:
        configs["configs[1] ImportanceK 1e6 (10-latent Gaussian)"] = config_entry(head)
    elif args.workload == "importance":
        configs[f"ImportanceK 1e6 per GPU x{out['n_gpus']}"] = config_entry(head)
    else:
        nm = {"smc_lgssm": "configs[2] SMC LGSSM T=100 1e6", "smc_hmm": "configs[4] SMC HMM-256 T=500 1e6"}[args.workload]
        configs[nm + (f" per GPU x{out['n_gpus']} (configs[3])" if out.get("n_gpus", 1) > 1 or FORCE_DIST else "")] = config_entry(head)
    small = {}
    for name, e in (extra or {}).items():
        if name == "smc_lgssm" and isinstance(e, dict) and "value" in e:
            configs["configs[2] SMC LGSSM T=100 1e6 (one filter)"] = config_entry(e)
        elif name == "smc_hmm" and isinstance(e, dict) and "value" in e:
            configs["configs[4] SMC HMM-256 T=500 1e6 (one filter)"] = config_entry(e)
        elif name == "smc_lgssm_sharded" and isinstance(e, dict) and "value" in e:
            configs[f"configs[3] SMC LGSSM T=100 1e6 per GPU x{out['n_gpus']}"] = config_entry(e)
        if isinstance(e, dict):
            sub = {k: _short(v) for k, v in e.items() if isinstance(v, dict) and ("value" in v or "error" in v or "step_us_device" in v)
                   and k not in ("roofline", "cpu_baseline", "config")}
            small[name] = dict(_short(e), **sub)
    if small:
        line["extra"] = small
    line["configs"] = configs
    return line


def run_rank(args):
    reserve_stdout()
    rank, world = init_dist(args.gpus)
    from genjax._amd.runtime import load_hip_ops

    ops = load_hip_ops()  # raises without libgjx_hip.so / a GPU: there is no CPU fallback
    sharded = world > 1 or FORCE_DIST
    if REHEARSE and (world < 2 or args.workload not in ("smc_lgssm", "smc_hmm")):
        print("bench.py: GJX_BENCH_REHEARSE rehearses `--gpus N --workload smc_lgssm|smc_hmm` (N >= 2) only", file=sys.stderr)
        sys.exit(2)
    smc_gpu = None
    if args.workload in ("scan_lgssm", "scan_hmm"):  # profiling passes of the one-launch scans (`extra` entries of the default run)
        if rank == 0:
            emit_line(bench_scan(args, ops, fast_math=args.fast_math, with_host_loop=False) if args.workload == "scan_lgssm"
                      else bench_scan_hmm(args, ops))
        return
    if args.workload == "importance":
        res, _ = bench_importance(args, ops, rank, world, fast_math=args.fast_math)
    elif sharded:
        res = bench_smc_sharded(args, ops, rank, world, args.workload)
    else:
        res, smc_gpu = bench_smc(args, ops, args.workload, filters=int(os.environ.get("GJX_BENCH_FILTERS", "1")))
    extra = {}
    out = None
    if rank == 0:
        out = {
            "metric": res.pop("metric"), "value": res.pop("value"), "unit": res.pop("unit"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": res.pop("config"), "roofline": res.pop("roofline"),
        }
        out.update(res)
        out["log_z_abs_err_vs_exact"] = abs(out["log_z"] - out["log_z_exact"])
        out["statistic"] = "median over timed_blocks repetitions of the K-step block"
    guard = None
    if not args.no_extra and args.workload == "importance":
        if sharded:
            # BASELINE configs[3]: the LGSSM filter with 1e6 particles per GPU, sharded (every rank takes part).  The
            # headline above is already measured: a rank that never comes back from the exchange must not cost the line.
            guard = ExtrasDeadline(rank, out, float(os.environ.get("GJX_BENCH_EXTRA_DEADLINE_S", "240")), "smc_lgssm_sharded", args)
            try:
                extra["smc_lgssm_sharded"] = entry(bench_smc_sharded(args, ops, rank, world, "smc_lgssm"))
            except Exception as ex:  # reported in the line; the other ranks are released by their own deadline
                extra["smc_lgssm_sharded"] = {"error": f"{type(ex).__name__}: {ex}"}
        elif rank == 0:
            # the literal single-GPU BASELINE configs, each first-class: roofline, CPU baseline, log Z vs CPU
            for kind in ("smc_lgssm", "smc_hmm"):
                r1, g1 = bench_smc(args, ops, kind, filters=1)
                e = entry(r1)
                if not args.no_cpu_baseline:
                    e["cpu_baseline"] = cpu_baseline_smc(args, kind, g1)
                # ESS-adaptive variant: resample only when ESS < N / 2
                ra, _ = bench_smc(args, ops, kind, filters=1, min_s=0.05, variant=dict(ess_threshold=0.5))
                e["ess_adaptive_0.5"] = entry(ra, ("value", "unit", "ms_per_step", "log_z", "log_z_exact", "resampling_steps", "steps"))
                e["ess_adaptive_0.5"]["step_ms"] = ra["roofline"]["step_ms"]
                r16, _ = bench_smc(args, ops, kind, filters=16, min_s=0.05)
                e["batched_16_filters_per_launch"] = entry(r16, ("value", "unit", "ms_per_step", "roofline", "log_z"))
                extra[kind] = e
            # worst case of the resampler: collapsing weights (a sharp observation model, observations jumping by tens of
            # standard deviations: at most steps ONE tile owns every output slot), against the normal LGSSM step
            try:
                import numpy as np

                from genjax._amd import abi as _abi

                yc = np.tile(np.array([0.1, 25.0, -40.0, -39.5, 60.0, 60.2, 0.0, 3.0], dtype=np.float32), 5)
                rc_, gc_ = bench_smc(args, ops, "smc_lgssm", filters=1, min_s=0.03, T=len(yc),
                                     variant=dict(y=yc, model=_abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.05), want_ancestors=True))
                anc = gc_["ancestors"]
                extra["smc_lgssm_collapsing_weights"] = {
                    "step_ms": rc_["roofline"]["step_ms"], "normal_step_ms": extra["smc_lgssm"]["roofline"]["step_ms"],
                    "ratio_to_normal_step": rc_["roofline"]["step_ms"] / extra["smc_lgssm"]["roofline"]["step_ms"],
                    "distinct_ancestors_at_step_2": int(anc[2].unique().numel()), "steps": len(yc)}
            except Exception as ex:
                extra["smc_lgssm_collapsing_weights"] = {"error": f"{type(ex).__name__}: {ex}"}
            # one rank of BASELINE configs[3] on this device: the world-8 step through the peer transport and through the kernels
            # a collective transport launches
            try:
                extra["smc_sharded_rank0_of_8_virtual"] = bench_sharded_rank0_virtual(args, ops)
            except Exception as ex:
                extra["smc_sharded_rank0_of_8_virtual"] = {"error": f"{type(ex).__name__}: {ex}"}
            # ImportanceK over a Scan model: the reference's literal semantics of the state-space configs (no resampling)
            try:
                extra["importance_scan_lgssm"] = bench_scan(args, ops)
                extra["importance_scan_lgssm"]["fast_math"] = entry(bench_scan(args, ops, min_s=0.04, fast_math=True, with_host_loop=False),
                                                                    ("value", "unit", "kernel_ms", "roofline", "log_z"))
                if not args.no_cpu_baseline:
                    extra["importance_scan_lgssm"]["cpu_baseline"] = cpu_baseline_scan(args)
            except Exception as ex:
                extra["importance_scan_lgssm"] = {"error": f"{type(ex).__name__}: {ex}"}
            try:
                extra["importance_scan_hmm"] = bench_scan_hmm(args, ops)
            except Exception as ex:
                extra["importance_scan_hmm"] = {"error": f"{type(ex).__name__}: {ex}"}
            # the rejection samplers (Beta = two gammas, Gamma) under ImportanceK
            for nm in ("beta_bernoulli", "gamma_normal"):
                try:
                    extra[f"importance_{nm}"] = bench_site_model(args, ops, nm)
                except Exception as ex:
                    extra[f"importance_{nm}"] = {"error": f"{type(ex).__name__}: {ex}"}
            # the reference-API call a user makes (tracing cache hit, plan lookup, one launch, fold, one scalar back)
            try:
                extra["importance_host_api_call"] = bench_host_api_call(args)
            except Exception as ex:
                extra["importance_host_api_call"] = {"error": f"{type(ex).__name__}: {ex}"}
            # ImportanceK variants: one pass per launch (the literal config), the other generator, fast math
            r, _ = bench_importance(args, ops, rank, world, launch_passes=1, steps=64, warmup=16, ramp=False, min_s=0.03)
            extra["importance_1_pass_per_launch"] = entry(r)
            r, _ = bench_importance(args, ops, rank, world, rng="threefry" if args.rng == "philox" else "philox", steps=24,
                                    warmup=8, ramp=False, min_s=0.03)
            extra[f"importance_{r['config']['rng']}"] = entry(r)
            try:
                r, _ = bench_importance(args, ops, rank, world, steps=64, warmup=16, min_s=0.05, fast_math=True)
                extra["importance_fast_math"] = entry(r)
                r, _ = bench_importance(args, ops, rank, world, launch_passes=1, steps=64, warmup=16, ramp=False, min_s=0.03,
                                        fast_math=True)
                extra["importance_fast_math_1_pass_per_launch"] = entry(r)
            except Exception as ex:  # reported, never silently dropped
                extra["importance_fast_math"] = {"error": f"{type(ex).__name__}: {ex}"}
    if rank == 0:
        if not args.no_cpu_baseline:  # (r04: N > 1 lines carry it too — rank 0's host cores, the same bounded sample)
            if args.workload == "importance":
                # (N > 1: the CPU sample is ONE GPU's share, 1e6 particles — its log Z is not the job's, so nothing is compared)
                out["cpu_baseline"] = cpu_baseline_importance(args, out["log_z"] if world == 1 else None)
                out["jax_cpu_plain"] = jax_cpu_plain(args)
            else:
                out["cpu_baseline"] = cpu_baseline_smc(args, args.workload, smc_gpu)
        if guard is None or guard.claim_line():
            write_detail(out, extra)
            emit_line(compact_line(out, extra, args))
    if sharded:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if guard is not None:
        guard.finish()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(parent_launch(args))  # the parent never touches the GPU
    run_rank(args)


if __name__ == "__main__":
    main()
