#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload importance|smc_lgssm|smc_hmm]
                  [--rng philox|threefry] [--no-cpu-baseline] [--no-extra]

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
  importance (default, BASELINE configs[1]): ImportanceK on the 10-latent Gaussian model, 1e6
      particles per GPU: fused `@gen`-body kernel (RNG -> samplers -> SoA trace columns -> score,
      log-weights, per-tile max) + fixed-point log-sum-exp.  value = particles/s.
  smc_lgssm (configs[2]): bootstrap SMC, T=100, 1e6 particles per GPU.  value = particle-steps/s.
  smc_hmm   (configs[4]): 256-state HMM, T=500.
For N>1 (launched by torch.distributed.run, one rank per GPU) the particle population is sharded
(weak scaling); importance exchanges one 520-byte record per pass (all-gather, bucketed and overlapped).
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
RAMP_PASSES = int(os.environ.get("GJX_BENCH_RAMP", "2048"))  # untimed passes before the warm-up (clock ramp, ~35 ms)
N_PER_GPU = 1_000_000
# Algorithmic bytes per unit (SURVEY §8d / DESIGN.md §5)
# SURVEY §8d counts 52 B/particle for the pass (the last 4 are the log-sum-exp's re-read of logw, which the
# fused row-anchored partial sums made unnecessary); the dominant kernel's algorithmic share:
BYTES_IMPORTANCE_KERNEL_PER_PARTICLE = 48  # 10 latent columns + score + logw written
BYTES_SMC_PER_PARTICLE_STEP = 44


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1024)  # multiples of the 8 passes per launch: no ragged launch by default
    p.add_argument("--warmup", type=int, default=64)
    p.add_argument("--workload", default="importance", choices=["importance", "smc_lgssm", "smc_hmm"])
    p.add_argument("--rng", default="philox", choices=["philox", "threefry"])
    p.add_argument("--particles", type=int, default=N_PER_GPU, help="particles per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true")
    return p.parse_args()


FORCE_DIST = bool(os.environ.get("GJX_BENCH_FORCE_DIST"))  # exercise the N>1 code path with one rank


def init_dist(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert world == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world}"
    else:
        torch.cuda.set_device(0)
    return rank, world


def barrier_sync(world):
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(seconds, world):
    if world == 1 and not FORCE_DIST:
        return seconds
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_importance(args, ops, rank, world):
    import torch.distributed as dist

    from genjax._amd import workloads as W

    impl = 1 if args.rng == "philox" else 0
    sharded = world > 1 or FORCE_DIST
    total_particles = args.particles * world
    if sharded:
        from genjax._amd import dist as gdist

        # weak scaling: world x 1e6 particles split at 256-particle row boundaries (per-rank counts differ
        # by at most one row), which keeps the exact log-normaliser independent of the number of ranks
        first, n = gdist.shard_rows(total_particles, rank, world)
    else:
        first, n = 0, args.particles
    wl = W.Gaussian10(ops, impl, seed=0, n_local=n, first=first, n_total=total_particles)
    kernel_ms = []
    # HIP events are created (and their pool grown) before the timed region
    from genjax._amd.ops import HipEvent

    ev_pool = [(HipEvent(), HipEvent()) for _ in range(args.steps + 8)]  # (events bracket timed launches only)

    # BATCH passes share one log-sum-exp launch (and, sharded, one exchanged block of records): 32 amortises a
    # ~35 us per-batch cost to ~1 us per pass.  LAUNCH independent passes (seeds 0, 1, ...) share one importance
    # launch: a single 1e6-particle pass is under two rounds of the machine (21.5 us), eight keep it full
    # (17.5 us per pass).  Persistent output buffers + pre-marshalled C calls: no host allocation per step.
    LAUNCH = int(os.environ.get("GJX_BENCH_LAUNCH", "8"))
    BATCH = int(os.environ.get("GJX_BENCH_BATCH", "32")) // LAUNCH * LAUNCH or LAUNCH
    prep = wl.prepare(fold_batch=BATCH, passes=LAUNCH)
    if sharded:
        # each pass leaves a 65-word record of its shard's weights; one asynchronous all-gather per BATCH
        # passes overlaps with the next batch's kernels (genjax/_amd/dist.py)
        pipe = gdist.BatchedImportance(ops, wl, batch=BATCH, world=world, always_exchange=True, passes=LAUNCH)
    last_buf = [0]

    launch_no = [0]

    def on_launch(phase, count, evs, timed):
        """HIP events around every fourth importance launch (a barrier packet each: ~2.5 us against ~120 us)."""
        if phase == 0:
            launch_no[0] += 1 if timed else 0  # (every fourth TIMED launch, the first one included)
            if timed and launch_no[0] % 4 == 1:
                evs.append(ev_pool.pop() + (count,))
                evs[-1][0].record(ops.stream())
            else:
                evs.append(None)
        elif evs[-1] is not None:
            evs[-1][1].record(ops.stream())
            if timed:
                kernel_ms.append(evs[-1])

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
            return prep.e_all[:1], prep.q_all[:1], prep.logw
        pipe.wait()  # every exchange has landed (stream-ordered; the host does not block)
        _, e_all, q_all = pipe.results(last_buf[0])
        return e_all[:1], q_all[:1], None

    # Clock ramp: the device reaches its sustained clocks only after tens of milliseconds of load (measured on the box:
    # 120-128 us per launch in the first milliseconds, a dip to ~140 us between ~4 and ~20 ms, 115-119 us from ~30 ms
    # on), so a fixed untimed run of the same launches precedes the W warm-up steps and the timed region sees the
    # steady state whatever W and K are.
    run_steps(max(0, RAMP_PASSES - args.warmup), False)
    run_steps(args.warmup, False)
    # An event record is a barrier packet in the HIP queue; a back-to-back pair with nothing in
    # between measures that fixed cost, which is subtracted from the kernel intervals.
    cal = [(HipEvent(), HipEvent()) for _ in range(16)]
    sh_cal = ops.stream()
    for a, b in cal:
        a.record(sh_cal)
        b.record(sh_cal)
    barrier_sync(world)
    ev_overhead_ms = sorted(a.elapsed_ms(b) for a, b in cal)[len(cal) // 2]
    t0 = time.perf_counter()
    m, q, logw = run_steps(args.steps, True)  # EXACTLY args.steps passes
    t_loop = time.perf_counter() - t0
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    host_ts = [t_loop / args.steps]
    if os.environ.get("GJX_BENCH_DEBUG"):
        print("host per-step ms:", ["%.3f" % (x * 1e3) for x in host_ts], "loop", t_loop * 1e3, "total", dt * 1e3,
              file=sys.stderr)
    # the dominant kernel: launches of `c` passes each; full launches (c == LAUNCH) define the quoted duration
    full = [(a, b, c) for a, b, c in kernel_ms if c == LAUNCH] or kernel_ms
    k_ms_raw = sum(a.elapsed_ms(b) for a, b, _ in full) / len(full)
    k_ms = max(k_ms_raw - ev_overhead_ms, 1e-6)
    passes_per_launch = full[0][2]
    ms_per_step = dt / args.steps * 1e3
    log_z = ops.log_z_from_rows(m, q, total_particles)  # exact (anchor, fixed-point sum) pair of pass 0
    bytes_per_launch = BYTES_IMPORTANCE_KERNEL_PER_PARTICLE * n * passes_per_launch
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
    # collected by profiles/collect.sh on this same command); null if none matches this kernel/size.
    traffic, traffic_src = None, None
    if n == N_PER_GPU:
        import glob

        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
            try:
                pj = json.load(open(f))
                if f"gjx_plan_kernel_{args.rng}" in pj.get("kernel", ""):
                    # the PMC figure is per 1e6-particle pass; a launch of several passes moves that many times it
                    per_pass = pj.get("hbm_bytes_per_pass", pj["hbm_bytes_per_launch"] / max(1, pj.get("passes_per_launch", 1)))
                    traffic, traffic_src = per_pass * passes_per_launch, os.path.relpath(f, ROOT)
                    break
            except Exception:
                pass
    res = {
        "metric": "particles/sec, ImportanceK log-marginal-likelihood estimate (1e6 particles per GPU)",
        "value": total_particles / (dt / args.steps),
        "unit": "particles/s",
        "ms_per_step": ms_per_step,
        "config": {"workload": "ImportanceK k_particles=1e6/GPU on a 10-latent Gaussian model (BASELINE configs[1])",
                   "particles_per_gpu": args.particles, "latent_sites": 10, "observed_sites": 10, "rng": args.rng,
                   "passes_per_launch": LAUNCH, "passes_per_fold": BATCH,
                   "clock_ramp_passes_before_warmup": max(0, RAMP_PASSES - args.warmup),
                   "parallelism": (f"particle-sharded x{world}, row-aligned; one 520 B all-gather per pass, "
                                   f"bucketed x{BATCH} and overlapped with the next batch") if sharded
                   else "single device"},
        "roofline": {"bound": "hbm", "kernel": f"gjx_plan_kernel_{args.rng}", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel_ms": k_ms, "kernel_ms_raw_event_interval": k_ms_raw,
                     "event_pair_overhead_ms": ev_overhead_ms, "kernel_launches_timed": len(full),
                     "passes_per_launch": passes_per_launch, "kernel_ms_per_pass": k_ms / passes_per_launch,
                     "algorithmic_bytes_per_launch": bytes_per_launch},
        "log_z": log_z,
        "log_z_exact": W.gaussian10_exact_log_z(wl.y),
    }
    return res, wl


def bench_smc(args, ops, rank, world, kind, filters=0):
    from genjax._amd import workloads as W

    impl = 1 if args.rng == "philox" else 0
    n = args.particles
    T = 100 if kind == "smc_lgssm" else 500
    if world > 1 or FORCE_DIST:
        from genjax._amd import dist as gdist

        # the sharded filter exchanges whole 1024-particle tiles: round the per-GPU population up
        n = -(-n // ops.tile) * ops.tile
        n_total = n * world
        exchange = os.environ.get("GJX_BENCH_SHUFFLE", "ranges")
        smc = gdist.ShardedSMC(ops, kind[4:], impl, 1 if kind == "smc_lgssm" else 2, n_total, T, rank, world,
                               exchange=exchange)
        smc.run()
        barrier_sync(world)
        steps = max(1, min(args.steps, 5))
        smc.received = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            r = smc.run()
        barrier_sync(world)
        dt = max_over_ranks((time.perf_counter() - t0) / steps, world)
        per_step_ms = dt * 1e3 / T
        achieved = BYTES_SMC_PER_PARTICLE_STEP * n / (per_step_ms * 1e-3) / 1e9
        shuffle = ("ancestor shuffle = grouped send/recv of each rank's contiguous source range (all-to-all-v, in place)"
                   if exchange == "ranges" else "all-gather of particles and weights")
        return {
            "metric": "particle-steps/sec, bootstrap SMC (1e6 particles per GPU)",
            "value": n_total * T / dt, "unit": "particle-steps/s", "ms_per_step": dt * 1e3,
            "config": {"workload": f"bootstrap SMC {kind} T={T} N={n_total} sharded x{world}", "rng": args.rng,
                       "parallelism": f"particle-sharded x{world}: all-reduce(max) + all-gather of tile masses + {shuffle}",
                       "particles_received_per_rank_step": r["received"] / steps / max(1, T - 1)},
            "roofline": {"bound": "hbm", "kernel": "one SMC step incl. exchange", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "step_ms": per_step_ms, "algorithmic_bytes_per_launch": BYTES_SMC_PER_PARTICLE_STEP * n},
            "log_z": r["log_z"], "log_z_exact": r["log_z_exact"],
        }
    # FILTERS independent filters (seeds s, s+1, ...) step in the same launches: a 1e6-particle step is ~1000
    # workgroups, under one round of the machine
    FILTERS = filters if filters else int(os.environ.get("GJX_BENCH_FILTERS", "16"))
    wl = W.LgssmSMC(ops, impl, 1, n, T, filters=FILTERS) if kind == "smc_lgssm" else W.HmmSMC(ops, impl, 2, n, T, filters=FILTERS)
    # warm-up long enough for the clock ramp (see bench_importance): ~50 ms of the same launches
    for _ in range(8 if kind == "smc_lgssm" else 2):
        out = wl.run()
    barrier_sync(world)
    steps = max(1, min(args.steps, 10 if kind == "smc_lgssm" else 3))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()
        out = wl.run()  # enqueue only: data, keys and tables were prepared above
        e1.record()
    barrier_sync(world)
    dt = (time.perf_counter() - t0) / steps
    r = wl.result(out)
    dev_ms = sum(a.elapsed_time(b) for a, b in evs) / steps
    per_step_ms = dev_ms / T
    achieved = BYTES_SMC_PER_PARTICLE_STEP * n * FILTERS / (per_step_ms * 1e-3) / 1e9
    log_z = r["log_z"][0] if FILTERS > 1 else r["log_z"]
    # HBM bytes of one step (k_resample + k_tile_sums) from the PMC passes committed under profiles/ (bytes per
    # particle-step of the many-filter launches; FETCH_SIZE x2 + WRITE_SIZE), scaled to this launch's particles
    traffic, traffic_src = None, None
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_smc_pmc.json")), reverse=True):
        try:
            ks = json.load(open(f)).get(kind, {})
            per = sum((v["hbm_read_bytes"] + v["hbm_write_bytes"]) / v.get("particles_per_launch", 8e6) for k, v in ks.items()
                      if "k_resample" in k or "k_tile_sums" in k)
            if per > 0:
                traffic, traffic_src = per * n * FILTERS, os.path.relpath(f, ROOT)
                break
        except Exception:
            pass
    return {
        "metric": "particle-steps/sec, bootstrap SMC (1e6 particles per GPU)",
        "value": n * T * FILTERS / dt,
        "unit": "particle-steps/s",
        "ms_per_step": dt * 1e3,
        "config": {"workload": f"bootstrap SMC {kind} T={T} N={n}", "rng": args.rng, "filters_per_launch": FILTERS},
        "roofline": {"bound": "hbm", "kernel": "k_resample+k_tile_sums (one SMC step of every filter)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src, "step_ms": per_step_ms, "step_ms_per_filter": per_step_ms / FILTERS,
                     "algorithmic_bytes_per_launch": BYTES_SMC_PER_PARTICLE_STEP * n * FILTERS},
        "log_z": log_z, "log_z_exact": r["log_z_exact"],
    }


def cpu_baseline(args):
    """The CPU oracle (a port: plain C + OpenMP restatement) timed on this box's host cores on a
    bounded sample of the same workload."""
    from genjax._amd import workloads as W
    from genjax._amd.abi import GjxLib
    from genjax._amd.ops import Ops

    lib = os.path.join(ROOT, "oracle", "libgjx_oracle.so")
    if not os.path.exists(lib):
        return None
    ora = Ops(GjxLib(lib, "cpu"))
    impl = 1 if args.rng == "philox" else 0
    cores = os.cpu_count() or 1
    if args.workload == "importance":
        n = args.particles
        wl = W.Gaussian10(ora, impl, seed=0, n_local=n)
        wl.step()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            wl.step()
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": n * reps / dt, "unit": "particles/s", "cores": cores, "kind": "port",
                "sample": f"{reps} full passes of the same {n}-particle ImportanceK workload (OpenMP, {cores} threads)"}
    T = 10
    n = args.particles
    fn = W.lgssm_smc if args.workload == "smc_lgssm" else W.hmm_smc
    seed = 1 if args.workload == "smc_lgssm" else 2
    fn(ora, impl, seed, n, 2)
    t0 = time.perf_counter()
    fn(ora, impl, seed, n, T)
    dt = time.perf_counter() - t0
    return {"value": n * T / dt, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"first {T} steps of the same {n}-particle filter (propagate/weight OpenMP over {cores} threads, "
                      "resampling scan sequential)"}


def main():
    args = parse()
    rank, world = init_dist(args.gpus)
    from genjax._amd.runtime import load_hip_ops

    ops = load_hip_ops()  # raises without libgjx_hip.so / a GPU: there is no CPU fallback
    if args.workload == "importance":
        res, _ = bench_importance(args, ops, rank, world)
    else:
        res = bench_smc(args, ops, rank, world, args.workload)
    if rank == 0:
        out = {
            "metric": res.pop("metric"), "value": res.pop("value"), "unit": res.pop("unit"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": res.pop("config"), "roofline": res.pop("roofline"),
        }
        out.update(res)
        out["log_z_abs_err_vs_exact"] = abs(out["log_z"] - out["log_z_exact"])
        if world == 1 and not args.no_extra and args.workload == "importance":
            extra = {}
            for kind in ("smc_lgssm", "smc_hmm"):
                a2 = argparse.Namespace(**vars(args))
                a2.steps, a2.warmup = (10, 1) if kind == "smc_lgssm" else (3, 1)
                r = bench_smc(a2, ops, rank, world, kind)
                extra[kind] = {k: r[k] for k in ("value", "unit", "ms_per_step", "roofline", "log_z", "log_z_exact")}
                extra[kind]["filters_per_launch"] = r["config"]["filters_per_launch"]
                r1 = bench_smc(a2, ops, rank, world, kind, filters=1)  # the literal BASELINE config: ONE filter of 1e6 particles
                extra[kind]["one_filter"] = {"value": r1["value"], "ms_per_step": r1["ms_per_step"],
                                             "roofline_frac": r1["roofline"]["frac"], "step_ms": r1["roofline"]["step_ms"]}
            a2 = argparse.Namespace(**vars(args))
            a2.rng = "threefry" if args.rng == "philox" else "philox"
            a2.steps, a2.warmup = 20, 3
            r, _ = bench_importance(a2, ops, rank, world)
            extra[f"importance_{a2.rng}"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "roofline", "log_z")}
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1 or FORCE_DIST:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
