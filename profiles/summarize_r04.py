#!/usr/bin/env python3
"""gpurun_out/profiles_<tag>/ (profiles/collect_r04.sh) -> the committed summaries under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5`
  <tag>_summary.md         the same per (kernel, grid) with average / median durations
  <tag>_pmc.json           ImportanceK kernel: HBM bytes per pass (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md
                           HBM section) and SQ counters per pass, exact and fast-math plans (bench.py reads it)
  <tag>_smc_pmc.json       one-filter SMC step kernels: HBM bytes and SQ counters per launch (bench.py reads it)
  <tag>_scan_pmc.json      the one-launch scan kernel
  <tag>_bench.json         the plain bench line of the same build"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

tag, src = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def newest(pattern):
    files = glob.glob(os.path.join(src, pattern), recursive=True)
    assert files, pattern
    return max(files, key=os.path.getmtime)


def counters(dirname):
    """{kernel: {counter: [values per dispatch]}} plus {kernel: grid}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"{dirname}/**/*counter_collection.csv"))):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def kernel_block(w, match, per_launch_units, unit_name):
    """Counters of the kernels of workload `w` whose name contains one of `match`: averages per launch."""
    out = {}
    sets = {k: counters(f"{w}_{k}") for k in ("fetch", "write", "sq1", "sq2")}
    keys = sorted({k for k in sets["write"] if any(m in k[0] for m in match)}, key=lambda k: -k[1])
    for name, grid in keys:
        e = {"grid_threads": grid}
        f = sets["fetch"].get((name, grid), {}).get("FETCH_SIZE", [])
        wv = sets["write"].get((name, grid), {}).get("WRITE_SIZE", [])
        if f and wv:
            e["launches_counted"] = len(wv)
            e["FETCH_SIZE_KB_raw"] = sum(f) / len(f)
            e["WRITE_SIZE_KB"] = sum(wv) / len(wv)
            e["hbm_read_bytes"] = e["FETCH_SIZE_KB_raw"] * 1024 * 2  # gfx950: FETCH_SIZE counts half of coalesced reads
            e["hbm_write_bytes"] = e["WRITE_SIZE_KB"] * 1024
        sq = {}
        for sname in ("sq1", "sq2"):
            for c, v in sets[sname].get((name, grid), {}).items():
                sq[c] = sum(v) / len(v)
        e["sq_per_launch"] = sq
        e[unit_name] = per_launch_units
        out[f"{name[:90]} [grid {grid}]"] = e
    return out


# ---- kernel trace of the driver's command
shutil.copy(newest("trace/**/*kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
stats = list(csv.DictReader(open(os.path.join(here, f"{tag}_kernel_stats.csv"))))
by = collections.defaultdict(list)
for r in csv.DictReader(open(newest("trace/**/*kernel_trace.csv"))):
    by[(r["Kernel_Name"][:90], int(r["Grid_Size_X"]), int(r["VGPR_Count"]), int(r.get("SGPR_Count", 0) or 0))].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(here, f"{tag}_summary.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 evidence (one MI355X)\n\n`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 "
            "--no-cpu-baseline` (profiles/collect_r04.sh) — the driver's command: headline ImportanceK (20 passes per launch at --steps 20), every "
            "`extra` entry (one-filter SMC LGSSM / HMM (ONE k_resample launch per step), ESS-adaptive, 16 filters per launch, collapsing weights, the one-launch Scan, "
            "1 pass per launch, threefry, fast math).\n\n## --stats (all launches of the run, warm-up and clock ramp included)\n\n")
    f.write("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
    for r in stats:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |\n")
    f.write("\n## per (kernel, grid): the launch shapes the bench line quotes\n\n| kernel | grid (threads) | VGPR | SGPR | launches | avg ns | median ns | min ns |\n|---|---|---|---|---|---|---|---|\n")
    for (name, grid, vg, sg), ts in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        if len(ts) < 3:
            continue
        f.write(f"| `{name}` | {grid} | {vg} | {sg} | {len(ts)} | {sum(ts) / len(ts):.0f} | {statistics.median(ts):.0f} | {min(ts)} |\n")

# ---- r04: one trace per workload (each BASELINE config has its own rows)
with open(os.path.join(here, f"{tag}_summary.md"), "a") as f:
    for w, what in (("smc_lgssm", "configs[2]: `bench.py --workload smc_lgssm` (one filter of 1e6 particles, T = 100; nothing else in the run)"),
                    ("smc_hmm", "configs[4]: `bench.py --workload smc_hmm` (one filter, HMM-256, T = 500)"),
                    ("scan_hmm", "`bench.py --workload scan_hmm` (ImportanceK over the HMM as a one-launch scan)"),
                    ("sharded_rank0", "one rank of configs[3] on one device: `tools/time_sharded_rank0.py` (8 virtual ranks x 1e6 particles, 7 816 tile records; peer transport and the kernels of a collective rank)")):
        try:
            by_w = collections.defaultdict(list)
            for r in csv.DictReader(open(newest(f"trace_{w}/**/*kernel_trace.csv"))):
                by_w[(r["Kernel_Name"][:90], int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            f.write(f"\n## {what}\n\n| kernel | grid (threads) | launches | avg ns | median ns | min ns |\n|---|---|---|---|---|---|\n")
            for (name, grid), ts in sorted(by_w.items(), key=lambda kv: -sum(kv[1]))[:8]:
                if len(ts) >= 3:
                    f.write(f"| `{name}` | {grid} | {len(ts)} | {sum(ts) / len(ts):.0f} | {statistics.median(ts):.0f} | {min(ts)} |\n")
        except Exception as ex:  # reported, not fatal
            f.write(f"\n## {what}\n\n(no trace: {ex!r})\n")
    for extra_txt, title in (("phases_smc.txt", "phase times of the shipped one-filter LGSSM step (tools/phases_smc.sh: cumulative us per step by early exit, profiling build)"),
                             ("pmc_samplers.txt", "rejection samplers: SQ counters per launch of 8 passes (tools/pmc_samplers.sh); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU)")):
        pth = os.path.join(src, extra_txt)
        if os.path.exists(pth):
            f.write(f"\n## {title}\n\n```\n{open(pth).read().strip()}\n```\n")

# ---- ImportanceK: per pass (every launch of these runs covers 8 passes of 1e6 particles)
imp = {}
for w, label in (("importance", "exact"), ("importance_fast", "fast_math")):
    try:
        blk = kernel_block(w, ("gjx_plan_kernel_philox",), 8, "passes_per_launch")
        name, e = next(iter(blk.items()))
        e["kernel"] = name
        e["hbm_bytes_per_pass"] = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / 8
        e["sq_per_pass"] = {c: v / 8 for c, v in e["sq_per_launch"].items()}
        imp[label] = e
    except Exception as ex:  # (a pass that was not collected this round)
        imp[label] = {"error": repr(ex)}
pmc = dict(imp["exact"])
pmc["fast_math"] = imp["fast_math"]
try:
    blk = kernel_block("importance_threefry", ("gjx_plan_kernel_threefry",), 8, "passes_per_launch")
    name, e = next(iter(blk.items()))
    e["kernel"] = name
    e["hbm_bytes_per_pass"] = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / 8
    pmc["threefry"] = e
except Exception as ex:  # reported, not fatal
    pmc["threefry"] = {"error": repr(ex)}
pmc["note"] = ("separate --pmc passes of `GJX_BENCH_LAUNCH=8 bench.py --steps 48 --warmup 8 --no-extra` (every launch: 8 passes of 1e6 particles); FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md; algorithmic bytes per pass = 48 MB (10 value columns + score + logw, 4 B each)")
json.dump(pmc, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)

# ---- one-filter SMC step kernels: per launch = one step of 1e6 particles
smc = {}
for w in ("smc_lgssm", "smc_hmm"):
    blk = kernel_block(w, ("k_resample",), 1_000_000, "particles_per_launch")
    smc[w] = {k: v for k, v in blk.items() if v["grid_threads"] >= 256 * 900}  # the full-population launches
    try:  # r04: parked wave-cycles (s_waitcnt / barriers), LDS instructions, lane utilisation
        for (name, grid), cs in counters(f"{w}_sq3").items():
            for k, e in smc[w].items():
                if k.startswith(name[:90]) and e["grid_threads"] == grid:
                    e["sq3_per_launch"] = {c: sum(v) / len(v) for c, v in cs.items()}
                    q = e["sq3_per_launch"]
                    if q.get("SQ_WAVE_CYCLES"):
                        e["parked_fraction_SQ_WAIT_ANY_over_WAVE_CYCLES"] = q.get("SQ_WAIT_ANY", 0.0) / q["SQ_WAVE_CYCLES"]
                    if q.get("SQ_INSTS_VALU"):
                        e["valu_lane_utilisation"] = q.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * q["SQ_INSTS_VALU"])
    except Exception as ex:
        smc[w + "_sq3_error"] = repr(ex)
smc["note"] = "one filter of 1e6 particles (the literal BASELINE configs): per launch = per SMC step; FETCH_SIZE doubled (gfx950)"
json.dump(smc, open(os.path.join(here, f"{tag}_smc_pmc.json"), "w"), indent=1)

scan = {}
for w, units in (("scan_lgssm", 100_000_000), ("scan_hmm", 500_000_000)):
    try:
        blk = kernel_block(w, ("gjx_scan_kernel",), units, "particle_steps_per_launch")
        try:  # L2 hits / misses / requests per launch (their own pass)
            for (name, grid), cs in counters(f"{w}_l2").items():
                for k, e in blk.items():
                    if k.startswith(name[:90]) and e["grid_threads"] == grid:
                        e["l2_per_launch"] = {c: sum(v) / len(v) for c, v in cs.items()}
        except Exception as ex:  # reported, not fatal
            for e in blk.values():
                e["l2_per_launch"] = {"error": repr(ex)}
        for e in blk.values():
            sq = e.get("sq_per_launch", {})
            if "SQ_INSTS_VALU" in sq:
                e["valu_wave_instructions_per_particle_step"] = sq["SQ_INSTS_VALU"] * 64 / e["particle_steps_per_launch"]
        scan[w] = blk
    except Exception as ex:
        scan[w] = {"error": repr(ex)}
scan["note"] = ("one launch = the whole scan of the population (LGSSM: 1e6 particles x 100 steps; HMM-256: 1e6 x 500); FETCH_SIZE doubled "
                "(gfx950); valu_wave_instructions_per_particle_step = SQ_INSTS_VALU x 64 lanes / particle-steps")
json.dump(scan, open(os.path.join(here, f"{tag}_scan_pmc.json"), "w"), indent=1)

bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(here, f"{tag}_bench.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "bench_detail.json")):
    shutil.copy(os.path.join(src, "bench_detail.json"), os.path.join(here, f"{tag}_bench_detail.json"))
print("summaries written:", sorted(x for x in os.listdir(here) if x.startswith(tag)))
