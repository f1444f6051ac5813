#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ (written by profiles/collect.sh) into the committed summaries:
profiles/<tag>_kernel_stats.csv, <tag>_pmc.json (HBM traffic + SQ counters per launch of the dominant
kernel, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md), <tag>_bench.json, <tag>_summary.md."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/profiles_{tag}"
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    files = glob.glob(os.path.join(src, pattern))
    assert files, pattern
    return max(files, key=os.path.getmtime)  # gpurun merges runs into the same tree: take the latest


shutil.copy(one("trace/*/*kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
stats = list(csv.DictReader(open(os.path.join(here, f"{tag}_kernel_stats.csv"))))


# A launch of the importance kernel covers several independent passes (bench.py: LAUNCH = 8; the warm-up and the
# ragged last launch cover fewer), so counters are totalled over the run and divided by the passes it made.
PASSES_PER_RUN = 56  # collect.sh: --steps 48 --warmup 8 (7 launches of 8 passes)


def pmc(dirname, per_pass=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(f"{dirname}/*/*counter_collection.csv"))):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / (PASSES_PER_RUN if per_pass and "gjx_plan_kernel" in k else len(v)) for c, v in cs.items()}
            for k, cs in agg.items()}


fetch, write, sq = pmc("fetch"), pmc("write"), pmc("sq")
dom = next(k for k in write if "gjx_plan_kernel_philox" in k)
fetch_kb, write_kb = fetch[dom]["FETCH_SIZE"], write[dom]["WRITE_SIZE"]
traffic = {
    "kernel": dom,
    "FETCH_SIZE_KB_raw": fetch_kb,
    "WRITE_SIZE_KB": write_kb,
    "per": "1e6-particle pass (a launch covers up to 8 passes)",
    "hbm_read_bytes_per_launch": fetch_kb * 1024 * 2,  # gfx950: FETCH_SIZE reports half of coalesced reads
    "hbm_write_bytes_per_launch": write_kb * 1024,
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "hbm_bytes_per_pass": fetch_kb * 1024 * 2 + write_kb * 1024,
    "passes_per_launch": 1,  # the three figures above are already per pass
    "note": "separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); 4-byte-per-lane "
            "column stores (8 bytes per lane); WRITE_SIZE equals the algorithmic 48 MB per pass within 1 %",
    "sq": sq.get(dom, {}),
}
json.dump(traffic, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
bench = json.loads(open(one("bench_plain.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(here, f"{tag}_bench.json"), "w"), indent=1)
with open(os.path.join(here, f"{tag}_summary.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 evidence (one MI355X)\n\n")
    f.write("`rocprofv3 --kernel-trace --stats -- python3 bench.py` (collect.sh; defaults: 2048-pass clock ramp, 64 warm-up, 1024 timed "
            "passes; every importance launch covers 8 passes, so the timed region is the last 128 launches of `gjx_plan_kernel_philox`; "
            "the first table averages over ALL launches, ramp included)\n\n")
    f.write("| kernel | calls | avg ns | % |\n|---|---|---|---|\n")
    for r in stats:
        f.write(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |\n")
    # the same kernels by launch size (the bench times SMC with 8 filters per launch and with one; an importance
    # launch covers 8 passes): average duration per (kernel, grid) from the kernel trace of the same run
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(one("trace/*/*kernel_trace.csv"))):
        name = r["Kernel_Name"]
        if any(k in name for k in ("k_resample", "k_tile_sums", "k_scan_tiles", "gjx_plan_kernel")):
            by[(name[:80], int(r["Grid_Size_X"]), int(r["VGPR_Count"]))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    by = {k: [d for _, d in sorted(v)] for k, v in by.items()}  # chronological
    f.write("\n| kernel | grid (threads) | VGPRs | launches | avg ns | avg ns, last 128 launches |\n|---|---|---|---|---|---|\n")
    for (name, grid, vg), ts in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        tail = ts[-128:]
        f.write(f"| `{name}` | {grid} | {vg} | {len(ts)} | {sum(ts) / len(ts):.0f} | {sum(tail) / len(tail):.0f} |\n")
    rf = bench["roofline"]
    f.write(f"\nbench.py (un-profiled run): value {bench['value']:.4g} {bench['unit']}, {bench['ms_per_step']*1e3:.1f} us/step; "
            f"dominant kernel `{rf['kernel']}` {rf['kernel_ms']*1e3:.1f} us per launch of {rf.get('passes_per_launch', 1)} passes by HIP events "
            f"(raw interval {rf['kernel_ms_raw_event_interval']*1e3:.1f} us - event-pair overhead {rf['event_pair_overhead_ms']*1e3:.1f} us), "
            f"achieved {rf['achieved']:.0f} GB/s = {rf['frac']:.3f} of 8 TB/s.\n")
    f.write(f"\nHBM traffic of `{dom}` per 1e6-particle pass (PMC, separate runs; a launch covers up to 8 passes): read {traffic['hbm_read_bytes_per_launch']/1e6:.2f} MB "
            f"(FETCH_SIZE x2), write {traffic['hbm_write_bytes_per_launch']/1e6:.2f} MB; algorithmic 48.0 MB.\n")
    if traffic["sq"]:
        f.write("\nSQ counters per pass: " + ", ".join(f"{k}={v:.3g}" for k, v in sorted(traffic["sq"].items())) + "\n")
# HBM traffic of the SMC step kernels, if collected (FETCH_SIZE doubled like above)
smc = {}
for w in ("smc_lgssm", "smc_hmm"):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, f"smc_{w}_{c}", "*", "*counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))
        big = max(int(r["Grid_Size"]) for r in rows if "k_resample" in r["Kernel_Name"])  # the many-filter launches
        top = {}  # every kernel's own largest grid = its many-filter launches (k_tile_sums_wave runs a wave per tile)
        for r in rows:
            top[r["Kernel_Name"]] = max(top.get(r["Kernel_Name"], 0), int(r["Grid_Size"]))
        for r in rows:
            if int(r["Grid_Size"]) == top[r["Kernel_Name"]] and ("k_resample" in r["Kernel_Name"] or "k_tile_sums_wave" in r["Kernel_Name"]):
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            per[k][c] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
            per[k]["particles_per_launch"] = big * 4  # 256 threads per 1024-particle tile
    if per:
        smc[w] = {k: {"hbm_read_bytes": v.get("FETCH_SIZE", 0.0), "hbm_write_bytes": v.get("WRITE_SIZE", 0.0),
                      "particles_per_launch": v["particles_per_launch"]} for k, v in per.items()}
if smc:
    json.dump(smc, open(os.path.join(here, f"{tag}_smc_pmc.json"), "w"), indent=1)
    with open(os.path.join(here, f"{tag}_summary.md"), "a") as f:
        f.write("\nHBM traffic of the SMC step kernels (launches of 16 filters x 1e6 particles; PMC; FETCH_SIZE x2), bytes per particle-step:\n\n")
        f.write("| workload | kernel | read B | write B |\n|---|---|---|---|\n")
        for w, ks in smc.items():
            for k, v in ks.items():
                if "k_resample" in k or "k_tile_sums" in k:
                    f.write(f"| {w} | `{k}` | {v['hbm_read_bytes'] / v['particles_per_launch']:.2f} | {v['hbm_write_bytes'] / v['particles_per_launch']:.2f} |\n")
        f.write("\n(SURVEY §8d counts 44 algorithmic bytes per particle-step for an unfused step; the fused step moves ~20: "
                "state + log-weights read and written once, log-weights read once more for the tile masses.)\n")
print(open(os.path.join(here, f"{tag}_summary.md")).read())
