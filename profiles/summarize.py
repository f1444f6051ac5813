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
    f.write("`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 48 --warmup 8` (collect.sh; every importance launch covers 8 passes)\n\n")
    f.write("| kernel | calls | avg ns | % |\n|---|---|---|---|\n")
    for r in stats:
        f.write(f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |\n")
    # the same kernels by launch size (the bench times SMC with 8 filters per launch and with one; an importance
    # launch covers 8 passes): average duration per (kernel, grid) from the kernel trace of the same run
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(one("trace/*/*kernel_trace.csv"))):
        name = r["Kernel_Name"]
        if any(k in name for k in ("k_resample", "k_tile_sums", "k_scan_tiles", "gjx_plan_kernel")):
            by[(name[:80], int(r["Grid_Size_X"]), int(r["VGPR_Count"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    f.write("\n| kernel | grid (threads) | VGPRs | launches | avg ns |\n|---|---|---|---|---|\n")
    for (name, grid, vg), ts in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"| `{name}` | {grid} | {vg} | {len(ts)} | {sum(ts) / len(ts):.0f} |\n")
    rf = bench["roofline"]
    f.write(f"\nbench.py (un-profiled run): value {bench['value']:.4g} {bench['unit']}, {bench['ms_per_step']*1e3:.1f} us/step; "
            f"dominant kernel `{rf['kernel']}` {rf['kernel_ms']*1e3:.1f} us per launch of {rf.get('passes_per_launch', 1)} passes by HIP events "
            f"(raw interval {rf['kernel_ms_raw_event_interval']*1e3:.1f} us - event-pair overhead {rf['event_pair_overhead_ms']*1e3:.1f} us), "
            f"achieved {rf['achieved']:.0f} GB/s = {rf['frac']:.3f} of 8 TB/s.\n")
    f.write(f"\nHBM traffic of `{dom}` per 1e6-particle pass (PMC, separate runs; a launch covers up to 8 passes): read {traffic['hbm_read_bytes_per_launch']/1e6:.2f} MB "
            f"(FETCH_SIZE x2), write {traffic['hbm_write_bytes_per_launch']/1e6:.2f} MB; algorithmic 48.0 MB.\n")
    if traffic["sq"]:
        f.write("\nSQ counters per pass: " + ", ".join(f"{k}={v:.3g}" for k, v in sorted(traffic["sq"].items())) + "\n")
print(open(os.path.join(here, f"{tag}_summary.md")).read())
