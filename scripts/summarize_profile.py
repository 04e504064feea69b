#!/usr/bin/env python3
"""Condense a gpurun_out/<run>/ directory of rocprofv3 CSVs into profiles/<tag>_summary.{md,json}.

Per (kernel, grid size): calls, avg/min/max duration from --kernel-trace; PMC
averages from the --pmc passes; HBM traffic per launch derived as the
MI355X_MICROARCH guide prescribes for gfx950 (FETCH_SIZE is in KiB and counts a
128-B fabric read as 64 B -> x2; WRITE_SIZE in KiB is exact)."""
import collections
import csv
import glob
import json
import os
import sys

run, tag = sys.argv[1], sys.argv[2]
out_md = os.path.join("profiles", tag + "_summary.md")
out_js = os.path.join("profiles", tag + "_summary.json")


def short(name):
    n = name.replace("void ", "").replace("spmvhip::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


trace = collections.defaultdict(list)
for p in glob.glob(os.path.join(run, "trace*", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(p)):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) if "Grid_Size_X" in r else int(r["Grid_Size"]))
        trace[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(run, "pmc*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        pmc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))

summary = {}
lines = [f"# rocprofv3 summary `{tag}` (source: {run})", "",
         "## kernel trace (`rocprofv3 --kernel-trace --stats`)", "",
         "| kernel | grid (work-items) | calls | avg us | min us | max us |", "|---|---|---|---|---|---|"]
for (k, g), d in sorted(trace.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"| {k} | {g} | {len(d)} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {max(d) / 1e3:.1f} |")
    summary.setdefault(f"{k}@{g}", {})["trace"] = {"calls": len(d), "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3}
lines += ["", "## PMC (`rocprofv3 --pmc ...`, separate passes; per-launch averages)", ""]
for (k, g), c in sorted(pmc.items()):
    if "stream" not in k and "vector" not in k and "scalar" not in k and "ell" not in k and "pb_" not in k:
        continue
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    lines.append(f"### {k} grid {g}")
    lines.append("")
    for n in sorted(avg):
        lines.append(f"- {n}: {avg[n]:.6g}")
    if "FETCH_SIZE" in avg:
        rd = avg["FETCH_SIZE"] * 1024 * 2
        wr = avg.get("WRITE_SIZE", 0) * 1024
        lines.append(f"- **HBM-side traffic per launch** = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = {rd / 1e9:.3f} GB read"
                     + (f" + {wr / 1e9:.3f} GB written" if wr else " (WRITE_SIZE not in this pass)"))
        avg["traffic_read_bytes"] = rd
        avg["traffic_write_bytes"] = wr
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        lines.append(f"- L2 hit rate = {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}")
    lines.append("")
    summary.setdefault(f"{k}@{g}", {})["pmc"] = avg
os.makedirs("profiles", exist_ok=True)
open(out_md, "w").write("\n".join(lines) + "\n")
json.dump(summary, open(out_js, "w"), indent=1)
print("wrote", out_md, out_js)
