#!/usr/bin/env python3
"""gpurun_out/<round>c (written by scripts/final_profile.sh on the GPU box) -> profiles/<round>_summary.md,
profiles/<round>_default_bench_kernel_stats.csv, profiles/<round>_bench_default.json and profiles/traffic.json
(stamped with the sha256 of the library that was profiled: bench.py reports a traffic figure only for that build).
    usage: python3 scripts/make_summary.py r03"""
import collections, csv, glob, json, os, shutil, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
O = f"gpurun_out/{RND}c"
def short(n): return n.replace("void ", "").replace("spmvhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
def latest(pattern):
    """gpurun MERGES a call's files into gpurun_out/: of several runs of the same pass keep the newest file only"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]
sha = open(f"{O}/libspmvhip.sha256").read().strip()
lines = [f"# {RND} profile summary (rocprofv3, one MI355X) -- `scripts/final_profile.sh {RND}` + `scripts/make_summary.py {RND}`", "",
         f"Library profiled: `libspmvhip.so` sha256 `{sha}`.", "",
         "Per workload: `rocprofv3 --kernel-trace --stats` durations and, from separate `--pmc` passes, HBM-side traffic",
         "per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md: a 128-B fabric read is",
         "tallied as 64 B).  Algorithmic bytes: nnz*12 + M*12 + N*8.", ""]
traffic = {"_doc": "HBM-side bytes per SpMV from rocprofv3 PMC passes (2 x FETCH_SIZE KiB + WRITE_SIZE KiB); bench.py copies the entry matching "
                   f"workload+kernel into roofline.traffic when the loaded library has this sha256; source profiles/{RND}_summary.md",
           "_libspmvhip_sha256": sha, "_measured": f"profiles/{RND}_summary.md"}
wlname = {"c5": "c5-powerlaw-80M-1.6G", "c3": "c3-powerlaw-10M-200M", "c3_tiles": "c3-powerlaw-10M-200M", "c3_onepass": "c3-powerlaw-10M-200M",
          "c2_stripes": "c2-uniform-1M-32", "c2_sell": "c2-uniform-1M-32", "c3b": "c3b-powerlaw-10M-200M-band",
          "c5_serial": "c5-powerlaw-80M-1.6G", "c3_serial": "c3-powerlaw-10M-200M", "c2_serial": "c2-uniform-1M-32"}
alg = {"c5": 20.8e9, "c3": 2.6e9, "c3_tiles": 2.6e9, "c3_onepass": 2.6e9, "c2_stripes": 0.404e9, "c2_sell": 0.404e9, "c3b": 2.6e9,
       "c5_serial": 20.8e9, "c3_serial": 2.6e9, "c2_serial": 0.404e9}
KEEP = ("csr_stream", "pb_expand", "pb_reduce", "csr_scalar", "csr_vector", "sell_spmv", "sell_long", "sb_spmv")
LAUNCHER = (("pb_reduce_det", "hipSpMVRowsCSR"), ("sb_spmv_kernel<false, true", "hipSpMVRowsCSR"), ("sb_spmv_kernel<true, true", "hipSpMVRowsCSR"),
            ("pb_", "hipSpMVTilesCSR"), ("sb_spmv", "hipSpMVStripesCSR"), ("sell", "hipSpMVRowsSELL"),
            ("csr_stream", "hipSpMVWarpPerRowCSR"))
for tag in wlname:
    tr = collections.defaultdict(list)
    for p in latest(f"{O}/{tag}/trace/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if any(x in k for x in KEEP): tr[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not tr: continue
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in [f for d in glob.glob(f"{O}/{tag}/pmc_*") for f in latest(f"{d}/*/*_counter_collection.csv")]:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k in tr: pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the block's timed launcher = the kernel with the most calls (the candidates of the bench's own pick and of the library's
    # two selections run a few times each, too); the two-phase launcher is two kernels
    top = max(tr, key=lambda k: len(tr[k]))
    maxcalls = len(tr[top])
    tr = {k: v for k, v in tr.items() if k == top or (top.startswith("pb_") and k.startswith("pb_"))}
    lines += [f"## {tag}: {wlname[tag]}  (algorithmic {alg[tag] / 1e9:.2f} GB)", "",
              "| kernel | calls | avg us | min us | HBM read GB | HBM written GB | L2 hit | WAIT_ANY / WAIT_INST_ANY / ACTIVE of wave cycles |",
              "|---|---|---|---|---|---|---|---|"]
    tot = tott = 0
    used = []
    for k, d in sorted(tr.items(), key=lambda kv: -len(kv[1])):
        if len(d) < maxcalls * 0.6: continue
        d = sorted(d)[: max(1, len(d) - 3)] if "sb_spmv" in k or "pb_" in k else d      # first calls of a format-building launcher are not slower, but drop warm-up outliers
        a = {n: sum(v) / len(v) for n, v in pm[k].items()}
        rd, wr = a.get("FETCH_SIZE", 0) * 2048, a.get("WRITE_SIZE", 0) * 1024
        hit = a.get("TCC_HIT_sum", 0) / max(a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0), 1)
        wc = a.get("SQ_WAVE_CYCLES", 0)
        sq = f"{a.get('SQ_WAIT_ANY', 0) / wc:.2f} / {a.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} / {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}" if wc else "-"
        lines.append(f"| {k} | {len(d)} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {hit:.2f} | {sq} |")
        tot += rd + wr; tott += sum(d) / len(d); used.append(k)
    lines += ["", f"SpMV = {tott / 1e3:.1f} us, HBM-side traffic {tot / 1e9:.2f} GB = {tot / alg[tag]:.2f} x algorithmic; algorithmic rate "
                  f"{alg[tag] / tott:.0f} GB/s = {100 * alg[tag] / tott / 8000:.1f} % of 8 TB/s", ""]
    launcher = next(l for key, l in LAUNCHER if any(key in k for k in used))
    if tot:
        traffic.setdefault(wlname[tag], {})[launcher] = tot
# the structured stand-ins: one SpMV kernel per launcher, all in one run -> one table per matrix (time + HBM-side traffic per launch)
STRUCT = {"stencil": ("stencil3d-500x100x100 (5 M rows, 88.9 M entries, 18 slots)", 88902800 * 12 + 5000000 * 20),
          "road": ("road-12M (12 M rows, 25.7 M entries, 9 slots)", 25733999 * 12 + 12000000 * 20),
          "stencil_pattern": ("stencil3d-500x100x100 as a PATTERN file (all values 1.0: no value stream; 0.456 GB are left to move)", 88902800 * 12 + 5000000 * 20),
          "road_pattern": ("road-12M as a PATTERN file (all values 1.0: no value stream; 0.343 GB are left to move)", 25733999 * 12 + 12000000 * 20)}
SKEEP = KEEP + ("ell_",)
for tag, (title, bcsr) in STRUCT.items():
    tr = collections.defaultdict(list)
    for p in latest(f"{O}/{tag}/trace/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("void ", "").replace("spmvhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
            if any(x in k for x in SKEEP): tr[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not tr: continue
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in [f for d in glob.glob(f"{O}/{tag}/pmc_*") for f in latest(f"{d}/*/*_counter_collection.csv")]:
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("void ", "").replace("spmvhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
            if k in tr: pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += [f"## structured: {title}  (B_csr {bcsr / 1e9:.3f} GB; ELL kernels without row lengths stream M*K*12 B)", "",
              "| kernel (template arguments kept) | calls | avg us | min us | HBM read GB | HBM written GB | L2 hit | B_csr / time, % of 8 TB/s |", "|---|---|---|---|---|---|---|---|"]
    for k, d in sorted(tr.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
        if len(d) < 5: continue
        a = {n: sum(v) / len(v) for n, v in pm[k].items()}
        rd, wr = a.get("FETCH_SIZE", 0) * 2048, a.get("WRITE_SIZE", 0) * 1024
        hit = a.get("TCC_HIT_sum", 0) / max(a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0), 1)
        avg = sum(d) / len(d)
        lines.append(f"| {k} | {len(d)} | {avg / 1e3:.1f} | {min(d) / 1e3:.1f} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {hit:.2f} | {100 * bcsr / avg / 8000:.1f} |")
    lines.append("")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{RND}_summary.md", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
for p in latest(f"{O}/default/trace/*/*_kernel_stats.csv") or latest(f"{O}/c5/trace/*/*_kernel_stats.csv"):
    shutil.copy(p, f"profiles/{RND}_default_bench_kernel_stats.csv")
if os.path.exists(f"{O}/default.trace.json"):          # the line the default command printed under the profiler, for the cross-check
    shutil.copy(f"{O}/default.trace.json", f"profiles/{RND}_default_bench_under_trace.json")
if os.path.exists(f"{O}/bench_default.json"):
    shutil.copy(f"{O}/bench_default.json", f"profiles/{RND}_bench_default.json")
print("\n".join(l for l in lines if l.startswith("SpMV =") or l.startswith("## ")))
