#!/usr/bin/env python3
"""gpurun_out/r01c (written by scripts/final_profile.sh on the GPU box) -> profiles/r01_summary.md + profiles/traffic.json"""
import collections, csv, glob, json
O = 'gpurun_out/r01c'
def short(n): return n.replace("void ", "").replace("spmvhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
lines = ["# r01 profile summary (rocprofv3, one MI355X) -- `scripts/final_profile.sh` + `scripts/make_summary.py`", "",
         "Per workload: `rocprofv3 --kernel-trace --stats` durations and, from separate `--pmc` passes, HBM-side traffic",
         "per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md: a 128-B fabric read is",
         "tallied as 64 B).  Algorithmic bytes: nnz*12 + M*12 + N*8.", ""]
traffic = {"_doc": "HBM-side bytes per SpMV from rocprofv3 PMC passes (2 x FETCH_SIZE KiB + WRITE_SIZE KiB); bench.py copies the entry "
                   "matching workload+kernel into roofline.traffic; source profiles/r01_summary.md"}
wlname = {"c5": "c5-powerlaw-80M-1.6G", "c3": "c3-powerlaw-10M-200M", "c3_onepass": "c3-powerlaw-10M-200M",
          "c3n": "c3n-powerlaw-10M-200M-band512", "c2": "c2-uniform-1M-32"}
alg = {"c5": 20.8e9, "c3": 2.6e9, "c3_onepass": 2.6e9, "c3n": 2.6e9, "c2": 0.404e9}
KEEP = ("csr_stream", "pb_expand", "pb_reduce", "csr_scalar", "csr_vector", "sell_spmv", "sell_long")
for tag in ("c5", "c3", "c3_onepass", "c3n", "c2"):
    tr = collections.defaultdict(list)
    for p in glob.glob(f"{O}/{tag}/trace/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if any(x in k for x in KEEP): tr[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not tr: continue
    pm = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(f"{O}/{tag}/pmc_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k in tr: pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    maxcalls = max(len(v) for v in tr.values())      # auto mode runs every candidate a few times during warm-up
    lines += [f"## {tag}: {wlname[tag]}  (algorithmic {alg[tag] / 1e9:.2f} GB)", "",
              "| kernel | calls | avg us | min us | HBM read GB | HBM written GB | L2 hit | WAIT_ANY / WAIT_INST_ANY / ACTIVE of wave cycles |",
              "|---|---|---|---|---|---|---|---|"]
    tot = tott = 0
    used = []
    for k, d in sorted(tr.items(), key=lambda kv: -len(kv[1])):
        if len(d) < maxcalls * 0.6: continue
        a = {n: sum(v) / len(v) for n, v in pm[k].items()}
        rd, wr = a.get("FETCH_SIZE", 0) * 2048, a.get("WRITE_SIZE", 0) * 1024
        hit = a.get("TCC_HIT_sum", 0) / max(a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0), 1)
        wc = a.get("SQ_WAVE_CYCLES", 1)
        lines.append(f"| {k} | {len(d)} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {hit:.2f} | "
                     f"{a.get('SQ_WAIT_ANY', 0) / wc:.2f} / {a.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} / {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} |")
        tot += rd + wr; tott += sum(d) / len(d); used.append(k)
    lines += ["", f"SpMV = {tott / 1e3:.1f} us, HBM-side traffic {tot / 1e9:.2f} GB = {tot / alg[tag]:.2f} x algorithmic; algorithmic rate "
                  f"{alg[tag] / tott:.0f} GB/s = {100 * alg[tag] / tott / 8000:.1f} % of 8 TB/s", ""]
    launcher = ("hipSpMVTilesCSR" if any("pb_" in k for k in used) else "hipSpMVRowsSELL" if any("sell" in k for k in used)
                else "hipSpMVRowsCSR" if tag == "c2" else "hipSpMVWarpPerRowCSR")
    traffic.setdefault(wlname[tag], {})[launcher] = tot
old = json.load(open("profiles/traffic.json"))
for wl, d in old.items():                      # keep entries measured in earlier passes (e.g. the one-pass kernel on c5)
    if wl != "_doc":
        for k, v in d.items(): traffic.setdefault(wl, {}).setdefault(k, v)
open("profiles/r01_summary.md", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
print("\n".join(l for l in lines if l.startswith("SpMV =") or l.startswith("## ")))
