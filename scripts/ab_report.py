#!/usr/bin/env python3
"""print the two-phase kernels' rows of every rocprofv3 kernel_stats.csv below a directory"""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)):
    tag = f[len(sys.argv[1]):].strip("/").split("/")[0]
    tot = 0.0
    for r in csv.DictReader(open(f)):
        for k in ("pb_expand", "pb_reduce", "csr_stream2", "sell_spmv"):
            if k in r["Name"] and int(r["Calls"]) > 1:
                tot += float(r["AverageNs"]) / 1e3
                print(f"{tag:28s} {k:12s} calls {r['Calls']:>3s}  avg_us {float(r['AverageNs'])/1e3:9.1f}  min_us {float(r['MinNs'])/1e3:9.1f}")
    print(f"{tag:28s} sum avg_us {tot:9.1f}")
