#!/usr/bin/env python3
"""Harness stdout -> CSV.

Parses the log grammar printed by tests/harness/test_SpMV_HIP.elf -- which is the grammar of the
reference's harness (test/SpMV_test.cu:93-96,139-143,254-257) that its scripts/parseLog.py consumes:

    #<matrix path>
    SpMV_OMP_test.c  AVG_TIMES_ITERATION:<n>  sparse matrix: <M>x<N>-<NNZ>NNZ-<K>=MAX_ROW_NZ
    omp sched gather:  kind: OMP_SCHED_*  omp chunkSize: <c>  monotonic: Y|N
    @computing SpMV   with func: <CUDA|OMP> <CSR|ELL> <i> at:<ptr>
    cudaBlockSize: x y z  cudaGridSize: x y z    timeAvg:.. timeVar:..  timeInternalAvg:.. timeInternalVar:..
    threadNum: <t>  ompGridSize: <r>x<c>  timeAvg:.. timeVar:..  timeInternalAvg:.. timeInternalVar:..

One CSV row per timing line, with the reference tool's column names
(source,funcID,timeAvg,timeVar,internalTimeAvg,internalTimeVar,matRows,matCols,NNZ,maxRowNNZ,sampleSize,
ompSchedKind,ompChunkSize,ompMonotonic,threadNum,ompGrid,blockSize_x..gridSize_z) followed by the columns
the reference computes later in spreadsheets: GFLOPS = 2*NNZ/timeAvg, GBps (algorithmic CSR bytes
NNZ*12 + M*12 + N*8 over timeAvg) and rooflineFrac (GBps / 8000).  Lines starting with '#perf' / '#tight'
/ '#auto' (extra lines of this repo's harness) and ANSI colour codes are ignored.

usage: parse_harness_log.py <logfile|-> [--json]
"""
import csv
import json
import re
import sys

FIELDS = ("source,funcID,timeAvg,timeVar,internalTimeAvg,internalTimeVar,matRows,matCols,NNZ,maxRowNNZ,sampleSize,"
          "ompSchedKind,ompChunkSize,ompMonotonic,threadNum,ompGrid,"
          "blockSize_x,blockSize_y,blockSize_z,gridSize_x,gridSize_y,gridSize_z,GFLOPS,GBps,rooflineFrac").split(",")
ANSI = re.compile(r"\x1b\[[0-9;]*m")
FP = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"
RE_SIZE = re.compile(r"AVG_TIMES_ITERATION:\s*(\d+).*sparse matrix:\s*(\d+)x(\d+)-(\d+)NNZ-(-?\d+)=MAX_ROW_NZ")
RE_SCHED = re.compile(r"kind:\s*(OMP_\w+)\s+omp chunkSize:\s*(\d+)\s+monotonic:\s*(\w)")
RE_FUNC = re.compile(r"func:\s*(.*?)\s+at:")
RE_TIMES = re.compile(rf"timeAvg:\s*({FP})\s+timeVar:\s*({FP})\s+timeInternalAvg:\s*({FP})\s+timeInternalVar:\s*({FP})")
RE_OMP = re.compile(r"threadNum:\s*(\d+)\s+ompGridSize:\s*(\d+x\d+)")
RE_GPU = re.compile(r"cudaBlockSize:\s*(\d+)\s+(\d+)\s+(\d+)\s+cudaGridSize:\s*(\d+)\s+(\d+)\s+(\d+)")


def parse(lines):
    rows, ctx = [], {}
    for raw in lines:
        line = ANSI.sub("", raw).rstrip("\n")
        if line.startswith(("#perf", "#tight", "#auto")):
            continue
        if line.startswith("#"):
            ctx = {"source": line[1:].strip()}
            continue
        m = RE_SIZE.search(line)
        if m:
            ctx.update(sampleSize=int(m.group(1)), matRows=int(m.group(2)), matCols=int(m.group(3)), NNZ=int(m.group(4)),
                       maxRowNNZ=int(m.group(5)))
            continue
        m = RE_SCHED.search(line)
        if m:
            ctx.update(ompSchedKind=m.group(1), ompChunkSize=int(m.group(2)), ompMonotonic=m.group(3))
            continue
        m = RE_FUNC.search(line)
        if m:
            ctx["funcID"] = m.group(1)
            continue
        m = RE_TIMES.search(line)
        if not m or "funcID" not in ctx:
            continue
        row = {k: ctx.get(k) for k in FIELDS}
        row.update(timeAvg=float(m.group(1)), timeVar=float(m.group(2)), internalTimeAvg=float(m.group(3)),
                   internalTimeVar=float(m.group(4)))
        o, g = RE_OMP.search(line), RE_GPU.search(line)
        if o:
            row.update(threadNum=int(o.group(1)), ompGrid=o.group(2))
        if g:
            for name, v in zip(FIELDS[16:22], g.groups()):
                row[name] = int(v)
        if row["timeAvg"] > 0 and row.get("NNZ") is not None:
            byts = row["NNZ"] * 12 + row["matRows"] * 12 + row["matCols"] * 8
            row["GFLOPS"] = 2.0 * row["NNZ"] / row["timeAvg"] * 1e-9
            row["GBps"] = byts / row["timeAvg"] * 1e-9
            row["rooflineFrac"] = row["GBps"] / 8000.0
        rows.append(row)
    return rows


def main(argv):
    if len(argv) < 2:
        sys.exit(__doc__)
    src = sys.stdin if argv[1] == "-" else open(argv[1], errors="replace")
    rows = parse(src)
    if "--json" in argv:
        json.dump(rows, sys.stdout, indent=1)
    else:
        w = csv.DictWriter(sys.stdout, fieldnames=FIELDS)
        w.writeheader()
        w.writerows(rows)


if __name__ == "__main__":
    main(sys.argv)
