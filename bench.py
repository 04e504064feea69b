#!/usr/bin/env python3
"""bench.py -- SpMV GFLOP/s + achieved HBM GB/s on synthetic CSR/ELL matrices, 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c3|c2|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one y = A.x over the whole matrix: every rank runs the hot kernel (a C-ABI launcher, picked among
hipSpMVWarpPerRowCSR / hipSpMVTilesCSR / hipSpMVStripesCSR / hipSpMVRowsSELL by timing them during warm-up) on its
nnz-balanced contiguous row block and, for N > 1, the ranks exchange y over xGMI so that each ends with the full
vector (ready to be the next x).  The default workload is BASELINE.json configs[4] (power-law CSR, 80 M rows /
1.6 G nnz, max row 50 k), the configuration the metric's 1/2/4/8-GPU curve is quoted on; it is kept whole at every N
(strong scaling).  At N = 1 the same run also measures configs[2] (10 M / 200 M: "headline_c3", the configuration the
roofline target is quoted on), its banded twin, configs[1] (c2) and configs[3] (c4: the clipped matrix in ELL against
CSR, with and without the row-length early exit).

EVERY number in the line has a parity block from the same run: head / middle / tail / heaviest-row windows against
the serial oracle (the reference's 7e-4 gate, NaN = failure, plus the tight figure max|dy| / sum|a x|), and for the
two headline workloads every row of the CPU baseline's sample.  A failed check aborts the bench.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects: "roofline" (algorithmic bytes /
HIP-event kernel time vs 8 TB/s) and "cpu_baseline" (the reference's own spmvRowsBasicCSR from oracle/_ref, or the
oracle port, timed on this box's host cores).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12           # bytes/s, /opt/skills/guides/MI355X_MICROARCH.md
GATE = 7e-4                 # the reference's DOUBLE_DIFF_THREASH (src/include/config.h:113)
STRIPES_X_LIMIT = 256 << 20  # the stripes kernel re-reads x once per XCD and round of bins: tried only while x fits the Infinity Cache


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=25)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="c5")
    p.add_argument("--scale", type=float, default=1.0, help="shrink/grow the workload (rows and nnz) by this factor")
    p.add_argument("--launcher", default="auto",
                   help="a C-ABI launcher name, or 'auto': time the candidates during warm-up and keep the fastest")
    p.add_argument("--variant", type=int, default=-1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the secondary workloads measured at N=1")
    p.add_argument("--cpu-sample-nnz", type=int, default=100_000_000)
    p.add_argument("--y-hash", action="store_true",
                   help="add `y_sha256`, the hash of the bytes of the timed kernel's full y (rank 0's copy), to the line: two runs with a "
                        "serial-order launcher (--launcher hipSpMVRowsCSR) must agree on it whatever the number of ranks")
    p.add_argument("--only-structured", default="",
                   help="measure only the structured stand-ins whose name contains this string, or equals it when it ends in '$' (stencil3d, road, blocks, pattern; 'all' = the "
                        "three) and print that block alone -- for profiling the ELL / CSR kernels on the reference's kind of matrix")
    p.add_argument("--groups", type=int, default=0, help="pin the row groups per rank of the exchange at N > 1")
    p.add_argument("--pieces", type=int, default=0, help="pin the pieces of y per rank for --exchange push")
    p.add_argument("--exchange", default="auto", choices=["auto", "rccl", "push", "fused", "auto-no-rccl"],
                   help="how y reaches the other ranks at N > 1: RCCL all-gather, copy-engine pushes into peer windows, or "
                        "stores fused into phase 2 of the two-phase kernel (auto = measure the candidates, keep the fastest)")
    p.add_argument("--exchange-extra", action="store_true",
                   help="also try the forms beyond the eight default candidates (more groups / pieces, smaller or tapered "
                        "bins, the push kernel)")
    p.add_argument("--exchange-budget", type=float, default=60.0, help="seconds the exchange search may take (slowest rank)")
    p.add_argument("--rehearse-shared-gpu", action="store_true",
                   help="development only: all ranks use GPU 0 and gloo is the control plane (RCCL refuses two ranks on one "
                        "device), so the N > 1 code path -- plan, candidates, peer windows, checks -- can run on a one-GPU "
                        "box; the RCCL all-gather candidates are skipped and the timings mean nothing")
    p.add_argument("--force-dist", action="store_true",
                   help="run the multi-rank code path (process group, all-gather) even with --gpus 1; for rehearsal")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------- timing
def time_kernel_loop(torch, dist, world, step, steps, warmup, ev_pairs):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides."""
    for _ in range(warmup):
        step(None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(ev_pairs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    return t1 - t0


def make_events(api, n):
    evs = [(C.c_void_p(), C.c_void_p()) for _ in range(n)]
    for a, b in evs:
        api.lib.spmvHipEventCreate(C.byref(a))
        api.lib.spmvHipEventCreate(C.byref(b))
    return evs


def free_events(api, evs):
    for a, b in evs:
        api.lib.spmvHipEventDestroy(a)
        api.lib.spmvHipEventDestroy(b)


def kernel_ms(api, ev_pairs):
    out = []
    ms = C.c_float()
    for a, b in ev_pairs:
        api.lib.spmvHipEventElapsedMs(a, b, C.byref(ms))
        out.append(ms.value)
    return out


def avg_var(api, values):
    """mean and population variance exactly as the reference reports them (statsAvgVar, src/commons/utils.c:340-348)"""
    import numpy as np
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.zeros(2)
    api.hostlib.statsAvgVar(v.ctypes.data_as(C.c_void_p), v.size, out.ctypes.data_as(C.c_void_p))
    return float(out[0]), float(out[1])


KERNEL_OF = {"hipSpMVTilesCSR": "pb_expand_kernel + pb_reduce_kernel", "hipSpMVWarpPerRowCSR": "csr_stream2_kernel",
             "hipSpMVRowsCSR": "csr_stream2_kernel", "hipSpMVRowsSELL": "sell_spmv_kernel", "hipSpMVStripesCSR": "sb_spmv_kernel",
             "hipSpMVRowsELL": "ell_colmajor_thread", "hipSpMVWarpsPerRowELLNTrasposed": "ell_rowmajor_group"}
AUTO_CANDIDATES = ("hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL")
THREAD_PER_ROW = ("hipSpMVRowsCSR", "hipSpMVRowsSELL")      # both add a row's products in ascending j with one lane


def auto_candidates(n_cols):
    return tuple(c for c in AUTO_CANDIDATES if c != "hipSpMVStripesCSR" or n_cols * 8 <= STRIPES_X_LIMIT)


def pick_launcher(api, torch, dm, x_ptr, y_ptr, requested, n_cols, eligible=None):
    """'auto' -> run each candidate 3x (the first call of a format-building launcher also builds its format) and keep
    the fastest of the `eligible` ones (default: all); the others are timed for information.  Returns (name, {name: ms})."""
    if requested != "auto":
        return requested, {}
    cfg = api.CONFIG()
    times = {}
    for name in dict.fromkeys(tuple(eligible or ()) + auto_candidates(n_cols)):
        fn = api.SPMV_LAUNCHERS[name]
        if fn(C.byref(dm.handle), x_ptr, cfg, y_ptr):
            continue                                  # e.g. 32-bit positions exceeded for this shape: skip
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn(C.byref(dm.handle), x_ptr, cfg, y_ptr)
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / 3 * 1e3
    best = min((n for n in times if not eligible or n in eligible), key=times.get)
    return best, times


def lib_sha256(api):
    h = hashlib.sha256()
    with open(api.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def pmc_traffic(api, workload_name, launcher):
    """HBM-side bytes per launch measured by separate rocprofv3 --pmc passes (profiles/traffic.json, written by
    scripts/summarize_profile.py).  The file is stamped with the sha256 of the library it was measured on: a number
    that belongs to another build of the kernels is NOT reported (None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None, "profiles/traffic.json missing"
    if t.get("_libspmvhip_sha256") != lib_sha256(api):
        return None, "profiles/traffic.json was measured on another build of libspmvhip.so"
    return t.get(workload_name, {}).get(launcher), t.get("_measured", "")


def format_info(api, dm, launcher):
    """device bytes and one-time build time of the launcher's private copy of the matrix"""
    if launcher == "hipSpMVTilesCSR":
        i = api.tiles_info(dm)
        return {"extra_device_bytes": int(i.bytes), "format_build_ms": float(i.buildMs), "format_build_alloc_ms": float(getattr(i, "allocMs", 0.0)),
                "format_build_peak_temporary_bytes": int(i.tempBytes), "format_build_peak_temporary_bytes_per_nnz": float(i.tempBytes) / max(int(dm.nnz), 1),
                "bins": int(i.nBins), "rows_per_bin": int(i.rowsPerBin),
                "phase1_work_item_entries": int(i.chunk), "product_workspace_bytes": int(dm.nnz) * 8}
    if launcher == "hipSpMVStripesCSR":
        nb, rpb, wide, ms = C.c_uint(), C.c_uint(), C.c_int(), C.c_double()
        api.lib.spmvHipStripesShape(C.byref(dm.handle), C.byref(nb), C.byref(rpb), C.byref(wide), C.byref(ms))
        return {"extra_device_bytes": int(api.lib.spmvHipStripesBytes(C.byref(dm.handle))), "format_build_ms": float(ms.value),
                "bins": int(nb.value), "rows_per_bin": int(rpb.value), "wide_columns": bool(wide.value)}
    if launcher == "hipSpMVRowsSELL":
        return {"extra_device_bytes": int(api.lib.spmvHipSellBytes(C.byref(dm.handle)))}
    return {"extra_device_bytes": 0}


def tiles_phase_ms(api, torch, dm, x_ptr, y_ptr, reps=5):
    """the two kernels of the two-phase launcher timed separately (events on the launch stream)"""
    nb = api.tiles_info(dm).nBins
    e = [C.c_void_p() for _ in range(3)]
    for v in e:
        api.lib.spmvHipEventCreate(C.byref(v))
    p1, p2 = [], []
    ms = C.c_float()
    for _ in range(reps):
        api.lib.spmvHipEventRecord(e[0])
        api.lib.hipSpMVTilesExpand(C.byref(dm.handle), x_ptr)
        api.lib.spmvHipEventRecord(e[1])
        api.lib.hipSpMVTilesReduce(C.byref(dm.handle), 0, nb, y_ptr, 0, None)
        api.lib.spmvHipEventRecord(e[2])
        torch.cuda.synchronize()
        api.lib.spmvHipEventElapsedMs(e[0], e[1], C.byref(ms)); p1.append(ms.value)
        api.lib.spmvHipEventElapsedMs(e[1], e[2], C.byref(ms)); p2.append(ms.value)
    for v in e:
        api.lib.spmvHipEventDestroy(v)
    return {"pb_expand_kernel": sum(p1) / reps, "pb_reduce_kernel": sum(p2) / reps}


def measure_single(api, synth, torch, w, launcher, steps, warmup, candidates=None):
    """1-GPU measurement of workload `w`; returns (dict, context for checks)."""
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    info = synth.describe(w, lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x_host = synth.make_x(w.N, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    y = torch.full((w.N,), float("nan"), dtype=torch.float64, device="cuda")
    launcher, tried = pick_launcher(api, torch, dm, x.data_ptr(), y.data_ptr(), launcher, w.N, candidates)
    y.fill_(float("nan"))
    res, kms = time_launcher(api, torch, dm, launcher, x, y, steps, warmup)
    nnz = int(irp[-1])
    bytes_alg = synth.algorithmic_bytes_csr(nnz, w.N, w.N)
    k_avg, k_var = avg_var(api, kms)
    res.update({
        "workload": info, "launcher": launcher, "auto_candidates_ms": tried,
        "eligible_launchers": list(candidates) if candidates else "all",
        "fastest_of_all_launchers": min(tried, key=tried.get) if tried else launcher,
        "kernel_ms_avg": k_avg, "kernel_ms_var": k_var, "kernel_ms_min": min(kms),
        "gflops": 2.0 * nnz / (res["ms_per_step"] * 1e-3) * 1e-9,
        "hbm_gbps": bytes_alg / (k_avg * 1e-3) * 1e-9, "hbm_frac": bytes_alg / (k_avg * 1e-3) / HBM_PEAK,
        "algorithmic_bytes": bytes_alg,
    })
    res.update(format_info(api, dm, launcher))
    if res.get("format_build_ms"):
        res["spmvs_to_amortise_format_build"] = res["format_build_ms"] / max(k_avg, 1e-9)
    if launcher == "hipSpMVRowsCSR" and _ROWS_VARIANT[0] == 2:
        nm = api.lib.spmvHipAutoChoiceRows(C.byref(dm.handle), None)
        res["launcher_resolved_to"] = nm.decode() if nm else None
    if launcher == "hipSpMVTilesCSR":
        res["kernel_ms_phases"] = tiles_phase_ms(api, torch, dm, x.data_ptr(), y.data_ptr())
    return res, dict(dm=dm, irp=irp, x_host=x_host, y=y, x=x, lens=lens)


def time_launcher(api, torch, dm, launcher, x, y, steps, warmup):
    fn = api.SPMV_LAUNCHERS[launcher]
    cfg = api.CONFIG()
    evs = make_events(api, steps)

    def step(ev):
        if ev:
            api.lib.spmvHipEventRecord(ev[0])
        rc = fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr())
        if ev:
            api.lib.spmvHipEventRecord(ev[1])
        if rc:
            raise RuntimeError(launcher + " failed")

    wall = time_kernel_loop(torch, None, 1, step, steps, warmup, evs)
    kms = kernel_ms(api, evs)
    free_events(api, evs)
    return {"ms_per_step": wall / steps * 1e3}, kms


# ------------------------------------------------------------------------------------------------- checker
class OracleWindows:
    """Serial-oracle y (sgemvSerial restated, oracle/spmv_oracle.c) and the per-row scale sum|a x| on a few row windows of
    a synthetic workload -- head, middle, tail and around the longest row -- whose entries are regenerated by the CPU twin
    of the device generator.  Checker use of oracle/ (allowed for bench.py)."""

    def __init__(self, synth, w, irp, x_host, lens=None, size=200_000, windows=None):
        import numpy as np
        self.np = np
        ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        vp = C.c_void_p
        ora.synthFillCsrRef.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
        ora.oracleCsrSerial64_32.argtypes = [C.c_ulong, vp, vp, vp, vp, vp]
        if windows is None:
            windows = [(0, min(w.N, size)), (w.N // 2, min(w.N, w.N // 2 + size)), (max(0, w.N - size), w.N)]
            if lens is not None and w.N:
                h = int(np.argmax(lens))
                windows.append((max(0, h - 64), min(w.N, h + 64)))
        self.windows = []
        for r0, r1 in windows:
            irp_r = np.ascontiguousarray(irp[r0:r1 + 1], dtype=np.uint64)
            nz = int(irp_r[-1] - irp_r[0])
            jr = np.empty(nz, dtype=np.uint32)
            ar = np.empty(nz, dtype=np.float64)
            ora.synthFillCsrRef(r1 - r0, w.N, r0, irp_r.ctypes.data_as(vp), jr.ctypes.data_as(vp), ar.ctypes.data_as(vp),
                                synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
            yr = np.empty(r1 - r0)
            irp_l = irp_r - irp_r[0]
            ora.oracleCsrSerial64_32(r1 - r0, irp_l.ctypes.data_as(vp), jr.ctypes.data_as(vp), ar.ctypes.data_as(vp),
                                     x_host.ctypes.data_as(vp), yr.ctypes.data_as(vp))
            prod = np.abs(ar * x_host[jr])
            scale = np.add.reduceat(np.concatenate([prod, [0.0]]), np.minimum(irp_l[:-1].astype(np.int64), prod.size))
            scale = np.where(np.diff(irp_l.astype(np.int64)) > 0, scale, 0.0)
            self.windows.append((int(r0), int(r1), yr, scale))

    def check(self, y_gpu_fn, bitwise=False):
        """-> parity dict.  gate: |dy| <= 7e-4 per element and no NaN (the reference's doubleVectorsDiff made NaN-aware);
        reported beside it: max|dy| / sum|a x| over rows with entries (empty rows must be exactly 0)."""
        np = self.np
        spot, ok = [], True
        for r0, r1, yr, scale in self.windows:
            yg = y_gpu_fn(r0, r1)
            nan = bool(np.isnan(yg).any())
            d = np.abs(yr - yg)
            nz = scale > 0
            tight = float(np.max(d[nz] / scale[nz])) if nz.any() else 0.0
            zeros_ok = bool(np.all(d[~nz] == 0)) if (~nz).any() else True
            c = {"rows": [r0, r1], "max_abs_diff": float(np.nanmax(d)) if d.size else 0.0, "max_diff_over_sum_abs_ax": tight, "nan": nan}
            if bitwise:
                c["bit_identical"] = bool(np.array_equal(yr, yg))
                ok = ok and c["bit_identical"]
            ok = ok and not nan and c["max_abs_diff"] <= GATE and zeros_ok
            spot.append(c)
        return {"ok": bool(ok), "gate_abs": GATE, "max_abs_diff": max(c["max_abs_diff"] for c in spot),
                "max_diff_over_sum_abs_ax": max(c["max_diff_over_sum_abs_ax"] for c in spot), "spot_checks": spot}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_CPUS = {}


def effective_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (a container on a 128-thread
    host is often entitled to far fewer: a parallel region with one thread per visible CPU then shares a few cores).
    Taken ONCE, at start-up: once libgomp has bound the calling thread to its place the mask shows that place only."""
    if _CPUS:
        return dict(_CPUS)
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                 # cgroup v2: "<quota|max> <period>"
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    eff = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    _CPUS.update({"affinity_cpus": aff, "cgroup_cpu_quota": quota, "effective_cpus": eff})
    return dict(_CPUS)


def cpu_sanity(ora, irp64, as_, rows):
    """what the OpenMP runtime really gives a parallel region here, and a gather-free streaming figure on the same arrays
    (row sums of AS under the same static partition: 8 B per entry + 16 B per row)"""
    import numpy as np
    vp = C.c_void_p
    ora.oracleRowSumSeconds.restype = C.c_double
    ora.oracleRowSumSeconds.argtypes = [C.c_ulong, vp, vp, vp]
    y = np.empty(rows)
    nnz = int(irp64[rows])
    ts = [ora.oracleRowSumSeconds(rows, irp64.ctypes.data_as(vp), as_.ctypes.data_as(vp), y.ctypes.data_as(vp)) for _ in range(4)]
    t = min(ts[1:])
    out = dict(effective_cpus())
    out.update({"threads_in_region": int(ora.oracleThreadsInRegion()), "distinct_cpus_in_region": int(ora.oracleDistinctCpusInRegion()),
                "stream_like_gbps": (nnz * 8 + rows * 16) / t * 1e-9,
                "stream_like_note": "row sums of AS, no gather: 8 B per entry + 16 B per row, best of 3 passes, same static partition"})
    return out


def cpu_baseline(synth, w, irp, x_host, rows, iters=5):
    """The reference's own spmvRowsBasicCSR (oracle/_ref/libspmvref.so, src/SpMV_CSR_OMP.c:34-63) -- or the oracle port --
    on rows [0, rows) of the workload with all host cores, the way SURVEY 8d asks: OMP_PROC_BIND=close, OMP_PLACES=cores
    (set before libgomp starts, see main()), the matrix in the reference's host layout (64-bit indices), static schedule;
    beside it the dynamic schedule with the reference's fair-folded chunk rewrite, the build with SIMD_ROWS_REDUCTION off,
    and the same static run with the pages of the matrix first touched by the threads that read them (the reference
    fills its matrix from one thread, src/lib/parser.c:318-326: that is `value`).  Returns (dict, y_cpu)."""
    import numpy as np
    from spmv_openmp_cuda_amd.ctypes_defs import ref_CONFIG, ref_spmat
    rows = max(1, min(int(rows), w.N))
    irp_s = np.ascontiguousarray(irp[:rows + 1], dtype=np.uint64)
    nnz_s = int(irp_s[-1])
    ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    vp = C.c_void_p
    ora.synthFillCsrRef.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    ja32 = np.empty(nnz_s, dtype=np.uint32)
    as_gen = np.empty(nnz_s, dtype=np.float64)
    ora.synthFillCsrRef(rows, w.N, 0, irp_s.ctypes.data_as(vp), ja32.ctypes.data_as(vp), as_gen.ctypes.data_as(vp),
                        synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
    # serial first touch, as the reference's loader: fresh arrays written by this one thread
    ja64 = ja32.astype(np.uint64)
    as_ = as_gen.copy()
    del as_gen
    y_cpu = np.full(rows, np.nan)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libspmvref.so")
    out = {"unit": "GFLOP/s", "cpu_model": cpu_model(),
           "omp_env": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS", "OMP_SCHEDULE")},
           "omp_places": int(ora.oracleOmpPlaces()), "omp_proc_bind": int(ora.oracleOmpProcBind())}

    def gflops(ts):
        return 2.0 * nnz_s / (sum(ts) / len(ts)) * 1e-9

    if os.path.exists(ref_path):
        kind = "reference"
        ref = C.CDLL(ref_path)
        for f in (ref.refChunksNOOP, ref.refChunksFairFolded):
            f.restype = vp
        ref.spmvRowsBasicCSR.argtypes = [C.POINTER(ref_spmat), vp, C.POINTER(ref_CONFIG), vp]
        rl = np.diff(irp_s).astype(np.uint64)

        def spmat_of(irp_a, ja_a, as_a):
            m = ref_spmat()
            m.M, m.N, m.NZ = rows, w.N, nnz_s
            m.IRP = irp_a.ctypes.data_as(C.POINTER(C.c_ulong))
            m.JA = ja_a.ctypes.data_as(C.POINTER(C.c_ulong))
            m.AS = as_a.ctypes.data_as(C.POINTER(C.c_double))
            m.RL = rl.ctypes.data_as(C.POINTER(C.c_ulong))
            return m

        def passes(lib, m, chunk_fn, y):
            cfg = ref_CONFIG()
            cfg.gridRows = cfg.gridCols = 8
            cfg.threadNum = lib.refMaxThreads()
            cfg.chunkDistrbFunc = chunk_fn
            ts = []
            for _ in range(iters + 1):
                t0 = time.perf_counter()
                rc = lib.spmvRowsBasicCSR(C.byref(m), x_host.ctypes.data_as(vp), C.byref(cfg), y.ctypes.data_as(vp))
                ts.append(time.perf_counter() - t0)
                assert rc == 0
            return ts[1:]                                           # first pass = page-in

        cores = int(ref.refMaxThreads())
        m = spmat_of(irp_s, ja64, as_)
        ref.refSetSchedule(1, 0)                                    # omp_sched_static, like OMP_SCHEDULE=static
        t_static = passes(ref, m, ref.refChunksNOOP(), y_cpu)
        ref.refSetSchedule(2, 1)                                    # omp_sched_dynamic, chunk 1 -> rewritten by chunksFairFolded
        out["omp_dynamic_fair_folded_gflops"] = gflops(passes(ref, m, ref.refChunksFairFolded(), np.empty(rows)))
        ref.refSetSchedule(1, 0)
        nosimd_path = os.path.join(ROOT, "oracle", "_ref", "libspmvref_nosimd.so")
        if os.path.exists(nosimd_path):
            ref2 = C.CDLL(nosimd_path)
            ref2.refChunksNOOP.restype = vp
            ref2.spmvRowsBasicCSR.argtypes = [C.POINTER(ref_spmat), vp, C.POINTER(ref_CONFIG), vp]
            ref2.refSetSchedule(1, 0)
            out["omp_static_no_simd_reduction_gflops"] = gflops(passes(ref2, m, ref2.refChunksNOOP(), np.empty(rows)))
        # parallel first touch: fresh arrays, every row's entries copied by the thread of the static partition that reads them
        irp_p, ja_p, as_p, y_p = np.empty(rows + 1, dtype=np.uint64), np.empty(nnz_s, dtype=np.uint64), np.empty(nnz_s), np.empty(rows)
        ora.oracleFirstTouchCsr.argtypes = [C.c_ulong, vp, vp, vp, vp, vp, vp, vp]
        ora.oracleFirstTouchCsr(rows, irp_s.ctypes.data_as(vp), ja64.ctypes.data_as(vp), as_.ctypes.data_as(vp),
                                irp_p.ctypes.data_as(vp), ja_p.ctypes.data_as(vp), as_p.ctypes.data_as(vp), y_p.ctypes.data_as(vp))
        out["omp_static_parallel_first_touch_gflops"] = gflops(passes(ref, spmat_of(irp_p, ja_p, as_p), ref.refChunksNOOP(), y_p))
        del irp_p, ja_p, as_p
    else:
        kind = "port"
        ora.oracleSetSchedule(1, 0)
        cores = int(ora.oracleMaxThreads())
        ora.oracleCsrOmp32.argtypes = [C.c_ulong, vp, vp, vp, vp, vp]
        irp32 = irp_s.astype(np.uint32)
        t_static = []
        for _ in range(iters + 1):
            t0 = time.perf_counter()
            ora.oracleCsrOmp32(rows, irp32.ctypes.data_as(vp), ja32.ctypes.data_as(vp), as_.ctypes.data_as(vp),
                               x_host.ctypes.data_as(vp), y_cpu.ctypes.data_as(vp))
            t_static.append(time.perf_counter() - t0)
        t_static = t_static[1:]
    t = sum(t_static) / len(t_static)
    avg, var = (lambda v: (float(np.mean(v)), float(np.var(v))))(np.array(t_static))
    out.update(cpu_sanity(ora, irp_s, as_, rows))
    out["value_is_sample"] = bool(rows != w.N)
    # how the figure is to be read: a CSR SpMV with uniformly spread columns pays one cache-missing 64-B line of x per
    # entry on top of the 16 B/entry stream, so it is latency-bound (threads x misses in flight / memory latency), far
    # below the streaming figure beside it
    out["gather_bound_note"] = (f"{2.0 * nnz_s / t * 1e-9 / 2:.2f} G entries/s on {out['threads_in_region']} threads = "
                                f"{nnz_s / t / max(out['threads_in_region'], 1) * 1e-6:.1f} M x-gathers/s per thread; the gather-free "
                                f"stream over the same arrays runs at {out['stream_like_gbps']:.0f} GB/s")
    out.update({"value": 2.0 * nnz_s / t * 1e-9, "cores": cores, "kind": kind,
                "sample": f"rows [0,{rows}) of {w.name} = {nnz_s} nnz ({'the whole matrix' if rows == w.N else 'a head sample'}), x full "
                          f"length, {len(t_static)} timed passes of spmvRowsBasicCSR after one page-in pass, OMP schedule static, matrix "
                          f"first touched by one thread as the reference's loader does, {t * 1e3:.2f} ms/pass",
                "pass_seconds_avg": avg, "pass_seconds_var": var,
                "hbm_like_gbps": synth.algorithmic_bytes_csr(nnz_s, rows, w.N) / t * 1e-9})
    return out, y_cpu


def cpu_baseline_ell_c4(synth, w, irp, x_host, rows, K):
    """The reference's spmvRowsBasicELL (src/SpMV_ELL_OMP.c:33-67; oracle/_ref, or the oracle port) beside the GPU's ELL
    kernels of config 4: rows [0, rows) of the clipped matrix as the loader would hold them (row-major, 64-bit indices,
    {0, 0.0} padding, row lengths), static schedule, and spmvRowsBasicCSR on the same rows for comparison."""
    import numpy as np
    from spmv_openmp_cuda_amd.ctypes_defs import ref_CONFIG, ref_spmat
    vp = C.c_void_p
    ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    ora.synthFillCsrRef.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    ora.oracleCsrSerial.argtypes = [C.c_ulong, vp, vp, vp, vp, vp]
    irp_s = np.ascontiguousarray(irp[:rows + 1], dtype=np.uint64)
    nnz_s = int(irp_s[-1])
    ja32, as_ = np.empty(nnz_s, dtype=np.uint32), np.empty(nnz_s)
    ora.synthFillCsrRef(rows, w.N, 0, irp_s.ctypes.data_as(vp), ja32.ctypes.data_as(vp), as_.ctypes.data_as(vp),
                        synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
    ja64 = ja32.astype(np.uint64)
    lens = np.diff(irp_s.astype(np.int64))
    eja, eas = np.zeros((rows, K), dtype=np.uint64), np.zeros((rows, K))
    r_of = np.repeat(np.arange(rows), lens)
    pos = np.arange(nnz_s) - np.repeat(irp_s[:-1].astype(np.int64), lens)
    eja[r_of, pos] = ja64
    eas[r_of, pos] = as_
    del r_of, pos
    rl = lens.astype(np.uint64)
    y_ref = np.empty(rows)
    ora.oracleCsrSerial(rows, irp_s.ctypes.data_as(vp), ja64.ctypes.data_as(vp), as_.ctypes.data_as(vp), x_host.ctypes.data_as(vp), y_ref.ctypes.data_as(vp))
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libspmvref.so")
    kind = "reference" if os.path.exists(ref_path) else "port"
    lib = C.CDLL(ref_path) if kind == "reference" else ora
    (lib.refSetSchedule if kind == "reference" else ora.oracleSetSchedule)(1, 0)
    cfg = ref_CONFIG()
    cfg.gridRows = cfg.gridCols = 8
    cfg.threadNum = int((lib.refMaxThreads if kind == "reference" else ora.oracleMaxThreads)())
    if kind == "reference":
        lib.refChunksNOOP.restype = vp
        cfg.chunkDistrbFunc = lib.refChunksNOOP()

    def mat(ell):
        m = ref_spmat()                              # (this repo's spmat starts with the same fields: the port reads it alike)
        m.M, m.N, m.NZ, m.MAX_ROW_NZ = rows, w.N, nnz_s, K if ell else 0
        m.JA = (eja if ell else ja64).ctypes.data_as(C.POINTER(C.c_ulong))
        m.AS = (eas if ell else as_).ctypes.data_as(C.POINTER(C.c_double))
        m.RL = rl.ctypes.data_as(C.POINTER(C.c_ulong))
        if not ell:
            m.IRP = irp_s.ctypes.data_as(C.POINTER(C.c_ulong))
        return m
    out = {"kind": kind, "cores": int(cfg.threadNum), "unit": "GFLOP/s", "schedule": "static",
           "sample": f"rows [0,{rows}) of {w.name}: {nnz_s} nnz in {rows} x {K} slots, 3 timed passes after one page-in pass", **effective_cpus()}
    if kind == "port":                               # the port's functions take this repo's (longer) struct: same prefix, pad the tail
        from spmv_openmp_cuda_amd.ctypes_defs import spmat as own_spmat

        def widen(m):
            o = own_spmat()
            for f, _ in ref_spmat._fields_:
                setattr(o, f, getattr(m, f))
            return o
    for label, fn, ell in (("spmvRowsBasicELL (row lengths)", lib.spmvRowsBasicELL, True), ("spmvRowsBasicCSR", lib.spmvRowsBasicCSR, False)):
        fn.argtypes = [vp, vp, vp, vp]
        m = mat(ell)
        if kind == "port":
            m = widen(m)
        y = np.empty(rows)
        ts = []
        for _ in range(4):
            y[:] = np.nan
            t0 = time.perf_counter()
            rc = fn(C.byref(m), x_host.ctypes.data_as(vp), C.byref(cfg), y.ctypes.data_as(vp))
            ts.append(time.perf_counter() - t0)
            if rc or not np.all(np.abs(y - y_ref) <= GATE):
                raise SystemExit(f"CPU baseline {label} failed the gate on {w.name}")
        out[label] = {"gflops": 2.0 * nnz_s / (sum(ts[1:]) / 3) * 1e-9, "pass_seconds_avg": sum(ts[1:]) / 3}
    log(f"c4 CPU ({kind}, {cfg.threadNum} threads, {rows} rows): ELL {out['spmvRowsBasicELL (row lengths)']['gflops']:.2f}  CSR {out['spmvRowsBasicCSR']['gflops']:.2f} GFLOP/s")
    return out


def full_rows_parity(y_cpu, y_gpu_head):
    import numpy as np
    d = np.abs(y_cpu - y_gpu_head)
    nan = bool(np.isnan(y_gpu_head).any())
    return {"rows_checked": int(y_cpu.size), "max_abs_diff": float(np.nanmax(d)), "nan_in_gpu_y": nan,
            "ok": bool(not nan and np.nanmax(d) <= GATE)}


_ROWS_VARIANT = [1]
_Y_HASH = [False]


def set_rows_variant(api, v):
    """hipSpMVRowsCSR: 1 = the LDS-stream kernel by name, 2 = the library default (fastest serial-order kernel)"""
    api.set_variant("hipSpMVRowsCSR", v)
    _ROWS_VARIANT[0] = v


def measure_block(api, synth, torch, w, launcher, steps, warmup, candidates=None, cpu_rows=0):
    """measure + check one workload; returns the block for the JSON line (with "parity", and "cpu_baseline" when
    cpu_rows > 0).  A failed check aborts the bench."""
    res, ctx = measure_single(api, synth, torch, w, launcher, steps, warmup, candidates)
    y_t = ctx["y"]
    win = OracleWindows(synth, w, ctx["irp"], ctx["x_host"], ctx["lens"])
    parity = win.check(lambda a, b: y_t[a:b].cpu().numpy())
    if _Y_HASH[0]:
        res["y_sha256"] = hashlib.sha256(y_t.cpu().numpy().tobytes()).hexdigest()
    if cpu_rows:
        import numpy as np
        rows = w.N if cpu_rows >= int(ctx["irp"][-1]) else int(np.searchsorted(ctx["irp"], cpu_rows, side="right") - 1)
        base, y_cpu = cpu_baseline(synth, w, ctx["irp"], ctx["x_host"], rows)
        head = full_rows_parity(y_cpu, y_t[:y_cpu.size].cpu().numpy())
        parity["cpu_baseline_rows"] = head
        parity["ok"] = bool(parity["ok"] and head["ok"])
        res["cpu_baseline"] = base
    res["parity"] = parity
    log(f"{w.name} [{res['launcher']}]: {res['gflops']:.1f} GFLOP/s  kernel {res['kernel_ms_avg']:.3f} ms  "
        f"{res['hbm_gbps']:.0f} GB/s = {100 * res['hbm_frac']:.1f}% of 8 TB/s   parity ok={parity['ok']} "
        f"max|dy|={parity['max_abs_diff']:.2e} max|dy|/sum|ax|={parity['max_diff_over_sum_abs_ax']:.2e}")
    if not parity["ok"]:
        raise SystemExit(f"PARITY FAILURE on {w.name} [{res['launcher']}]: {parity}")
    if res["auto_candidates_ms"] and hasattr(api.lib, "hipSpMVAutoCSR"):
        # what a caller of the C-ABI's own selector gets for this matrix (hipSpMVAutoCSR measures inside the library at
        # its first call; SELL is not among its candidates) -- after the timed launcher's y has been checked, checked itself
        ms3 = (C.c_double * 4)()
        y_t.fill_(float("nan"))
        if api.lib.hipSpMVAutoCSR(C.byref(ctx["dm"].handle), ctx["x"].data_ptr(), api.CONFIG(), y_t.data_ptr()) == 0:
            torch.cuda.synchronize()
            name = api.lib.spmvHipAutoChoice(C.byref(ctx["dm"].handle), ms3)
            par2 = win.check(lambda a, b: y_t[a:b].cpu().numpy())
            res["library_auto_choice"] = {"launcher": name.decode() if name else None, "parity_ok": bool(par2["ok"]),
                                          "ms": dict(zip(("hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR"), [float(v) for v in ms3]))}
            if not par2["ok"]:
                raise SystemExit(f"PARITY FAILURE on {w.name} [hipSpMVAutoCSR -> {name}]: {par2}")
    if res["auto_candidates_ms"] and hasattr(api.lib, "spmvHipAutoChoiceRows"):
        # ... and what the serial-order default of hipSpMVRowsCSR (variant 2: LDS-stream kernel / deterministic two-phase /
        # deterministic stripes) resolves to and costs on this matrix: every candidate must give the serial oracle's BITS
        keep = _ROWS_VARIANT[0]
        set_rows_variant(api, 2)
        try:
            y_t.fill_(float("nan"))
            if api.lib.hipSpMVRowsCSR(C.byref(ctx["dm"].handle), ctx["x"].data_ptr(), api.CONFIG(), y_t.data_ptr()) == 0:
                torch.cuda.synchronize()
                ms3 = (C.c_double * 4)()
                name = api.lib.spmvHipAutoChoiceRows(C.byref(ctx["dm"].handle), ms3)
                par3 = win.check(lambda a, b: y_t[a:b].cpu().numpy(), bitwise=True)
                res["library_serial_order_choice"] = {
                    "launcher": name.decode() if name else None, "bit_identical_to_serial_oracle": bool(par3["ok"]),
                    "ms": dict(zip(("hipSpMVRowsCSR(LDS-stream)", "hipSpMVTilesCSR(deterministic)", "hipSpMVStripesCSR(owner wavefronts)",
                                    "hipSpMVStripesCSR(ordered tickets)"), [float(v) for v in ms3]))}
                if not par3["ok"]:
                    raise SystemExit(f"PARITY FAILURE on {w.name} [hipSpMVRowsCSR variant 2 -> {name}]: {par3}")
        finally:
            set_rows_variant(api, keep)
    ctx["dm"].free()
    del ctx
    torch.cuda.empty_cache()
    return res


def measure_c4(api, synth, torch, args, steps, warmup):
    """BASELINE config 4: the power-law matrix clipped to 64 slots (pure ELL of the unclipped one is 6 TB and is refused,
    as by the reference's loader) in ELL transposed+pitched thread-per-row and ELL row-major lanes-per-row, each with
    and without the row-length early exit, against the CSR launchers on the SAME clipped matrix.  Bytes: SURVEY 8d's
    B_ell = M.K.12 + M.8 + N.8 for the all-slots runs, B_ell_rl = B_csr with row lengths."""
    w = synth.WORKLOADS["c4"]
    if args.scale != 1.0:
        w = synth.scaled(w, args.scale)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    info = synth.describe(w, lens)
    nnz, M, K = int(irp[-1]), w.N, int(lens.max())
    dm = synth.device_csr(w, irp, 0, M)
    x_host = synth.make_x(M, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    y = torch.full((M,), float("nan"), dtype=torch.float64, device="cuda")
    win = OracleWindows(synth, w, irp, x_host, lens)
    b_csr = synth.algorithmic_bytes_csr(nnz, M, M)
    b_ell = M * K * 12 + M * 8 + M * 8
    out = {"workload": info, "slots_K": K, "padding_ratio_MK_over_nnz": M * K / nnz,
           "algorithmic_bytes": {"B_csr = B_ell_rl": b_csr, "B_ell (all slots)": b_ell}, "runs": []}

    def run(label, launcher, mat, rl, nbytes, bitwise=False):
        if rl is not None:
            api.lib.spmvHipSetEllRowLens(1 if rl else 0)
        y.fill_(float("nan"))
        res, kms = time_launcher(api, torch, mat, launcher, x, y, steps, warmup)
        k_avg, k_var = avg_var(api, kms)
        par = win.check(lambda a, b: y[a:b].cpu().numpy(), bitwise=bitwise)
        if bool(torch.isnan(y).any()):
            par["ok"] = False
            par["nan_left_in_y"] = True
        r = {"format": label, "launcher": launcher, "row_lens_early_exit": rl, "kernel_ms_avg": k_avg, "kernel_ms_var": k_var,
             "gflops": 2.0 * nnz / (res["ms_per_step"] * 1e-3) * 1e-9, "bytes_convention": "B_csr" if nbytes == b_csr else "B_ell",
             "hbm_gbps": nbytes / (k_avg * 1e-3) * 1e-9, "hbm_frac": nbytes / (k_avg * 1e-3) / HBM_PEAK, "parity": par}
        out["runs"].append(r)
        log(f"c4 {label} rl={rl}: kernel {k_avg:.3f} ms  {r['gflops']:.0f} GFLOP/s  {100 * r['hbm_frac']:.1f}% ({r['bytes_convention']})  parity ok={par['ok']}")
        if not par["ok"]:
            raise SystemExit(f"PARITY FAILURE on c4 [{label}, row lens {rl}]: {par}")

    best, tried = pick_launcher(api, torch, dm, x.data_ptr(), y.data_ptr(), args.launcher if args.launcher in AUTO_CANDIDATES else "auto", M)
    out["csr_auto_candidates_ms"] = tried
    run("CSR " + best, best, dm, None, b_csr)
    run("CSR hipSpMVRowsCSR (serial order)", "hipSpMVRowsCSR", dm, None, b_csr, bitwise=True)
    ell_t = api.csr_to_ell_device(dm, True)
    for rl in (True, False):       # thread per row, ascending slots: bit-identical to the serial oracle (padding adds +0.0 * x[0])
        run("ELL transposed+pitched, thread per row", "hipSpMVRowsELL", ell_t, rl, b_csr if rl else b_ell, bitwise=True)
    ell_t.free()
    ell = api.csr_to_ell_device(dm, False)
    for rl in (True, False):
        run("ELL row-major, lanes per row (wavefront-per-row family)", "hipSpMVWarpsPerRowELLNTrasposed", ell, rl, b_csr if rl else b_ell)
    ell.free()
    api.lib.spmvHipSetEllRowLens(1)
    if not args.no_cpu_baseline:
        out["cpu_baseline_ell"] = cpu_baseline_ell_c4(synth, w, irp, x_host, min(M, 2_000_000), K)
    # the size guard on the UNCLIPPED matrix (10 M rows x 50 k slots): refused through the C-ABI before any allocation
    w3 = synth.WORKLOADS["c3"] if args.scale == 1.0 else synth.scaled(synth.WORKLOADS["c3"], args.scale)
    irp3 = synth.prefix(synth.row_lengths(w3))
    dm3 = synth.device_csr(w3, irp3, 0, w3.N)
    bad = api.DeviceMatrix()
    rc = api.lib.spmvHipCsrToEll(C.byref(dm3.handle), 1, C.byref(bad.handle))
    out["unclipped_ell_refused"] = bool(rc != 0) if args.scale == 1.0 else None
    dm3.free()
    dm.free()
    if args.scale == 1.0 and rc == 0:
        raise SystemExit("the ELL size guard accepted the unclipped power-law matrix")
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------- structured matrices
# The matrices the reference's report publishes numbers for are structured SuiteSparse matrices (matrix list
# doc/relazione.tex:463); the files cannot be fetched here, so stand-ins with the same SHAPE are generated
# (csrc/host/structured.c), written as MatrixMarket files and taken through the loader (MMtoCSR / MMtoELL), the uploads
# (spMatCpyCSR / ellTranspose + spMatCpyELL) and every launcher -- the path a user of the reference's CLI walks.
# `published`: what the reference's tables show for the matrix the stand-in is shaped after (unstated GPU, probably a
# Quadro RTX 5000 with 448 GB/s; "CSR 1" / "ELL 2" columns are invalid there -- rows 0..31 only, BASELINE.md caveat 1).
STRUCTURED = (
    {"name": "stencil3d-500x100x100", "kind": 0, "p": (500, 100, 100), "after": "channel-500x100x100-b050 (4.8 M rows, 85.4 M nnz, max row 18)",
     "published": {"CUDA ELL 0 (transposed, thread per row)": "3.86e-3 s = 44.25 GFLOP/s", "CUDA ELL 1 (row-major, thread per row)": "9.04e-3 s = 18.9 GFLOP/s",
                   "OMP ELL 0 spmvRowsBasicELL": "2.81e-2 s = 6.1 GFLOP/s"}},
    {"name": "road-12M", "kind": 1, "p": (12_000_000, 0, 0), "after": "asia_osm (11.95 M rows, 25.4 M nnz, max row 9)",
     "published": {"CUDA CSR 0 (thread per row)": "2.13e-3 s = 23.87 GFLOP/s", "OMP CSR 1 spmvRowsBasicCSR": "3.09e-2 s = 1.6 GFLOP/s"}},
    # the two graphs above are distributed as MatrixMarket PATTERN files (DIMACS10 collection): the loader gives every entry
    # 1.0 (parser.c:59-61).  The same shapes written as pattern files: the upload recognises "all values equal" and the CSR
    # kernels stream no values (include/spmvHip.h, spmvHipSetUnitValues)
    {"name": "stencil3d-500x100x100-pattern", "kind": 0, "pattern": True, "p": (500, 100, 100), "after": "channel-500x100x100-b050 as distributed: a pattern file",
     "published": {"CUDA ELL 0 (transposed, thread per row)": "3.86e-3 s = 44.25 GFLOP/s"}},
    {"name": "road-12M-pattern", "kind": 1, "pattern": True, "p": (12_000_000, 0, 0), "after": "asia_osm as distributed: a pattern file",
     "published": {"CUDA CSR 0 (thread per row)": "2.13e-3 s = 23.87 GFLOP/s"}},
    {"name": "blocks-36k", "kind": 2, "p": (36_417, 24, 0), "after": "pdb1HYS (36 417 rows, 4.34 M nnz, max row 204)",
     "published": {"CUDA CSR 0 (thread per row)": "1.04e-3 s = 8.4 GFLOP/s", "CUDA CSR 1 (warp per row)": "8.92e-5 s = 97.42 GFLOP/s (INVALID: rows 0..31 only)"}},
)


def scratch_dir():
    for d in ("/dev/shm", os.environ.get("TMPDIR", ""), "/tmp"):
        if d and os.path.isdir(d) and os.access(d, os.W_OK):
            return d
    return "."


def measure_structured(api, synth, torch, args, steps, warmup, only=""):
    import numpy as np
    from spmv_openmp_cuda_amd.ctypes_defs import ref_CONFIG
    H, vp = api.hostlib, C.c_void_p
    ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    ora.oracleCsrSerial.argtypes = [C.c_ulong, vp, vp, vp, vp, vp]
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libspmvref.so")
    cpu = C.CDLL(ref_path) if os.path.exists(ref_path) else ora          # both export spmvRowsBasicCSR / spmvRowsBasicELL
    cpu_kind = "reference" if cpu is not ora else "port"
    for fn in (cpu.spmvRowsBasicCSR, cpu.spmvRowsBasicELL):
        fn.argtypes = [vp, vp, vp, vp]
    (cpu.refSetSchedule if cpu_kind == "reference" else ora.oracleSetSchedule)(1, 0)              # static
    noop = (cpu.refChunksNOOP if cpu_kind == "reference" else None)
    if noop is not None:
        noop.restype = vp
    out = {"note": "stand-ins shaped like the matrices of the reference's report, generated -> .mtx -> MMtoCSR/MMtoELL -> upload -> launchers; "
                   "`published` = the reference's own tables for the original matrix (other, unstated hardware)", "matrices": []}
    for spec in STRUCTURED:
        if only and only != "all" and (only[:-1] != spec["name"] if only.endswith("$") else only not in spec["name"]):
            continue
        p0, p1, p2 = spec["p"]
        if args.scale != 1.0:
            p0 = max(8, int(p0 * args.scale))
        path = os.path.join(scratch_dir(), f"spmv_bench_{os.getpid()}_{spec['name']}.mtx")
        Mv, NZv, mxv = C.c_ulong(), C.c_ulong(), C.c_ulong()
        blk = {"name": spec["name"], "shaped_after": spec["after"], "published_for_the_original": spec["published"], "runs": []}
        csr = ell = ell_t = None
        dms = []
        try:
            t0 = time.perf_counter()
            pattern = bool(spec.get("pattern"))
            if H.spmvSynthWriteMtx(path.encode(), spec["kind"] | (16 if pattern else 0), p0, p1, p2, 0x57A7 + spec["kind"], C.byref(Mv), C.byref(NZv), C.byref(mxv)):
                raise RuntimeError("spmvSynthWriteMtx failed")
            t1 = time.perf_counter()
            csr = H.MMtoCSR(path.encode())
            t2 = time.perf_counter()
            ell = H.MMtoELL(path.encode())
            t3 = time.perf_counter()
            if not csr or not ell:
                raise RuntimeError("the loader refused the generated file")
            M, nnz, K = int(Mv.value), int(NZv.value), int(mxv.value)
            m = csr.contents
            assert (m.M, m.NZ) == (M, nnz) and ell.contents.MAX_ROW_NZ == K
            blk.update({"M": M, "N": M, "nnz": nnz, "row_len_avg": nnz / M, "row_len_max": K, "mtx_bytes": os.path.getsize(path),
                        "write_mtx_s": t1 - t0, "MMtoCSR_s": t2 - t1, "MMtoELL_s": t3 - t2,
                        "loader_entries_per_s": nnz / (t2 - t1), "ell_padding_ratio_MK_over_nnz": M * K / nnz})
            irp = np.ctypeslib.as_array(m.IRP, shape=(M + 1,))
            ja = np.ctypeslib.as_array(m.JA, shape=(nnz,))
            as_ = np.ctypeslib.as_array(m.AS, shape=(nnz,))
            x_host = synth.make_x(M, 20 + spec["kind"])
            y_ref = np.empty(M)
            ora.oracleCsrSerial(M, irp.ctypes.data_as(vp), ja.ctypes.data_as(vp), as_.ctypes.data_as(vp), x_host.ctypes.data_as(vp), y_ref.ctypes.data_as(vp))
            prod = np.abs(as_ * x_host[ja])
            scale = np.add.reduceat(np.concatenate([prod, [0.0]]), np.minimum(irp[:-1].astype(np.int64), nnz))
            scale = np.where(np.diff(irp.astype(np.int64)) > 0, scale, 0.0)
            del prod
            b_csr = synth.algorithmic_bytes_csr(nnz, M, M)
            b_ell = M * K * 12 + M * 8 + M * 8
            blk["algorithmic_bytes"] = {"B_csr = B_ell_rl": b_csr, "B_ell (all slots)": b_ell}
            if pattern:
                blk["algorithmic_bytes"]["B_csr without the value stream"] = b_csr - nnz * 8
            x = torch.from_numpy(x_host).cuda()
            y = torch.full((M,), float("nan"), dtype=torch.float64, device="cuda")

            def run(label, launcher, mat, nbytes, variant=-1, rl=None, bitwise=False):
                if variant >= 0:
                    api.set_variant(launcher, variant)
                if rl is not None:
                    api.lib.spmvHipSetEllRowLens(1 if rl else 0)
                y.fill_(float("nan"))
                res, kms = time_launcher(api, torch, mat, launcher, x, y, steps, warmup)
                k_avg, k_var = avg_var(api, kms)
                yg = y.cpu().numpy()
                d = np.abs(yg - y_ref)
                nz_rows = scale > 0
                par = {"rows_checked": M, "nan": bool(np.isnan(yg).any()), "max_abs_diff": float(np.nanmax(d)),
                       "max_diff_over_sum_abs_ax": float(np.max(d[nz_rows] / scale[nz_rows])) if nz_rows.any() else 0.0}
                if bitwise:
                    par["bit_identical"] = bool(np.array_equal(yg, y_ref + 0.0))
                par["ok"] = bool(not par["nan"] and par["max_abs_diff"] <= GATE and par.get("bit_identical", True))
                r = {"kernel": label, "launcher": launcher, "variant": variant if variant >= 0 else None, "row_lens_early_exit": rl,
                     "kernel_ms_avg": k_avg, "kernel_ms_var": k_var, "gflops": 2.0 * nnz / (k_avg * 1e-3) * 1e-9,
                     "bytes_convention": "B_csr" if nbytes == b_csr else "B_ell", "hbm_gbps": nbytes / (k_avg * 1e-3) * 1e-9,
                     "hbm_frac": nbytes / (k_avg * 1e-3) / HBM_PEAK, "parity": par}
                if pattern and r["bytes_convention"] == "B_csr":
                    r["hbm_frac_without_value_stream"] = (b_csr - nnz * 8) / (k_avg * 1e-3) / HBM_PEAK
                if launcher == "hipSpMVRowsCSR" and variant == 2:
                    nm = api.lib.spmvHipAutoChoiceRows(C.byref(mat.handle), None)
                    r["resolved_to"] = nm.decode() if nm else None
                if launcher == "hipSpMVWarpPerRowCSR" and variant == 2:
                    nm = api.lib.spmvHipAutoChoice(C.byref(mat.handle), None)
                    r["resolved_to"] = nm.decode() if nm else None
                blk["runs"].append(r)
                log(f"{spec['name']} {label}: kernel {k_avg:.4f} ms  {r['gflops']:.0f} GFLOP/s  {100 * r['hbm_frac']:.1f}% ({r['bytes_convention']})  "
                    f"parity ok={par['ok']}" + (f"  -> {r['resolved_to']}" if r.get("resolved_to") else ""))
                if not par["ok"]:
                    raise SystemExit(f"PARITY FAILURE on {spec['name']} [{label}]: {par}")

            t4 = time.perf_counter()
            dm = api.DeviceMatrix()
            if api.lib.spMatCpyCSR(csr, C.byref(dm.handle)):
                raise RuntimeError("spMatCpyCSR failed")
            dm.rows = M
            dms.append(dm)
            blk["spMatCpyCSR_s"] = time.perf_counter() - t4
            uv = C.c_double(0.0)
            blk["all_values_equal"] = {"recognised": api.lib.spmvHipUnitValue(C.byref(dm.handle), C.byref(uv)) == 1, "value": uv.value}
            if pattern and not blk["all_values_equal"]["recognised"]:
                raise SystemExit(f"{spec['name']}: a pattern file was not recognised as an all-values-equal matrix")
            run("CSR thread per row, as named by the reference (hipSpMVRowsCSR, default: fastest serial-order kernel)", "hipSpMVRowsCSR", dm, b_csr, 2, bitwise=True)
            run("CSR thread per row, LDS-stream kernel", "hipSpMVRowsCSR", dm, b_csr, 1, bitwise=True)
            run("CSR wavefront per row, as named by the reference (hipSpMVWarpPerRowCSR, default: fastest reduction-order kernel)", "hipSpMVWarpPerRowCSR", dm, b_csr, 2)
            run("CSR wavefront per row, LDS-stream kernel + LDS segmented reduction", "hipSpMVWarpPerRowCSR", dm, b_csr, 1)
            run("CSR one wavefront per row (the reference kernel's intent)", "hipSpMVWarpPerRowCSR", dm, b_csr, 0)
            run("CSR two-phase", "hipSpMVTilesCSR", dm, b_csr)
            run("CSR stripes", "hipSpMVStripesCSR", dm, b_csr)
            if not pattern:
                run("SELL-C-sigma", "hipSpMVRowsSELL", dm, b_csr)
            set_rows_variant(api, 1)
            api.set_variant("hipSpMVWarpPerRowCSR", 1)
            dm.free()
            dms.clear()
            # ELL: the loader's row-major matrix, and its ellTranspose()d form for the coalesced thread-per-row kernel
            t5 = time.perf_counter()
            ell_t = H.ellTranspose(ell)
            blk["ellTranspose_s"] = time.perf_counter() - t5
            if not ell_t:
                raise RuntimeError("ellTranspose failed")
            de_t = api.DeviceMatrix()
            if api.lib.spMatCpyELL(ell_t, C.byref(de_t.handle)):
                raise RuntimeError("spMatCpyELL (transposed) failed")
            dms.append(de_t)
            for rl in ((True,) if pattern else (True, False)):
                run("ELL transposed+pitched, thread per row", "hipSpMVRowsELL", de_t, b_csr if rl else b_ell, rl=rl, bitwise=True)
            de_t.free()
            dms.clear()
            if pattern:                                   # the row-major ELL family and the CPU passes were measured on the valued twin
                best = min(blk["runs"], key=lambda r: r["kernel_ms_avg"])
                blk["fastest"] = {"kernel": best["kernel"], "kernel_ms_avg": best["kernel_ms_avg"], "gflops": best["gflops"],
                                  "hbm_frac_B_csr": b_csr / (best["kernel_ms_avg"] * 1e-3) / HBM_PEAK,
                                  "hbm_frac_without_value_stream": (b_csr - nnz * 8) / (best["kernel_ms_avg"] * 1e-3) / HBM_PEAK}
                out["matrices"].append(blk)
                continue                                  # (the finally below still frees and removes)
            de = api.DeviceMatrix()
            if api.lib.spMatCpyELL(ell, C.byref(de.handle)):
                raise RuntimeError("spMatCpyELL failed")
            dms.append(de)
            for rl in (True, False):
                run("ELL row-major, thread per row", "hipSpMVRowsELLNNTransposed", de, b_csr if rl else b_ell, rl=rl, bitwise=True)
                run("ELL row-major, lanes per row (wavefront-per-row family)", "hipSpMVWarpsPerRowELLNTrasposed", de, b_csr if rl else b_ell, rl=rl)
            de.free()
            dms.clear()
            api.lib.spmvHipSetEllRowLens(1)
            best = min(blk["runs"], key=lambda r: r["kernel_ms_avg"])
            blk["fastest"] = {"kernel": best["kernel"], "kernel_ms_avg": best["kernel_ms_avg"], "gflops": best["gflops"], "hbm_frac_B_csr": b_csr / (best["kernel_ms_avg"] * 1e-3) / HBM_PEAK}
            # the reference's OpenMP row-parallel kernels on the loader's own host matrices (static schedule, all entitled cores)
            if not args.no_cpu_baseline:
                cfg = ref_CONFIG()
                cfg.gridRows = cfg.gridCols = 8
                cfg.threadNum = int((cpu.refMaxThreads if cpu_kind == "reference" else ora.oracleMaxThreads)())
                cfg.chunkDistrbFunc = noop() if noop is not None else None
                y_cpu = np.empty(M)
                base = {"kind": cpu_kind, "cores": int(cfg.threadNum), "unit": "GFLOP/s", "schedule": "static", **effective_cpus()}
                for label, fn, mat in (("spmvRowsBasicCSR", cpu.spmvRowsBasicCSR, csr), ("spmvRowsBasicELL (row lengths)", cpu.spmvRowsBasicELL, ell)):
                    ts = []
                    for _ in range(4):
                        y_cpu[:] = np.nan
                        t0 = time.perf_counter()
                        rc = fn(mat, x_host.ctypes.data_as(vp), C.byref(cfg), y_cpu.ctypes.data_as(vp))
                        ts.append(time.perf_counter() - t0)
                        if rc or not np.all(np.abs(y_cpu - y_ref) <= GATE):
                            raise SystemExit(f"CPU baseline {label} failed the gate on {spec['name']}")
                    base[label] = {"gflops": 2.0 * nnz / (sum(ts[1:]) / 3) * 1e-9, "pass_seconds_avg": sum(ts[1:]) / 3}
                blk["cpu_baseline"] = base
                log(f"{spec['name']} CPU ({cpu_kind}, {cfg.threadNum} threads): CSR {base['spmvRowsBasicCSR']['gflops']:.2f}  ELL {base['spmvRowsBasicELL (row lengths)']['gflops']:.2f} GFLOP/s")
        finally:
            for dmx in dms:
                dmx.free()
            for ptr in (csr, ell, ell_t):
                if ptr:
                    H.freeSpmat(ptr)
            if os.path.exists(path):
                os.remove(path)
            torch.cuda.empty_cache()
        out["matrices"].append(blk)
    return out


def emit(line):
    """The ONE JSON line goes to the real stdout (saved before RCCL could write its banner there)."""
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


_REAL_STDOUT = None


def run_single(args, api, synth, torch, w):
    steps, warmup = args.steps, args.warmup
    if args.only_structured:
        emit({"metric": "spmv_gflops", "unit": "GFLOP/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "dtype": "f64", "data": "synthetic",
              "extra_structured": measure_structured(api, synth, torch, args, min(steps, 10), min(warmup, 2), only=args.only_structured)})
        return
    res = measure_block(api, synth, torch, w, args.launcher, steps, warmup,
                        cpu_rows=0 if args.no_cpu_baseline else args.cpu_sample_nnz)
    extra = {}
    if not args.no_extra:
        for key in ("c3", "c3b", "c2"):
            if key == args.workload:
                continue
            we = synth.WORKLOADS[key]
            if args.scale != 1.0:
                we = synth.scaled(we, args.scale)
            # config 2 is quoted on a thread-per-row kernel: pick among the one-lane-per-row launchers only (all are timed)
            # -- hipSpMVRowsCSR in its library default (variant 2: the fastest SERIAL-ORDER kernel, bit-identical to one thread
            # walking the row) and SELL
            cands = THREAD_PER_ROW if key.startswith("c2") else None
            set_rows_variant(api, 2 if key.startswith("c2") else 1)
            # the configuration the roofline target is quoted on also gets the CPU baseline, on the WHOLE matrix
            extra[key] = measure_block(api, synth, torch, we, args.launcher, steps, warmup, cands,
                                       cpu_rows=(1 << 62) if key == "c3" and not args.no_cpu_baseline else 0)
            set_rows_variant(api, 1)
        extra["c4"] = measure_c4(api, synth, torch, args, min(steps, 10), min(warmup, 2))
        extra["structured"] = measure_structured(api, synth, torch, args, min(steps, 10), min(warmup, 2))
    traffic, traffic_note = pmc_traffic(api, res["workload"]["workload"], res["launcher"])
    line = {
        "metric": "spmv_gflops", "value": res["gflops"], "unit": "GFLOP/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": res["workload"]["workload"], **{k: v for k, v in res["workload"].items() if k != "workload"},
                   "kernel": res["launcher"], "auto_candidates_ms": res["auto_candidates_ms"],
                   "library_auto_choice": res.get("library_auto_choice"),
                   "library_serial_order_choice": res.get("library_serial_order_choice"),
                   "parallelism": "1 GPU", "libspmvhip_sha256": lib_sha256(api)},
        "hbm_gbps": res["hbm_gbps"], "hbm_roofline_frac": res["hbm_frac"],
        "roofline": {"bound": "hbm", "achieved": res["hbm_gbps"], "peak": HBM_PEAK * 1e-9, "unit": "GB/s",
                     "frac": res["hbm_frac"], "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": KERNEL_OF.get(res["launcher"], res["launcher"]),
                     "algorithmic_bytes_per_launch": res["algorithmic_bytes"],
                     "kernel_ms_avg": res["kernel_ms_avg"], "kernel_ms_var": res["kernel_ms_var"],
                     "kernel_ms_phases": res.get("kernel_ms_phases"),
                     "format_build_ms": res.get("format_build_ms"), "format_build_alloc_ms": res.get("format_build_alloc_ms"),
                     "format_build_peak_temporary_bytes_per_nnz": res.get("format_build_peak_temporary_bytes_per_nnz"),
                     "spmvs_to_amortise_format_build": res.get("spmvs_to_amortise_format_build"),
                     "extra_device_bytes": res.get("extra_device_bytes")},
        "parity": res["parity"],
    }
    if "y_sha256" in res:
        line["y_sha256"] = res["y_sha256"]
    if "cpu_baseline" in res:
        line["cpu_baseline"] = res["cpu_baseline"]
    if "c3" in extra:
        line["headline_c3"] = extra["c3"]
    for k in ("c3b", "c2", "c4", "structured"):
        if k in extra:
            line["extra_" + k] = extra[k]
    emit(line)


# ------------------------------------------------------------------------------------------------- N > 1
def run_multi(args, api, synth, torch, dist, w, world, rank, local):
    import numpy as np
    from spmv_openmp_cuda_amd import sharding
    steps, warmup = args.steps, args.warmup
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    info = synth.describe(w, lens)
    nnz_total = int(irp[-1])
    x_host = synth.make_x(w.N, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    cfg = api.CONFIG()
    windows = OracleWindows(synth, w, irp, x_host, lens)       # every rank checks its OWN copy of y against the oracle
    del lens

    # this rank's matrices per (groups, rdiv, taper): built when a candidate first needs them, freed when no candidate
    # still to come (and not the best so far) does
    formats = {}

    def format_of(fkey):
        if fkey not in formats:
            groups, rdiv, taper = fkey
            plan_g = sharding.make_plan(irp, world, groups)
            dms = [synth.device_csr(w, irp, *plan_g.block(rank, g)) for g in range(groups)]
            if rdiv > 1 or taper:
                for dm in dms:
                    if dm.nnz > 0:
                        rpb = api.tiles_info(dm).rowsPerBin
                        if rpb == 0:
                            api.build_tiles(dm)                     # automatic format: the height to divide
                            rpb = api.tiles_info(dm).rowsPerBin
                        api.build_tiles(dm, rowsPerBin=max(64, (rpb // rdiv + 63) // 64 * 64) if rdiv > 1 else 0, taper=taper)
            formats[fkey] = (plan_g, dms)
        return formats[fkey]

    def drop_formats(keep):
        for fkey in [k for k in formats if k not in keep]:
            for dm in formats.pop(fkey)[1]:
                dm.free()
        torch.cuda.empty_cache()

    class RcclExchange:
        """this rank's `groups` row groups as device matrices + gather buffers; kernel(g) -> async RCCL all-gather(g)"""
        def __init__(self, key, launcher_name):
            self.key, self.groups = key, key.groups
            self.fn = api.SPMV_LAUNCHERS[launcher_name]
            self.plan, self.dms = format_of(key.format_key)
            self.bufs = sharding.GatherBuffers(self.plan, rank, torch, "cuda")
            self.y = self.bufs.y
            self.desc = (f"{world} ranks x {self.groups} nnz-balanced row group(s); per group: kernel then async RCCL "
                         f"all-gather(y) overlapping the next group" + ("" if self.plan.equal_blocks else "; padded blocks + compaction"))

        def poison(self):
            for b in self.bufs.ypad:
                b.fill_(float("nan"))
            self.bufs.y.fill_(float("nan"))
            torch.cuda.synchronize()

        def step(self, ev=None):
            def compute_group(g, slot):
                if ev:
                    api.lib.spmvHipEventRecord(ev[g][0])
                rc = self.fn(C.byref(self.dms[g].handle), x.data_ptr(), cfg, slot.data_ptr())
                if ev:
                    api.lib.spmvHipEventRecord(ev[g][1])
                if rc:
                    raise RuntimeError("launcher failed")
            return sharding.step(self.plan, dist, self.bufs, compute_group)

        def sync(self):
            torch.cuda.synchronize()

        def free(self):
            self.dms, self.bufs, self.y = [], None, None

    class PushExchange:
        """y lives in a peer window; this rank's rows are delivered to the other ranks' windows by copy-engine pushes
        behind each piece of y ("push"), by stores fused into phase 2 of the two-phase kernel ("fused") or by a push
        kernel beside phase 2 ("pushk"); with G > 1 row groups per rank the rows of one group travel while the next
        group is computed"""
        def __init__(self, key, launcher_name):
            self.key, self.groups = key, key.groups
            self.plan, dms = format_of(key.format_key)
            self.runs = [sharding.PushSpMV(api, px, dms[g], self.plan.block(rank, g)[0], launcher_name, x.data_ptr(), key.mode, key.pieces)
                         for g in range(self.groups)]
            self.y = px.y
            self.desc = (f"{world} ranks x {self.groups} nnz-balanced row group(s), y in peer windows (device IPC over xGMI): " +
                         ("phase 2 stores every finished bin of y to all ranks itself" if key.mode == "fused" else
                          "a push kernel beside phase 2 copies every bin of y to all ranks as soon as it is flagged" if key.mode == "pushk" else
                          f"up to {key.pieces} piece(s) of y (this rank: {self.runs[0].pieces}), each pushed to all ranks by the copy engines "
                          f"while the next is reduced") +
                         (f"; bins of 1/{key.rdiv} the automatic height" if key.rdiv > 1 else "") +
                         ("; tapered bins (a round of quarter-height bins first and last)" if key.taper else "") +
                         "; step ends with a 4-byte all-reduce as barrier")

        def poison(self):
            self.y.fill_(float("nan"))
            torch.cuda.synchronize()

        def step(self, ev=None):
            try:
                for g, run in enumerate(self.runs):
                    run.enqueue(ev[g] if ev else None)
            finally:
                px.finish()                          # also when a launcher failed here: the other ranks are in this barrier
            return self.y

        def sync(self):
            torch.cuda.synchronize()

        def free(self):
            self.runs, self.y = [], None

    # every rank must run the same kernel: rank 0 decides (on its block of the 1-group plan)
    base_plan, (base_dm,) = format_of((1, 1, False))
    probe = torch.empty(max(1, base_dm.rows), dtype=torch.float64, device="cuda")
    launcher, tried = pick_launcher(api, torch, base_dm, x.data_ptr(), probe.data_ptr(), args.launcher, w.N)
    del probe
    choice = torch.tensor([AUTO_CANDIDATES.index(launcher) if launcher in AUTO_CANDIDATES else -1], device="cuda")
    dist.broadcast(choice, 0)
    if int(choice) >= 0:
        launcher = AUTO_CANDIDATES[int(choice)]
    tiles = launcher == "hipSpMVTilesCSR"

    px = None
    if args.exchange in ("auto", "auto-no-rccl", "push", "fused"):
        px = sharding.PeerExchange(api, dist, torch, rank, world, local, w.N)
        if not px.ok:
            log("peer windows unavailable, RCCL only:", px.why)
            px = None
    use_rccl = args.exchange in ("auto", "rccl") and not args.rehearse_shared_gpu
    keys = sharding.default_candidates(world, tiles, rccl=use_rccl, windows=px is not None and args.exchange != "rccl",
                                       extra=args.exchange_extra)
    K = sharding.ExchangeKey
    if args.exchange == "push":
        keys = [k for k in keys if k.mode == "push"]
    if args.exchange == "fused":
        keys = [K("push", 1)] + [k for k in keys if k.mode in ("fused", "pushk")]
    if args.groups > 0:
        keys = [k for k in keys if k.groups == args.groups] or [K(keys[0].mode, keys[0].pieces, args.groups)]
    if args.pieces > 0:
        keys = [k for k in keys if k.mode != "push" or k.pieces == args.pieces] or [K("push", args.pieces)]
    if not keys:
        raise SystemExit("no exchange candidate is available (no RCCL and no peer windows)")
    log("exchange candidates:", [k.name for k in keys])

    def make(key):
        return RcclExchange(key, launcher) if key.mode == "rccl" else PushExchange(key, launcher)

    serial = launcher in ("hipSpMVRowsCSR", "hipSpMVRowsSELL")

    def validate(cand, ref):
        """the candidate's y on THIS rank: complete, equal to the oracle on the windows, and -- element by element -- equal
        to the first validated candidate's y (bit for bit for a serial-order kernel, to rounding otherwise); read back by
        fresh kernels launched after the step's barrier, so rows other ranks stored into this rank's window are seen as
        the owner's next kernel will see them"""
        torch.cuda.synchronize()
        y_c = cand.y
        if bool(torch.isnan(y_c).any()):
            return False
        if not windows.check(lambda a, b: y_c[a:b].cpu().numpy())["ok"]:
            return False
        if ref is None:
            return True
        if serial:
            return bool(torch.equal(y_c, ref))
        return float((y_c - ref).abs().max()) <= 1e-12 * max(float(ref.abs().max()), 1e-300)

    def on_resolved(i, best_key):
        keep = {k.format_key for k in keys[i + 1:]} | {(1, 1, False)} | ({best_key.format_key} if best_key else set())
        drop_formats(keep)

    best, report = sharding.search_exchange(keys, make, dist, torch, "cuda", validate, budget_s=args.exchange_budget, steps=3,
                                            log=log, on_resolved=on_resolved)
    if best is None:
        raise SystemExit(f"no exchange candidate produced a complete y: {report}")
    log("exchange search:", report)

    # the exchange alone (no kernels), for the record: what the links give an all-gather of y at this N
    def alone(fn_once, reps=5):
        fn_once()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn_once()
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0])
    exchange_alone_ms = {}
    if use_rccl:
        gb = sharding.GatherBuffers(base_plan, rank, torch, "cuda")
        exchange_alone_ms["rccl_all_gather"] = alone(lambda: dist.all_gather_into_tensor(gb.ypad[0], gb.slot[0]))
        del gb
    if px is not None:
        r0_, r1_ = base_plan.rows(rank)

        def push_once():
            px.push(r0_, r1_)
            px.finish()
        exchange_alone_ms["push_copy_engines"] = alone(push_once)
        exchange_alone_ms["barrier_only"] = alone(lambda: px.finish())
    log("exchange alone (ms, slowest rank):", exchange_alone_ms)

    setup = best
    groups, plan = setup.groups, setup.plan
    r0, r1 = plan.rows(rank)
    nnz_local = int(irp[r1] - irp[r0])
    evs = [make_events(api, groups) for _ in range(steps)]

    setup.poison()
    dist.barrier()                          # the warm-up steps of a fast rank must not land before a slow rank's poison
    wall = time_kernel_loop(torch, dist, world, lambda ev: setup.step(ev), steps, warmup, evs)
    kms = [sum(kernel_ms(api, per_step)) for per_step in evs]
    for per_step in evs:
        free_events(api, per_step)
    y = setup.y
    t = torch.tensor([wall, sum(kms) / len(kms)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, kernel_ms_max = float(t[0]), float(t[1])

    def identical_on_all_ranks(vec):
        """bytes of `vec` compared with rank 0's across the job (rank 0's vector is broadcast once)"""
        y0 = vec.clone()
        dist.broadcast(y0, 0)
        same = torch.tensor([1.0 if torch.equal(vec.view(torch.int64), y0.view(torch.int64)) else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        return float(same[0]) == 1.0

    nan_left = torch.isnan(y).sum().to(torch.float64)
    dist.all_reduce(nan_left, op=dist.ReduceOp.MAX)
    timed_par = windows.check(lambda a, b: y[a:b].cpu().numpy())
    ok_t = torch.tensor([1.0 if timed_par["ok"] else 0.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
    # arrival-order kernels (LDS atomics) are not bit-reproducible, so their y may differ in the last bits between ranks
    # only if ranks recomputed rows -- they do not: every row is computed once and delivered, so the bytes must agree
    same_timed = identical_on_all_ranks(y)

    # Bit-exactness of the sharding and of the exchange on these links, independently of any kernel's summation order:
    # ONE step with the serial-order launcher (hipSpMVRowsCSR: a row's products added in ascending j, bit-identical to
    # sgemvSerial) through the chosen exchange family; every rank's y must equal rank 0's byte for byte and the serial
    # oracle bit for bit on the windows.  ("gathered y equal to the 1-GPU y bitwise".)
    vkey = K("rccl", groups=1) if setup.key.mode == "rccl" else K("push", 1)
    vcand = RcclExchange(vkey, "hipSpMVRowsCSR") if vkey.mode == "rccl" else PushExchange(vkey, "hipSpMVRowsCSR")
    vcand.poison()
    dist.barrier()
    vcand.step()
    torch.cuda.synchronize()
    yv = vcand.y
    serial_par = windows.check(lambda a, b: yv[a:b].cpu().numpy(), bitwise=True)
    ok_s = torch.tensor([1.0 if serial_par["ok"] and not bool(torch.isnan(yv).any()) else 0.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(ok_s, op=dist.ReduceOp.MIN)
    same_serial = identical_on_all_ranks(yv)
    vcand.free()

    if rank == 0:
        bytes_alg_local = synth.algorithmic_bytes_csr(nnz_local, r1 - r0, w.N)
        bytes_alg_total = synth.algorithmic_bytes_csr(nnz_total, w.N, w.N)
        k_avg = kernel_ms_max * 1e-3
        parity_ok = bool(same_timed and same_serial and float(nan_left) == 0 and float(ok_t[0]) == 1.0 and float(ok_s[0]) == 1.0)
        line = {
            "metric": "spmv_gflops", "value": 2.0 * nnz_total / (wall_max / steps) * 1e-9, "unit": "GFLOP/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": wall_max / steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": info["workload"], **{k: v for k, v in info.items() if k != "workload"},
                       "kernel": launcher, "auto_candidates_ms": tried, "exchange": setup.key.name,
                       **report, "exchange_alone_ms": exchange_alone_ms, "parallelism": setup.desc,
                       "libspmvhip_sha256": lib_sha256(api)},
            "hbm_gbps": bytes_alg_total / (wall_max / steps) * 1e-9,
            "hbm_roofline_frac": bytes_alg_total / (wall_max / steps) / (HBM_PEAK * world),
            "roofline": {"bound": "hbm", "achieved": bytes_alg_local / k_avg * 1e-9, "peak": HBM_PEAK * 1e-9,
                         "unit": "GB/s", "frac": bytes_alg_local / k_avg / HBM_PEAK, "traffic": None,
                         "kernel": KERNEL_OF.get(launcher, launcher),
                         "algorithmic_bytes_per_launch": bytes_alg_local,
                         "kernel_ms_avg": kernel_ms_max, "note": "per-rank kernels (all row groups), slowest rank"},
            "exposed_exchange_ms_per_step": wall_max / steps * 1e3 - kernel_ms_max,
            "parity": {"ok": parity_ok, "gate_abs": GATE, "nan_left": float(nan_left),
                       "all_ranks_hold_identical_y": bool(same_timed), "every_rank_matches_oracle_windows": bool(float(ok_t[0]) == 1.0),
                       "timed_kernel_rank0": timed_par,
                       "serial_order_step": {"launcher": "hipSpMVRowsCSR", "exchange": vkey.name,
                                             "all_ranks_hold_identical_bytes": bool(same_serial),
                                             "bit_identical_to_serial_oracle_on_every_rank": bool(float(ok_s[0]) == 1.0),
                                             "rank0": serial_par}},
        }
        if args.y_hash:
            line["y_sha256"] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()
        emit(line)
        if not parity_ok:
            raise SystemExit(f"PARITY FAILURE at N = {world}: {line['parity']}")
    dist.barrier()
    setup.free()
    drop_formats(set())
    if px is not None:
        px.close()


def spawn_ranks(world):
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes (torch.distributed.run on the
    loopback address) and relay rank 0's JSON line and the job's exit code.  Runs before torch or the HIP library are
    imported -- this parent never touches a GPU and never replaces itself with another program."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL and the peer windows need it on this driver
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    out, _ = proc.communicate()
    lines = []
    for raw in out.splitlines():
        try:
            if "metric" in json.loads(raw):
                lines.append(raw)
                continue
        except ValueError:
            pass
        if raw.strip():
            print(raw, file=sys.stderr)                      # anything else a rank wrote to stdout is not the result line
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    if proc.returncode == 0 and not lines:
        print("[bench] the ranks ended without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    global _REAL_STDOUT
    args = parse()
    world = args.gpus
    multi = world > 1 or args.force_dist
    if world > 1 and ("RANK" not in os.environ or "WORLD_SIZE" not in os.environ):
        sys.exit(spawn_ranks(world))
    if multi:
        # RCCL prints a version banner on fd 1 when the communicator comes up; the contract is ONE JSON
        # line on stdout, so everything else of this process is sent to stderr
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    else:
        # the CPU baseline runs the way SURVEY 8d asks; libgomp reads these when it starts, i.e. before the first
        # OpenMP library is loaded (ranks of a multi-GPU job share the host cores and are not bound)
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
        # one OpenMP thread per CPU this process is ENTITLED to (affinity mask capped by the cgroup quota): 128 threads
        # on a 16-CPU share would time the scheduler, not the loop
        os.environ.setdefault("OMP_NUM_THREADS", str(effective_cpus()["effective_cpus"]))
    import torch                      # BEFORE the HIP library: one HIP runtime per process (see api.py)
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if multi:
        if int(os.environ.get("WORLD_SIZE", "1")) != world:
            raise SystemExit(f"--gpus {world} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_shared_gpu:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if world > 1:                                    # host-side generators are OpenMP: share the cores between the ranks
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // world)))
    from spmv_openmp_cuda_amd import api, synth
    api.spmvHipInit(local)
    api.lib.spmvHipSetStream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    api.lib.spmvHipSetSync(0)
    # this program compares the library's kernels by NAME itself: "hipSpMVWarpPerRowCSR" here is the LDS-stream kernel
    # (variant 1), not the default variant 2 that picks among them -- what that default picks is reported per block as
    # `library_auto_choice`
    api.set_variant("hipSpMVWarpPerRowCSR", 1)
    set_rows_variant(api, 1)
    if args.variant >= 0 and args.launcher != "auto":
        if args.launcher == "hipSpMVRowsCSR":
            set_rows_variant(api, args.variant)
        else:
            api.set_variant(args.launcher, args.variant)
    w = synth.WORKLOADS[args.workload]
    if args.scale != 1.0:
        w = synth.scaled(w, args.scale)
    _Y_HASH[0] = bool(args.y_hash)
    if multi:
        run_multi(args, api, synth, torch, dist, w, world, rank, local)
        api.spmvHipFinalize()
        dist.destroy_process_group()
    else:
        run_single(args, api, synth, torch, w)
        api.spmvHipFinalize()


if __name__ == "__main__":
    main()
