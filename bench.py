#!/usr/bin/env python3
"""bench.py -- SpMV GFLOP/s + achieved HBM GB/s on synthetic CSR matrices, 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c3|c2|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one y = A.x over the whole matrix: every rank runs the hot kernel
(a C-ABI launcher, picked among hipSpMVWarpPerRowCSR / hipSpMVTilesCSR / hipSpMVRowsSELL
by timing them during warm-up) on its nnz-balanced contiguous row block and, for
N > 1, the ranks exchange y over xGMI so that each ends with the full vector (ready
to be the next x): an RCCL all-gather, copy-engine pushes into peer windows, stores
fused into the producing kernel or a push kernel beside it -- whichever whole step
measures fastest at start-up.  The default workload is
BASELINE.json configs[4] (power-law CSR, 80 M rows / 1.6 G nnz, max row 50 k),
the configuration the metric's 1/2/4/8-GPU curve is quoted on; it is kept
whole at every N (strong scaling).  At N = 1 the same run also measures
configs[2] (10 M / 200 M) and reports it under "headline_c3".

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" (algorithmic bytes / HIP-event kernel time vs 8 TB/s) and
"cpu_baseline" (the reference's own spmvRowsBasicCSR from oracle/_ref, or the
oracle port, timed on this box's host cores on a bounded row sample, and used
as the checker for the GPU result on those rows).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12           # bytes/s, /opt/skills/guides/MI355X_MICROARCH.md


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
                   help="a C-ABI launcher name, or 'auto': time hipSpMVWarpPerRowCSR (one-pass LDS-stream kernel) and "
                        "hipSpMVTilesCSR (column-sliced two-phase kernel) during warm-up and keep the faster")
    p.add_argument("--variant", type=int, default=-1)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the secondary workloads measured at N=1")
    p.add_argument("--cpu-sample-nnz", type=int, default=100_000_000)
    p.add_argument("--groups", type=int, default=0,
                   help="row groups per rank for kernel/all-gather overlap at N > 1 (0 = measure 1/2[/4] and keep the fastest)")
    p.add_argument("--exchange", default="auto", choices=["auto", "rccl", "push", "fused", "auto-no-rccl"],
                   help="how y reaches the other ranks at N > 1: RCCL all-gather, copy-engine pushes into peer windows, or "
                        "stores fused into phase 2 of the two-phase kernel (auto = measure all, keep the fastest)")
    p.add_argument("--pieces", type=int, default=0, help="pieces of y per rank for --exchange push (0 = measure 1/2/4/8)")
    p.add_argument("--rehearse-shared-gpu", action="store_true",
                   help="development only: all ranks use GPU 0 and gloo is the control plane (RCCL refuses two ranks on one "
                        "device), so the N > 1 code path -- plan, candidates, peer windows, checks -- can run on a one-GPU "
                        "box; the RCCL all-gather candidates are skipped and the timings mean nothing")
    p.add_argument("--force-dist", action="store_true",
                   help="run the multi-rank code path (process group, all-gather) even with --gpus 1; for rehearsal")
    return p.parse_args()


def time_kernel_loop(api, torch, dist, world, step, steps, warmup, ev_pairs):
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


def kernel_ms(api, ev_pairs):
    out = []
    ms = C.c_float()
    for a, b in ev_pairs:
        api.lib.spmvHipEventElapsedMs(a, b, C.byref(ms))
        out.append(ms.value)
    return out


KERNEL_OF = {"hipSpMVTilesCSR": "pb_expand_kernel + pb_reduce_kernel", "hipSpMVWarpPerRowCSR": "csr_stream2_kernel",
             "hipSpMVRowsCSR": "csr_stream2_kernel", "hipSpMVRowsSELL": "sell_spmv_kernel"}
AUTO_CANDIDATES = ("hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVRowsSELL")


THREAD_PER_ROW = ("hipSpMVRowsCSR", "hipSpMVRowsSELL")      # both add a row's products in ascending j with one lane


def pick_launcher(api, torch, dm, x_ptr, y_ptr, requested, eligible=None):
    """'auto' -> run each candidate 3x (first call of the tiles launcher also builds its
    slice-major format) and keep the fastest of the `eligible` ones (default: all); the others
    are timed for information.  Returns (name, {name: ms})."""
    if requested != "auto":
        return requested, {}
    cfg = api.CONFIG()
    times = {}
    for name in dict.fromkeys(tuple(eligible or ()) + AUTO_CANDIDATES):
        fn = api.SPMV_LAUNCHERS[name]
        if fn(C.byref(dm.handle), x_ptr, cfg, y_ptr):
            continue                                  # e.g. tiles unsupported for this shape: skip
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn(C.byref(dm.handle), x_ptr, cfg, y_ptr)
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / 3 * 1e3
    best = min((n for n in times if not eligible or n in eligible), key=times.get)
    return best, times


def pmc_traffic(workload_name, launcher):
    """HBM bytes per launch measured by separate rocprofv3 --pmc passes of this command
    (profiles/traffic.json, written by scripts/summarize_profile.py runs); None if not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload_name, {}).get(launcher)
    except OSError:
        return None


def measure_single(api, synth, torch, w, launcher, steps, warmup, candidates=None):
    """1-GPU measurement of workload `w`; returns (dict, context for checks)."""
    import numpy as np
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    info = synth.describe(w, lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x_host = synth.make_x(w.N, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    y = torch.full((w.N,), float("nan"), dtype=torch.float64, device="cuda")
    launcher, tried = pick_launcher(api, torch, dm, x.data_ptr(), y.data_ptr(), launcher, candidates)
    y.fill_(float("nan"))
    fn = api.SPMV_LAUNCHERS[launcher]
    cfg = api.CONFIG()
    evs = [(C.c_void_p(), C.c_void_p()) for _ in range(steps)]
    for a, b in evs:
        api.lib.spmvHipEventCreate(C.byref(a))
        api.lib.spmvHipEventCreate(C.byref(b))

    def step(ev):
        if ev:
            api.lib.spmvHipEventRecord(ev[0])
        rc = fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr())
        if ev:
            api.lib.spmvHipEventRecord(ev[1])
        if rc:
            raise RuntimeError(launcher + " failed")

    wall = time_kernel_loop(api, torch, None, 1, step, steps, warmup, evs)
    kms = kernel_ms(api, evs)
    for a, b in evs:
        api.lib.spmvHipEventDestroy(a)
        api.lib.spmvHipEventDestroy(b)
    nnz = int(irp[-1])
    bytes_alg = synth.algorithmic_bytes_csr(nnz, w.N, w.N)
    k_avg = sum(kms) / len(kms) * 1e-3
    res = {
        "workload": info, "launcher": launcher, "auto_candidates_ms": tried,
        "eligible_launchers": list(candidates) if candidates else "all",
        "extra_device_bytes": int(api.lib.spmvHipTilesBytes(C.byref(dm.handle))) if launcher == "hipSpMVTilesCSR" else
                              int(api.lib.spmvHipSellBytes(C.byref(dm.handle))) if launcher == "hipSpMVRowsSELL" else 0,
        "ms_per_step": wall / steps * 1e3, "kernel_ms_avg": k_avg * 1e3, "kernel_ms_min": min(kms),
        "gflops": 2.0 * nnz / (wall / steps) * 1e-9,
        "hbm_gbps": bytes_alg / k_avg * 1e-9, "hbm_frac": bytes_alg / k_avg / HBM_PEAK,
        "algorithmic_bytes": bytes_alg,
    }
    return res, dict(dm=dm, irp=irp, x_host=x_host, y=y, x=x, lens=lens)


def oracle_spot_checks(synth, w, irp, x_host, y_gpu_fn, windows=None):
    """GPU y against the serial oracle on 200 k-row windows (head, middle, tail by default); the rows are
    regenerated by the CPU twin of the device generator.  Checker use of oracle/ (allowed for bench.py)."""
    import numpy as np
    ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    vp = C.c_void_p
    ora.synthFillCsrRef.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    ora.oracleCsrSerial64_32.argtypes = [C.c_ulong, vp, vp, vp, vp, vp]
    spot = []
    for r0 in (windows if windows is not None else (0, w.N // 2, max(0, w.N - 200_000))):
        r1 = min(w.N, r0 + 200_000)
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
        yg = y_gpu_fn(r0, r1)
        spot.append({"rows": [int(r0), int(r1)], "max_abs_diff": float(np.max(np.abs(yr - yg))) if r1 > r0 else 0.0,
                     "nan": bool(np.isnan(yg).any())})
    return spot


def cpu_baseline_and_check(api, synth, w, irp, x_host, y_gpu_fn, sample_nnz, iters=5):
    """Time the reference's spmvRowsBasicCSR (oracle/_ref) -- or the oracle port --
    on the first rows of the workload (<= sample_nnz nnz) with all host cores, and
    use its y as the checker for the GPU's y on those rows."""
    import numpy as np
    from spmv_openmp_cuda_amd.ctypes_defs import ref_CONFIG, ref_spmat
    rows = int(np.searchsorted(irp, sample_nnz, side="right") - 1)
    rows = max(1, min(rows, w.N))
    irp_s = np.ascontiguousarray(irp[:rows + 1], dtype=np.uint64)
    nnz_s = int(irp_s[-1])
    ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    ja32 = np.empty(nnz_s, dtype=np.uint32)
    as_ = np.empty(nnz_s, dtype=np.float64)
    vp = C.c_void_p
    ora.synthFillCsrRef.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
    ora.synthFillCsrRef(rows, w.N, 0, irp_s.ctypes.data_as(vp), ja32.ctypes.data_as(vp), as_.ctypes.data_as(vp),
                        synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
    ja64 = ja32.astype(np.uint64)
    del ja32
    y_cpu = np.full(rows, np.nan)
    cores = len(os.sched_getaffinity(0))
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libspmvref.so")
    times = []
    extra_sched = {}
    if os.path.exists(ref_path):
        kind = "reference"
        ref = C.CDLL(ref_path)
        ref.refChunksNOOP.restype = vp
        ref.refSetSchedule(1, 0)                                  # omp_sched_static, like OMP_SCHEDULE=static
        m = ref_spmat()
        m.M, m.N, m.NZ = rows, w.N, nnz_s
        rl = np.diff(irp_s).astype(np.uint64)
        m.IRP = irp_s.ctypes.data_as(C.POINTER(C.c_ulong))
        m.JA = ja64.ctypes.data_as(C.POINTER(C.c_ulong))
        m.AS = as_.ctypes.data_as(C.POINTER(C.c_double))
        m.RL = rl.ctypes.data_as(C.POINTER(C.c_ulong))
        cfg = ref_CONFIG()
        cfg.gridRows = cfg.gridCols = 8
        cfg.threadNum = ref.refMaxThreads()
        cfg.chunkDistrbFunc = ref.refChunksNOOP()
        cores = cfg.threadNum
        ref.spmvRowsBasicCSR.argtypes = [C.POINTER(ref_spmat), vp, C.POINTER(ref_CONFIG), vp]
        ref.refChunksFairFolded.restype = vp

        def timed_passes():
            ts = []
            for _ in range(iters + 1):
                t0 = time.perf_counter()
                rc = ref.spmvRowsBasicCSR(C.byref(m), x_host.ctypes.data_as(vp), C.byref(cfg), y_cpu.ctypes.data_as(vp))
                ts.append(time.perf_counter() - t0)
                assert rc == 0
            return ts[1:]                                         # first pass = page-in

        times = [0.0] + timed_passes()                            # OMP_SCHEDULE=static (the faster one in the reference's report)
        # second schedule of BASELINE.md section 3: dynamic with the reference's fair-folded chunk rewrite
        ref.refSetSchedule(2, 1)                                  # omp_sched_dynamic, chunk 1 -> rewritten by chunksFairFolded
        cfg.chunkDistrbFunc = ref.refChunksFairFolded()
        dyn = timed_passes()
        extra_sched = {"omp_dynamic_fair_folded_gflops": 2.0 * nnz_s / (sum(dyn) / len(dyn)) * 1e-9}
        ref.refSetSchedule(1, 0)
        cfg.chunkDistrbFunc = ref.refChunksNOOP()
        # third figure of SURVEY 8d: the reference compiled with SIMD_ROWS_REDUCTION off, static schedule
        nosimd_path = os.path.join(ROOT, "oracle", "_ref", "libspmvref_nosimd.so")
        if os.path.exists(nosimd_path):
            ref2 = C.CDLL(nosimd_path)
            ref2.refChunksNOOP.restype = vp
            ref2.refSetSchedule(1, 0)
            cfg2 = ref_CONFIG()
            cfg2.gridRows = cfg2.gridCols = 8
            cfg2.threadNum = ref2.refMaxThreads()
            cfg2.chunkDistrbFunc = ref2.refChunksNOOP()
            ref2.spmvRowsBasicCSR.argtypes = [C.POINTER(ref_spmat), vp, C.POINTER(ref_CONFIG), vp]
            y2 = np.empty(rows)
            ts2 = []
            for _ in range(iters + 1):
                t0 = time.perf_counter()
                rc = ref2.spmvRowsBasicCSR(C.byref(m), x_host.ctypes.data_as(vp), C.byref(cfg2), y2.ctypes.data_as(vp))
                ts2.append(time.perf_counter() - t0)
                assert rc == 0
            extra_sched["omp_static_no_simd_reduction_gflops"] = 2.0 * nnz_s / (sum(ts2[1:]) / len(ts2[1:])) * 1e-9
    else:
        kind = "port"
        ora.oracleSetSchedule(1, 0)
        cores = ora.oracleMaxThreads()
        ora.oracleCsrOmp32.argtypes = [C.c_ulong, vp, vp, vp, vp, vp]
        irp32 = irp_s.astype(np.uint32)
        ja32 = ja64.astype(np.uint32)
        for _ in range(iters + 1):
            t0 = time.perf_counter()
            ora.oracleCsrOmp32(rows, irp32.ctypes.data_as(vp), ja32.ctypes.data_as(vp), as_.ctypes.data_as(vp),
                               x_host.ctypes.data_as(vp), y_cpu.ctypes.data_as(vp))
            times.append(time.perf_counter() - t0)
    # spot checks away from the head of the matrix (a wrapped grid or a wrong row-block table would show here)
    spot = oracle_spot_checks(synth, w, irp, x_host, y_gpu_fn, windows=(w.N // 2, max(0, w.N - 200_000)))
    times = times[1:]                                             # first pass = page-in
    t = sum(times) / len(times)
    y_gpu_head = y_gpu_fn(0, rows)
    diff = np.abs(y_cpu - y_gpu_head)
    parity = {"rows_checked": rows, "max_abs_diff": float(np.nanmax(diff)), "gate_abs": 7e-4,
              "nan_in_gpu_y": bool(np.isnan(y_gpu_head).any()), "spot_checks": spot,
              "ok": bool(not np.isnan(y_gpu_head).any() and np.nanmax(diff) <= 7e-4 and
                         all(not c["nan"] and c["max_abs_diff"] <= 7e-4 for c in spot))}
    base = {"value": 2.0 * nnz_s / t * 1e-9, "unit": "GFLOP/s", "cores": int(cores), "kind": kind,
            "sample": f"rows [0,{rows}) of the workload = {nnz_s} nnz, x full length, {len(times)} timed passes of "
                      f"spmvRowsBasicCSR, OMP schedule static, {t * 1e3:.2f} ms/pass",
            "hbm_like_gbps": synth.algorithmic_bytes_csr(nnz_s, rows, w.N) / t * 1e-9}
    base.update(extra_sched)
    return base, parity


def emit(line):
    """The ONE JSON line goes to the real stdout (saved before RCCL could write its banner there)."""
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


_REAL_STDOUT = None


def main():
    global _REAL_STDOUT
    args = parse()
    if args.gpus > 1 or args.force_dist:
        # RCCL prints a version banner on fd 1 when the communicator comes up; the contract is ONE JSON
        # line on stdout, so everything else of this process is sent to stderr
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    import numpy as np
    import torch                      # BEFORE the HIP library: one HIP runtime per process (see api.py)
    import torch.distributed as dist
    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        assert int(os.environ.get("WORLD_SIZE", "1")) == world, "launch with torch.distributed.run --nproc-per-node N"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_shared_gpu:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
            if args.exchange in ("auto", "rccl"):
                args.exchange = "auto-no-rccl"
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
    if args.variant >= 0 and args.launcher != "auto":
        api.set_variant(args.launcher, args.variant)

    w = synth.WORKLOADS[args.workload]
    if args.scale != 1.0:
        w = synth.scaled(w, args.scale)
    steps, warmup = args.steps, args.warmup
    t_setup = time.perf_counter()

    extra = {}
    if world == 1 and not args.force_dist:
        res, ctx = measure_single(api, synth, torch, w, args.launcher, steps, warmup)
        log(f"{w.name} [{res['launcher']}]: {res['gflops']:.1f} GFLOP/s  kernel {res['kernel_ms_avg']:.3f} ms  "
            f"{res['hbm_gbps']:.0f} GB/s = {100 * res['hbm_frac']:.1f}% of 8 TB/s   (setup {time.perf_counter() - t_setup:.1f}s)")
        cpu_base, parity = None, None
        if not args.no_cpu_baseline:
            y_t = ctx["y"]
            cpu_base, parity = cpu_baseline_and_check(api, synth, w, ctx["irp"], ctx["x_host"],
                                                      lambda a, b: y_t[a:b].cpu().numpy(), args.cpu_sample_nnz)
            if not parity["ok"]:
                raise SystemExit(f"PARITY FAILURE against the CPU checker: {parity}")
            log("cpu_baseline", cpu_base, "parity", parity)
        ctx["dm"].free()
        del ctx
        torch.cuda.empty_cache()
        if not args.no_extra:
            for key in ("c3", "c3b", "c2"):
                if key == args.workload:
                    continue
                we = synth.WORKLOADS[key]
                if args.scale != 1.0:
                    we = synth.scaled(we, args.scale)
                # config 2 is quoted on a thread-per-row kernel: pick among the one-lane-per-row launchers only
                cands = THREAD_PER_ROW if key.startswith("c2") else None
                r, c = measure_single(api, synth, torch, we, args.launcher, steps, warmup, cands)
                launcher = r["launcher"]
                c["dm"].free()
                del c
                torch.cuda.empty_cache()
                extra[key] = r
                log(f"{we.name} [{launcher}]: {r['gflops']:.1f} GFLOP/s  kernel {r['kernel_ms_avg']:.3f} ms  "
                    f"{100 * r['hbm_frac']:.1f}% of 8 TB/s")
        line = {
            "metric": "spmv_gflops", "value": res["gflops"], "unit": "GFLOP/s", "n_gpus": 1, "steps": steps,
            "warmup": warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": res["workload"]["workload"], **{k: v for k, v in res["workload"].items() if k != "workload"},
                       "kernel": res["launcher"], "auto_candidates_ms": res["auto_candidates_ms"],
                       "parallelism": "1 GPU"},
            "hbm_gbps": res["hbm_gbps"], "hbm_roofline_frac": res["hbm_frac"],
            "roofline": {"bound": "hbm", "achieved": res["hbm_gbps"], "peak": HBM_PEAK * 1e-9, "unit": "GB/s",
                         "frac": res["hbm_frac"], "traffic": pmc_traffic(res["workload"]["workload"], res["launcher"]),
                         "kernel": KERNEL_OF.get(res["launcher"], res["launcher"]),
                         "algorithmic_bytes_per_launch": res["algorithmic_bytes"],
                         "kernel_ms_avg": res["kernel_ms_avg"]},
        }
        if cpu_base:
            line["cpu_baseline"] = cpu_base
            line["parity"] = parity
        if "c3" in extra:
            line["headline_c3"] = extra["c3"]
        for k in ("c3b", "c2"):
            if k in extra:
                line["extra_" + k] = extra[k]
        emit(line)
        api.spmvHipFinalize()
        return

    # ------------------------------------------------------------------ N > 1
    from spmv_openmp_cuda_amd import sharding
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    info = synth.describe(w, lens)
    del lens
    nnz_total = int(irp[-1])
    x_host = synth.make_x(w.N, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    cfg = api.CONFIG()

    group_cache = {}

    def group_setup(groups, rdiv=1, taper=False):
        """(plan, this rank's device matrices) for `groups` row groups per rank; built once, freed at the end.
        rdiv > 1 / taper: separate copies whose two-phase format has bins of 1/rdiv the automatic height, or tapered
        bins (one round of quarter-height bins first and last) -- more, lower rounds of bins in phase 2, so that the
        exchange of y can start earlier and has less left to send when the kernel ends"""
        key = (groups, rdiv, taper)
        if key not in group_cache:
            plan_g = sharding.make_plan(irp, world, groups)
            dms = [synth.device_csr(w, irp, *plan_g.block(rank, g)) for g in range(groups)]
            if rdiv > 1 or taper:
                _, auto_dms = group_setup(groups)
                for dm, ref in zip(dms, auto_dms):
                    nb, rpb = C.c_uint(), C.c_uint()
                    ok = dm.nnz > 0 and api.lib.spmvHipTilesShape(C.byref(ref.handle), C.byref(nb), C.byref(rpb)) == 0
                    if ok:
                        if rdiv > 1:
                            api.lib.spmvHipSetTilesRowsPerBin(max(64, (rpb.value // rdiv + 63) // 64 * 64))
                        api.lib.spmvHipSetTilesTaper(1 if taper else 0)
                        api.lib.spmvHipBuildTiles(C.byref(dm.handle))
                        api.lib.spmvHipSetTilesRowsPerBin(0)
                        api.lib.spmvHipSetTilesTaper(0)
            group_cache[key] = (plan_g, dms)
        return group_cache[key]

    class RcclExchange:
        """this rank's `groups` row groups as device matrices + gather buffers; kernel(g) -> async RCCL all-gather(g)"""
        def __init__(self, groups):
            self.groups = self.events = groups
            self.name = f"rccl-g{groups}"
            self.plan, self.dms = group_setup(groups)
            self.bufs = sharding.GatherBuffers(self.plan, rank, torch, "cuda")
            self.y = self.bufs.y
            self.desc = (f"{world} ranks x {groups} nnz-balanced row groups; per group: kernel then async RCCL "
                         f"all-gather(y) overlapping the next group" + ("" if self.plan.equal_blocks else "; padded blocks + compaction"))

        def poison(self):
            for b in self.bufs.ypad:
                b.fill_(float("nan"))
            self.bufs.y.fill_(float("nan"))

        def step(self, fn, ev=None):
            def compute_group(g, slot):
                if ev:
                    api.lib.spmvHipEventRecord(ev[g][0])
                rc = fn(C.byref(self.dms[g].handle), x.data_ptr(), cfg, slot.data_ptr())
                if ev:
                    api.lib.spmvHipEventRecord(ev[g][1])
                if rc:
                    raise RuntimeError("launcher failed")
            return sharding.step(self.plan, dist, self.bufs, compute_group)

        def free(self):
            self.dms, self.bufs, self.y = [], None, None

    class PushExchange:
        """y lives in a peer window; this rank's rows are delivered to the other ranks' windows by copy-engine pushes
        behind each piece of y ("push-pQ"), by stores fused into phase 2 of the two-phase kernel ("fused") or by a push
        kernel beside phase 2 ("pushk"); with G > 1 row groups per rank ("-gG") the rows of one group travel while the
        next group is computed"""
        def __init__(self, px, mode, pieces, groups=1, rdiv=1, taper=False):
            self.groups = self.events = groups
            self.plan, dms = group_setup(groups, rdiv, taper)
            self.runs = [sharding.PushSpMV(api, px, dms[g], self.plan.block(rank, g)[0], launcher, x.data_ptr(), mode, pieces)
                         for g in range(groups)]
            self.px = px
            self.name = (mode if mode in ("fused", "pushk") else f"push-p{self.runs[0].pieces}") + (f"-g{groups}" if groups > 1 else "") + \
                (f"-r{rdiv}" if rdiv > 1 else "") + ("-t" if taper else "")
            self.y = px.y
            self.desc = (f"{world} ranks x {groups} nnz-balanced row group(s), y in peer windows (device IPC over xGMI): " +
                         ("phase 2 stores every finished bin of y to all ranks itself" if mode == "fused" else
                          "a push kernel beside phase 2 copies every bin of y to all ranks as soon as it is flagged" if mode == "pushk" else
                          f"{self.runs[0].pieces} piece(s) of y, each pushed to all ranks by the copy engines while the next is reduced") +
                         (f"; bins of 1/{rdiv} the automatic height" if rdiv > 1 else "") +
                         ("; tapered bins (a round of quarter-height bins first and last)" if taper else "") +
                         "; step ends with a 4-byte RCCL all-reduce as barrier")

        def poison(self):
            self.y.fill_(float("nan"))

        def step(self, fn, ev=None):
            for g, run in enumerate(self.runs):
                run.enqueue(ev[g] if ev else None)
            self.px.finish()
            return self.y

        def free(self):
            self.runs, self.y = [], None

    # every rank must run the same kernel: rank 0 decides (on its block of the 1-group plan)
    base_plan, (base_dm,) = group_setup(1)
    first = RcclExchange(1)
    launcher, tried = pick_launcher(api, torch, base_dm, x.data_ptr(), first.bufs.slot[0].data_ptr(), args.launcher)
    choice = torch.tensor([AUTO_CANDIDATES.index(launcher) if launcher in AUTO_CANDIDATES else -1], device="cuda")
    dist.broadcast(choice, 0)
    if int(choice) >= 0:
        launcher = AUTO_CANDIDATES[int(choice)]
    fn = api.SPMV_LAUNCHERS[launcher]

    # How y is exchanged is MEASURED, not assumed (no multi-GPU node at development time): RCCL all-gather with 1/2[/4]
    # row groups per rank (more groups overlap more of the gather but make every group's kernels less efficient), the
    # push exchange over peer windows with 1..8 pieces, and the fused store.  Whole steps, slowest rank, 3 steps each;
    # a candidate whose y differs from the first candidate's is dropped.
    px = None
    if args.exchange in ("auto", "auto-no-rccl", "push", "fused"):
        px = sharding.PeerExchange(api, dist, torch, rank, world, local, w.N)
        if not px.ok:
            log("peer windows unavailable, RCCL only:", px.why)
            px = None
    def candidates():
        if args.exchange in ("auto", "rccl"):
            for G in ([args.groups] if args.groups > 0 else ([1, 2, 4] if world == 2 else [1, 2])):
                yield (lambda G=G: first if G == 1 else RcclExchange(G))
        if px is not None and args.exchange in ("auto", "auto-no-rccl", "push"):
            for q in ([args.pieces] if args.pieces > 0 else ([1, 2, 4, 8] if launcher == "hipSpMVTilesCSR" else [1])):
                yield (lambda q=q: PushExchange(px, "push", q))
        if px is not None and args.exchange in ("auto", "auto-no-rccl", "push") and world > 1 and args.pieces <= 0:
            # copy engines + row groups: the rows of one group travel while the next group is computed.  On the PCIe
            # stand-in (scripts/slowlink_groups.py) ONLY the copy engines overlap a transfer with the next group's
            # kernels; stores issued by CUs (push kernel, fused store) to a slow destination hold those kernels up.
            for q, G in ((1, 2), (1, 4), (2, 2)) + (((1, 8),) if world == 2 else ()):   # N = 2: one link carries half of y
                yield (lambda q=q, G=G: PushExchange(px, "push", q, G))
            if launcher == "hipSpMVTilesCSR":
                # tapered bins + copy engines, one group: the low first round leaves early, the low last round is the
                # only part of y still to be sent when phase 2 ends, and phase 2 itself never waits for a link
                yield (lambda: PushExchange(px, "push", 4, 1, 1, True))
                yield (lambda: PushExchange(px, "push", 4, 2, 1, True))
        # the fused store is tried only after a copy-engine push through the same mappings delivered a correct y
        if px is not None and args.exchange in ("auto", "auto-no-rccl", "fused") and launcher == "hipSpMVTilesCSR" and \
                (args.exchange == "fused" or any(k.startswith("push") for k in exchange_ms)):
            yield (lambda: PushExchange(px, "fused", 1))
            if world > 1:
                yield (lambda: PushExchange(px, "pushk", 1))
                # two row groups per rank: the rows of the first travel under the kernels of the second
                yield (lambda: PushExchange(px, "fused", 1, 2))
                yield (lambda: PushExchange(px, "pushk", 1, 2))
                # smaller bins = more rounds of bins in phase 2: rows start to travel earlier (scripts/slowlink_probe.py)
                yield (lambda: PushExchange(px, "fused", 1, 1, 2))
                yield (lambda: PushExchange(px, "fused", 1, 1, 4))
                yield (lambda: PushExchange(px, "fused", 1, 2, 2))
                yield (lambda: PushExchange(px, "fused", 1, 1, 1, True))
                yield (lambda: PushExchange(px, "fused", 1, 2, 1, True))
    exchange_ms, rejected = {}, {}
    best, ref_sum = None, None
    for k_cand, make in enumerate(candidates()):
        try:                                    # a candidate that cannot be set up on SOME rank is dropped on ALL of them
            cand = make()
        except Exception as e:                  # noqa: BLE001 -- e.g. out of memory for one more copy of the format
            cand = None
            log(f"exchange candidate #{k_cand} could not be set up on rank {rank}: {e}")
        built = torch.tensor([1.0 if cand is not None else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(built, op=dist.ReduceOp.MIN)
        if float(built[0]) == 0.0:
            rejected[f"candidate#{k_cand}"] = "set-up failed on some rank"
            if cand is not None and cand is not first:
                cand.free()
            continue
        if cand.name in exchange_ms or cand.name in rejected:      # e.g. fewer pieces than asked for
            continue
        cand.poison()
        torch.cuda.synchronize()
        dist.barrier()                      # peers write into this rank's y: nobody steps before everybody has poisoned
        cand.step(fn)
        torch.cuda.synchronize()
        y_c = cand.y
        sums = torch.stack([y_c.abs().sum(), torch.isnan(y_c).sum().to(torch.float64)])
        if ref_sum is None:
            ref_sum = sums[0].clone()
        good = torch.tensor([1.0 if (float(sums[1]) == 0 and abs(float(sums[0] - ref_sum)) <= 1e-9 * float(ref_sum)) else 0.0],
                            dtype=torch.float64, device="cuda")
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        if float(good[0]) == 0.0:
            rejected[cand.name] = "y incomplete or different on some rank"
            if cand is not first:
                cand.free()
            continue
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            cand.step(fn)
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exchange_ms[cand.name] = float(tt[0])
        if best is None or exchange_ms[cand.name] < exchange_ms[best.name]:
            if best is not None and best is not first:
                best.free()
            best = cand
        elif cand is not first:
            cand.free()
        torch.cuda.empty_cache()
    if best is None:
        raise SystemExit(f"no exchange candidate produced a complete y: {rejected}")

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
    if not args.rehearse_shared_gpu and first.bufs is not None:
        exchange_alone_ms["rccl_all_gather"] = alone(lambda: dist.all_gather_into_tensor(first.bufs.ypad[0], first.bufs.slot[0]))
    if px is not None:
        r0_, r1_ = base_plan.rows(rank)
        def push_once():
            px.push(r0_, r1_)
            px.finish()
        exchange_alone_ms["push_copy_engines"] = alone(push_once)
        exchange_alone_ms["barrier_only"] = alone(lambda: px.finish())
    log("exchange alone (ms, slowest rank):", exchange_alone_ms)
    setup = best
    log("exchange candidates (ms/step, slowest rank):", exchange_ms, "rejected:", rejected, "->", setup.name)
    groups, plan = setup.events, setup.plan
    r0, r1 = plan.rows(rank)
    nnz_local = int(irp[r1] - irp[r0])
    evs = [[(C.c_void_p(), C.c_void_p()) for _ in range(groups)] for _ in range(steps)]
    for per_step in evs:
        for a, b in per_step:
            api.lib.spmvHipEventCreate(C.byref(a))
            api.lib.spmvHipEventCreate(C.byref(b))

    def step(ev):
        setup.step(fn, ev)

    setup.poison()
    torch.cuda.synchronize()
    dist.barrier()                          # as above: the warm-up steps of a fast rank must not land before a slow rank's poison
    wall = time_kernel_loop(api, torch, dist, world, step, steps, warmup, evs)
    kms = [sum(kernel_ms(api, per_step)) for per_step in evs]
    y = setup.y
    t = torch.tensor([wall, sum(kms) / len(kms)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, kernel_ms_max = float(t[0]), float(t[1])
    # cross-rank consistency: every rank must hold the same full y, with no NaN left
    chk = torch.stack([y.sum(), y.abs().sum(), torch.isnan(y).sum().to(torch.float64)])
    chk_all = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(chk_all, chk)
    same = all(torch.equal(c, chk_all[0]) for c in chk_all)
    if rank == 0:
        spot = oracle_spot_checks(synth, w, irp, x_host, lambda a, b: y[a:b].cpu().numpy())
        bytes_alg_local = synth.algorithmic_bytes_csr(nnz_local, r1 - r0, w.N)
        bytes_alg_total = synth.algorithmic_bytes_csr(nnz_total, w.N, w.N)
        k_avg = kernel_ms_max * 1e-3
        line = {
            "metric": "spmv_gflops", "value": 2.0 * nnz_total / (wall_max / steps) * 1e-9, "unit": "GFLOP/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": wall_max / steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": info["workload"], **{k: v for k, v in info.items() if k != "workload"},
                       "kernel": launcher, "auto_candidates_ms": tried, "exchange": setup.name,
                       "exchange_step_ms": exchange_ms, "exchange_rejected": rejected,
                       "exchange_alone_ms": exchange_alone_ms, "parallelism": setup.desc},
            "hbm_gbps": bytes_alg_total / (wall_max / steps) * 1e-9,
            "hbm_roofline_frac": bytes_alg_total / (wall_max / steps) / (HBM_PEAK * world),
            "roofline": {"bound": "hbm", "achieved": bytes_alg_local / k_avg * 1e-9, "peak": HBM_PEAK * 1e-9,
                         "unit": "GB/s", "frac": bytes_alg_local / k_avg / HBM_PEAK, "traffic": None,
                         "kernel": KERNEL_OF.get(launcher, launcher),
                         "algorithmic_bytes_per_launch": bytes_alg_local,
                         "kernel_ms_avg": kernel_ms_max, "note": "per-rank kernels (all row groups), slowest rank"},
            "exposed_gather_ms_per_step": wall_max / steps * 1e3 - kernel_ms_max,
            "parity": {"all_ranks_hold_identical_y": bool(same), "nan_left": float(chk_all[0][2]), "spot_checks": spot,
                       "gate_abs": 7e-4, "ok": bool(same and float(chk_all[0][2]) == 0 and
                                                    all(not c["nan"] and c["max_abs_diff"] <= 7e-4 for c in spot))},
        }
        emit(line)
    dist.barrier()
    if setup is not first:
        setup.free()
    first.free()
    for _, dms_g in group_cache.values():
        for dm in dms_g:
            dm.free()
    if px is not None:
        px.close()
    api.spmvHipFinalize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
