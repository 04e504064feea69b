#!/usr/bin/env python3
"""Kernel time (HIP events) of several launchers on one synthetic workload, same matrix, same x:
    python3 scripts/time_launchers.py c3 hipSpMVTilesCSR hipSpMVStripesCSR [--steps 20] [--check]
A launcher may also be named the reference's way (CUDA_CSR_ROWS, CUDA_CSR_ROWS_WARP: what the CLI and the function
tables resolve those names to, default variants) or carry a suffix: hipSpMVWarpPerRowCSR:1 (kernel variant),
hipSpMVStripesCSR:det / hipSpMVTilesCSR:det (the deterministic form of the format is built first).
--check compares head / middle / tail 200 k-row windows of every launcher's y with the serial oracle."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("workload")
    p.add_argument("launchers", nargs="+")
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--check", action="store_true")
    a = p.parse_args()
    import numpy as np
    import torch
    from spmv_openmp_cuda_amd import api, synth
    import bench
    torch.cuda.set_device(0)
    api.spmvHipInit(0)
    api.lib.spmvHipSetStream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    api.lib.spmvHipSetSync(0)
    w = synth.WORKLOADS[a.workload]
    if a.scale != 1.0:
        w = synth.scaled(w, a.scale)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x_host = synth.make_x(w.N, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    y = torch.empty(w.N, dtype=torch.float64, device="cuda")
    nnz = int(irp[-1])
    alg = synth.algorithmic_bytes_csr(nnz, w.N, w.N)
    cfg = api.CONFIG()
    windows = None
    modes = {"CUDA_CSR_ROWS": "hipSpMVRowsCSR", "CUDA_CSR_ROWS_WARP": "hipSpMVWarpPerRowCSR"}
    for spec in a.launchers:
        name, _, suffix = modes.get(spec, spec).partition(":")
        fn = api.SPMV_LAUNCHERS[name]
        api.set_variant("hipSpMVRowsCSR", 2)                 # the library defaults
        api.set_variant("hipSpMVWarpPerRowCSR", 2)
        if suffix.isdigit():
            api.set_variant(name, int(suffix))
        elif suffix in ("det", "det2") and name == "hipSpMVStripesCSR":
            api.build_stripes(dm, deterministic=2 if suffix == "det2" else 1)
        elif suffix == "det" and name == "hipSpMVTilesCSR":
            api.build_tiles(dm, deterministic=True)
        elif suffix == "" and name == "hipSpMVStripesCSR" and api.stripes_info(dm).nBins and api.stripes_info(dm).deterministic:
            api.build_stripes(dm)
        elif suffix == "" and name == "hipSpMVTilesCSR" and api.tiles_info(dm).nBins and getattr(api.tiles_info(dm), "deterministic", 0):
            api.build_tiles(dm)
        y.fill_(float("nan"))
        if fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr()):
            print(f"{w.name} {spec}: FAILED")
            continue
        torch.cuda.synchronize()
        extra = ""
        if name == "hipSpMVStripesCSR":
            i = api.stripes_info(dm)
            extra = f" bins={i.nBins} rows/bin<={i.rowsPerBin} wide={i.wide} det={i.deterministic} build={i.buildMs:.1f}ms"
        if name in ("hipSpMVWarpPerRowCSR", "hipSpMVAutoCSR", "hipSpMVRowsCSR"):
            ms4 = (C.c_double * 4)()
            pick = (api.lib.spmvHipAutoChoiceRows if name == "hipSpMVRowsCSR" else api.lib.spmvHipAutoChoice)(C.byref(dm.handle), ms4)
            extra = f" pick={pick.decode() if pick else None} candidates_ms={[round(v, 4) for v in ms4]}"
        if a.check:
            if windows is None:
                windows = bench.OracleWindows(synth, w, irp, x_host, lens)
            par = windows.check(lambda r0, r1: y[r0:r1].cpu().numpy())
            extra += " check=" + ("ok" if par["ok"] and par["max_diff_over_sum_abs_ax"] <= 1e-12 else "FAIL " + str(par))
        for _ in range(3):
            fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr())
        evs = [(C.c_void_p(), C.c_void_p()) for _ in range(a.steps)]
        for e0, e1 in evs:
            api.lib.spmvHipEventCreate(C.byref(e0))
            api.lib.spmvHipEventCreate(C.byref(e1))
        torch.cuda.synchronize()
        for e0, e1 in evs:
            api.lib.spmvHipEventRecord(e0)
            fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr())
            api.lib.spmvHipEventRecord(e1)
        torch.cuda.synchronize()
        ms = bench.kernel_ms(api, evs)
        for e0, e1 in evs:
            api.lib.spmvHipEventDestroy(e0)
            api.lib.spmvHipEventDestroy(e1)
        avg = sum(ms) / len(ms)
        if os.environ.get("SPMV_PRINT_STEPS"):
            print("steps ms:", " ".join(f"{t:.3f}" for t in ms))
        print(f"{w.name} {spec}: avg {avg:.4f} ms  min {min(ms):.4f}  max {max(ms):.4f}  = {alg / avg / 8e9 * 100:.1f}% of 8 TB/s "
              f"({2 * nnz / avg * 1e-6:.0f} GFLOP/s){extra}", flush=True)
    dm.free()
    api.spmvHipFinalize()


if __name__ == "__main__":
    main()
