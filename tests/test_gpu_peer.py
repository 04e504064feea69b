"""Push exchange over peer windows (include/spmvHip.h "peer windows", sharding.PeerExchange / PushSpMV) on the
one-GPU box: `world` processes share GPU 0, map each other's y windows over device IPC and deliver their rows
with (a) copy-engine pushes behind every piece of y and (b) the stores fused into hipSpMVTilesReduce.  The
control plane is gloo here (RCCL refuses two ranks on one device); on the 8-GPU node bench.py runs the same
classes with RCCL as the control plane and xGMI under the copies.  Every rank must end with the full y:
bit-identical on all ranks, equal to the serial oracle (bitwise for the serial-order launcher)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import ctypes as C
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import Oracle
    from spmv_openmp_cuda_amd import api, sharding, synth
    api.spmvHipInit(0)
    api.lib.spmvHipSetStream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    api.lib.spmvHipSetSync(0)
    oracle = Oracle()
    w = synth.Workload("p", 300_007, 6_000_000, "powerlaw", 20000, 9)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    x_host = synth.make_x(w.N, w.cfg)
    x = torch.from_numpy(x_host).cuda()
    ja_all, as_all = oracle.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
    y_ref = oracle.csr_serial_dev(irp.astype(np.uint32), ja_all, as_all, x_host)
    scale = np.add.reduceat(np.abs(as_all * x_host[ja_all]), irp[:-1].astype(np.int64))

    def host_barrier():
        torch.cuda.synchronize()
        dist.barrier()

    plan = sharding.make_plan(irp, world, 1)
    r0, r1 = plan.rows(rank)
    dm = synth.device_csr(w, irp, r0, r1)
    px = sharding.PeerExchange(api, dist, torch, rank, world, 0, w.N)
    verdict = []
    ok = px.ok
    if not ok:
        verdict.append("setup: " + px.why)
    cases = [("hipSpMVRowsCSR", "push", 1, True), ("hipSpMVWarpPerRowCSR", "push", 1, False),
             ("hipSpMVTilesCSR", "push", 1, False), ("hipSpMVTilesCSR", "push", 3, False),
             ("hipSpMVTilesCSR", "fused", 1, False), ("hipSpMVTilesCSR", "pushk", 1, False)]
    for launcher, mode, pieces, exact in (cases if ok else []):
        run = sharding.PushSpMV(api, px, dm, r0, launcher, x.data_ptr(), mode, pieces, barrier=host_barrier, unit=4)
        px.y.fill_(float("nan"))
        host_barrier()
        for _ in range(6 if mode == "pushk" else 2):     # several steps: windows and flags are re-used, consumer L1 warm
            y = run.step()
        if mode == "pushk":
            good_push = api.lib.spmvHipTilesPushFailed(C.byref(dm.handle)) == 0
        else:
            good_push = True
        host_barrier()
        yh = y.cpu().numpy()
        good = good_push and not np.isnan(yh).any()
        good = good and (np.array_equal(yh, y_ref) if exact else bool(np.all(np.abs(yh - y_ref) <= 1e-13 * scale + 1e-300)))
        sums = [None] * world
        dist.all_gather_object(sums, yh.tobytes() if w.N < 1_000_000 else float(yh.sum()))
        good = good and all(s == sums[0] for s in sums)  # every rank holds the same bytes
        if not good:
            verdict.append(f"{launcher}/{mode}/{pieces}: mismatch (max |dy| {np.nanmax(np.abs(yh - y_ref)):.3e}, "
                           f"nan {int(np.isnan(yh).sum())})")
        ok = ok and good
        host_barrier()
    dm.free()
    px.close()
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "FAILED: " + "; ".join(verdict))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_push_exchange_shared_gpu(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def test_tapered_bins_single_process():
    """tapered format (low bins in the first and last round, spmvTilesOpts.taper): same y, consistent bin -> row map"""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    from spmv_openmp_cuda_amd import api, synth
    api.spmvHipInit(0)
    api.lib.spmvHipSetStream(None)
    api.lib.spmvHipSetSync(1)
    oracle = Oracle()
    w = synth.Workload("p", 2_400_011, 12_000_000, "powerlaw", 20000, 9)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    x_host = synth.make_x(w.N, w.cfg)
    ja, as_ = oracle.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
    y_ref = oracle.csr_serial_dev(irp.astype(np.uint32), ja, as_, x_host)
    scale = np.add.reduceat(np.abs(as_ * x_host[ja]), irp[:-1].astype(np.int64))
    dx = api.DeviceVector(w.N).up(x_host)
    dm = synth.device_csr(w, irp, 0, w.N)
    api.build_tiles(dm, taper=True)
    nb, rpb = C.c_uint(), C.c_uint()
    assert api.lib.spmvHipTilesShape(C.byref(dm.handle), C.byref(nb), C.byref(rpb)) == 0
    info = api.tiles_info(dm)
    assert info.taper == 1 and info.nBins == nb.value and info.buildMs > 0
    rows = []
    for b in range(nb.value + 1):
        r = C.c_ulong()
        assert api.lib.spmvHipTilesBinRow(C.byref(dm.handle), b, C.byref(r)) == 0
        rows.append(int(r.value))
    h = np.diff(rows)
    assert rows[0] == 0 and rows[-1] == w.N and (h > 0).all() and h.max() <= rpb.value
    assert h[0] < h[len(h) // 2] and h[-2] < h[len(h) // 2]            # low bins at both ends, high bins between
    assert (h[:256] == h[0]).all()                                     # one full round of low bins first
    dy = api.DeviceVector(w.N)
    dy.poison()
    api.spmv("hipSpMVTilesCSR", dm, dx, dy)
    y = dy.down()
    assert not np.isnan(y).any() and np.all(np.abs(y - y_ref) <= 1e-13 * scale + 1e-300)
    # the split launches on a tapered format: bin ranges in any order, fused store and push kernel
    scratch = api.DeviceVector(w.N)
    extra = (C.c_void_p * 1)(scratch.ptr.value)
    from spmv_openmp_cuda_amd.sharding import bin_ranges
    for mode in ("ranges", "fused", "pushk"):
        dy.poison(); scratch.poison()
        assert api.lib.hipSpMVTilesExpand(C.byref(dm.handle), dx.ptr) == 0
        if mode == "ranges":
            for b0, b1 in reversed(bin_ranges(nb.value, 3, 256)):
                assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), b0, b1, dy.ptr, 0, None) == 0
        elif mode == "fused":
            assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), 0, nb.value, dy.ptr, 1, extra) == 0
        else:
            assert api.lib.hipSpMVTilesReducePush(C.byref(dm.handle), dy.ptr, 1, extra) == 0
            assert api.lib.spmvHipTilesPushFailed(C.byref(dm.handle)) == 0
        y2 = dy.down()
        assert not np.isnan(y2).any() and np.all(np.abs(y2 - y_ref) <= 1e-13 * scale + 1e-300), mode
        if mode != "ranges":
            assert np.array_equal(scratch.down(), y2), mode
    dm.free()


def test_tiles_reduce_bin_ranges_single_process():
    """phase 2 cut into bin ranges == the one-launch result (same kernel, same order inside a bin up to atomics)"""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    from spmv_openmp_cuda_amd import api, synth
    api.spmvHipInit(0)
    api.lib.spmvHipSetStream(None)
    api.lib.spmvHipSetSync(1)
    oracle = Oracle()
    w = synth.Workload("p", 200_003, 4_000_000, "powerlaw", 20000, 9)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    x_host = synth.make_x(w.N, w.cfg)
    dm = synth.device_csr(w, irp, 0, w.N)
    dx = api.DeviceVector(w.N).up(x_host)
    dy = api.DeviceVector(w.N)
    dy.poison()
    nb, rpb = C.c_uint(), C.c_uint()
    assert api.lib.spmvHipTilesShape(C.byref(dm.handle), C.byref(nb), C.byref(rpb)) == 0
    assert nb.value * rpb.value >= w.N > (nb.value - 1) * rpb.value
    assert api.lib.hipSpMVTilesExpand(C.byref(dm.handle), dx.ptr) == 0
    from spmv_openmp_cuda_amd.sharding import bin_ranges
    for b0, b1 in reversed(bin_ranges(nb.value, 5)):     # any order of the ranges
        assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), b0, b1, dy.ptr, 0, None) == 0
    y = dy.down()
    ja, as_ = oracle.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
    y_ref = oracle.csr_serial_dev(irp.astype(np.uint32), ja, as_, x_host)
    scale = np.add.reduceat(np.abs(as_ * x_host[ja]), irp[:-1].astype(np.int64))
    assert not np.isnan(y).any()
    assert np.all(np.abs(y - y_ref) <= 1e-13 * scale + 1e-300)
    # push kernel and fused store with an ODD first row (8-byte aligned vectors: the 16-byte copy peels one row) and
    # several destinations; every destination must equal y bit for bit
    big = [api.DeviceVector(w.N + 1) for _ in range(4)]
    for mode in ("fused", "pushk"):
        for v in big:
            v.poison()
        ptrs = [C.c_void_p(v.ptr.value + 8) for v in big]
        extra = (C.c_void_p * 3)(*[p.value for p in ptrs[1:]])
        assert api.lib.hipSpMVTilesExpand(C.byref(dm.handle), dx.ptr) == 0
        if mode == "fused":
            assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), 0, nb.value, ptrs[0], 3, extra) == 0
        else:
            assert api.lib.hipSpMVTilesReducePush(C.byref(dm.handle), ptrs[0], 3, extra) == 0
            assert api.lib.spmvHipTilesPushFailed(C.byref(dm.handle)) == 0
        got = [v.down()[1:] for v in big]
        assert not np.isnan(got[0]).any() and np.all(np.abs(got[0] - y_ref) <= 1e-13 * scale + 1e-300)
        for g in got[1:]:
            assert np.array_equal(g, got[0]), mode
    # the same two exchange forms on the DETERMINISTIC form of the format: every destination is the serial oracle's bits
    api.build_tiles(dm, deterministic=True)
    nbd, rpbd = C.c_uint(), C.c_uint()
    assert api.lib.spmvHipTilesShape(C.byref(dm.handle), C.byref(nbd), C.byref(rpbd)) == 0 and rpbd.value % 4 == 0
    for mode in ("fused", "pushk"):
        for v in big:
            v.poison()
        ptrs = [C.c_void_p(v.ptr.value + 8) for v in big]
        extra = (C.c_void_p * 3)(*[p.value for p in ptrs[1:]])
        assert api.lib.hipSpMVTilesExpand(C.byref(dm.handle), dx.ptr) == 0
        if mode == "fused":
            assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), 0, nbd.value, ptrs[0], 3, extra) == 0
        else:
            assert api.lib.hipSpMVTilesReducePush(C.byref(dm.handle), ptrs[0], 3, extra) == 0
            assert api.lib.spmvHipTilesPushFailed(C.byref(dm.handle)) == 0
        for v in big:
            assert np.array_equal(v.down()[1:], y_ref), mode
    api.build_tiles(dm)                                       # back to the arrival-order form for the checks below
    assert api.lib.hipSpMVTilesReducePush(C.byref(dm.handle), dy.ptr, 0, None) == 1        # needs a destination
    # invalid ranges fail loudly
    assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), 2, 1, dy.ptr, 0, None) == 1
    assert api.lib.hipSpMVTilesReduce(C.byref(dm.handle), 0, nb.value + 1, dy.ptr, 0, None) == 1
    dm.free()


@pytest.mark.parametrize("world,launcher,extra", [(2, "hipSpMVTilesCSR", True), (3, "hipSpMVWarpPerRowCSR", False), (2, "hipSpMVTilesCSR", False)])
def test_bench_multirank_path_on_shared_gpu(world, launcher, extra):
    """bench.py's N > 1 code end to end (plan, kernel pick + broadcast, exchange search through peer windows, cross-rank
    byte comparison, oracle windows on every rank, the serial-order validating step, the ONE JSON line) with `world`
    ranks sharing the GPU; timings mean nothing.  Default list: at most eight candidates, safest first; --exchange-extra
    adds the other families."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rehearse-shared-gpu",
           "--workload", "c3", "--scale", "0.05", "--steps", "3", "--warmup", "1", "--launcher", launcher] + (["--exchange-extra"] if extra else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line on stdout
    j = json.loads(lines[0])
    c, p = j["config"], j["parity"]
    assert j["n_gpus"] == world and p["ok"] and p["all_ranks_hold_identical_y"] and p["every_rank_matches_oracle_windows"]
    assert p["serial_order_step"]["all_ranks_hold_identical_bytes"] and p["serial_order_step"]["bit_identical_to_serial_oracle_on_every_rank"]
    assert c["exchange"] in c["exchange_step_ms"] and not c["exchange_rejected"] and c["exchange_search_s"] > 0
    assert any(k.startswith("push") for k in c["exchange_step_ms"])
    if not extra:
        assert len(c["exchange_step_ms"]) + len(c["exchange_skipped"]) <= 8
    if launcher == "hipSpMVTilesCSR":
        assert "fused" in c["exchange_step_ms"] and "fused-g2" in c["exchange_step_ms"] and "push-p4" in c["exchange_step_ms"]
    if extra:                                             # every exchange family ran and delivered a complete y
        for fam in ("pushk", "pushk-g2", "fused-r2", "fused-t", "push-p1-g2", "push-p4-t"):
            assert fam in c["exchange_step_ms"], fam
