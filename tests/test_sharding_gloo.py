"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups run the same
sharding code bench.py uses on RCCL (spmv_openmp_cuda_amd/sharding.py): nnz-balanced
row blocks, padded in-place all-gather of y, compaction.  The per-rank SpMV is
done by the oracle here (no GPU in this container); the gathered y must equal
the serial oracle's y on the whole matrix BITWISE on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, equal_rows, groups, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import Oracle
    from spmv_openmp_cuda_amd import sharding, synth
    oracle = Oracle()
    if equal_rows:
        w = synth.Workload("u", 6000, 6000 * 8, "uniform", cfg=8)
    else:
        w = synth.Workload("p", 5003, 90000, "powerlaw", 3000, 9)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    plan = sharding.make_plan(irp, world, groups)
    x = synth.make_x(w.N, w.cfg)
    bufs = sharding.GatherBuffers(plan, rank, torch, "cpu")

    def compute_group(g, slot):
        # this rank's rows of group g, generated independently with their row offset (as on the GPUs)
        b0, b1 = plan.block(rank, g)
        ja, as_ = oracle.synth_fill(w.N, b0, irp[b0:b1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
        irp_local = (irp[b0:b1 + 1] - irp[b0]).astype(np.uint32)
        slot[: b1 - b0].copy_(torch.from_numpy(oracle.csr_serial_dev(irp_local, ja, as_, x)))

    for _ in range(2):                                   # two steps: buffers are re-used
        y = sharding.step(plan, dist, bufs, compute_group)
    # reference: the whole matrix on one "device"
    ja_all, as_all = oracle.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
    y_ref = oracle.csr_serial_dev(irp.astype(np.uint32), ja_all, as_all, x)
    ok = np.array_equal(y.numpy(), y_ref) and not np.isnan(y.numpy()).any()
    ok = ok and plan.equal_blocks == bool(equal_rows and groups == 1 and w.N % world == 0)
    r0, r1 = plan.rows(rank)
    ok = ok and (r0, r1) == (plan.block(rank, 0)[0], plan.block(rank, groups - 1)[1])
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,equal_rows,groups", [(2, False, 1), (2, True, 1), (3, False, 1), (2, False, 4), (3, True, 2)])
def test_sharded_spmv_gloo(tmp_path, world, equal_rows, groups):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, equal_rows, groups, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def test_bin_ranges_cover_and_align():
    """phase-2 bin ranges of the push exchange: consecutive, complete, multiples of the unit except the last"""
    sys.path.insert(0, ROOT)
    from spmv_openmp_cuda_amd.sharding import bin_ranges
    for n, pieces, unit in [(610, 3, 256), (610, 8, 256), (1221, 4, 256), (19, 5, 1), (19, 3, 4), (5, 1, 256), (1, 8, 256),
                            (4880, 2, 256), (256, 4, 256)]:
        r = bin_ranges(n, pieces, unit)
        assert r[0][0] == 0 and r[-1][1] == n and len(r) <= pieces
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all((b - a) % unit == 0 for a, b in r[:-1]) and all(b > a for a, b in r)
    assert bin_ranges(610, 3, 256) == [(0, 256), (256, 512), (512, 610)]
