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


class _CpuExchange:
    """a candidate of search_exchange on CPU: the oracle computes this rank's rows, gloo gathers them"""

    def __init__(self, key, plan, rank, irp, w, x, oracle, sharding, corrupt=False, fail_from_step=0):
        import torch as T
        self.key, self.plan, self.rank, self.irp, self.w, self.x, self.oracle, self.sharding = key, plan, rank, irp, w, x, oracle, sharding
        self.bufs = sharding.GatherBuffers(plan, rank, T, "cpu")
        self.y = self.bufs.y
        self.corrupt = corrupt
        self.freed = False
        self.fail_from_step, self.steps_done = fail_from_step, 0

    def poison(self):
        for b in self.bufs.ypad:
            b.fill_(float("nan"))
        self.bufs.y.fill_(float("nan"))

    def step(self):
        from spmv_openmp_cuda_amd import synth

        self.steps_done += 1

        def compute_group(g, slot):
            if self.fail_from_step and self.steps_done >= self.fail_from_step:
                raise RuntimeError("launcher failed")             # e.g. a kernel launch returning EXIT_FAILURE on this rank
            b0, b1 = self.plan.block(self.rank, g)
            ja, as_ = self.oracle.synth_fill(self.w.N, b0, self.irp[b0:b1 + 1], synth.SEED_STRUCT + self.w.cfg, synth.SEED_VAL + self.w.cfg, 0)
            il = (self.irp[b0:b1 + 1] - self.irp[b0]).astype(np.uint32)
            y = self.oracle.csr_serial_dev(il, ja, as_, self.x)
            if self.corrupt and b1 > b0:
                y[0] += 1.0
            slot[: b1 - b0].copy_(torch.from_numpy(y))
        self.sharding.step(self.plan, dist, self.bufs, compute_group)
        return self.y

    def free(self):
        self.freed = True


def _search_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import Oracle
    from spmv_openmp_cuda_amd import sharding, synth
    oracle = Oracle()
    w = synth.Workload("p", 5003, 90000, "powerlaw", 3000, 9)          # 5003 rows: no (world x groups) divides them
    irp = synth.prefix(synth.row_lengths(w))
    x = synth.make_x(w.N, w.cfg)
    ja_all, as_all = oracle.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
    y_ref = oracle.csr_serial_dev(irp.astype(np.uint32), ja_all, as_all, x)
    K = sharding.ExchangeKey
    keys = [K("rccl", groups=1), K("rccl", groups=3), K("push", 1), K("fused"), K("rccl", groups=2), K("rccl", groups=4),
            K("rccl", groups=5)]
    made = []

    def make(key):
        if key.mode == "push":
            if rank == world - 1:
                raise RuntimeError("no peer windows on this rank")        # set-up fails on ONE rank
            cand = _CpuExchange(key, sharding.make_plan(irp, world, 1), rank, irp, w, x, oracle, sharding)
        else:
            # groups = 2: a wrong y on rank 0 only
            # groups = 5: the SECOND step (the first timed one) raises on the last rank only, before its first kernel
            cand = _CpuExchange(key, sharding.make_plan(irp, world, key.groups), rank, irp, w, x, oracle, sharding,
                                corrupt=(key.groups == 2 and rank == 0),
                                fail_from_step=2 if key.groups == 5 and rank == world - 1 else 0)
        made.append(cand)
        return cand

    def validate(cand, ref):
        y = cand.y.numpy()
        if np.isnan(y).any():
            return False
        return np.array_equal(y, y_ref) if ref is None else bool(torch.equal(cand.y, ref))

    resolved = []
    best, rep = sharding.search_exchange(keys, make, dist, torch, "cpu", validate, budget_s=1e9, steps=2,
                                         on_resolved=lambda i, bk: resolved.append((i, bk.name if bk else None)))
    ok = best is not None and not best.freed and rep["chosen"] in ("rccl-g1", "rccl-g3", "rccl-g4")
    ok = ok and set(rep["exchange_step_ms"]) == {"rccl-g1", "rccl-g3", "rccl-g4"}
    ok = ok and rep["exchange_rejected"] == {"push-p1": "set-up failed on some rank", "rccl-g2": "y incomplete or different on some rank",
                                             "rccl-g5": "a timed step failed on some rank"}
    ok = ok and list(rep["exchange_skipped"]) == ["fused"]             # no push candidate was validated
    ok = ok and [i for i, _ in resolved] == list(range(len(keys)))
    ok = ok and all(c.freed for c in made if c is not best)
    ok = ok and np.array_equal(best.step().numpy(), y_ref)
    # an exhausted budget stops the search behind the first candidate that delivered, on every rank alike
    best2, rep2 = sharding.search_exchange(keys, make, dist, torch, "cpu", validate, budget_s=0.0, steps=1)
    ok = ok and rep2["chosen"] == "rccl-g1" and list(rep2["exchange_step_ms"]) == ["rccl-g1"] and len(rep2["exchange_skipped"]) == len(keys) - 1
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else f"mismatch {rep} {rep2} {resolved}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_search_takes_the_same_branch_on_every_rank(tmp_path, world):
    """bench.py's start-up search over exchange candidates (sharding.search_exchange) on gloo, with a plan whose row
    count no (world x groups) divides: a candidate that cannot be built on ONE rank and one whose y is wrong on ONE rank
    are dropped on ALL ranks -- as is one whose step raises on ONE rank during the timed phase (the step keeps its
    collectives matched, nobody hangs) --, kernel-issued stores are skipped while no push candidate has delivered, losers are freed,
    and an exhausted budget ends the search on every rank at the same candidate."""
    port = _free_port()
    mp.spawn(_search_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok"


def test_exchange_keys_are_rank_independent():
    sys.path.insert(0, ROOT)
    from spmv_openmp_cuda_amd.sharding import default_candidates
    for tiles in (True, False):
        d = default_candidates(8, tiles)
        assert len(d) <= 8 and len({k.name for k in d}) == len(d)
        assert [k.name for k in d][:3] == ["rccl-g1", "rccl-g2", "push-p1"]                  # safest first
        e = default_candidates(8, tiles, extra=True)
        assert e[:len(d)] == d and len({k.name for k in e}) == len(e)
    assert not any(k.mode in ("fused", "pushk") for k in default_candidates(8, False, extra=True))
