"""Row-block sharding of y = A.x across the GPUs of one node.

The reference is single-GPU; this part is new.  Rows are independent, so the
matrix is cut into contiguous row blocks balanced by nnz (boundaries on the IRP
prefix sum), x is replicated, every rank computes its rows and the ranks
all-gather y so that each ends with the full vector.  One exchange step, one
kind of collective: all-gather over RCCL/xGMI (gloo in the CPU tests).

xGMI is point-to-point (one link per GPU pair), so at N = 2..8 the gather costs
as much as the kernel.  To overlap the two, every rank's rows are cut into
`groups` consecutive row GROUPS (world x groups blocks in all, still nnz
balanced); a step runs  kernel(g) -> async all-gather(g)  for g = 0..groups-1,
so gather g travels while group g+1 is computed, and waits at the end.

`all_gather_into_tensor` needs equal-sized pieces, so for each group the ranks'
blocks are padded to the largest one; the kernel writes straight into the
rank's slot of the group's padded buffer, the gather is in place, and the
blocks are then copied back to back into y (skipped when nothing is padded).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class ShardPlan:
    world: int
    groups: int
    bounds: np.ndarray          # int64[world*groups+1]; block k = rank k // groups, group k % groups
    max_rows: list              # per group: largest block of that group over the ranks

    def block(self, rank, group):
        k = rank * self.groups + group
        return int(self.bounds[k]), int(self.bounds[k + 1])

    def rows(self, rank):
        """all rows of `rank` (its groups are consecutive)"""
        return int(self.bounds[rank * self.groups]), int(self.bounds[(rank + 1) * self.groups])

    @property
    def M(self):
        return int(self.bounds[-1])

    @property
    def equal_blocks(self):
        """no padding anywhere and one group: the gather buffer IS y"""
        d = np.diff(self.bounds)
        return self.groups == 1 and bool((d == d[0]).all())


def partition_by_nnz(irp, parts):
    """Python twin of spmvHipPartitionRows (csrc/hip/abi.hip): boundary p is the row
    whose starting offset is closest to p/parts of the nnz."""
    irp = np.asarray(irp)
    M = irp.size - 1
    nnz = int(irp[-1]) - int(irp[0])
    bounds = np.zeros(parts + 1, dtype=np.int64)
    for p in range(1, parts):
        target = int(irp[0]) + (nnz * p) // parts
        r = int(np.searchsorted(irp, target, side="left"))
        r = min(r, M)
        if r > 0 and target - int(irp[r - 1]) < int(irp[r]) - target:
            r -= 1
        bounds[p] = max(r, bounds[p - 1])
    bounds[parts] = M
    return bounds


def make_plan(irp, world, groups=1, snap_tol=0.01):
    """nnz-balanced bounds; when cutting the rows into EQUAL blocks is just as balanced (largest block within
    `snap_tol` of the mean nnz -- true for matrices whose heavy rows are scattered) the equal cut is used instead:
    with one group per rank the gather buffer then IS y and no compaction pass is needed."""
    parts = world * groups
    bounds = partition_by_nnz(irp, parts)
    M = len(irp) - 1
    if snap_tol and M % parts == 0 and M > 0:
        eq = np.arange(parts + 1, dtype=np.int64) * (M // parts)
        share = np.diff(np.asarray(irp)[eq].astype(np.int64))
        if share.max() <= (1.0 + snap_tol) * share.mean():
            bounds = eq
    d = np.diff(bounds).reshape(world, groups)
    return ShardPlan(world, groups, bounds, [int(d[:, g].max()) for g in range(groups)])


class GatherBuffers:
    """Per group g: ypad[g] (world x max_rows[g]) and this rank's slot in it; y = the full vector."""

    def __init__(self, plan, rank, torch, device, dtype=None):
        dtype = dtype or torch.float64
        self.plan, self.rank = plan, rank
        self.ypad, self.slot = [], []
        for g in range(plan.groups):
            k = plan.max_rows[g]
            buf = torch.full((plan.world * max(k, 1),), float("nan"), dtype=dtype, device=device)
            self.ypad.append(buf)
            self.slot.append(buf[rank * max(k, 1):(rank + 1) * max(k, 1)])
        self.y = self.ypad[0][:plan.M] if plan.equal_blocks else \
            torch.full((plan.M,), float("nan"), dtype=dtype, device=device)

    def compact(self, g):
        """copy group g's gathered blocks to their rows of y"""
        plan = self.plan
        if plan.equal_blocks:
            return
        k = max(plan.max_rows[g], 1)
        for p in range(plan.world):
            b0, b1 = plan.block(p, g)
            if b1 > b0:
                self.y[b0:b1].copy_(self.ypad[g][p * k:p * k + (b1 - b0)])


def step(plan, dist, bufs, compute_group):
    """One sharded SpMV: for every group run `compute_group(g, slot_tensor)` (enqueues the kernels that
    write this rank's rows of group g into the slot), then all-gather the group asynchronously; finally
    wait for the gathers in order and compact.  Returns bufs.y."""
    works = []
    for g in range(plan.groups):
        compute_group(g, bufs.slot[g])
        works.append(dist.all_gather_into_tensor(bufs.ypad[g], bufs.slot[g], async_op=True))
    for g, w in enumerate(works):
        w.wait()
        bufs.compact(g)
    return bufs.y
