"""Row-block sharding of y = A.x across the GPUs of one node.

The reference is single-GPU; this part is new.  Rows are independent, so the
matrix is cut into `world` contiguous row blocks balanced by nnz (boundaries on
the IRP prefix sum), x is replicated, every rank computes its block and the
ranks all-gather y so that each ends with the full vector.  One exchange step,
one collective: an all-gather over RCCL/xGMI (gloo in the CPU tests).

`all_gather_into_tensor` needs equal-sized blocks, so each rank's block is
padded to the largest block (`max_rows`); the kernel writes straight into the
rank's slot of the padded buffer and `compact` copies the blocks back to back.
When every block has the same number of rows the padded buffer IS y and the
compaction disappears.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class ShardPlan:
    world: int
    bounds: np.ndarray          # int64[world+1], rows of rank p = bounds[p]..bounds[p+1]
    max_rows: int

    @property
    def equal_blocks(self):
        return bool((np.diff(self.bounds) == self.max_rows).all())

    def rows(self, rank):
        return int(self.bounds[rank]), int(self.bounds[rank + 1])


def partition_by_nnz(irp, world):
    """Python twin of spmvHipPartitionRows (csrc/hip/abi.hip): boundary p is the row
    whose starting offset is closest to p/world of the nnz."""
    irp = np.asarray(irp)
    M = irp.size - 1
    nnz = int(irp[-1]) - int(irp[0])
    bounds = np.zeros(world + 1, dtype=np.int64)
    for p in range(1, world):
        target = int(irp[0]) + (nnz * p) // world
        r = int(np.searchsorted(irp, target, side="left"))
        r = min(r, M)
        if r > 0 and target - int(irp[r - 1]) < int(irp[r]) - target:
            r -= 1
        bounds[p] = max(r, bounds[p - 1])
    bounds[world] = M
    return bounds


def make_plan(irp, world):
    bounds = partition_by_nnz(irp, world)
    return ShardPlan(world, bounds, int(np.diff(bounds).max()))


def alloc_buffers(plan, rank, torch, device, dtype=None):
    """(ypad, slot, y): padded gather buffer, this rank's slot in it, and the
    contiguous result (aliases ypad when blocks are equal)."""
    dtype = dtype or torch.float64
    M = int(plan.bounds[-1])
    ypad = torch.full((plan.world * plan.max_rows,), float("nan"), dtype=dtype, device=device)
    slot = ypad[rank * plan.max_rows:(rank + 1) * plan.max_rows]
    y = ypad[:M] if plan.equal_blocks else torch.full((M,), float("nan"), dtype=dtype, device=device)
    return ypad, slot, y


def gather_y(plan, dist, ypad, slot, y):
    """All-gather the row blocks (in place) and compact them into y."""
    dist.all_gather_into_tensor(ypad, slot)
    if not plan.equal_blocks:
        for p in range(plan.world):
            b0, b1 = plan.rows(p)
            y[b0:b1].copy_(ypad[p * plan.max_rows:p * plan.max_rows + (b1 - b0)])
    return y
