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
import time
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
    wait for the gathers in order and compact.  Returns bufs.y.
    A `compute_group` that raises on THIS rank does not strand the other ranks in their collectives: every gather of
    the step is still issued (the slot then holds whatever it held), and the exception is re-raised afterwards."""
    works, failure = [], None
    for g in range(plan.groups):
        if failure is None:
            try:
                compute_group(g, bufs.slot[g])
            except Exception as e:                  # noqa: BLE001 -- re-raised below, after the collectives
                failure = e
        works.append(dist.all_gather_into_tensor(bufs.ypad[g], bufs.slot[g], async_op=True))
    for g, w in enumerate(works):
        w.wait()
        bufs.compact(g)
    if failure is not None:
        raise failure
    return bufs.y


# ---------------------------------------------------------------------------------------------------------
# Push exchange over peer windows (include/spmvHip.h "peer windows"): xGMI is a point-to-point mesh, so the
# all-gather of y can also be written as every rank storing its rows straight into the other ranks' vectors --
# with the copy engines behind every finished piece of y (`PeerExchange.push`) or from inside the producing
# kernel (`PeerExchange.extra_pointers` handed to hipSpMVTilesReduce).  No padding, no compaction: every rank's
# y is one window of M doubles and rank r owns rows plan.rows(r) of ALL of them.
class _RawDeviceArray:
    """a raw device pointer as something torch.as_tensor() understands"""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PeerExchange:
    """Collective constructor (every rank of `dist`'s default group calls it): creates this rank's window of
    `n` doubles, ships the handles with all_gather_object and maps the others' windows.  `ok` is the same on
    every rank; when False nothing is left allocated and the caller stays on the RCCL path."""

    def __init__(self, api, dist, torch, rank, world, device_index, n):
        import ctypes as C
        self.api, self.dist, self.torch = api, dist, torch
        self.rank, self.world, self.n = rank, world, int(n)
        self.base = None
        self.peer_base = {}
        self.ok = False
        self.why = ""
        handle = (C.c_ubyte * api.IPC_HANDLE_BYTES)()
        base = C.c_void_p()
        good = world - 1 <= api.MAX_PEERS and api.lib.spmvHipWindowCreate(self.n * 8, C.byref(base), handle) == 0
        if good:
            self.base = base
        mine = (bytes(handle), int(device_index), bool(good))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if all(e[2] for e in everyone):
            for p, (h, dev, _) in enumerate(everyone):
                if p == rank:
                    continue
                buf = (C.c_ubyte * api.IPC_HANDLE_BYTES).from_buffer_copy(h)
                ptr = C.c_void_p()
                if api.lib.spmvHipWindowOpen(buf, dev, C.byref(ptr)) != 0:
                    good = False
                    self.why = f"rank {rank}: cannot map the window of rank {p} (device {dev})"
                    break
                self.peer_base[p] = ptr
        else:
            good = False
            self.why = "window allocation failed on some rank"
        verdict = [None] * world
        dist.all_gather_object(verdict, (bool(good), self.why))
        self.ok = all(v[0] for v in verdict)
        if not self.ok:
            self.why = "; ".join(v[1] for v in verdict if v[1]) or self.why
            self.close()
            return
        # peers in ring order starting behind this rank: at any moment the ranks write to different targets
        self.order = [(rank + k) % world for k in range(1, world)]
        self._peer_array = (C.c_void_p * max(len(self.order), 1))(*[self.peer_base[p].value for p in self.order])
        self.y = torch.as_tensor(_RawDeviceArray(self.base.value, self.n), device="cuda")
        self._flag = torch.zeros(1, dtype=torch.int32, device="cuda")

    def extra_pointers(self, row0):
        """(count, array of double*) -- the peers' y vectors offset to this rank's first row, for the fused store"""
        import ctypes as C
        arr = (C.c_void_p * max(len(self.order), 1))(*[self.peer_base[p].value + 8 * int(row0) for p in self.order])
        return len(self.order), arr

    def push(self, row0, row1):
        """rows [row0, row1) of the own y to every peer, behind what is enqueued on the library stream"""
        if row1 > row0 and self.api.lib.spmvHipPeerPush(self.base, 8 * int(row0), 8 * int(row1 - row0), len(self.order), self._peer_array):
            raise RuntimeError("spmvHipPeerPush failed")

    def finish(self, barrier=None):
        """own pushes done (in stream order), then a node-wide barrier: afterwards every rank's y is complete"""
        if self.api.lib.spmvHipPeerPushJoin() or self.api.lib.spmvHipTilesPushJoin():
            raise RuntimeError("spmvHipPeerPushJoin / spmvHipTilesPushJoin failed")
        if barrier is not None:
            barrier()
        else:
            self.dist.all_reduce(self._flag)         # RCCL, in stream order: returns on a rank only after every rank entered it

    def close(self):
        self.y = None
        for ptr in self.peer_base.values():
            self.api.lib.spmvHipWindowClose(ptr)
        self.peer_base = {}
        try:
            self.dist.barrier()                          # nobody frees a window that is still mapped elsewhere
        except Exception:
            pass
        if self.base is not None:
            self.api.lib.spmvHipWindowFree(self.base)
            self.base = None


def bin_ranges(n_bins, pieces, unit=1):
    """cut [0, n_bins) into at most `pieces` consecutive ranges whose lengths are multiples of `unit` (the last
    one takes the remainder).  Phase 2 runs one workgroup per CU and bin, so with unit = number of CUs a cut costs
    no extra round of workgroups: 610 bins on 256 CUs are 3 rounds whether launched as one grid or as 256+256+98."""
    n_bins, unit = int(n_bins), max(1, int(unit))
    units = -(-n_bins // unit)
    pieces = max(1, min(int(pieces), units))
    per = -(-units // pieces) * unit
    return [(b, min(b + per, n_bins)) for b in range(0, n_bins, per)]


class PushSpMV:
    """This rank's rows of y = A.x written into its window and delivered to the peers.
    mode "push":  kernel pieces -> copy-engine pushes behind each piece (tiles launcher: phase 1 once, phase 2 cut
                  into `pieces` bin ranges so that the push of one range travels under the reduction of the next;
                  other launchers: one piece);
    mode "fused": hipSpMVTilesReduce stores every finished bin to all the peers itself (tiles launcher only);
    mode "pushk": hipSpMVTilesReducePush -- a push kernel beside phase 2 copies every bin to the peers as soon as the
                  reduction flags it (tiles launcher only, needs at least one peer)."""

    def __init__(self, api, px, dm, row0, launcher, x_ptr, mode="push", pieces=1, barrier=None, unit=256):
        import ctypes as C
        self.api, self.px, self.dm, self.row0, self.launcher = api, px, dm, int(row0), launcher
        self.x_ptr, self.mode, self.barrier = x_ptr, mode, barrier
        self.cfg = api.CONFIG()
        self.rows = int(dm.rows)
        self.y_own = C.c_void_p(px.base.value + 8 * self.row0)
        self.tiles = launcher == "hipSpMVTilesCSR" and self.rows > 0 and int(dm.nnz) > 0
        if mode in ("fused", "pushk") and not self.tiles:
            raise ValueError("the fused / push-kernel exchange exists for hipSpMVTilesCSR only")
        if self.tiles:
            nb, rpb = C.c_uint(), C.c_uint()
            if api.lib.spmvHipTilesShape(C.byref(dm.handle), C.byref(nb), C.byref(rpb)):
                raise RuntimeError("spmvHipTilesShape failed")
            self.rpb = int(rpb.value)
            self.ranges = bin_ranges(int(nb.value), pieces if mode == "push" else 1, unit)

            def first_row(b):                       # bins need not be equally high (tapered formats)
                r = C.c_ulong()
                if api.lib.spmvHipTilesBinRow(C.byref(dm.handle), b, C.byref(r)):
                    raise RuntimeError("spmvHipTilesBinRow failed")
                return int(r.value)
            self.range_rows = [(first_row(b0), first_row(b1)) for b0, b1 in self.ranges]
            self.n_extra, self.extra = px.extra_pointers(self.row0)
        self.pieces = len(self.ranges) if self.tiles else 1

    def step(self, ev=None):
        self.enqueue(ev)
        self.px.finish(self.barrier)
        return self.px.y

    def enqueue(self, ev=None):
        """kernels and pushes of this block of rows, without the final join + barrier (PeerExchange.finish)"""
        import ctypes as C
        api, lib, h = self.api, self.api.lib, C.byref(self.dm.handle)
        if ev:
            lib.spmvHipEventRecord(ev[0])
        if self.tiles:
            rc = lib.hipSpMVTilesExpand(h, self.x_ptr)
            if self.mode == "fused":
                b0, b1 = self.ranges[0][0], self.ranges[-1][1]
                rc = rc or lib.hipSpMVTilesReduce(h, b0, b1, self.y_own, self.n_extra, self.extra)
            elif self.mode == "pushk":
                rc = rc or lib.hipSpMVTilesReducePush(h, self.y_own, self.n_extra, self.extra)
            else:
                for (b0, b1), (ra, rb) in zip(self.ranges, self.range_rows):
                    rc = rc or lib.hipSpMVTilesReduce(h, b0, b1, self.y_own, 0, None)
                    self.px.push(self.row0 + ra, self.row0 + rb)
        else:
            rc = api.SPMV_LAUNCHERS[self.launcher](h, self.x_ptr, self.cfg, self.y_own) if self.rows else 0
            self.px.push(self.row0, self.row0 + self.rows)
        if ev:
            lib.spmvHipEventRecord(ev[1])
        if rc:
            raise RuntimeError(self.launcher + " failed")


# ---------------------------------------------------------------------------------------------------------
# Choosing the exchange at start-up.  How y reaches the other ranks is MEASURED (whole steps, slowest rank), not
# assumed; this is the collective control flow of that search, kept free of anything GPU-specific so that the
# world_size-2/3 gloo tests drive exactly the code bench.py runs on RCCL.
@dataclass(frozen=True)
class ExchangeKey:
    """A candidate, named by rank-INDEPENDENT parameters only (what a rank derives from its own rows -- its number of
    bins, the pieces that fit them -- never enters the name: every rank must walk the same list in the same order)."""
    mode: str               # "rccl" | "push" | "fused" | "pushk"
    pieces: int = 1         # push: pieces of phase 2 whose rows are pushed while the next piece is reduced
    groups: int = 1         # row groups per rank (rows of one group travel while the next group is computed)
    rdiv: int = 1           # two-phase format with bins of 1/rdiv the automatic height
    taper: bool = False     # two-phase format with tapered bins

    @property
    def name(self):
        base = {"rccl": "rccl", "push": f"push-p{self.pieces}", "fused": "fused", "pushk": "pushk"}[self.mode]
        return base + (f"-g{self.groups}" if self.groups > 1 or self.mode == "rccl" else "") + \
            (f"-r{self.rdiv}" if self.rdiv > 1 else "") + ("-t" if self.taper else "")

    @property
    def format_key(self):
        """which copy of the rank's matrices (row groups, bin layout) the candidate runs on"""
        return (self.groups, self.rdiv, self.taper)


def default_candidates(world, tiles, rccl=True, windows=True, extra=False):
    """Safest first: RCCL all-gather (the portable collective), then copy-engine pushes into peer windows, then stores
    issued by the kernels themselves.  At most eight by default; `extra` appends the forms that only make sense once
    a first hardware run has shown which family wins (more groups / pieces, smaller or tapered bins, push kernel)."""
    K = ExchangeKey
    out = []
    if rccl:
        out += [K("rccl", groups=1), K("rccl", groups=2)]
    if windows:
        out += [K("push", 1)]
        if tiles:
            out += [K("push", 4)]
        out += [K("push", 1, groups=2)]
        if tiles:
            out += [K("fused"), K("fused", groups=2)]
    if extra:
        if rccl:
            out += [K("rccl", groups=4)]
        if windows:
            out += [K("push", 1, groups=4)]
            if tiles:
                out += [K("push", 2), K("push", 8), K("push", 2, groups=2), K("push", 4, taper=True), K("push", 4, groups=2, taper=True),
                        K("pushk"), K("pushk", groups=2), K("fused", rdiv=2), K("fused", rdiv=4), K("fused", groups=2, rdiv=2),
                        K("fused", taper=True), K("fused", groups=2, taper=True)]
    return out


def search_exchange(keys, make, dist, torch, device, validate, budget_s=60.0, steps=3, log=lambda *a: None,
                    on_resolved=lambda i, best_key: None, clock=time.perf_counter):
    """Collective.  For every key in order: stop when the slowest rank has spent `budget_s` in the search; build the
    candidate (`make(key)`; an exception on ANY rank drops it on ALL); one validating step behind a poison of y and a
    barrier (peers write into this rank's y: nobody steps before everybody has poisoned) checked by
    `validate(cand, ref_y_or_None)` on every rank (a failure on ANY rank drops it on ALL; the first candidate that
    passes provides the reference y the later ones are compared with element by element); then `steps` timed steps,
    slowest rank.  Kernel-issued stores ("fused", "pushk") are only tried after a copy-engine push through the same
    mappings has delivered a correct y.  Every decision is taken on a reduced value, so all ranks take the same
    branch.  Returns (best candidate or None, report)."""
    t_start = clock()
    times, rejected, skipped = {}, {}, {}
    best, best_key, ref = None, None, None

    def agree(value, op):
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=op)
        return float(t[0])

    for i, key in enumerate(keys):
        if agree(clock() - t_start, dist.ReduceOp.MAX) >= budget_s and best is not None:
            for k in keys[i:]:
                skipped[k.name] = "search budget spent"
            break
        if key.mode in ("fused", "pushk") and not any(k.mode == "push" and k.name in times for k in keys):
            skipped[key.name] = "no copy-engine push through the peer windows has delivered a correct y"
            on_resolved(i, best_key)
            continue
        cand = None
        try:
            cand = make(key)
        except Exception as e:                      # noqa: BLE001 -- e.g. out of memory for one more copy of the format
            log(f"exchange candidate {key.name} could not be set up here: {e}")
        if agree(1.0 if cand is not None else 0.0, dist.ReduceOp.MIN) == 0.0:
            rejected[key.name] = "set-up failed on some rank"
            if cand is not None:
                cand.free()
            on_resolved(i, best_key)
            continue
        good = False
        try:
            cand.poison()
        except Exception as e:                      # noqa: BLE001 -- agreed on below: nobody is left alone in the barrier
            log(f"exchange candidate {key.name} could not poison y here: {e}")
            cand_ok = False
        else:
            cand_ok = True
        dist.barrier()
        try:
            if cand_ok:
                cand.step()
                good = bool(validate(cand, ref))
        except Exception as e:                      # noqa: BLE001
            log(f"exchange candidate {key.name} failed its validating step here: {e}")
        if agree(1.0 if good else 0.0, dist.ReduceOp.MIN) == 0.0:
            rejected[key.name] = "y incomplete or different on some rank"
            cand.free()
            on_resolved(i, best_key)
            continue
        if ref is None:
            ref = cand.y.clone()
        dist.barrier()
        t0 = clock()
        timed_ok = True
        # A step that raises on ONE rank (a launcher returning failure, a push kernel giving up) must not leave the others
        # waiting in a collective: the candidates' step() keep their collectives matched even then, this rank goes on
        # through ALL the steps the others run, and the drop is agreed on afterwards.
        for _ in range(steps):
            try:
                cand.step()
            except Exception as e:                  # noqa: BLE001
                timed_ok = False
                log(f"exchange candidate {key.name} failed a timed step here: {e}")
        try:
            if hasattr(cand, "sync"):
                cand.sync()
        except Exception as e:                      # noqa: BLE001
            timed_ok = False
            log(f"exchange candidate {key.name} failed to synchronise here: {e}")
        elapsed_ms = (clock() - t0) / steps * 1e3
        if agree(1.0 if timed_ok else 0.0, dist.ReduceOp.MIN) == 0.0:
            rejected[key.name] = "a timed step failed on some rank"
            cand.free()
            on_resolved(i, best_key)
            continue
        times[key.name] = agree(elapsed_ms, dist.ReduceOp.MAX)
        if best is None or times[key.name] < times[best_key.name]:
            if best is not None:
                best.free()
            best, best_key = cand, key
        else:
            cand.free()
        on_resolved(i, best_key)
    report = {"exchange_step_ms": times, "exchange_rejected": rejected, "exchange_skipped": skipped,
              "exchange_search_s": agree(clock() - t_start, dist.ReduceOp.MAX), "chosen": best_key.name if best_key else None}
    return best, report
