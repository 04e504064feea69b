"""Probe: can two processes on this box map each other's device buffers (torch CUDA IPC) and copy into them?
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 scripts/ipc_probe.py"""
import os, sys, time
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
dev = int(os.environ.get("LOCAL_RANK", "0")) % ndev
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
n = 10_000_000
y = torch.full((n,), float(rank + 1) * 0 - 1.0, dtype=torch.float64, device="cuda")
h = y.untyped_storage()._share_cuda_()
print(rank, "handle fields", [type(a).__name__ for a in h], file=sys.stderr)
allh = [None] * world
dist.all_gather_object(allh, h)
peers = []
for p in range(world):
    if p == rank:
        peers.append(y); continue
    st = torch.UntypedStorage._new_shared_cuda(*allh[p])
    t = torch.empty(0, dtype=torch.float64, device=st.device).set_(st, 0, (n,))
    peers.append(t)
per = n // world
mine = slice(rank * per, (rank + 1) * per)
y[mine] = float(rank + 1)
torch.cuda.synchronize(); dist.barrier()
streams = [torch.cuda.Stream() for _ in range(world)]
for it in range(3):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    ev = torch.cuda.Event(); ev.record()
    for p in range(world):
        if p == rank: continue
        with torch.cuda.stream(streams[p]):
            streams[p].wait_event(ev)
            peers[p][mine].copy_(y[mine], non_blocking=True)
    for p in range(world):
        if p != rank: torch.cuda.current_stream().wait_stream(streams[p])
    torch.cuda.synchronize(); dist.barrier()
    t1 = time.perf_counter()
    print(rank, f"push {per * 8 / 1e6:.0f} MB to {world - 1} peers: {(t1 - t0) * 1e3:.3f} ms", file=sys.stderr)
exp = torch.arange(n, device="cuda") // per + 1
ok = bool((y == exp.to(torch.float64)).all())
print(rank, "full y assembled:", ok, file=sys.stderr)
dist.barrier()
del peers
dist.destroy_process_group()
sys.exit(0 if ok else 1)
