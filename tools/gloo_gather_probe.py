"""Developer probe (GPU box): `python -m torch.distributed.run --nproc-per-node 2 tools/gloo_gather_probe.py plain`.
Two ranks on device 0 call torch.distributed.gather (gloo) on 12.5 MB DEVICE tensors thirty times; on the pool's
boxes it stops returning after a few calls, with nothing of the renderer involved: gloo is a CPU backend and reads
the tensors' device memory from host threads.  The renderer's BandGather therefore stages device bands through
pinned host memory whenever the process group is not nccl (mode `staged` here does the same and runs to the end)."""
import os, sys, time, faulthandler
import torch, torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
mode = sys.argv[1]
src = torch.full((68, 184320), rank, dtype=torch.uint8, device=dev)
recv = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
comm = torch.cuda.Stream(device=dev)
for k in range(30):
    if mode == "plain":
        dist.gather(src, recv, dst=0)
    elif mode == "staged":
        host = src.cpu()
        hrecv = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, hrecv, dst=0)
        if rank == 0:
            for r in range(world):
                recv[r].copy_(hrecv[r])
    else:
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            comm.wait_event(ev)
            dist.gather(src, recv, dst=0)
    if rank == 0:
        print(mode, "gather", k, "ok", flush=True)
torch.cuda.synchronize()
dist.barrier()
if rank == 0:
    print(mode, "done", flush=True)
