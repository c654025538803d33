"""Developer probe (GPU box): `python -m torch.distributed.run --nproc-per-node 1 tools/nccl_single_rank_probe.py`.
The collectives bench.py issues with several ranks -- init with device_id, gather on a side stream with events, the
all_reduce forms, barrier -- on an RCCL communicator of one rank: checks that this torch build accepts the calls as
written (a one-GPU box cannot run two nccl ranks)."""
import os

import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=dev)
src = torch.arange(68 * 1000, dtype=torch.uint8, device=dev).reshape(68, 1000)
recv = [torch.empty_like(src)]
comm = torch.cuda.Stream(device=dev)
ev = torch.cuda.Event()
for k in range(40):
    ev.record(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
        comm.wait_event(ev)
        dist.gather(src, recv, dst=0)
torch.cuda.current_stream().wait_stream(comm)
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(recv[0], src) and t.tolist() == [1.5, 2.5]
dist.destroy_process_group()
print("nccl single-rank probe ok")
