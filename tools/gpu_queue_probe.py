"""Do two torch streams land on two hardware queues once a process group exists? (run under rocprofv3 --kernel-trace)"""
import os, sys, torch
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if os.environ.get("PG") == "1":
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    t = torch.ones(1024, device=dev)
    dist.all_reduce(t)
side = torch.cuda.Stream(device=dev)
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
torch.cuda.synchronize()
for _ in range(5):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = a @ b
    d = a + b
    e = d * 2
torch.cuda.synchronize()
print("done")
