"""dev: N ranks on ONE device over gloo, repeated all-reduce of a (2, 1024, 512) float32 CUDA tensor + a float64 scalar
(the two collectives of cdk.ShardedCdkStep): does the emulation itself stall?"""
import os, sys, time
import torch, torch.distributed as dist
import datetime
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="env://", timeout=datetime.timedelta(seconds=40))
t = torch.randn(2, 1024, 512, device="cuda")
s = torch.zeros(1, dtype=torch.float64, device="cuda")
w = torch.randn(4096, 4096, device="cuda")
t0 = time.time()
for i in range(int(sys.argv[1])):
    for _ in range(20):
        w2 = w @ w  # some GPU work between the collectives
    dist.all_reduce(t)
    dist.all_reduce(s)
    if rank == 0 and i % 50 == 0:
        print(i, round(time.time() - t0, 2), flush=True)
torch.cuda.synchronize()
print("rank", rank, "done", flush=True)
dist.destroy_process_group()
