"""dev (GPU box, one GPU): the RCCL calls of parallel.Communicator on a one-rank "nccl" group - the collectives are no-ops
with one rank, but the backend's support for every call the multi-GPU step makes is exercised (ReduceOp.AVG, async
all-reduce on slices of a flat buffer, all_gather_into_tensor, broadcast, device_id init)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch
from neural_svd_amd import parallel
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
comm = parallel.Communicator.from_env(dev)
assert comm.backend == "nccl" and comm.world == 1
m = torch.arange(513, dtype=torch.float32, device=dev)
comm.all_reduce_mean(m)
g = torch.ones(1 << 20, dtype=torch.float32, device=dev)
works = [comm.all_reduce_sum(g[lo:lo + (1 << 18)], async_op=True) for lo in range(0, 1 << 20, 1 << 18)]
for w in works:
    w.wait()
out = torch.empty((1, 2, 64, 4), dtype=torch.float32, device=dev)
inp = torch.randn(2, 64, 4, device=dev)
comm.all_gather(out, inp, async_op=True).wait()
comm.broadcast(g, 0)
comm.barrier()
torch.cuda.synchronize()
assert torch.equal(out[0], inp) and float(m[512]) == 512.0 and float(g.sum()) == float(1 << 20)
print("rccl one-rank smoke ok; max_float:", comm.max_float(1.5))
comm.close()
