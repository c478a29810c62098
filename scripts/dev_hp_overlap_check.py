"""dev (run under torch.distributed.run with NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo on one GPU): the head-parallel
step with the all-gather overlapped by the next batch's feature kernel gives bit-identical parameters to the plain
ordering."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H, parallel
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda", int(os.environ.get("NSVD_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
torch.cuda.set_device(dev)
comm = parallel.Communicator.from_env(dev, backend=os.environ.get("NSVD_DIST_BACKEND"))
shape = H.ModelShape(L=8, D=2, m=64, hidden=(128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
res = []
for ov in (True, False):
    tr = FusedTrainer(shape, prob, 64, sequential=False, device=dev, comm=comm, parallelism="hp", overlap_gather=ov,
                      seed=3)
    assert tr.overlap_gather == ov
    for _ in range(25):
        tr.step()
    torch.cuda.synchronize()
    res.append((tr.P.flat.clone(), tr.P.ema.clone(), float(tr.loss[0])))
ok = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
print(f"rank {comm.rank}: overlap == plain: {ok}; loss {res[0][2]:.4f}; params finite {bool(torch.isfinite(res[0][0]).all())}")
comm.barrier()
comm.close()
sys.exit(0 if ok else 1)
