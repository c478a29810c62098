"""dev: run the per-rank head-parallel compute for one world size N (argv[1]) for profiling under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
N = int(sys.argv[1])
dev = torch.device("cuda:0")
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
shape = H.ModelShape(L=16 // N, D=2, m=1024, hidden=(128, 128, 128))
tr = FusedTrainer(shape, prob, 512 * N, sequential=False, device=dev)
for _ in range(300): tr.step()
torch.cuda.synchronize()
