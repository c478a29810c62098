"""dev: 300 steps of the headline step with backward_windows = int(sys.argv[1]) for rocprofv3 --stats"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
t = FusedTrainer(shape, prob, 512, sequential=False, lr=1e-4, num_iters=500000, seed=0, device=dev,
                 backward_windows=int(sys.argv[1]))
for _ in range(300): t.step()
torch.cuda.synchronize()
