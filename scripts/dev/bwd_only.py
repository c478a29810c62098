"""dev: the backward pair alone in a loop (the forward run once): do the weight-gradient / chain kernels take as long
without a forward kernel in front of each of them? (power / clock coupling between the kernels of a step)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
tr = FusedTrainer(shape, prob, 512, sequential=False, device=dev, lr=0.0)
for _ in range(50):
    tr.step()
tr._own_batch = False
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    tr.backward(None, True)
torch.cuda.synchronize()
print("done")
