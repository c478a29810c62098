"""dev helper (GPU): run only the fused forward N times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import reference_init
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
fB, ws, bs, sc = reference_init(shape, 0.1, None, 0)
ws = [w.to(dev) for w in ws]; bs = [b.to(dev) for b in bs]; fB = fB.to(dev)
p = H.pack_params(shape, ws, bs, fB, None)
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
x = (16 * torch.randn(512, 2)).to(dev)
wsb = H.new_workspace(shape, 512, dev)
for _ in range(n):
    H.operator_forward(shape, p, prob, x, wsb, True, H.PATH_FUSED)
torch.cuda.synchronize()
print("done")
