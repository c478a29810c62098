import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from neural_svd_amd import hip_ops as H
from oracle import nsvd_oracle as O
dev = "cuda:0"
L, B, m, hidden = 2, 64, 1024, (128,)
p = O.init_params(L, 2, m, hidden, 0.1, seed=0)
torch.manual_seed(1)
x = 16 * torch.randn(B, 2)
p64 = p.to(torch.float64)
phi = O.fourier_features(x.double(), p64.fourier_B)
base64 = O.mlp_forward(phi, p64)
base32 = O.mlp_forward(O.fourier_features(x, p.fourier_B), p)
print("cpu fp32 vs fp64 base rel", float((base32.double()-base64).norm()/base64.norm()))
shape = H.ModelShape(L=L, D=2, m=m, hidden=hidden)
ws = [w.to(dev) for w in p.ws]; bs = [b.to(dev) for b in p.bs]
pp = H.pack_params(shape, ws, bs, p.fourier_B.to(dev), None)
out = H.model_forward(shape, pp, x.to(dev), 1.0, H.new_workspace(shape, B, dev)).cpu()
print("hip model_forward vs fp64 rel", float((out.double()-base64).norm()/base64.norm()))
prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.5, sigma=16.0)
c64 = O.operator_forward(x.double(), p64, prob_o)
c32 = O.operator_forward(x, p, prob_o)
print("cpu fp32 vs fp64 f rel", float((c32.f.double()-c64.f).norm()/c64.f.norm()), "spc0 min", float(c64.spc0.min()), float(c64.sp0.min()))
print("f64 f absmax", float(c64.f.abs().max()), "rows by |x|:", x.norm(dim=1).topk(3))
