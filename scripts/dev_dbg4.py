import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from neural_svd_amd import hip_ops as H
from oracle import nsvd_oracle as O
dev = "cuda:0"
hidden = (128,)
for (L, B, m) in [(1, 32, 1024), (1, 64, 1024), (2, 32, 1024), (2, 64, 1024), (2, 128, 1024), (2, 64, 896), (2, 64, 1000), (2, 64, 1023), (2, 64, 1025), (2,64,2048), (16, 128, 1024), (3, 64, 1024)]:
    p = O.init_params(L, 2, m, hidden, 0.1, seed=0)
    torch.manual_seed(1)
    x = 16 * torch.randn(B, 2)
    p64 = p.to(torch.float64)
    base64 = O.mlp_forward(O.fourier_features(x.double(), p64.fourier_B), p64)
    shape = H.ModelShape(L=L, D=2, m=m, hidden=hidden)
    ws = [w.to(dev) for w in p.ws]; bs = [b.to(dev) for b in p.bs]
    fB = p.fourier_B.to(dev)
    pp = H.pack_params(shape, ws, bs, fB, None)
    xd = x.to(dev)
    wsb = H.new_workspace(shape, B, dev)
    out = H.model_forward(shape, pp, xd, 1.0, wsb).cpu()
    # torch-on-GPU recomputation of the same thing
    phi = torch.cat([torch.sin(xd @ fB), torch.cos(xd @ fB)], 1)
    z0 = torch.einsum("lhd,bd->lhb", ws[0], phi) + bs[0]
    o2 = (torch.einsum("lhp,lpb->lhb", ws[1], torch.nn.functional.softplus(z0)) + bs[1]).permute(2, 0, 1).reshape(B, L).cpu()
    r = lambda a: float((a.double()-base64).norm()/base64.norm())
    print(L, B, m, "hip", f"{r(out):.2e}", "torch-gpu", f"{r(o2):.2e}")
