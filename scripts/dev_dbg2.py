import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from neural_svd_amd import hip_ops as H
from oracle import nsvd_oracle as O
dev = "cuda:0"
for (B, m) in [(64, 1024), (32, 1024), (64, 512), (128, 1024), (512, 1024)]:
    torch.manual_seed(1)
    x = 16 * torch.randn(B, 2)
    fB = 2 * np.pi * 0.1 * torch.randn(2, m)
    out = H.fourier_features(x.to(dev), fB.to(dev), 0.01, 5).cpu()
    pts = O.stencil_points(x.double(), 0.01)
    ref = torch.cat([O.fourier_features(p, fB.double()) for p in pts], dim=0).T
    d = (out.double() - ref).abs()
    bad = (d > 1e-4).nonzero()
    print(B, m, "max err", float(d.max()), "n bad", bad.shape[0], bad[:5].tolist(), bad[-3:].tolist())
