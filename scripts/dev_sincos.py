import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H
dev = "cuda:0"
for scale in (1.0, 10.0, 60.0, 1000.0, 60000.0, 1e6):
    x = (torch.rand(1 << 20, 1) * 2 - 1) * scale
    fB = torch.ones(1, 1)
    out = H.fourier_features(x.to(dev), fB.to(dev), 0.0, 1).cpu().double()  # (2, N)
    xs = x.double().view(-1)
    es = (out[0] - torch.sin(xs)).abs(); ec = (out[1] - torch.cos(xs)).abs()
    t = torch.sin(x.to(dev)).cpu().double().view(-1)
    print(f"scale {scale:8.0f}: sin max {es.max():.2e} mean {es.mean():.2e} | cos max {ec.max():.2e} mean {ec.mean():.2e} | torch.sin(gpu) max {(t-torch.sin(xs)).abs().max():.2e} mean {(t-torch.sin(xs)).abs().mean():.2e}")
