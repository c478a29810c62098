"""dev: repeat ONE mixed-precision CDK step from identical weights; where in the workspace do repetitions differ?"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from neural_svd_amd import hip_ops as H  # noqa: E402
from neural_svd_amd.cdk import FusedCdkStep, HeteroNetwork, NestedLoRAForCDK, get_mlp  # noqa: E402

dev = "cuda:0"
sizes, B = [128, 256, 256], 256
d0, d1, d2 = sizes
g = torch.Generator().manual_seed(77)
x, y = torch.randn(B, d0, generator=g).to(dev), torch.randn(B, d0, generator=g).to(dev)
torch.manual_seed(11)
model = HeteroNetwork([get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                       get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                      [nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(dev).train()
sd0 = copy.deepcopy(model.state_dict())
method = NestedLoRAForCDK(model, neigs=d2, step=1, sequential=False, set_first_mode_const=True).to(dev)
fs = FusedCdkStep(method, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=B, use_amp=True)
tb = H.tower_workspace(B, d0, d1, d2, dev).numel()
al = lambda n: (n + 255) // 256 * 256  # noqa: E731
names = [("Y1", B * d1), ("A1", B * d1), ("A1T", B * d1), ("Y2p", 2 * B * d2), ("Y2", B * d2), ("XT", B * d0),
         ("W2T", d1 * d2), ("dY2", B * d2), ("dY2T", B * d2), ("dA1", B * d1), ("dY1T", B * d1), ("mean1", d1),
         ("inv1", d1), ("mean2", d2), ("inv2", d2)]
regions, off = [], 0
for t in range(2):
    off = t * tb
    for n, nf in names:
        regions.append((off, off + nf * 4, f"tower{t}.{n}"))
        off += al(nf * 4)
    regions.append((off, (t + 1) * tb, f"tower{t}.operand copies"))
regions.append((2 * tb, fs.ws.numel(), "step scratch"))
snaps = []
for rep in range(30):
    model.load_state_dict(sd0)
    fs.t = 0
    fs._weight_versions = None
    for b in fs.bufs:
        for v in b.values():
            v.zero_()
    if rep % 3 == 1:
        torch.empty(1 << 22, device=dev).normal_()  # (perturb the timing / cache state between repetitions)
    fs.step(x, y)
    torch.cuda.synchronize()
    snaps.append(fs.ws.cpu().numpy().copy())
base = snaps[0]
for rep in range(1, len(snaps)):
    d = np.nonzero(snaps[rep] != base)[0]
    if d.size == 0:
        continue
    hit = {}
    for lo, hi, n in regions:
        k = int(((d >= lo) & (d < hi)).sum())
        if k:
            first = int(d[(d >= lo) & (d < hi)][0] - lo)
            hit[n] = (k, first)
    print(f"rep {rep}: {d.size} bytes differ:", hit)
print("done")
