"""dev: ONE mixed-precision CDK step from identical weights, repeated; which outputs differ between repetitions?"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from neural_svd_amd.cdk import FusedCdkStep, HeteroNetwork, NestedLoRAForCDK, get_mlp  # noqa: E402

dev = "cuda:0"
sizes, B, N = [128, 256, 256], 256, int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(77)
x, y = torch.randn(B, sizes[0], generator=g).to(dev), torch.randn(B, sizes[0], generator=g).to(dev)
torch.manual_seed(11)
model = HeteroNetwork([get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                       get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                      [nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(dev).train()
sd0 = copy.deepcopy(model.state_dict())
method = NestedLoRAForCDK(model, neigs=sizes[-1], step=1, sequential=False, set_first_mode_const=True).to(dev)
fs = FusedCdkStep(method, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=B, use_amp=True)
ref = None
names = [k for k in sd0 if "num_batches" not in k]
for rep in range(N):
    model.load_state_dict(sd0)
    fs.t = 0
    fs._weight_versions = None
    for b in fs.bufs:
        for v in b.values():
            v.zero_()
    out = fs.step(x, y).clone()
    torch.cuda.synchronize()
    cur = [out.cpu()] + [model.state_dict()[k].detach().cpu().clone() for k in names]
    if ref is None:
        ref = cur
        continue
    d = [("out" if i == 0 else names[i - 1], float((a - b).abs().max())) for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
    if d:
        print(f"rep {rep}: out {cur[0].tolist()} vs {ref[0].tolist()}")
        print("    ", d[:10])
print("done")

# ---- which workspace regions differ between two repetitions of the same step?
import numpy as np  # noqa: E402
snaps = []
for rep in range(24):
    model.load_state_dict(sd0)
    fs.t = 0
    fs._weight_versions = None
    for b in fs.bufs:
        for v in b.values():
            v.zero_()
    fs.ws.zero_()
    fs.step(x, y)
    torch.cuda.synchronize()
    snaps.append(fs.ws.cpu().numpy().copy())
base = snaps[0]
print("workspace bytes", base.size)
for rep in range(1, len(snaps)):
    d = np.nonzero(snaps[rep] != base)[0]
    if d.size == 0:
        continue
    # coalesce into ranges
    cuts = np.nonzero(np.diff(d) > 4096)[0]
    starts = np.concatenate([[d[0]], d[cuts + 1]])
    ends = np.concatenate([d[cuts], [d[-1]]])
    print(f"rep {rep}: {d.size} bytes differ in ranges", [(int(s), int(e)) for s, e in zip(starts, ends)][:12])
