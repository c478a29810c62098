"""dev: which never-written buffer of a tower workspace does the mixed-precision CDK step READ? Each buffer of the
float32 layout (csrc/tower.hip: carve_tower) is filled with NaN in turn before one step from identical weights; a NaN
in the step's outputs names the buffer."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


def align(n):
    return (n + 255) // 256 * 256


S = 2  # (fwd2 slices of this shape)
names = [("Y1", B * d1), ("A1", B * d1), ("A1T", B * d1), ("Y2p", S * B * d2), ("Y2", B * d2), ("XT", B * d0),
         ("W2T", d1 * d2), ("dY2", B * d2), ("dY2T", B * d2), ("dA1", B * d1), ("dY1T", B * d1), ("mean1", d1),
         ("inv1", d1), ("mean2", d2), ("inv2", d2)]
offs, off = {}, 0
for n, nf in names:
    offs[n] = (off, nf * 4)
    off += align(nf * 4)
print("tower workspace", tb, "carved so far", off)


def one(poison):
    model.load_state_dict(sd0)
    fs.t = 0
    fs._weight_versions = None
    for b in fs.bufs:
        for v in b.values():
            v.zero_()
    fs.ws.zero_()
    if poison is not None:
        o, nb = offs[poison]
        for t in range(2):
            fs.ws[t * tb + o:t * tb + o + nb].view(torch.float32).fill_(float("nan"))
    out = fs.step(x, y).clone()
    torch.cuda.synchronize()
    return out.cpu()


ref = one(None)
print("reference", ref.tolist())
for n, _ in names:
    out = one(n)
    bad = not bool(torch.isfinite(out).all())
    print(f"poisoned {n:6s}: {'NaN in the outputs' if bad else ('same' if torch.equal(out, ref) else 'DIFFERENT but finite')}  {out.tolist() if bad or not torch.equal(out, ref) else ''}")
# the area behind the carved buffers (hA / hB operand copies) and the rest of the step's workspace
for lo, hi, n in ((off, tb, "operand copies"), (2 * tb, fs.ws.numel(), "step scratch behind the towers")):
    model.load_state_dict(sd0)
    fs.t = 0
    fs._weight_versions = None
    for b in fs.bufs:
        for v in b.values():
            v.zero_()
    fs.ws.zero_()
    if n == "operand copies":
        for t in range(2):
            fs.ws[t * tb + lo:t * tb + hi].view(torch.float32).fill_(float("nan"))
    else:
        fs.ws[lo:hi - (hi - lo) % 4].view(torch.float32).fill_(float("nan"))
    out = fs.step(x, y).clone().cpu()
    print(f"poisoned {n}: {'NaN' if not bool(torch.isfinite(out).all()) else ('same' if torch.equal(out, ref) else 'DIFFERENT')} {out.tolist()}")
