"""dev: per-workgroup timeline of the whole-column tower kernels (both towers in one launch, as the training step runs
them): start, end of the K loop and end of every workgroup on the 100 MHz clock (nsvd_debug_tcol_stamps)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from neural_svd_amd import _lib  # noqa: E402
from neural_svd_amd.cdk import FusedCdkStep, HeteroNetwork, NestedLoRAForCDK, get_mlp  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
sizes = [512, 8192, 512]
model = HeteroNetwork([get_mlp(sizes, nonlinearity="lrelu0.2"), get_mlp(sizes, nonlinearity="lrelu0.2")],
                      [nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(dev)
method = NestedLoRAForCDK(model, neigs=512, step=1, sequential=False, set_first_mode_const=True).to(dev)
fs = FusedCdkStep(method, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=1024, use_amp=True)
x, y = torch.randn(1024, 512, device=dev), torch.randn(1024, 512, device=dev)
for _ in range(20):
    fs.step(x, y)
torch.cuda.synchronize()
lib = _lib.load()
n = 4 + 4 * 1024
st = (C.c_ulonglong * n)()
lib.nsvd_debug_tcol_stamps.argtypes = [C.c_void_p]
lib.nsvd_debug_tcol_stamps(st)  # arm
for rep in range(3):
    fs.step(x, y)  # (the backward launch is the last to write: its timeline is what is read)
    torch.cuda.synchronize()
    lib.nsvd_debug_tcol_stamps(st)
    a = np.array(st[4:4 + 4 * 256], dtype=np.float64).reshape(256, 4)
    t0 = a[:, 0].min()
    s, l, e = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0
    print(f"rep {rep}: start min/med/max {s.min():.1f}/{np.median(s):.1f}/{s.max():.1f} us | loop end {l.min():.1f}/{np.median(l):.1f}/{l.max():.1f}"
          f" | end {e.min():.1f}/{np.median(e):.1f}/{e.max():.1f} us")
    xcd = np.arange(256) % 8
    print("   end by XCD:", [round(float(e[xcd == k].max()), 1) for k in range(8)], " loop end by XCD:", [round(float(l[xcd == k].max()), 1) for k in range(8)])
