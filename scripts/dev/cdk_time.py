"""dev: time the CDK loss forward / backward at the reference's size (B=1024, L=512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.nested_lowrank import get_joint_nesting_masks, joint_step_weights

B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = "cuda:0"
f = torch.randn(B, L, device=dev) / L ** 0.5
g = torch.randn(B, L, device=dev) / L ** 0.5
v, M = get_joint_nesting_masks(joint_step_weights(L, 1), True)
v, M = v.to(dev), M.to(dev)
ws = H.cdk_workspace(B, L, True, dev)
loss = torch.empty(3, device=dev)
rj, ri = torch.empty(B, device=dev), torch.empty(B * (B - 1), device=dev)
gf, gg = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev)

def fwd(): H.cdk_loss_forward(f, g, None, v, M, True, loss, rj, ri, ws)
def fwd_nogram(): H.cdk_loss_forward(f, g, None, v, M, True, loss, None, None, ws)
def bwd(): H.cdk_loss_backward(v, B, L, True, None, gf, gg, ws)

NIT = int(os.environ.get("NIT", "4000"))
def timeit(fn, n=None):
    n = n or NIT
    for _ in range(NIT): fn()  # ~0.2 s: let the clocks ramp
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

Lp = L + 1
fl_f = 2 * 2 * Lp * Lp * B + 2 * B * B * Lp
fl_b = 2 * 2 * B * Lp * Lp
tf, tn, tb = timeit(fwd), timeit(fwd_nogram), timeit(bwd)
print(f"B={B} L={L}: fwd {tf:.1f} us ({fl_f / tf * 1e-6:.1f} TF/s)  fwd(no gram) {tn:.1f} us  bwd {tb:.1f} us ({fl_b / tb * 1e-6:.1f} TF/s)")
# torch eager comparison of the same math on the GPU (rocBLAS fp32)
def eager():
    one = torch.ones(B, 1, device=dev)
    ft, gt = torch.cat([one, f], 1), torch.cat([one, g], 1)
    lf, lg = ft.T @ ft / B, gt.T @ gt / B
    lm = (M * lf * lg).sum()
    lo = -2 * torch.einsum('l,bl,bl->b', v, ft, gt).mean()
    gram = ft @ gt.T
    a = -(2 / B) * gt * v + (2 / B) * ft @ (M * lg)
    b = -(2 / B) * ft * v + (2 / B) * gt @ (M * lf)
    return lm + lo, gram.diag(), a[:, 1:], b[:, 1:]
print(f"torch eager (rocBLAS) same math: {timeit(eager, max(NIT // 8, 10)):.1f} us")
