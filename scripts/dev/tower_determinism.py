"""dev: one mixed-precision tower forward / backward repeated N times on the same inputs, compared bit for bit"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from neural_svd_amd import hip_ops as H  # noqa: E402

dev = torch.device("cuda:0")
B, d0, d1, d2 = [int(v) for v in (sys.argv[2:6] or (256, 128, 256, 256))]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(0)
P = dict(W1=torch.randn(d1, d0, generator=g) / d0 ** 0.5, b1=0.1 * torch.randn(d1, generator=g),
         g1=1.0 + 0.3 * torch.randn(d1, generator=g), be1=0.2 * torch.randn(d1, generator=g),
         W2=torch.randn(d2, d1, generator=g) / d1 ** 0.5, b2=0.1 * torch.randn(d2, generator=g),
         g2=1.0 + 0.3 * torch.randn(d2, generator=g), be2=0.2 * torch.randn(d2, generator=g))
P = {k: v.to(dev).contiguous() for k, v in P.items()}
x, dz = torch.randn(B, d0, generator=g).to(dev), torch.randn(B, d2, generator=g).to(dev)
for fresh_ws in (False, True):
    ws = H.tower_workspace(B, d0, d1, d2, dev)
    ref, badf, badb = None, 0, 0
    for rep in range(N):
        if fresh_ws:
            ws = torch.full_like(ws, rep * 37 % 251)  # a different garbage pattern every time
        z = H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=1)
        gr = H.tower_backward(x, P, dz, 0.2, ws, gemm_bf16=1)
        torch.cuda.synchronize()
        cur = (z.cpu(), {k: v.cpu() for k, v in gr.items()})
        if ref is None:
            ref = cur
            continue
        if not torch.equal(cur[0], ref[0]):
            badf += 1
        db = [k for k in ref[1] if not torch.equal(cur[1][k], ref[1][k])]
        if db:
            badb += 1
            if badb <= 3:
                print(f"  rep {rep}: gradients differ: {db}, max {max(float((cur[1][k] - ref[1][k]).abs().max()) for k in db):.3e}")
    print(f"fresh workspace contents each call: {fresh_ws}: forward differs {badf}, backward differs {badb} of {N - 1}")
