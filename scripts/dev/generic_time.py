"""dev: the generic kernels (any hidden width, any batch) - time of forward + loss + backward at configs[1]'s sizes with
hidden width 128 (forced onto the generic path) and at widths the MFMA kernels do not take (64, 256, 96)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
dev = "cuda:0"
cases = (((128, 128, 128), H.PATH_GENERIC), ((64, 64, 64), H.PATH_AUTO), ((256, 256, 256), H.PATH_AUTO), ((96, 96), H.PATH_AUTO))
if len(sys.argv) > 1:  # one width only (for a profile): generic_time.py 64
    cases = [c for c in cases if c[0][0] == int(sys.argv[1])]
for hidden, path in cases:
    L, D, m, B = 16, 2, 1024, 512
    shape = H.ModelShape(L=L, D=D, m=m, hidden=hidden)
    g = torch.Generator().manual_seed(0)
    dims = [2 * m] + list(hidden) + [1]
    ws_t = [(torch.randn(L, dims[i + 1], dims[i], generator=g) * (2.0 / dims[i]) ** 0.5).to(dev) for i in range(len(dims) - 1)]
    bs_t = [torch.zeros(L, dims[i + 1], 1, device=dev) for i in range(len(dims) - 1)]
    fB = (0.1 * 6.2831853 * torch.randn(D, m, generator=g)).to(dev)
    params = H.pack_params(shape, ws_t, bs_t, fB, None)
    grads = H.pack_params(shape, [torch.empty_like(w) for w in ws_t], [torch.empty_like(b) for b in bs_t], None, None)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    x = (16.0 * torch.randn(B, D, generator=g)).to(dev)
    ws = H.new_workspace(shape, B, dev)
    name = H.path_name(shape, B, path, prob)
    def step():
        f, Tf = H.operator_forward(shape, params, prob, x, ws, True, path)
        mom = H.evd_moments(f, Tf, H.MASK_JOINT, None)
        loss, df = H.evd_loss_grad(f, Tf, H.MASK_JOINT, None, None, mom)
        H.operator_backward(shape, params, prob, x, df, grads, ws, path=path)
    for _ in range(20): step()
    n, blocks = 40, []
    for _ in range(5):  # median of five blocks (a single block now and then carries a one-off stall of the box)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / n)
    dt = sorted(blocks)[2]
    E = 1 + 2 * D
    M = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    flops = 2 * B * L * (E * M + M + (M - dims[0] * dims[1]))
    print(f"hidden {hidden} path {name}: {dt * 1e3:.3f} ms per forward + loss + backward = {flops / dt / 1e12:.1f} TFLOP/s")
