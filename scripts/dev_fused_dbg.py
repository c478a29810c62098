import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H
from oracle import nsvd_oracle as O
dev = "cuda:0"
for (L, B, m, hidden) in [(2, 64, 1024, (128,)), (2, 64, 512, (128,)), (2, 64, 768, (128,)), (1, 32, 1024, (128,))]:
    p = O.init_params(L, 2, m, hidden, 0.1, seed=0)
    prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.5, sigma=16.0)
    torch.manual_seed(1)
    x = 16 * torch.randn(B, 2)
    c = O.operator_forward(x.double(), p.to(torch.float64), prob_o)
    shape = H.ModelShape(L=L, D=2, m=m, hidden=hidden)
    ws = [w.to(dev) for w in p.ws]; bs = [b.to(dev) for b in p.bs]
    pp = H.pack_params(shape, ws, bs, p.fourier_B.to(dev), None)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.5, 16.0)
    wsb = H.new_workspace(shape, B, dev)
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    for path, nm in ((H.PATH_GENERIC, "generic"), (H.PATH_FUSED, "fused")):
        f, Tf = H.operator_forward(shape, pp, prob, x.to(dev), wsb, False, path)
        torch.cuda.synchronize()
        err = (f.double().cpu() - c.f).abs()
        print(f"m={m} L={L} B={B} {nm}: f rel {rel(f, c.f):.2e}; worst rows {err.max(1).values.topk(3).indices.tolist()} per-head {err.max(0).values.tolist()}")
