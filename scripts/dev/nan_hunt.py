"""dev: find the first step at which the cfg2 training run produces non-finite values, and where."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, float(os.environ.get("EPS", "0.01")), 100.0, 0.0, 16.0)
kw = dict(fused_step=os.environ.get("FUSED", "1") == "1", path={"auto": H.PATH_AUTO, "generic": H.PATH_GENERIC}[os.environ.get("PATHN", "auto")])
tr = FusedTrainer(shape, prob, 512, sequential=False, step=1, lr=1e-4, num_iters=int(os.environ.get("ITERS", "20000")), seed=0, device=dev, **kw)
start = int(os.environ.get("START", "0"))
for _ in range(start):
    tr.step()
torch.cuda.synchronize()
print("at", start, "params finite", bool(torch.isfinite(tr.P.flat).all()))
every = int(os.environ.get("EVERY", "50"))
for it in range(start, 8000, every):
    for _ in range(every):
        tr.step()
    torch.cuda.synchronize()
    fin_p = bool(torch.isfinite(tr.P.flat).all()); fin_f = bool(torch.isfinite(tr.f).all()); fin_T = bool(torch.isfinite(tr.Tf).all())
    if it % 500 == 0 or not (fin_p and fin_f and fin_T):
        print(it + every, "loss", [round(float(v), 3) for v in tr.loss], "finite p/f/Tf", fin_p, fin_f, fin_T,
              "max|p|", float(tr.P.flat.abs().max()), "max|f|", float(tr.f.abs().max()), "max|Tf|", float(tr.Tf.abs().max()),
              "min|x|", float(tr.x.norm(dim=1).min()), flush=True)
    if not (fin_p and fin_f and fin_T):
        names = ["fourier_B"] + [f"W{i}" for i in range(4)] + [f"b{i}" for i in range(4)]
        bad = ~torch.isfinite(tr.P.flat)
        print("non-finite params:", int(bad.sum()), "of", bad.numel(), "first idx", int(bad.nonzero()[0]) if bad.any() else -1)
        print("sq finite", bool(torch.isfinite(tr.P.sq).all()), "ema finite", bool(torch.isfinite(tr.P.ema).all()))
        badf = ~torch.isfinite(tr.f); badT = ~torch.isfinite(tr.Tf)
        print("non-finite f:", int(badf.sum()), "Tf:", int(badT.sum()), "rows:", badT.any(dim=1).nonzero().flatten()[:10].tolist(),
              "cols:", badT.any(dim=0).nonzero().flatten()[:16].tolist())
        rows = (badf | badT).any(dim=1).nonzero().flatten()[:5]
        print("x of bad rows:", tr.x[rows].tolist(), "|x|", tr.x[rows].norm(dim=1).tolist())
        break
