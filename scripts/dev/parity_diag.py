"""dev: where the remaining eigenvalue-parity error at configs[1] comes from - per-head errors of f and Tf on a sub-grid
against the float64 oracle, and Rayleigh quotients from HIP outputs accumulated in float64 on the host (no float32
accumulator, no atomics) beside the product spectrum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
from oracle import nsvd_oracle as O
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
tr = FusedTrainer(shape, prob, 512, sequential=False, step=1, lr=1e-4, num_iters=steps, seed=0, device=dev)
for _ in range(steps): tr.step()
torch.cuda.synchronize()
sd = tr.P.state_dict(ema=True)
p64 = O.Params([sd[f"model.base.ws.{i}"].double().cpu() for i in range(4)],
               [sd[f"model.base.bs.{i}"].double().cpu() for i in range(4)],
               sd["model.base.feature_map._B"].double().cpu(), None)
prob_o = O.Problem(potential=O.POT_HYDROGEN, charge_or_k=1.0, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
ax = np.arange(-50.0, 50.0, 0.4)
xx = np.meshgrid(ax, ax)
grid = torch.tensor(np.array(list(zip(*[v.flatten() for v in xx])))).float()
sub = grid[::7][:8192].contiguous()
ref = O.operator_forward(sub.double(), p64, prob_o)
# parameters as the trainer evaluates them (EMA weights)
ws = [sd[f"model.base.ws.{i}"].to(dev).contiguous() for i in range(4)]
bs = [sd[f"model.base.bs.{i}"].to(dev).contiguous() for i in range(4)]
params = H.pack_params(shape, ws, bs, sd["model.base.feature_map._B"].to(dev).contiguous(), None)
wsp = H.new_workspace(shape, 2048, dev)
w = (torch.exp(-(sub.double() ** 2).sum(1) / (4 * 16.0 ** 2))).unsqueeze(1)  # ~ sqrt p (constant factors cancel in the quotient)
def quot(f, Tf):
    f, Tf = f.double().cpu() * w, Tf.double().cpu() * w
    return ((f * Tf).sum(0) / (f * f).sum(0)).numpy()
q64 = quot(ref.f, ref.Tf)
for name, path in (("fp32", H.PATH_FUSED), ("bf16x3", H.PATH_FUSED_BF16X3)):
    outs = [H.operator_forward(shape, params, prob, sub[i:i + 2048].to(dev).contiguous(), wsp, False, path) for i in range(0, 8192, 2048)]
    outs = [(a.clone(), b.clone()) for a, b in outs] if False else outs
    torch.cuda.synchronize()
    f, Tf = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    ef = ((f.double().cpu() - ref.f).norm(dim=0) / ref.f.norm(dim=0)).numpy()
    et = ((Tf.double().cpu() - ref.Tf).norm(dim=0) / ref.Tf.norm(dim=0)).numpy()
    print(name, "f   rel err per head x1e7:", np.round(ef * 1e7, 1).tolist())
    print(name, "Tf  rel err per head x1e5:", np.round(et * 1e5, 2).tolist())
    q = quot(f, Tf)
    print(name, "quotient rel err x1e5 (float64 accumulation on the host):", np.round(np.abs(q - q64) / np.abs(q64) * 1e5, 2).tolist())
    # mixed: HIP f with oracle Tf, oracle f with HIP Tf
    print(name, "  with oracle Tf:", np.round(np.abs(quot(f, ref.Tf) - q64) / np.abs(q64) * 1e5, 2).tolist())
    print(name, "  with oracle f :", np.round(np.abs(quot(ref.f, Tf) - q64) / np.abs(q64) * 1e5, 2).tolist())
    # where the Tf error sits: by radius
    r = sub.double().norm(dim=1)
    for lo, hi in ((0, 10), (10, 25), (25, 45), (45, 80)):
        msk = (r >= lo) & (r < hi)
        d = (Tf.double().cpu() - ref.Tf)[msk]
        print(name, f"  r in [{lo},{hi}): Tf abs err rms {float(d.pow(2).mean().sqrt()):.3e}  |Tf| rms {float(ref.Tf[msk].pow(2).mean().sqrt()):.3e}  |f| rms {float(ref.f[msk].pow(2).mean().sqrt()):.3e}")
