#!/usr/bin/env python3
"""Build-container only (needs /root/reference): time the imported REFERENCE training step and
oracle/torch_port.py (what bench.py reports as cpu_baseline) side by side, same weights, same batches.

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_port_vs_reference.py [B]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402  (installs the stubs and imports the reference)
import torch  # noqa: E402
from oracle import nsvd_oracle as O  # noqa: E402
from oracle import torch_port as TP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.set_num_threads(os.cpu_count())
over = dict(potential_type="hydrogen", neigs=16, mlp_hidden_dims="128,128,128", fourier_mapping_size=1024,
            fourier_scale=0.1, sampling_scale=16.0, batch_size=B, operator_scale=100.0, operator_shift=0.0,
            sequential=0, seed=0, num_iters=500000)
args = MG.make_args(**over)
operator, gt, method, *_ = MG.build(args, torch.float32)
imp = MG.importance_for(args, torch.float32)
opt = MG.get_optimizer(args, method)
sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, args.num_iters)

p = O.init_params(16, 2, 1024, (128, 128, 128), 0.1, seed=0)
prob = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
v, M = O.joint_nesting_masks(16, 1)
port = TP.PortStep(p, prob, v, M, lr=1e-4, num_iters=500000)

g = torch.Generator().manual_seed(1)
xs = [16.0 * torch.randn(B, 2, generator=g) for _ in range(8)]


def ref_step(x):
    opt.zero_grad()
    loss, _ = method.compute_loss_operator(operator, x, importance=imp)
    loss.backward()
    opt.step()
    sched.step()
    return float(loss)


def port_step(x):
    return float(port.step(x))


for name, fn in (("reference", ref_step), ("port", port_step)):
    fn(xs[0])
    ts = []
    for x in xs[1:]:
        t0 = time.perf_counter()
        l = fn(x)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name:10s} B={B}: median {ts[len(ts)//2]*1e3:8.1f} ms/step  ({1/ts[len(ts)//2]:.3f} step/s)  last loss {l:.4f}")
