"""(2a) of BASELINE.json's metric: eigenvalues of the SAME weights on the SAME grid, HIP float32 path vs the float64
CPU oracle, at the configs[1] model size. The weights are the EMA weights after --steps optimiser steps of the fused
trainer; the grid is arange(-50, 50, --val-eps)^2 (0.4 -> 62 500 points keeps the float64 oracle to about a minute).

    python scripts/parity_spectrum_cfg2.py --steps 20000 --out gpurun_out/parity_spectrum_cfg2.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
from oracle import nsvd_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--val-eps", type=float, default=0.4)
    ap.add_argument("--laplacian-eps", type=float, default=0.01, help="0: exact-Laplacian mode (no stencil noise)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = 16
    shape = H.ModelShape(L=L, D=2, m=1024, hidden=(128, 128, 128))
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, a.laplacian_eps, 100.0, 0.0, 16.0)
    tr = FusedTrainer(shape, prob, 512, sequential=False, step=1, lr=1e-4, num_iters=a.steps, seed=0, device=dev)
    for _ in range(a.steps):
        tr.step()
    torch.cuda.synchronize()
    out = {}
    for name, path in (("fp32", H.PATH_AUTO), ("bf16x3", H.PATH_FUSED_BF16X3)):
        tr.path = path
        out[name] = tr.spectrum(50.0, a.val_eps, use_ema=True)["eigvals"].numpy()
    sd = tr.P.state_dict(ema=True)
    nl = 4
    p64 = O.Params([sd[f"model.base.ws.{i}"].double().cpu() for i in range(nl)],
                   [sd[f"model.base.bs.{i}"].double().cpu() for i in range(nl)],
                   sd["model.base.feature_map._B"].double().cpu(), None)
    prob_o = O.Problem(potential=O.POT_HYDROGEN, charge_or_k=1.0, eps=a.laplacian_eps, op_scale=100.0, op_shift=0.0,
                       sigma=16.0)
    ax = np.arange(-50.0, 50.0, a.val_eps)
    xx = np.meshgrid(ax, ax)
    grid = torch.tensor(np.array(list(zip(*[v.flatten() for v in xx])))).float().double()  # the float32 grid values
    t0 = time.perf_counter()
    ref = O.spectrum_evd(grid, p64, prob_o, 50.0)
    tor = time.perf_counter() - t0
    e64 = np.asarray(ref["eigvals"], dtype=np.float64)
    # yardstick: the same oracle in float32 (what the float32 reference computes; its finite-difference Laplacian
    # carries a per-point error of ~ |f| at eps = 0.01)
    p32 = p64.to(torch.float32)
    e32 = np.asarray(O.spectrum_evd(grid.float(), p32, prob_o, 50.0)["eigvals"], dtype=np.float64)
    out["oracle_f32"] = e32
    rec = dict(what="eigenvalues of identical (EMA) weights on an identical grid: HIP float32 vs float64 oracle",
               steps=a.steps, laplacian_eps=a.laplacian_eps, grid_points=int(grid.shape[0]), val_eps=a.val_eps, oracle_seconds=round(tor, 1),
               eig_oracle_f64=[float(v) for v in e64])
    for name, e in out.items():
        r = np.abs(e - e64) / np.abs(e64)
        rec[name] = dict(eigvals=[float(v) for v in e], rel_diff_max=float(r.max()), rel_diff_mean=float(r.mean()))
        print(name, "max rel diff %.2e mean %.2e" % (r.max(), r.mean()))
    print("oracle eigvals", np.round(e64, 4).tolist(), "(%.0f s)" % tor)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
