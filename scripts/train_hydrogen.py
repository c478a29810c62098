"""End-to-end run of BASELINE.json configs[1] (2D hydrogen, L = 16, batch 512, joint nesting) with the reference's
training schedule (scripts/exps/pde/hydrogen.sh: RMSprop lr 1e-4, cosine over --steps, EMA 0.995) and its evaluation
(Rayleigh quotients of the EMA model on the arange(-50, 50, 0.1)^2 grid, methods/spectrum.py:29-102), reporting the
relative eigenvalue error against the analytic spectrum -Z^2 / (4 (n + 1/2)^2) (x operator_scale = 100).

    python scripts/train_hydrogen.py --steps 500000 --out gpurun_out/train_cfg2.json [--path bf16x3] [--laplacian-eps 0]

--problem oscillator: BASELINE.json configs[2] on one GPU (2D harmonic oscillator, L = 32, the GLOBAL batch of 4096,
sequential nesting, exponential mask; scripts/exps/pde/oscillator.sh: 100000 steps, grid arange(-5, 5, 0.1)^2),
spectrum 16 - (2n + 2) with degeneracy n + 1 (modes 29..32 have eigenvalue 0: reported as absolute errors).
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
from neural_svd_amd.operators import HarmonicOscillator, Hydrogen2D
from neural_svd_amd.trainer import FusedTrainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="hydrogen", choices=["hydrogen", "oscillator"])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--evals", default="10000,50000,100000,250000,500000")
    ap.add_argument("--path", default="auto", choices=["auto", "bf16x3"])
    ap.add_argument("--laplacian-eps", type=float, default=0.01)
    ap.add_argument("--sequential", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--eval-exact", action="store_true",
                    help="also evaluate the trained weights with the exact Laplacian (separates the finite-difference "
                         "noise of the EVALUATION from what training converged to)")
    ap.add_argument("--neigs", type=int, default=None, help="oscillator: number of modes (oscillator.sh: 55)")
    ap.add_argument("--batch-size", type=int, default=None, help="override the configuration's batch size")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.problem == "hydrogen":
        a.steps = a.steps or 500000
        L, B, lim = 16, a.batch_size or 512, 50.0
        shape = H.ModelShape(L=L, D=2, m=1024, hidden=(128, 128, 128))
        prob = H.make_problem(H.POT_HYDROGEN, 1.0, a.laplacian_eps, 100.0, 0.0, 16.0)
        kw = dict(sampling_scale=16.0, fourier_scale=0.1)
        gt = 100.0 * -Hydrogen2D(1.0).get_eigvals(L)  # [100, 11.11 x3, 4 x5, 2.04 x7]
        label = "configs[1]: 2D hydrogen L=16 B=512 %s nesting" % ("sequential" if a.sequential else "joint")
    else:
        a.steps = a.steps or 100000
        a.sequential = True
        L, B, lim = a.neigs or 32, a.batch_size or 4096, 5.0
        shape = H.ModelShape(L=L, D=2, m=256, hidden=(128, 128, 128), has_exp_mask=True)
        prob = H.make_problem(H.POT_HARMONIC, 1.0, a.laplacian_eps, 1.0, 16.0, 4.0)
        kw = dict(sampling_scale=4.0, fourier_scale=1.0, exp_mask_init=10.0)
        gt = 16.0 - HarmonicOscillator(1.0, 2).get_eigvals(L)[:L]  # [14, 12 x2, 10 x3, ..., 2 x7, 0 x4]
        label = "configs[2] on one GPU: 2D oscillator L=%d B=%d sequential nesting, exponential mask" % (L, B)
    path = H.PATH_FUSED_BF16X3 if a.path == "bf16x3" else H.PATH_AUTO
    tr = FusedTrainer(shape, prob, B, sequential=a.sequential, step=1, lr=1e-4, rmsprop_decay=0.999, ema_decay=0.995,
                      num_iters=a.steps, seed=a.seed, device=dev, path=path, **kw)
    evals = sorted({int(e) for e in a.evals.split(",") if int(e) <= a.steps} | {a.steps})
    rec = dict(config="%s, lr 1e-4 cosine over %d steps, EMA 0.995, eps=%g, path=%s, seed %d"
               % (label, a.steps, a.laplacian_eps, a.path, a.seed), ground_truth=gt.tolist(), evals=[])
    done, t_train = 0, 0.0
    for target in evals:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(target - done):
            tr.step()
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        done = target
        sp = tr.spectrum(lim, 0.1, use_ema=True)
        ev = sp["eigvals"].numpy()
        nz = np.abs(gt) > 0  # the oscillator's 8th shell sits at eigenvalue 0: no relative error there
        rel = (np.abs(ev - gt) / np.where(nz, np.abs(gt), 1.0))[nz]
        rel_sorted = (np.abs(np.sort(ev)[::-1] - gt) / np.where(nz, np.abs(gt), 1.0))[nz]
        e = dict(step=done, train_seconds=round(t_train, 2), steps_per_s=round(done / t_train, 1),
                 loss=float(tr.loss[0]), eigvals=[round(float(v), 4) for v in ev],
                 rel_err_mean=float(rel.mean()), rel_err_max=float(rel.max()), rel_err_mean_sorted=float(rel_sorted.mean()),
                 rel_err_first4=float(rel[:4].mean()), abs_err_zero_modes=[round(float(v), 4) for v in (ev - gt)[~nz]])
        if a.eval_exact and a.laplacian_eps > 0:
            trained = tr.problem
            tr.problem = H.make_problem(trained.potential, trained.charge_or_k, 0.0, trained.op_scale, trained.op_shift,
                                        trained.sigma)
            evx = tr.spectrum(lim, 0.1, use_ema=True)["eigvals"].numpy()
            tr.problem = trained
            relx = (np.abs(evx - gt) / np.where(nz, np.abs(gt), 1.0))[nz]
            e.update(exact_eval_eigvals=[round(float(v), 4) for v in evx], exact_eval_rel_err_mean=float(relx.mean()),
                     exact_eval_rel_err_max=float(relx.max()))
        rec["evals"].append(e)
        print(json.dumps(e), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
