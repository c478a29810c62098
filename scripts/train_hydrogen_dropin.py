"""configs[1] through the REFERENCE API end to end: get_problem / get_wavefunctions / get_evd_method / get_dataloader /
train_operator with the hyper-parameters of scripts/exps/pde/hydrogen.sh (neigs 16, batch 512), evaluation by
compute_spectrum_evd under the EMA weights inside train_operator, relative eigenvalue error of its last evaluation.

    python scripts/train_hydrogen_dropin.py --steps 500000 --out gpurun_out/train_cfg2_dropin.json [--sequential]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from neural_svd_amd.drop_in import train_operator
from neural_svd_amd.models import get_wavefunctions
from neural_svd_amd.nested_lowrank import get_evd_method
from neural_svd_amd.operators import get_dataloader, get_problem


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=500000)
    ap.add_argument("--eval-freq", type=int, default=100000)
    ap.add_argument("--sequential", action="store_true")
    ap.add_argument("--plain-loop", action="store_true", help="torch autograd loop body instead of the fused one")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--problem", default="hydrogen", choices=["hydrogen", "oscillator"],
                    help="oscillator: configs[2] on one GPU (scripts/exps/pde/oscillator.sh with neigs 32, batch 4096)")
    ap.add_argument("--out", default=None)
    o = ap.parse_args()
    dev = "cuda:0"
    osc = o.problem == "oscillator"
    a = argparse.Namespace(
        problem="sch", potential_type="hydrogen", charge=1.0, ndim=2, n_particles=1, neigs=16, laplacian_eps=0.01,
        operator_scale=100.0, operator_shift=0.0, sampling_mode="gaussian", sampling_scale=16.0, batch_size=512, lim=50.0,
        val_eps=0.1, use_fourier_feature=True, fourier_mapping_size=1024, fourier_scale=0.1, fourier_deterministic=False,
        fourier_append_raw=False, mlp_hidden_dims="128,128,128", parallel=1, nonlinearity="softplus", apply_exp_mask=0,
        exp_mask_init_scale=1.0, hard_mul_const=1.0, apply_boundary=0, sort=0, optimizer="rmsprop", lr=1e-4,
        rmsprop_decay=0.999, momentum=0.0, adam_eps=1e-7, num_iters=o.steps, ema_decay=0.995, use_lr_scheduler=True,
        print_freq=10 ** 9, eval_freq=o.eval_freq, log_dir=None, fused_loop=not o.plain_loop)
    if osc:
        vars(a).update(potential_type="harmonic_oscillator", neigs=32, operator_scale=1.0, operator_shift=16.0,
                       sampling_scale=4.0, batch_size=4096, lim=5.0, fourier_mapping_size=256, fourier_scale=1.0,
                       apply_exp_mask=1, exp_mask_init_scale=10.0)
    a.loss = argparse.Namespace(name="neuralsvd", neuralsvd=argparse.Namespace(step=1, sequential=o.sequential))
    torch.manual_seed(o.seed)
    operator, gt = get_problem(a, dev)
    model = get_wavefunctions(a)
    make_batch, val_data, batch_ftn_val, imp_train, imp_val = get_dataloader(a, dev)
    method = get_evd_method(a, "neuralsvd", model).to(dev)
    t0 = time.perf_counter()
    eigs, norms = train_operator(a, method, operator, make_batch, val_data, batch_ftn_val, None, None, dev, imp_train,
                                 imp_val, ground_truth_spectrum=gt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev = np.asarray(eigs[-1], dtype=np.float64)
    L = a.neigs
    gt = np.asarray(gt, dtype=np.float64)[:L]
    nz = np.abs(gt) > 0  # the oscillator's 8th shell sits at eigenvalue 0: no relative error there
    rel = (np.abs(ev - gt) / np.where(nz, np.abs(gt), 1.0))[nz]
    rec = dict(api="drop_in.train_operator (fused loop body)" if a.fused_loop else "drop_in.train_operator (plain loop body)",
               nesting="sequential" if o.sequential else "joint", steps=o.steps, evaluations=len(eigs),
               wall_seconds_including_evaluations=round(dt, 1), eigvals=[float(v) for v in ev],
               ground_truth=[float(v) for v in gt], problem=o.problem, seed=o.seed, rel_err_mean=float(rel.mean()), rel_err_max=float(rel.max()))
    print(json.dumps(rec))
    if o.out:
        os.makedirs(os.path.dirname(o.out), exist_ok=True)
        json.dump(rec, open(o.out, "w"), indent=1)


if __name__ == "__main__":
    main()
