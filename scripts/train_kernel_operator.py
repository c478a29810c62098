"""End-to-end check of the kernel-operator path (NestedLoRA.compute_loss_kernel on DenseKernelOperator, §3.5): learn
the top-L eigenfunctions of a Gaussian kernel matrix on N random 2-D points and compare the Rayleigh quotients of the
learned functions with numpy's eigendecomposition of K / N. (cfg4's own operator, K = A A^T / r with A independent of
the coordinates, is a throughput workload: its eigenvectors are not functions of z, nothing can learn them.)

    python scripts/train_kernel_operator.py [--steps 20000]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace as NS

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from neural_svd_amd.kernel_ops import DenseKernelOperator
from neural_svd_amd.models import get_wavefunctions
from neural_svd_amd.nested_lowrank import get_evd_method


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--N", type=int, default=4000)
    ap.add_argument("--L", type=int, default=8)
    ap.add_argument("--B", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--scale", type=float, default=20.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    z = torch.randn(a.N, 2, generator=g)
    d2 = (z[:, None, :] - z[None, :, :]).pow(2).sum(-1)
    K = (a.scale * torch.exp(-d2 / (2 * 0.75 ** 2))).double()  # scaled: the loss pulls |f_l|^2 to lambda_l
    ev = np.linalg.eigvalsh((K / a.N).numpy())[::-1][:a.L].copy()
    op = DenseKernelOperator(K.float().to(dev), z.to(dev))
    args = NS(ndim=2, n_particles=1, use_fourier_feature=True, fourier_mapping_size=64, fourier_scale=0.3,
              fourier_deterministic=False, fourier_append_raw=False, mlp_hidden_dims="128,128,128", neigs=a.L, parallel=1,
              nonlinearity="softplus", apply_exp_mask=0, exp_mask_init_scale=1.0, hard_mul_const=1.0, apply_boundary=0,
              sort=0, loss=NS(neuralsvd=NS(step=1, sequential=True)))
    torch.manual_seed(0)
    net = get_wavefunctions(args).to(dev)
    method = get_evd_method(args, "neuralsvd", op.index_model(net)).to(dev)
    opt = torch.optim.RMSprop(method.parameters(), lr=a.lr, alpha=0.999, eps=1e-10)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=a.steps)
    gen = torch.Generator(device=dev).manual_seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.steps):
        idx = op.sample_indices(a.B, gen)
        opt.zero_grad(set_to_none=True)
        loss, _ = method.compute_loss_kernel(op.get_approx_kernel_op, idx, None, split_batch=False)
        loss.backward()
        opt.step()
        sched.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # Rayleigh quotients on ALL points: lambda_l = f_l^T (K / N) f_l / (f_l^T f_l)
    with torch.no_grad():
        allidx = torch.arange(a.N, device=dev)
        F = torch.cat([method(allidx[i:i + 1024]) for i in range(0, a.N, 1024)]).double().cpu()
    KF = (K / a.N) @ F
    rq = ((F * KF).sum(0) / (F * F).sum(0)).numpy()
    rel = np.abs(rq - ev) / ev
    rec = dict(N=a.N, L=a.L, B=a.B, steps=a.steps, seconds=round(dt, 1), steps_per_s=round(a.steps / dt, 1),
               eig_numpy=[float(v) for v in ev], rayleigh_learned=[float(v) for v in rq],
               rel_err=[float(v) for v in rel], rel_err_mean=float(rel.mean()), loss=float(loss.detach()))
    print(json.dumps(rec))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
