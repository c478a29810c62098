"""compute_spectrum_evd with the reference's signature (methods/spectrum.py:29-102): streaming
cov = Phi^T Phi / n, quad = Phi^T T Phi / n on the validation grid, Rayleigh-quotient eigenvalues
diag(quad)/diag(cov) and norms diag(cov). The operator application and the accumulation are HIP
kernels (nsvd_operator_forward, nsvd_spectrum_accumulate)."""
from __future__ import annotations

import numpy as np
import torch

from . import hip_ops as H
from ._lib import NsvdError
from .operators import GaussianImportance, UniformBoxImportance


@torch.no_grad()
def compute_spectrum_evd(model, dataloader, operator, importance_train=None, importance_val=None,
                         set_first_mode_const=False, post_align=False, normalize=False, sort=False, gpu=None,
                         device=None):
    if (gpu is None) == (device is None):
        raise ValueError("exactly one of gpu / device")
    if not isinstance(importance_val, UniformBoxImportance):
        raise NsvdError("HIP path: importance_val must be UniformBoxImportance (the validation grid's density)")
    # Gaussian (or no) training density: the accumulation kernel evaluates sqrt(p_train) itself; any other callable
    # density (Laplace / uniform samplers, main_pde.py:101-118): the rows are weighted here, the kernel divides by
    # sqrt(p_val) only
    in_kernel = importance_train is None or isinstance(importance_train, GaussianImportance)
    dev = torch.device(f"cuda:{gpu}") if gpu is not None else torch.device(device)
    L = model.neigs + int(bool(set_first_mode_const))
    # float64 accumulators, rounded to the reference's float32 once at the end (include/nsvd.h:
    # nsvd_spectrum_accumulate_f64 - a float32 running sum carries an excited state's quotient to ~1e-4 only)
    cov = torch.zeros((L, L), dtype=torch.float64, device=dev)
    quad = torch.zeros_like(cov)
    eigfuncs, n = [], 0
    sigma = importance_train.sigma if (importance_train is not None and in_kernel) else 1.0
    for (x, _) in dataloader:
        if isinstance(x, list):
            x = x[0]
        x = x.to(dev).reshape(x.shape[0], -1).float().contiguous()
        Tphi, phi = operator(model, x, importance=importance_train)
        sw = importance_train(x).sqrt() if importance_train is not None else 1.0
        eigfuncs.append(sw * phi)
        if not in_kernel:
            phi, Tphi = sw * phi, sw * Tphi
        H.spectrum_accumulate(phi.float().contiguous(), Tphi.float().contiguous(), x, sigma,
                              importance_train is not None and in_kernel, importance_val.lim, cov, quad,
                              first_mode_const=bool(set_first_mode_const))
        n += len(x)
    out = dict()
    cov64 = (cov / n).cpu().numpy()
    quad64 = (quad / n).cpu().numpy()
    cov, quad = cov64.astype(np.float32), quad64.astype(np.float32)
    eigfuncs = torch.cat(eigfuncs, dim=0).cpu().numpy()
    out["eigfuncs"], out["cov"], out["quad"] = eigfuncs, cov, quad
    out["eigvals"] = eigvals = (np.diag(quad64) / np.diag(cov64)).astype(np.float32)
    out["norms"] = norms = np.diag(cov)
    if normalize:
        s = np.sqrt(norms)
        out["cov"] = cov / (s[:, None] * s[None, :])
        out["eigfuncs"] = eigfuncs / s.reshape(1, -1)
    if sort:
        idx = np.argsort(eigvals)[::-1]
        out["eigvals"] = out["eigvals"][idx]
        out["eigfuncs"] = out["eigfuncs"][:, idx, ...]
        out["cov"] = out["cov"][:, idx][idx, :]
        out["quad"] = out["quad"][:, idx][idx, :]
        out["norms"] = out["norms"][idx]
    if post_align:
        out["eigfuncs_aligned"], out["eigvals_aligned"], out["cov_aligned"] = post_alignment(
            out["eigfuncs"], out["cov"], out["quad"])
    return out


def post_alignment(eigfuncs, cov, quad):
    """methods/spectrum.py:161-169: whiten with cov^{-1/2}, diagonalise the whitened quad; the aligned functions
    eigfuncs @ (V^T cov^{-1/2})^T, sqrt of the (descending) eigenvalues, and the identity as their Gram matrix."""
    from scipy.linalg import eigh
    ec, vc = eigh(cov)
    whitening = vc @ np.diag(1 / np.sqrt(ec)) @ vc.T
    ev, V = eigh(whitening @ quad @ whitening)
    ev = np.sqrt(ev[::-1])
    V = V[:, ::-1]
    return eigfuncs @ (V.T @ whitening).T, ev, np.eye(quad.shape[0])
