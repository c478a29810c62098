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
    if set_first_mode_const or post_align:
        raise NotImplementedError("set_first_mode_const / post_align: not used by the PDE path")
    if not isinstance(importance_val, UniformBoxImportance):
        raise NsvdError("HIP path: importance_val must be UniformBoxImportance (the validation grid's density)")
    if importance_train is not None and not isinstance(importance_train, GaussianImportance):
        raise NsvdError("HIP path: importance_train must be None or GaussianImportance")
    dev = torch.device(f"cuda:{gpu}") if gpu is not None else torch.device(device)
    L = model.neigs
    # float64 accumulators, rounded to the reference's float32 once at the end (include/nsvd.h:
    # nsvd_spectrum_accumulate_f64 - a float32 running sum carries an excited state's quotient to ~1e-4 only)
    cov = torch.zeros((L, L), dtype=torch.float64, device=dev)
    quad = torch.zeros_like(cov)
    eigfuncs, n = [], 0
    sigma = importance_train.sigma if importance_train is not None else 1.0
    for (x, _) in dataloader:
        if isinstance(x, list):
            x = x[0]
        x = x.to(dev).reshape(x.shape[0], -1).float().contiguous()
        Tphi, phi = operator(model, x, importance=importance_train)
        sw = importance_train(x).sqrt() if importance_train is not None else 1.0
        eigfuncs.append(sw * phi)
        H.spectrum_accumulate(phi, Tphi, x, sigma, importance_train is not None, importance_val.lim, cov, quad)
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
    return out
