"""Eigenvalue parity at the configs[1] MODEL SIZE (BASELINE.json's "(2a)": same weights, same grid): the HIP float32
path against the float64 oracle, with the float32 oracle - the reference's own arithmetic - as the yardstick.

  * laplacian_eps = 0.01 (the scripts' setting): a float32 central difference at eps = 0.01 taken point-wise - the
    reference's arithmetic, the float32 oracle here - carries a per-point error of about |f| and
    its eigenvalues are percent-level noisy. The fused kernels carry the stencil in even / odd form and
    are held to north_star's 1e-4 against the FLOAT64 stencil on every eigenvalue (measured 3e-6 worst on the 62 500-point
    grid of scripts/parity_spectrum_cfg2.py), with the float32 oracle's own distance printed beside it.
  * laplacian_eps = 0 (exact Laplacian, reference diff_ops.py:54-61): nothing is differenced; the bar is 1e-5 relative
    on every eigenvalue (north_star asks for 1e-4).

Weights: the EMA weights after 1500 fused steps from the seed-0 reference initialisation (so the 16 functions are
not at their random start); grid arange(-50, 50, 1.0)^2 = 10^4 points (float64 oracle: ~20 s of CPU)."""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(laplacian_eps):
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, laplacian_eps, 100.0, 0.0, 16.0)
    tr = FusedTrainer(shape, prob, 512, sequential=False, step=1, lr=1e-4, num_iters=1500, seed=0, device=DEV)
    assert H.path_name(shape, 512, H.PATH_AUTO, prob) == "fused_mfma"
    for _ in range(1500):
        tr.step()
    torch.cuda.synchronize()
    sd = tr.P.state_dict(ema=True)
    p64 = O.Params([sd[f"model.base.ws.{i}"].double().cpu() for i in range(4)],
                   [sd[f"model.base.bs.{i}"].double().cpu() for i in range(4)],
                   sd["model.base.feature_map._B"].double().cpu(), None)
    prob_o = O.Problem(potential=O.POT_HYDROGEN, charge_or_k=1.0, eps=laplacian_eps, op_scale=100.0, op_shift=0.0,
                       sigma=16.0)
    ax = np.arange(-50.0, 50.0, 1.0)
    xx = np.meshgrid(ax, ax)
    grid = torch.tensor(np.array(list(zip(*[v.flatten() for v in xx])))).float().double()  # the float32 grid values
    return tr, p64, prob_o, grid


def _rel(e, e64):
    return np.abs(np.asarray(e, dtype=np.float64) - e64) / np.abs(e64)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("path", ["auto", "bf16x3"])
def test_eigenvalue_parity_stencil_1e4_of_float64(path):
    from neural_svd_amd import hip_ops as H
    tr, p64, prob_o, grid = _setup(0.01)
    e64 = np.asarray(O.spectrum_evd(grid, p64, prob_o, 50.0)["eigvals"], dtype=np.float64)
    e32 = O.spectrum_evd(grid.float(), p64.to(torch.float32), prob_o, 50.0)["eigvals"]
    tr.path = {"auto": H.PATH_AUTO, "bf16x3": H.PATH_FUSED_BF16X3}[path]
    got = tr.spectrum(50.0, 1.0, use_ema=True)["eigvals"].numpy()
    r_hip, r_ref = _rel(got, e64), _rel(e32, e64)
    assert np.all(np.isfinite(got))
    # Measured values go to gpurun_out/ (bench.py quotes the 62 500-point run of scripts/parity_spectrum_cfg2.py).
    import json
    import os
    rec = dict(grid_points=int(grid.shape[0]), path=path, hip_f32_vs_f64_mean=float(r_hip.mean()),
               hip_f32_vs_f64_max=float(r_hip.max()), oracle_f32_vs_f64_mean=float(r_ref.mean()),
               oracle_f32_vs_f64_max=float(r_ref.max()), north_star_tolerance=1e-4)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rec, open(os.path.join(out, f"parity_test_stencil_{path}.json"), "w"), indent=1)
    # north_star's tolerance, every eigenvalue, in the scripts' default mode (the reference's own float32 arithmetic is
    # two to three orders of magnitude further from float64: rec)
    assert r_hip.max() < 1e-4, rec
    assert r_hip.max() < r_ref.max(), rec


@pytest.mark.timeout(900)
@pytest.mark.parametrize("path", ["auto", "bf16x3"])
def test_eigenvalue_parity_exact_laplacian_1e5(path):
    from neural_svd_amd import hip_ops as H
    tr, p64, prob_o, grid = _setup(0.0)
    e64 = np.asarray(O.spectrum_evd(grid, p64, prob_o, 50.0)["eigvals"], dtype=np.float64)
    tr.path = {"auto": H.PATH_AUTO, "bf16x3": H.PATH_FUSED_BF16X3}[path]
    got = tr.spectrum(50.0, 1.0, use_ema=True)["eigvals"].numpy()
    assert _rel(got, e64).max() < 1e-5, _rel(got, e64).max()
