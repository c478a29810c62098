"""GPU parity of ``neural_svd_amd.cdk.normalize`` / ``HeteroNetwork`` (reference examples/models/siam.py:132-183; the
row-wise l2_ball / l2_sphere projection of the CDK towers runs on nsvd_row_normalize_forward / _backward) against the
golden vectors captured from the reference (tests/golden/normalize.npz, float64 values) and against the CPU oracle.
Tolerance: float32 elementwise arithmetic vs float64 truth, 2e-6 relative (L2)."""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-6


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("mode", ["l2_ball", "l2_sphere"])
@pytest.mark.parametrize("case", list("abcd"))
def test_normalize_golden(case, mode):
    from neural_svd_amd.cdk import normalize
    z = G.load("normalize")
    B, L, r = z[f"norm_{case}_cfg"]
    x = torch.tensor(z[f"norm_{case}_z"]).float().to(DEV).requires_grad_(True)
    y = normalize(x, float(r), mode)
    y.backward(torch.tensor(z[f"norm_{case}_dout"]).float().to(DEV))
    assert rel(y.detach(), z[f"norm_{case}_{mode}_f64_out"]) <= TOL
    assert rel(x.grad, z[f"norm_{case}_{mode}_f64_dz"]) <= TOL
    assert torch.equal(y[0].detach().cpu(), torch.zeros(int(L)))  # the all-zero row stays zero (no 0/0)


@pytest.mark.parametrize("B,L,mode", [(1024, 512, "l2_ball"), (1000, 513, "l2_ball"), (257, 3, "l2_sphere")])
def test_normalize_vs_oracle(B, L, mode):
    from neural_svd_amd.cdk import normalize
    g = torch.Generator().manual_seed(B + L)
    r = 4.0
    z64 = torch.randn(B, L, generator=g, dtype=torch.float64) * (1.5 * r / np.sqrt(L)) * \
        (0.3 + 1.4 * torch.rand(B, 1, generator=g, dtype=torch.float64))
    d64 = torch.randn(B, L, generator=g, dtype=torch.float64)
    zr = z64.clone().requires_grad_(True)
    yr = O.row_normalize(zr, r, mode)
    yr.backward(d64)
    x = z64.float().to(DEV).requires_grad_(True)
    y = normalize(x, r, mode)
    y.backward(d64.float().to(DEV))
    # rows within float32 rounding of the radius may take the other branch: compare the rest
    nrm = z64.norm(dim=1)
    ok = ((nrm - r).abs() > 1e-5 * r) if mode == "l2_ball" else torch.ones(B, dtype=torch.bool)
    assert ok.float().mean() > 0.99
    assert rel(y.detach().cpu()[ok], yr.detach()[ok]) <= TOL
    assert rel(x.grad.cpu()[ok], zr.grad[ok]) <= TOL


def test_hetero_network_mirror_and_other_modes():
    """HeteroNetwork: same constructor / outputs as the reference's; half-precision embeddings come back in their own
    dtype; 'clip' and 'tanh' are the reference's single torch ops; r_up <= 0 is the identity."""
    import torch.nn as nn
    from neural_svd_amd.cdk import HeteroNetwork, normalize

    class Tower(nn.Linear):
        output_dim = 16

    torch.manual_seed(0)
    net = HeteroNetwork([Tower(8, 16), Tower(8, 16)], [nn.Identity(), nn.Identity()], mu=4.0,
                        regularize_mode="l2_ball").to(DEV)
    x, y = torch.randn(32, 8, device=DEV) * 3, torch.randn(32, 8, device=DEV) * 3
    xr, xe, yr_, ye = net(x, y)
    assert net.output_dims == {"x": 16, "y": 16} and xe.shape == (32, 16)
    want = O.row_normalize(xr.detach().double().cpu(), 2.0, "l2_ball")
    assert rel(xe.detach(), want) <= TOL and float(xe.detach().norm(dim=1).max()) <= 2.0 * (1 + 1e-6)
    (xe.sum() + ye.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    h = normalize(xr.detach().half(), 2.0, "l2_sphere")
    assert h.dtype == torch.float16 and abs(float(h.float().norm(dim=1).mean()) - 2.0) < 1e-2
    assert torch.equal(normalize(xr, 0.0, "l2_ball"), xr)
    assert torch.equal(normalize(xr, 1.5, "clip"), torch.clip(xr, -1.5, 1.5))
    assert torch.allclose(normalize(xr, 1.5, "tanh"), 1.5 * torch.tanh(xr))


def test_sketchy_style_towers_end_to_end():
    """The model construction of examples/cdk/sketchy/main_sketchy.py:107-116 with this package's names only
    (get_mlp, HeteroNetwork, get_cdk_method-style NestedLoRAForCDK): one loss + backward, finite gradients everywhere,
    embeddings inside the ball of radius sqrt(mu)."""
    import torch.nn as nn
    from neural_svd_amd.cdk import HeteroNetwork, NestedLoRAForCDK, get_mlp
    torch.manual_seed(1)
    sizes = [64, 256, 32]
    model = HeteroNetwork(backbones=[get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                                     get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                          projectors=[nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(DEV)
    assert model.output_dims == {"x": 32, "y": 32}
    kinds = [type(m).__name__ for m in model.backbones["x"]]
    assert kinds == ["Linear", "BatchNorm1d", "LeakyReLU", "Linear", "BatchNorm1d"]
    method = NestedLoRAForCDK(model, neigs=32, step=1, sequential=False, set_first_mode_const=True).to(DEV)
    x, y = torch.randn(128, 64, device=DEV), torch.randn(128, 64, device=DEV)
    _, fx, _, fy = method(x, y)
    assert float(fx.detach().norm(dim=1).max()) <= 4.0 * (1 + 1e-6)
    out = method.compute_loss(fx, fy)
    loss = out[0] if isinstance(out, (tuple, list)) else out
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
