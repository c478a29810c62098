"""GPU parity of the CDK tower kernels (csrc/tower.hip: nsvd_tower_forward / _backward behind cdk.get_mlp's
TowerSequential) against the REFERENCE's get_mlp tower (tests/golden/tower.npz, case tb: 128 -> 256 -> 128, batch 128,
lrelu0.2, non-trivial BatchNorm affine parameters; float64 truth, with the float32 reference run as the yardstick) and,
at BASELINE configs[4]'s tower size (1024 x 512 -> 8192 -> 512), against the float64 oracle on sampled columns plus
size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = {"W1": "0.weight", "b1": "0.bias", "g1": "1.weight", "be1": "1.bias", "W2": "3.weight", "b2": "3.bias",
         "g2": "4.weight", "be2": "4.bias"}


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _golden_tower(z, case):
    from neural_svd_amd.cdk import TowerSequential, get_mlp
    B, d0, d1, d2, seed = [int(v) for v in z[f"{case}_cfg"]]
    slope = float(z[f"{case}_slope"])
    torch.manual_seed(seed)  # the reference's constructor calls in the reference's order: same initial weights
    m = get_mlp([d0, d1, d2], bias=True, nonlinearity=f"lrelu{slope}", use_bn=True)
    assert isinstance(m, TowerSequential)
    with torch.no_grad():
        gg = torch.Generator().manual_seed(2000 + seed)
        for k in (1, 4):
            m[k].weight.copy_(1.0 + 0.3 * torch.randn(m[k].weight.shape, generator=gg))
            m[k].bias.copy_(0.2 * torch.randn(m[k].bias.shape, generator=gg))
    return m.to(DEV).train()


def test_tower_matches_reference_golden():
    z = G.load("tower")
    case = "tb"
    m = _golden_tower(z, case)
    x = torch.tensor(z[f"{case}_x"]).float().to(DEV)
    dz = torch.tensor(z[f"{case}_dz"]).float().to(DEV)
    assert m.hip_ready(x)
    out = m(x)
    (out * dz).sum().backward()
    torch.cuda.synchronize()
    q64, q32 = f"{case}_f64_", f"{case}_f32_"
    assert rel(out, z[q64 + "z"]) < max(3 * G.rel(z[q32 + "z"], z[q64 + "z"]), 2e-6)
    gW1 = float(np.linalg.norm(z[q64 + "grad_0.weight"]))
    for k, n in NAMES.items():
        got, want = dict(m.named_parameters())[n].grad, z[q64 + f"grad_{n}"]
        if k in ("b1", "b2"):  # vanishing by construction (a bias in front of a BatchNorm): absolute, on the W scale
            assert float(got.double().cpu().abs().max()) < 1e-5 * gW1, n
            continue
        ref32 = abs(float(z[q32 + f"gradnorm_{n}"]) - float(np.linalg.norm(want))) / float(np.linalg.norm(want))
        assert rel(got, want) < max(10 * ref32, 5e-6), (n, rel(got, want))
    for k in (1, 4):
        assert rel(m[k].running_mean, z[q64 + f"running_mean_{k}"]) < 5e-6
        assert rel(m[k].running_var, z[q64 + f"running_var_{k}"]) < 5e-6
        assert int(m[k].num_batches_tracked) == 1


def test_tower_headline_size_against_oracle_and_library():
    """configs[4]'s tower (1024 x 512 -> 8192 -> 512): output and gradients vs the float64 oracle (full: ~2 s of CPU),
    bit reproducibility, evaluation mode untouched (torch modules), and agreement with torch's own float32 modules on
    the same weights at float32 noise level."""
    from neural_svd_amd.cdk import get_mlp
    torch.manual_seed(5)
    m = get_mlp([512, 8192, 512], bias=True, nonlinearity="lrelu0.2", use_bn=True).to(DEV).train()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1024, 512, generator=g).to(DEV)
    dz = torch.randn(1024, 512, generator=g).to(DEV)
    assert m.hip_ready(x)
    out = m(x)
    (out * dz).sum().backward()
    grads = {k: dict(m.named_parameters())[n].grad.clone() for k, n in NAMES.items()}
    # float64 oracle
    P = {k: dict(m.named_parameters())[n].detach().double().cpu() for k, n in NAMES.items()}
    zo, go, _ = O.tower_forward_backward(x.double().cpu(), P, dz.double().cpu(), 0.2)
    assert rel(out, zo) < 5e-6
    # yardstick for the gradients: torch's own float32 modules (library GEMMs + its BatchNorm) on the same weights. The
    # first layer's gradients pass through two BatchNorm backward passes (differences of nearly equal batch means):
    # float32 carries them to ~1e-4, whoever computes them.
    m.zero_grad()
    (torch.nn.Sequential.forward(m, x) * dz).sum().backward()
    lib = {k: dict(m.named_parameters())[n].grad.clone() for k, n in NAMES.items()}
    for k in ("W1", "g1", "be1", "W2", "g2", "be2"):
        mine, theirs = rel(grads[k], go[k]), rel(lib[k], go[k])
        assert mine < max(2.0 * theirs, 2e-5), (k, mine, theirs)
    # bit reproducibility of a second identical call
    m.zero_grad()
    out2 = m(x)
    (out2 * dz).sum().backward()
    assert torch.equal(out, out2)
    for k, n in NAMES.items():
        assert torch.equal(grads[k], dict(m.named_parameters())[n].grad), k
    # the library path on the same weights (plain nn.Sequential forward of the very same modules)
    m.zero_grad()
    ref = torch.nn.Sequential.forward(m, x)
    assert rel(out, ref.detach()) < 2e-5
    assert not m.eval().hip_ready(x)


def test_tower_behind_a_trainable_module_passes_the_gradient_upstream():
    """A tower used as a projector behind a backbone (reference siam.py:156-166 allows any backbone): its input requires
    a gradient, which the tower kernels do not produce - the torch modules must run then, and x.grad must be the
    library's. Slicing the tower gives a plain nn.Sequential."""
    from neural_svd_amd.cdk import TowerSequential, get_mlp
    torch.manual_seed(7)
    m = get_mlp([128, 256, 128], bias=True, nonlinearity="lrelu0.2", use_bn=True).to(DEV).train()
    assert isinstance(m, TowerSequential)
    g = torch.Generator().manual_seed(8)
    x0 = torch.randn(128, 128, generator=g).to(DEV)
    dz = torch.randn(128, 128, generator=g).to(DEV)
    assert m.hip_ready(x0)
    x = x0.clone().requires_grad_(True)
    assert not m.hip_ready(x)
    (m(x) * dz).sum().backward()
    got = x.grad.clone()
    x2 = x0.clone().requires_grad_(True)
    m.zero_grad()
    (torch.nn.Sequential.forward(m, x2) * dz).sum().backward()
    assert got is not None and rel(got, x2.grad) < 1e-6
    with torch.no_grad():
        assert m.hip_ready(x)  # no graph is being recorded: nothing to cut
    head = m[:3]
    assert type(head) is torch.nn.Sequential and len(head) == 3 and head[0] is m[0]
