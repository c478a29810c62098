"""GPU parity of the drop-in ``NestedLoRALossFunctionSVD`` (reference methods/nestedlora.py:114-164), which runs on the
EVD loss kernels (nsvd_evd_moments / nsvd_evd_loss_grad on the stacked [f; g]), against the golden vectors captured
from the reference (tests/golden/svd_loss.npz, float64 values) and against the CPU oracle at a larger size.

Tolerance: float32 kernels vs the float64 truth, <= 3e-5 relative (L2 for the gradients, relative to max(1, |loss|)
for the scalar) - the bound the float32 reference itself is held to in tests/test_oracle_golden.py::test_svd_loss.
"""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 3e-5


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run(f, Tg, g, Ta, v, M, scale=1.0):
    from neural_svd_amd.nested_lowrank import NestedLoRALossFunctionSVD
    fd = f.to(DEV, torch.float32).requires_grad_(True)
    gd = g.to(DEV, torch.float32).requires_grad_(True)
    loss = NestedLoRALossFunctionSVD.apply(fd, Tg.to(DEV, torch.float32), gd, Ta.to(DEV, torch.float32), v, M)
    (scale * loss).backward()
    return float(loss.detach()), fd.grad.cpu(), gd.grad.cpu()


@pytest.mark.parametrize("case", list("abcde"))
def test_svd_golden(case):
    z = G.load("svd_loss")
    f, Tg, g, Ta, v, M = [torch.tensor(z[f"svd_{case}_{k}"]) for k in ("f", "Tg", "g", "Tadjf", "v", "M")]
    loss, gf, gg = run(f, Tg, g, Ta, v, M)
    p = f"svd_{case}_f64_"
    assert abs(loss - float(z[p + "loss"][0])) <= TOL * max(1.0, abs(float(z[p + "loss"][0])))
    assert rel(gf, z[p + "grad_f"]) <= TOL and rel(gg, z[p + "grad_g"]) <= TOL


@pytest.mark.parametrize("B,L,seq", [(1024, 128, False), (500, 37, True)])
def test_svd_vs_oracle(B, L, seq):
    gen = torch.Generator().manual_seed(B + L)
    f, Tg, g, Ta = [torch.randn(B, L, generator=gen, dtype=torch.float64) * s for s in (0.5, 3.0, 0.7, 2.0)]
    v, M = (O.sequential_nesting_masks(L) if seq else O.joint_nesting_masks(L, 1))
    want = O.svd_loss(f, Tg, g, Ta, v.double(), M.double())
    loss, gf, gg = run(f, Tg, g, Ta, v, M, scale=2.5)  # grad_output is propagated
    assert abs(loss - float(want[0])) <= TOL * max(1.0, abs(float(want[0])))
    assert rel(gf, 2.5 * want[1]) <= TOL and rel(gg, 2.5 * want[2]) <= TOL


def test_svd_module_entry_and_errors():
    """NestedLoRA._compute_loss(..., evd=False) reaches the Function (the compute_loss_* wrappers raise for evd=False,
    like the reference's); unequal batches are refused."""
    from neural_svd_amd._lib import NsvdError
    from neural_svd_amd.nested_lowrank import NestedLoRA
    m = NestedLoRA(model=None, neigs=6, step=1, sequential=False)
    gen = torch.Generator().manual_seed(0)
    f, Tg, g, Ta = [torch.randn(16, 6, generator=gen).to(DEV) for _ in range(4)]
    loss = m._compute_loss(f, Tg, g, Ta, evd=False)
    want = O.svd_loss(f.double().cpu(), Tg.double().cpu(), g.double().cpu(), Ta.double().cpu(),
                      m.vector_mask.double(), m.matrix_mask.double())[0]
    assert abs(float(loss) - float(want)) <= TOL * max(1.0, abs(float(want)))
    with pytest.raises(NsvdError):
        m._compute_loss(f, Tg, g[:8], Ta[:8], evd=False)
