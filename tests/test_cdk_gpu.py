"""GPU parity of the CDK NestedLoRA loss (nsvd_cdk_loss_forward / _backward through ctypes and the drop-in
``NestedLoRAForCDK``) against the golden vectors captured from the reference (tests/golden/cdk_loss.npz,
float64 values) and against the CPU oracle at the reference's full size (B = 1024, L = 512).

Tolerance: float32 MFMA contractions vs the float64 truth, <= 3e-5 relative (L2 for arrays, relative to
max(1, |value|) for the loss scalars) - the same bound the float32 reference itself is held to in
tests/test_oracle_golden.py::test_cdk_loss.
"""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 3e-5


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    yield


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run_abi(f, g, v, M, first, bw):
    from neural_svd_amd import hip_ops as H
    B, L = f.shape
    d = lambda t: None if t is None else t.to(DEV, torch.float32).contiguous()  # noqa: E731
    fd, gd, vd, Md, bwd = d(f), d(g), d(v), d(M), d(None if bw is None else bw.reshape(-1))
    ws = H.cdk_workspace(B, L, first, DEV)
    ws.fill_(0xFF)  # the library must not rely on a zeroed workspace
    loss = torch.empty(3, device=DEV)
    rj, ri = torch.empty(B, device=DEV), torch.empty(B * (B - 1), device=DEV)
    H.cdk_loss_forward(fd, gd, bwd, vd, Md, first, loss, rj, ri, ws)
    gf, gg = torch.empty(B, L, device=DEV), torch.empty(B, L, device=DEV)
    H.cdk_loss_backward(vd, B, L, first, None, gf, gg, ws)
    torch.cuda.synchronize()
    return loss.cpu(), rj.cpu(), ri.cpu(), gf.cpu(), gg.cpu()


@pytest.mark.parametrize("case", list("abcde"))
def test_cdk_golden(case):
    z = G.load("cdk_loss")
    B, L, seq, step, first, has_bw = [int(t) for t in z[f"cdk_{case}_cfg"]]
    f, g = torch.tensor(z[f"cdk_{case}_f"]), torch.tensor(z[f"cdk_{case}_g"])
    bw = torch.tensor(z[f"cdk_{case}_bw"]) if has_bw else None
    v, M = torch.tensor(z[f"cdk_{case}_v"]), torch.tensor(z[f"cdk_{case}_M"])
    loss, rj, ri, gf, gg = run_abi(f, g, v, M, bool(first), bw)
    p = f"cdk_{case}_f64_"
    for got, want in zip(loss.tolist(), z[p + "loss"].tolist()):
        assert abs(got - want) <= TOL * max(1.0, abs(want))
    assert rel(rj, z[p + "rs_joint"]) <= TOL and rel(ri, z[p + "rs_indep"]) <= TOL
    assert rel(gf, z[p + "grad_f"]) <= TOL and rel(gg, z[p + "grad_g"]) <= TOL


@pytest.mark.parametrize("B,L,seq,first,use_bw", [(1024, 512, False, True, False), (1000, 500, True, True, True),
                                                   (193, 64, False, False, False), (2, 1, False, True, False)])
def test_cdk_vs_oracle(B, L, seq, first, use_bw):
    gen = torch.Generator().manual_seed(B + L)
    f = torch.randn(B, L, generator=gen, dtype=torch.float64) / np.sqrt(L)
    g = 0.7 * f + 0.3 * torch.randn(B, L, generator=gen, dtype=torch.float64) / np.sqrt(L)
    bw = (0.5 + torch.rand(B, 1, generator=gen, dtype=torch.float64)) if use_bw else None
    v, M = O.cdk_masks(L, seq, 1, first)
    # the HIP path sees the float32 roundings of the inputs; so does the float64 oracle
    f, g = f.float().double(), g.float().double()
    bw = None if bw is None else bw.float().double()
    want = O.cdk_loss(f, g, v.double(), M.double(), first, bw)
    loss, rj, ri, gf, gg = run_abi(f, g, v, M, first, bw)
    for got, w in zip(loss.tolist(), [float(t) for t in want[:3]]):
        assert abs(got - w) <= TOL * max(1.0, abs(w))
    assert rel(rj, want[3]) <= TOL and rel(ri, want[4]) <= TOL
    assert rel(gf, want[5]) <= TOL and rel(gg, want[6]) <= TOL


def test_cdk_module_autograd_and_half_inputs():
    """NestedLoRAForCDK.compute_loss: autograd through the Function, grad_output scaling (AMP loss scale),
    half-precision tower outputs (upcast; gradients returned in the input dtype), reruns are bit-identical."""
    from neural_svd_amd.cdk import NestedLoRAForCDK, get_cdk_method
    from types import SimpleNamespace as NS
    B, L = 256, 96
    args = NS(neigs=L, loss=NS(name="neuralsvd", neuralsvd=NS(step=1, sequential=False, set_first_mode_const=True)))
    method = get_cdk_method(args, model=torch.nn.Identity())
    assert isinstance(method, NestedLoRAForCDK) and method.vector_mask.numel() == L + 1
    gen = torch.Generator().manual_seed(5)
    f0 = (torch.randn(B, L, generator=gen) / np.sqrt(L))
    g0 = (0.5 * f0 + 0.5 * torch.randn(B, L, generator=gen) / np.sqrt(L))
    f = f0.to(DEV).requires_grad_(True)
    g = g0.to(DEV).requires_grad_(True)
    loss, lop, lmet, rj, ri = method.compute_loss(f, g)
    (loss * 128.0).backward()
    want = O.cdk_loss(f0.double(), g0.double(), method.vector_mask.double(), method.matrix_mask.double(), True, None)
    assert abs(float(loss.detach()) - float(want[0])) <= TOL * max(1.0, abs(float(want[0])))
    assert abs(float(lop.detach()) - float(want[1])) <= TOL and abs(float(lmet.detach()) - float(want[2])) <= TOL
    assert rel(f.grad, 128.0 * want[5]) <= TOL and rel(g.grad, 128.0 * want[6]) <= TOL
    assert rj.shape == (B,) and ri.shape == (B * (B - 1),) and not rj.requires_grad
    # bit-reproducible (fixed-order reductions, no float atomics)
    loss2, *_ = method.compute_loss(f.detach(), g.detach())
    assert float(loss2) == float(loss.detach())
    # half inputs
    fh = f0.to(DEV).half().requires_grad_(True)
    gh = g0.to(DEV).half().requires_grad_(True)
    lh, *_ = method.compute_loss(fh, gh)
    lh.backward()
    wh = O.cdk_loss(fh.detach().double().cpu(), gh.detach().double().cpu(), method.vector_mask.double(),
                    method.matrix_mask.double(), True, None)
    assert fh.grad.dtype == torch.float16 and abs(float(lh.detach()) - float(wh[0])) <= TOL * max(1.0, abs(float(wh[0])))
    assert rel(fh.grad.float(), wh[5]) <= 2e-3  # fp16 rounding of the returned gradient
    # only one tower needs a gradient
    f1 = f0.to(DEV).requires_grad_(True)
    l1, *_ = method.compute_loss(f1, g0.to(DEV))
    l1.backward()
    assert rel(f1.grad, want[5]) <= TOL


def test_cdk_argument_errors():
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd._lib import NsvdError
    f = torch.zeros(8, 4, device=DEV)
    v, M = torch.ones(5, device=DEV), torch.ones(5, 5, device=DEV)
    loss = torch.empty(3, device=DEV)
    with pytest.raises(NsvdError):  # workspace too small
        H.cdk_loss_forward(f, f, None, v, M, True, loss, None, None, torch.empty(256, dtype=torch.uint8, device=DEV))
    with pytest.raises(NsvdError):  # CPU tensor
        H.cdk_loss_forward(f.cpu(), f, None, v, M, True, loss, None, None, H.cdk_workspace(8, 4, True, DEV))
    with pytest.raises(NsvdError):  # mask shape
        H.cdk_loss_forward(f, f, None, v[:4], M, True, loss, None, None, H.cdk_workspace(8, 4, True, DEV))
