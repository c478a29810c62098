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


@pytest.mark.parametrize("B,d0,d1,d2", [(256, 128, 256, 256), (1024, 512, 8192, 512),
                                        # bottleneck towers (d1 < d0): the bfloat16 copy of X is the largest cast
                                        # operand there (round 3's scratch was sized by d1 only and overran)
                                        (512, 512, 256, 256), (256, 1024, 256, 512)])
@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("form", ["fused", "strips"])
def test_tower_mixed_precision_against_the_oracle_with_the_same_rounding(B, d0, d1, d2, form, half, monkeypatch):
    """form: "fused" = the wide layer with BatchNorm inside the contraction (csrc/tower_col.h: Y1 and dA1 never stored,
    the backward recovers the normalised value from the stored activation; what slope > 0 runs); "strips" = the
    contraction + strip kernels (NSVD_TOWER16_FUSED=0; what slope == 0 runs). Each against the float64 oracle that
    restates ITS roundings (include/nsvd.h: nsvd_tower_mixed_fused), at the same tolerances.
    half: the 16-bit type - bfloat16, or IEEE float16 (gemm_bf16 flag bit 4: the reference's autocast dtype,
    main_sketchy.py:182; 8 x finer roundings: the float32-mode comparison's lower bound is scaled accordingly).
    gemm_bf16 (the counterpart of the reference's autocast branch, main_sketchy.py:161,182): operands and the wide
    activations / gradients stored as bfloat16, float32 accumulation and statistics (include/nsvd.h). Against the
    float64 oracle that rounds the same tensors (its own intermediates differ from the float32 ones by 1e-7, which moves
    a few values in 10^4 across a bfloat16 rounding boundary: the tolerances below, not float32 noise level) - and
    against the float32 mode, from which it must differ by about the bfloat16 rounding (2^-9 per stored value), no more."""
    from neural_svd_amd import hip_ops as H
    if form == "strips":
        monkeypatch.setenv("NSVD_TOWER16_FUSED", "0")
    g = torch.Generator().manual_seed(B + d1)
    P = dict(W1=torch.randn(d1, d0, generator=g) / d0 ** 0.5, b1=0.1 * torch.randn(d1, generator=g),
             g1=1.0 + 0.3 * torch.randn(d1, generator=g), be1=0.2 * torch.randn(d1, generator=g),
             W2=torch.randn(d2, d1, generator=g) / d1 ** 0.5, b2=0.1 * torch.randn(d2, generator=g),
             g2=1.0 + 0.3 * torch.randn(d2, generator=g), be2=0.2 * torch.randn(d2, generator=g))
    x = torch.randn(B, d0, generator=g)
    dz = torch.randn(B, d2, generator=g)
    fused = H.tower_mixed_fused(B, d0, d1, d2, 0.2)
    assert fused == (form == "fused")
    zo, go, _ = O.tower_forward_backward(x.double(), {k: v.double() for k, v in P.items()}, dz.double(), 0.2,
                                         gemm_bf16="fused" if fused else True, half=half)
    flag = 1 | (H.TOWER16_F16 if half == "f16" else 0)
    Pd = {k: v.to(DEV).contiguous() for k, v in P.items()}
    for k, n in (("rm1", d1), ("rv1", d1), ("rm2", d2), ("rv2", d2)):
        Pd[k] = torch.zeros(n, device=DEV) if k.startswith("rm") else torch.ones(n, device=DEV)
    # the workspace followed by a canary: nothing may be written past nsvd_tower_workspace_bytes
    nws = H.tower_workspace(B, d0, d1, d2, DEV).numel()
    buf = torch.full((nws + 4096,), 0xA5, dtype=torch.uint8, device=DEV)
    ws = buf[:nws]
    xd, dzd = x.to(DEV), dz.to(DEV)
    res = {}
    for mixed in (True, False):
        z = H.tower_forward(xd, Pd, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=flag if mixed else 0)
        res[mixed] = (z.clone(), H.tower_backward(xd, Pd, dzd, 0.2, ws, gemm_bf16=flag if mixed else 0))
    torch.cuda.synchronize()
    assert bool((buf[nws:] == 0xA5).all()), "a tower kernel wrote past its workspace"
    z, grads = res[True]
    assert rel(z, zo) < 2e-4, rel(z, zo)
    for k in ("W1", "g1", "be1", "W2", "g2", "be2"):
        assert rel(grads[k], go[k]) < 2e-3, (k, rel(grads[k], go[k]))
    # the two modes: different by the operand rounding, and only by that
    z32, g32 = res[False]
    lo = 1e-4 if half == "bf16" else 1e-5
    assert lo < rel(z, z32) < 2e-2, rel(z, z32)
    assert lo < rel(grads["W2"], g32["W2"]) < 5e-2
    # bit reproducibility
    z2 = H.tower_forward(xd, Pd, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=flag)
    assert torch.equal(z, z2)


@pytest.mark.parametrize("form", ["fused", "strips"])
def test_tower_float16_against_the_references_autocast_run(form, monkeypatch):
    """tests/golden/amp.npz: the REFERENCE's tower (get_mlp, examples/models/mlp.py:129-164) forward + backward under
    torch.autocast(float16) - its AMP branch (main_sketchy.py:182), run on the CPU where the fixture was made. The HIP
    float16 mode, both forms, must be at least as close to that run as that run is to float64 arithmetic."""
    from neural_svd_amd import hip_ops as H
    if form == "strips":
        monkeypatch.setenv("NSVD_TOWER16_FUSED", "0")
    z = G.load("amp")
    B, d0, d1, d2, _ = [int(v) for v in z["amp_tower_cfg"]]
    assert H.tower_mixed_fused(B, d0, d1, d2, 0.2) == (form == "fused")
    P = {k: torch.tensor(z["amp_tower_param0_" + n]).float().to(DEV).contiguous() for k, n in NAMES.items()}
    for k, n in (("rm1", "1.running_mean"), ("rv1", "1.running_var"), ("rm2", "4.running_mean"), ("rv2", "4.running_var")):
        P[k] = torch.tensor(z["amp_tower_param0_" + n]).float().to(DEV).contiguous()
    x, dz = torch.tensor(z["amp_tower_x"]).to(DEV), torch.tensor(z["amp_tower_dz"]).to(DEV)
    ws = H.tower_workspace(B, d0, d1, d2, DEV)
    flag = 1 | H.TOWER16_F16
    zz = H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=flag)
    gr = H.tower_backward(x, P, dz, 0.2, ws, gemm_bf16=flag)
    torch.cuda.synchronize()
    assert rel(zz, z["amp_tower_f16_z"]) <= rel(z["amp_tower_f16_z"], z["amp_tower_f64_z"]) + 2e-4
    for k, n in NAMES.items():
        if k in ("b1", "b2"):
            continue  # (a bias in front of a BatchNorm: zero gradient, rounding noise only)
        ref, exact = z["amp_tower_f16_grad_" + n], z["amp_tower_f64_grad_" + n]
        assert rel(gr[k], ref) <= rel(ref, exact) + 2e-4, (k, rel(gr[k], ref), rel(ref, exact))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tower_module_under_autocast_runs_the_mixed_precision_mode(dtype):
    """the reference's Sketchy loop wraps method(x, y) in torch.cuda.amp.autocast (+ GradScaler) unless --disable_amp:
    inside autocast the tower module takes the mixed-precision mode with the AUTOCAST dtype as its half type (float16 -
    the reference's - or bfloat16: same bits as the C call with gemm_bf16 = 1 | 16 / 1), its output stays float32, and
    a GradScaler-style scaled backward gives the unscaled gradients back exactly (the scale is a power of two; float16:
    as long as nothing overflows - a scale that does overflow comes back as non-finite gradients, which is what
    torch's GradScaler looks for)."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.cdk import get_mlp
    torch.manual_seed(3)
    m = get_mlp([128, 256, 256], bias=True, nonlinearity="lrelu0.2", use_bn=True).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(256, 128, generator=g).to(DEV)
    dz = torch.randn(256, 256, generator=g).to(DEV)
    P = {k: dict(m.named_parameters())[n].detach().clone() for k, n in NAMES.items()}
    for k, mod, attr in (("rm1", m[1], "running_mean"), ("rv1", m[1], "running_var"), ("rm2", m[4], "running_mean"),
                         ("rv2", m[4], "running_var")):
        P[k] = getattr(mod, attr).detach().clone()
    ws = H.tower_workspace(256, 128, 256, 256, DEV)
    flag = 1 | (H.TOWER16_F16 if dtype == torch.float16 else 0)
    scale = 65536.0 if dtype == torch.bfloat16 else 8.0  # (dz ~ N(0, 1) here: 65536 dz overflows float16)
    want = H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=flag)
    gwant = H.tower_backward(x, P, dz, 0.2, ws, gemm_bf16=flag)
    with torch.autocast("cuda", dtype=dtype):
        out = m(x)
    assert out.dtype == torch.float32 and torch.equal(out, want)
    (out * dz * scale).sum().backward()
    for k, n in NAMES.items():
        got = dict(m.named_parameters())[n].grad / scale
        if dtype == torch.bfloat16:
            assert torch.equal(got, gwant[k]), k
        elif k not in ("b1", "b2"):  # (float16 has subnormals at 6e-5: the smallest stored gradients round differently
            # under a scale; the biases in front of a BatchNorm have zero gradients - rounding noise only)
            assert rel(got, gwant[k]) < 1e-3, (k, rel(got, gwant[k]))
    m.zero_grad()
    if dtype == torch.float16:  # a loss scale the float16 gradients cannot hold: non-finite parameter gradients
        with torch.autocast("cuda", dtype=dtype):
            out = m(x)
        (out * dz * 2.0 ** 20).sum().backward()
        assert not bool(torch.isfinite(dict(m.named_parameters())[NAMES["W1"]].grad).all())
        m.zero_grad()
    out32 = m(x)  # outside autocast: float32 contractions
    assert not torch.equal(out32, out) and rel(out32.detach(), out.detach()) < 2e-2


def test_tower_mixed_precision_shapes_and_fallback():
    """shapes the mixed-precision kernels do not take (a width that is not a multiple of 256): the C call refuses with
    NSVD_EUNSUPPORTED, and the module under autocast runs the float32 kernels instead (never less precise)."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.cdk import get_mlp
    assert H.tower_mixed_supported(1024, 512, 8192, 512) and H.tower_mixed_supported(256, 128, 256, 256)
    assert not H.tower_mixed_supported(128, 128, 256, 256) and not H.tower_mixed_supported(256, 128, 256, 128)
    assert H.tower_supported(128, 128, 256, 128)
    torch.manual_seed(5)
    m = get_mlp([128, 256, 128], bias=True, nonlinearity="lrelu0.2", use_bn=True).to(DEV).train()
    x = torch.randn(128, 128, generator=torch.Generator().manual_seed(6)).to(DEV)
    P = {k: dict(m.named_parameters())[n].detach().clone() for k, n in NAMES.items()}
    for k, n in (("rm1", 256), ("rv1", 256), ("rm2", 128), ("rv2", 128)):
        P[k] = torch.zeros(n, device=DEV) if k.startswith("rm") else torch.ones(n, device=DEV)
    ws = H.tower_workspace(128, 128, 256, 128, DEV)
    with pytest.raises(H.NsvdError):
        H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=True)
    want = H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=False)
    with torch.autocast("cuda", dtype=torch.float16):
        out = m(x)
    assert torch.equal(out, want)
