"""Pin the CPU oracle (oracle/nsvd_oracle.py) against golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G


# ------------------------------------------------------------------ masks (exact)
@pytest.mark.parametrize("L", [1, 4, 16, 33])
def test_sequential_masks(L):
    z = G.load("masks")
    v, M = O.sequential_nesting_masks(L)
    assert np.array_equal(v.numpy(), z[f"seq_L{L}_v"])
    assert np.array_equal(M.numpy(), z[f"seq_L{L}_M"])


@pytest.mark.parametrize("L,step", [(4, 1), (16, 1), (10, 4), (16, 4), (7, 3), (5, 7)])
def test_joint_masks(L, step):
    z = G.load("masks")
    v, M = O.joint_nesting_masks(L, step)
    assert v.dtype == torch.float32 and M.dtype == torch.float32
    assert np.array_equal(v.numpy(), z[f"joint_L{L}_s{step}_v"])
    assert np.array_equal(M.numpy(), z[f"joint_L{L}_s{step}_M"])


# ------------------------------------------------------------------ loss fwd/bwd
@pytest.mark.parametrize("case", list("abcdef"))
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-13), ("f32", torch.float32, 2e-5)])
def test_evd_loss(case, tag, dtype, tol):
    z = G.load("evd_loss")
    f = torch.tensor(z[f"loss_{case}_f"]).to(dtype)
    Tf = torch.tensor(z[f"loss_{case}_Tf"]).to(dtype)
    v = torch.tensor(z[f"loss_{case}_v"]).to(dtype)
    M = torch.tensor(z[f"loss_{case}_M"]).to(dtype)
    loss, lam1, lam2, _, _ = O.evd_loss_forward(f, Tf, v, M)
    g = O.evd_loss_backward(f, Tf, v, M, lam1, lam2)
    p = f"loss_{case}_{tag}_"
    assert abs(float(loss) - float(z[p + "loss"])) <= tol * max(1.0, abs(float(z[p + "loss"])))
    assert G.rel(lam1.numpy(), z[p + "lam1"]) <= tol
    assert G.rel(lam2.numpy(), z[p + "lam2"]) <= tol
    assert G.rel(g.numpy(), z[p + "grad_f"]) <= tol


@pytest.mark.parametrize("case", list("abcdef"))
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-13), ("f32", torch.float32, 2e-5)])
def test_evd_loss_independent_f1_f2(case, tag, dtype, tol):
    """the reference's loss Function called with f1, f2 that are not chunks of f (tests/golden/evd_loss_indep.npz:
    grad_output = 1.5; cases a, e pass f itself as f1 - autograd then adds the two gradients)"""
    z = G.load("evd_loss_indep")
    B, B1, B2, L, seq, step, f1_is_f = [int(t) for t in z[f"indep_{case}_cfg"]]
    f = torch.tensor(z[f"indep_{case}_f"]).to(dtype)
    Tf = torch.tensor(z[f"indep_{case}_Tf"]).to(dtype)
    f1 = f if f1_is_f else torch.tensor(z[f"indep_{case}_f1"]).to(dtype)
    f2 = torch.tensor(z[f"indep_{case}_f2"]).to(dtype)
    v = torch.tensor(z[f"indep_{case}_v"]).to(dtype)
    M = torch.tensor(z[f"indep_{case}_M"]).to(dtype)
    loss, g, g1, g2 = O.evd_loss_independent(f, Tf, f1, f2, v, M, grad_output=1.5)
    p = f"indep_{case}_{tag}_"
    assert abs(float(loss) - float(z[p + "loss"])) <= tol * max(1.0, abs(float(z[p + "loss"])))
    if f1_is_f:
        assert G.rel((g + g1).numpy(), z[p + "grad_f"]) <= tol
    else:
        assert G.rel(g.numpy(), z[p + "grad_f"]) <= tol
        assert G.rel(g1.numpy(), z[p + "grad_f1"]) <= tol
    assert G.rel(g2.numpy(), z[p + "grad_f2"]) <= tol


# ------------------------------------------------------------------ full model, float64 truth
SMALL = ["hyd_small", "osc_small", "hyd_ragged"]


@pytest.mark.parametrize("case", SMALL)
def test_model_f64_two_steps(case):
    """operator fwd, loss, every parameter gradient and two RMSprop+cosine steps in float64 must
    reproduce the reference to round-off."""
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case).to(torch.float64)
    v, M = G.masks_of(z, case)
    names = G.trainable_names(z, case)
    sq = [torch.zeros_like(t) for t in p.trainable()]
    for it in range(2):
        x = torch.tensor(z[f"{case}_x"][it]).double()
        r = O.loss_and_grads(x, p, prob, v, M)
        pre = f"{case}_f64_step{it}_"
        assert G.rel(r["f"], z[pre + "f"]) < 1e-12
        assert G.rel(r["Tf"], z[pre + "Tf"]) < 1e-9       # FD cancellation amplifies round-off
        assert abs(float(r["loss"]) - float(z[pre + "loss"])) < 1e-9 * abs(float(z[pre + "loss"]))
        for n, g in zip(names, r["grads"]):
            assert G.rel(g, z[pre + "grad_" + n]) < 1e-9, n
        lr = O.cosine_lr(cfg["lr"], it, cfg["num_iters"])
        O.rmsprop_step(p.trainable(), r["grads"], sq, lr, alpha=cfg["rmsprop_decay"], eps=1e-10)
        for n, t in zip(names, p.trainable()):
            assert G.rel(t, z[pre + "param_" + n]) < 1e-9, n  # grads carry FD round-off


@pytest.mark.parametrize("case", SMALL)
def test_model_f32_matches_reference_f32(case):
    """float32 oracle vs float32 reference: same arithmetic up to summation order."""
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    r = O.loss_and_grads(x, p, prob, v, M)
    pre = f"{case}_f32_step0_"
    assert G.rel(r["f"], z[pre + "f"]) < 5e-6
    # Tf: two correct float32 evaluations differ at the FD-noise level; compare against float64
    # truth with the float32 reference's own error as the yardstick
    t64 = z[f"{case}_f64_step0_Tf"]
    ref_err = G.rel(z[pre + "Tf"], t64)
    assert G.rel(r["Tf"], t64) < max(3 * ref_err, 1e-4)


@pytest.mark.parametrize("case", SMALL)
def test_spectrum_f64(case):
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    # the spectrum golden was taken after the two optimiser steps
    p = G.params_from_golden(z, case, prefix="f64_step1_param_").to(torch.float64)
    grid = O.validation_grid(cfg["lim"], cfg["val_eps"], cfg["ndim"])
    assert np.array_equal(grid.numpy(), z[f"{case}_val_data"])
    r = O.spectrum_evd(grid.double(), p, prob, cfg["lim"], chunk=cfg["batch_size"])
    assert G.rel(r["eigvals"], z[f"{case}_f64_spec_eigvals"]) < 1e-9
    assert G.rel(r["norms"], z[f"{case}_f64_spec_norms"]) < 1e-10
    assert G.rel(r["quad"], z[f"{case}_f64_spec_quad"]) < 1e-9
    nrm = r["norms"].sqrt()
    assert G.rel(r["cov"] / (nrm[:, None] * nrm[None, :]), z[f"{case}_f64_spec_cov_normalized"]) < 1e-10


# ------------------------------------------------------------------ headline shapes from the seed recipe
@pytest.mark.parametrize("case", ["hyd_med", "cfg1"])
def test_headline_from_seed(case):
    """Weights regenerated from torch.manual_seed(seed) in the reference's draw order must give the
    reference's outputs (this is how the 18.9 MB cfg1 weights travel: as a recipe)."""
    z = G.load("model_headline")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p32 = G.params_from_seed(cfg)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    r = O.loss_and_grads(x.double(), p32.to(torch.float64), prob, v, M)
    pre = f"{case}_f64_step0_"
    assert G.rel(r["f"], z[pre + "f"]) < 1e-11
    assert G.rel(r["Tf"], z[pre + "Tf"]) < 1e-8
    assert abs(float(r["loss"]) - float(z[pre + "loss"])) < 1e-8 * abs(float(z[pre + "loss"]))
    names = G.trainable_names(z, case)
    stride = 997 if case == "hyd_med" else 9973
    for n, g in zip(names, r["grads"]):
        g = g.numpy()
        assert abs(np.linalg.norm(g) - float(z[pre + "gradnorm_" + n])) < 1e-8 * float(z[pre + "gradnorm_" + n]), n
        assert G.rel(g.reshape(-1)[::stride], z[pre + "gradsample_" + n]) < 1e-8, n


# ------------------------------------------------------------------ misc
def test_ground_truth_spectra():
    z = G.load("misc")
    assert np.allclose(O.hydrogen2d_eigvals(64), z["gt_hydrogen2d_64"], rtol=0, atol=0)
    assert np.allclose(O.hydrogen2d_eigvals(9, charge=2.0), z["gt_hydrogen2d_z2_9"], rtol=0, atol=0)
    for n in (1, 6, 16, 32, 55):
        assert np.array_equal(O.oscillator2d_eigvals(n), z[f"gt_oscillator_{n}"]), n


def test_debug_model():
    """RNG-free model (all weights .1, integer Fourier harmonics)."""
    z = G.load("misc")
    fB = torch.tensor(z["debug_B"])
    L, hid = 3, [8, 8]
    ws, bs, prev = [], [], fB.shape[1] * 2
    for h in hid + [1]:
        ws.append(0.1 * torch.ones(L, h, prev))
        bs.append(0.1 * torch.ones(L, h, 1))
        prev = h
    p = O.Params(ws, bs, fB)
    x = torch.tensor(z["debug_x"])
    y = O.mlp_forward(O.fourier_features(x.double(), fB.double()), p.to(torch.float64))
    assert G.rel(y, z["debug_f64_y"]) < 1e-13
    y32 = O.mlp_forward(O.fourier_features(x, fB), p)
    assert G.rel(y32, z["debug_f32_y"]) < 1e-6


def test_ema_formula():
    s = [torch.zeros(3)]
    p = [torch.ones(3)]
    n = 0
    n = O.ema_update(s, p, 0.995, n)
    assert n == 1 and torch.allclose(s[0], torch.full((3,), 1 - 2 / 11))


# ------------------------------------------------------------------ the eager op-sequence port (CPU baseline)
@pytest.mark.parametrize("case", SMALL)
def test_torch_port_matches_reference_f32(case):
    """oracle/torch_port.py (what bench.py times as cpu_baseline) reproduces the reference's own
    float32 outputs: same op sequence => f, loss and the two-step RMSprop trajectory agree to
    float32 round-off."""
    from oracle import torch_port as TP
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case)
    v, M = G.masks_of(z, case)
    st = TP.PortStep(p, prob, v, M, lr=cfg["lr"], alpha=cfg["rmsprop_decay"], num_iters=cfg["num_iters"])
    names = G.trainable_names(z, case)
    for it in range(2):
        x = torch.tensor(z[f"{case}_x"][it])
        pre = f"{case}_f32_step{it}_"
        loss, f, Tf = st.loss(x)
        assert G.rel(f.detach(), z[pre + "f"]) < 2e-6
        ref_err = G.rel(z[pre + "Tf"], z[f"{case}_f64_step{it}_Tf"])
        assert G.rel(Tf.detach(), z[f"{case}_f64_step{it}_Tf"]) < max(3 * ref_err, 1e-4)
        st.step(x)
        got = dict(st.model.named_parameters())
        params = {n: got[n.replace("model.base.", "").replace("model.boundary_mask.", "")] for n in names}
        for n in names:
            # RMSprop's first steps are +-lr/sqrt(1-alpha) * sign(g): insensitive to gradient noise
            assert G.rel(params[n].detach(), z[pre + "param_" + n]) < 1e-3, n


# ------------------------------------------------------------------ CDK loss (next row: methods/cdk.py path)
@pytest.mark.parametrize("case", list("abcde"))
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 3e-5)])
def test_cdk_loss(case, tag, dtype, tol):
    z = G.load("cdk_loss")
    B, L, seq, step, first, has_bw = [int(t) for t in z[f"cdk_{case}_cfg"]]
    v, M = O.cdk_masks(L, bool(seq), step, bool(first))
    assert np.array_equal(v.numpy(), z[f"cdk_{case}_v"]) and np.array_equal(M.numpy(), z[f"cdk_{case}_M"])
    f = torch.tensor(z[f"cdk_{case}_f"]).to(dtype)
    g = torch.tensor(z[f"cdk_{case}_g"]).to(dtype)
    bw = torch.tensor(z[f"cdk_{case}_bw"]).to(dtype) if has_bw else None
    loss, lop, lmet, rj, ri, gf, gg = O.cdk_loss(f, g, v.to(dtype), M.to(dtype), bool(first), bw)
    p = f"cdk_{case}_{tag}_"
    want = z[p + "loss"]
    for got, w in zip((loss, lop, lmet), want):
        assert abs(float(got) - float(w)) <= tol * max(1.0, abs(float(w)))
    assert G.rel(rj, z[p + "rs_joint"]) <= tol and G.rel(ri, z[p + "rs_indep"]) <= tol
    assert G.rel(gf, z[p + "grad_f"]) <= tol and G.rel(gg, z[p + "grad_g"]) <= tol


# ------------------------------------------------------------------ SVD loss (methods/nestedlora.py:114-164)
@pytest.mark.parametrize("case", list("abcde"))
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 3e-5)])
def test_svd_loss(case, tag, dtype, tol):
    z = G.load("svd_loss")
    B, L, seq, step = [int(t) for t in z[f"svd_{case}_cfg"]]
    v, M = (O.sequential_nesting_masks(L) if seq else O.joint_nesting_masks(L, step))
    assert np.allclose(v.numpy(), z[f"svd_{case}_v"]) and np.allclose(M.numpy(), z[f"svd_{case}_M"])
    f, Tg, g, Ta = [torch.tensor(z[f"svd_{case}_{k}"]).to(dtype) for k in ("f", "Tg", "g", "Tadjf")]
    loss, gf, gg = O.svd_loss(f, Tg, g, Ta, v.to(dtype), M.to(dtype))
    p = f"svd_{case}_{tag}_"
    assert abs(float(loss) - float(z[p + "loss"][0])) <= tol * max(1.0, abs(float(z[p + "loss"][0])))
    assert G.rel(gf, z[p + "grad_f"]) <= tol and G.rel(gg, z[p + "grad_g"]) <= tol


def test_kernel_apply_definition():
    """oracle kernel_apply (parity unpinned: the reference has no kernel operator) against the literal double sum of
    its definition Kf[i, l] = (1 / B2) sum_k K[rows_i, cols_k] f[k, l], duplicates included."""
    g = torch.Generator().manual_seed(0)
    K = torch.randn(7, 7, generator=g, dtype=torch.float64)
    rows, cols = torch.tensor([3, 3, 0, 6]), torch.tensor([1, 5, 5, 2, 0])
    f = torch.randn(5, 3, generator=g, dtype=torch.float64)
    want = torch.zeros(4, 3, dtype=torch.float64)
    for i in range(4):
        for k in range(5):
            want[i] += K[rows[i], cols[k]] * f[k] / 5
    assert torch.allclose(O.kernel_apply(K, rows, cols, f), want, rtol=1e-13, atol=1e-15)


# ------------------------------------------------------------------ exact-Laplacian mode (laplacian_eps = 0)
@pytest.mark.parametrize("case", ["hyd_exact", "osc_exact", "osc_exact_small"])
def test_exact_laplacian_mode(case):
    """oracle closed-form jets (exact_jets) vs the reference's double-autograd exact mode (diff_ops.py:54-111):
    f, Tf, loss and gradients of one step in float64."""
    z = G.load("model_exact")
    cfg = G.cfg_of(z, case)
    assert cfg["laplacian_eps"] == 0.0
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case).to(torch.float64)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0]).double()
    out = O.loss_and_grads(x, p, prob, v.double(), M.double())
    pre = f"{case}_f64_step0_"
    assert G.rel(out["f"], z[pre + "f"]) < 1e-11
    assert G.rel(out["Tf"], z[pre + "Tf"]) < 1e-9
    assert abs(float(out["loss"]) - float(z[pre + "loss"])) < 1e-9 * abs(float(z[pre + "loss"]))
    for n, g in zip(G.trainable_names(z, case), out["grads"]):
        if pre + f"grad_{n}" in z.files:
            assert G.rel(g, z[pre + f"grad_{n}"]) < 1e-8, n
        else:
            gs = g.reshape(-1).numpy()
            assert abs(np.linalg.norm(gs) - float(z[pre + f"gradnorm_{n}"])) < 1e-8 * float(z[pre + f"gradnorm_{n}"])
            assert G.rel(gs[::13], z[pre + f"gradsample_{n}"]) < 1e-8, n


def test_exact_jets_against_autograd():
    """the same closed form against torch.autograd on the oracle's own model (independent of the reference)."""
    p = O.init_params(2, 3, 5, (7, 6), 0.7, exp_mask_init=3.0, seed=1).to(torch.float64)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=0.0, op_scale=1.0, op_shift=2.0, sigma=1.5, hard_mul_const=0.8)
    x = torch.randn(4, 3, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    g, lap, _ = O.exact_jets(x, p, prob)

    def gfun(xx):
        base = O.mlp_forward(O.fourier_features(xx, p.fourier_B), p)
        return O.sqrt_importance(xx, prob.sigma) * prob.hard_mul_const * base * O.boundary_mask(xx, p)

    for b in range(4):
        for l in range(2):
            H = torch.autograd.functional.hessian(lambda xx: gfun(xx.view(1, -1))[0, l], x[b])
            assert abs(float(torch.trace(H)) - float(lap[b, l])) < 1e-9 * max(1.0, abs(float(lap[b, l])))
    assert torch.allclose(gfun(x), g, rtol=1e-12, atol=1e-14)


# ------------------------------------------------------------------ normalize() of the CDK towers (siam.py:170-183)
@pytest.mark.parametrize("mode", ["l2_ball", "l2_sphere"])
@pytest.mark.parametrize("case", list("abcd"))
def test_row_normalize(case, mode):
    z = G.load("normalize")
    B, L, r = z[f"norm_{case}_cfg"]
    x = torch.tensor(z[f"norm_{case}_z"]).requires_grad_(True)
    y = O.row_normalize(x, float(r), mode)
    y.backward(torch.tensor(z[f"norm_{case}_dout"]))
    assert G.rel(y.detach(), z[f"norm_{case}_{mode}_f64_out"]) <= 1e-14
    assert G.rel(x.grad, z[f"norm_{case}_{mode}_f64_dz"]) <= 1e-13


# ----------------------------------------------------------------------------- compute_loss_kernel
@pytest.mark.parametrize("case", ["ka", "kb", "kc"])
@pytest.mark.parametrize("split", [False, True])
def test_kernel_loss_matches_reference(case, split):
    """oracle kernel_loss_and_grads vs the reference's NestedLoRA.compute_loss_kernel (methods/nestedlora.py:230-252)
    on its own WaveFunctions model with the toy Gaussian-kernel operator of make_golden.py: loss, f, Kf, gradients."""
    z = G.load("kernel_loss")
    cfg = G.cfg_of(z, case)
    p = (G.params_from_golden(z, case) if f"{case}_param0_model.base.ws.0" in z.files else G.params_from_seed(cfg))
    p = p.to(torch.float64)
    v, M = G.masks_of(z, case)
    ell = float(z[f"{case}_ell"])
    x = torch.tensor(z[f"{case}_x"]).double()
    r = O.kernel_loss_and_grads(x, p, lambda a, b, fb: O.gaussian_kernel_apply(a, b, fb, ell), v, M, split,
                                hard_mul_const=cfg["hard_mul_const"])
    q = f"{case}_f64_split{int(split)}_"
    assert abs(float(r["loss"]) - float(z[q + "loss"])) < 1e-11 * abs(float(z[q + "loss"]))
    assert G.rel(r["f"], z[q + "f"]) < 1e-11 and G.rel(r["Kf"], z[q + "Kf"]) < 1e-11
    for n, g in zip(G.trainable_names(z, case), r["grads"]):
        if q + f"grad_{n}" in z.files:
            assert G.rel(g.reshape(z[q + f"grad_{n}"].shape), z[q + f"grad_{n}"]) < 1e-9, n
        else:
            assert abs(float(g.norm()) - float(z[q + f"gradnorm_{n}"])) < 1e-9 * float(z[q + f"gradnorm_{n}"]), n
            assert G.rel(g.reshape(-1)[::61], z[q + f"gradsample_{n}"]) < 1e-9, n


# ----------------------------------------------------------------------------- CDK tower
def _tower_params_from_golden(z, case):
    g = lambda k: torch.tensor(z[f"{case}_param0_{k}"])  # noqa: E731
    return dict(W1=g("0.weight"), b1=g("0.bias"), g1=g("1.weight"), be1=g("1.bias"), W2=g("3.weight"),
                b2=g("3.bias"), g2=g("4.weight"), be2=g("4.bias"))


@pytest.mark.parametrize("case", ["ta", "tc"])
def test_tower_matches_reference(case):
    """oracle tower_forward_backward vs the reference's get_mlp tower (examples/models/mlp.py:129-164) in training mode:
    output, every parameter gradient, and the BatchNorm running statistics after one step."""
    z = G.load("tower")
    P = _tower_params_from_golden(z, case)
    x, dz, slope = torch.tensor(z[f"{case}_x"]), torch.tensor(z[f"{case}_dz"]), float(z[f"{case}_slope"])
    out, grads, (st1, st2) = O.tower_forward_backward(x, P, dz, slope)
    q = f"{case}_f64_"
    assert G.rel(out, z[q + "z"]) < 1e-12
    names = {"W1": "0.weight", "b1": "0.bias", "g1": "1.weight", "be1": "1.bias", "W2": "3.weight", "b2": "3.bias",
             "g2": "4.weight", "be2": "4.bias"}
    for k, n in names.items():
        want = z[q + f"grad_{n}"]
        if k in ("b1", "b2"):  # a bias in front of a BatchNorm has a vanishing gradient: compare absolutely
            assert np.abs(grads[k].numpy() - want).max() < 1e-12 * max(1.0, float(np.abs(z[q + "grad_0.weight"]).max()))
        else:
            assert G.rel(grads[k], want) < 1e-11, k
    for st, k in ((st1, 1), (st2, 4)):  # running = 0.9 * init + 0.1 * batch (init: mean 0, var 1), unbiased variance
        assert G.rel(0.1 * st[0], z[q + f"running_mean_{k}"]) < 1e-12
        assert G.rel(0.9 + 0.1 * st[2], z[q + f"running_var_{k}"]) < 1e-12


# ---------------------------------------------------------------------------------------------- CDK training step
CDK_KEYS = {"W1": "0.weight", "b1": "0.bias", "g1": "1.weight", "be1": "1.bias", "W2": "3.weight", "b2": "3.bias",
            "g2": "4.weight", "be2": "4.bias"}


def cdk_step_case(z, name, dtype=torch.float64):
    """initial state of golden case `name` of cdk_step.npz: towers, momentum buffers, running statistics, inputs"""
    B, d0, d1, d2, seed, nstep, T = [int(v) for v in z[f"{name}_cfg"]]
    towers, running = [], []
    if name == "sa":
        for side in "xy":
            p = f"{name}_param0_backbones.{side}."
            towers.append({k: torch.tensor(z[p + n]).to(dtype).clone() for k, n in CDK_KEYS.items()})
            running.append({"rm1": torch.tensor(z[p + "1.running_mean"]).to(dtype).clone(),
                            "rv1": torch.tensor(z[p + "1.running_var"]).to(dtype).clone(),
                            "rm2": torch.tensor(z[p + "4.running_mean"]).to(dtype).clone(),
                            "rv2": torch.tensor(z[p + "4.running_var"]).to(dtype).clone()})
        xs, ys = torch.tensor(z[f"{name}_x"]).to(dtype), torch.tensor(z[f"{name}_y"]).to(dtype)
    else:
        # the reference's constructor calls in the reference's order reproduce its initial weights (torch.nn.Linear's
        # default init draws from the global generator): two towers, x then y
        torch.manual_seed(seed)
        for _ in range(2):
            l1, l2 = torch.nn.Linear(d0, d1), None
            l2 = torch.nn.Linear(d1, d2)
            towers.append({"W1": l1.weight.detach().to(dtype).clone(), "b1": l1.bias.detach().to(dtype).clone(),
                           "g1": torch.ones(d1, dtype=dtype), "be1": torch.zeros(d1, dtype=dtype),
                           "W2": l2.weight.detach().to(dtype).clone(), "b2": l2.bias.detach().to(dtype).clone(),
                           "g2": torch.ones(d2, dtype=dtype), "be2": torch.zeros(d2, dtype=dtype)})
            running.append({"rm1": torch.zeros(d1, dtype=dtype), "rv1": torch.ones(d1, dtype=dtype),
                            "rm2": torch.zeros(d2, dtype=dtype), "rv2": torch.ones(d2, dtype=dtype)})
        g = torch.Generator().manual_seed(3000 + seed)
        xs = torch.randn(nstep, B, d0, generator=g, dtype=torch.float64).to(dtype)
        ys = torch.randn(nstep, B, d0, generator=g, dtype=torch.float64).to(dtype)
    bufs = [{k: torch.zeros_like(v) for k, v in t.items()} for t in towers]
    return towers, bufs, running, xs, ys, nstep, T


@pytest.mark.parametrize("name", ["sa", "sb"])
def test_cdk_train_step_matches_reference(name):
    """oracle cdk_train_step == the reference's Sketchy step (towers, l2_ball, CDK loss, clip_grad_norm_, SGD momentum,
    cosine schedule) over three steps: losses, total gradient norms, parameters, momentum buffers, running statistics"""
    z = G.load("cdk_step")
    towers, bufs, running, xs, ys, nstep, T = cdk_step_case(z, name)
    mu, lr0, mom, max_norm, slope = [float(v) for v in z[f"{name}_hyper"]]
    v, M = torch.tensor(z[f"{name}_v"]).double(), torch.tensor(z[f"{name}_M"]).double()
    q = f"{name}_f64_"
    for t in range(nstep):
        lr = O.cosine_lr(lr0, t, T)
        (loss, lop, lmet), total = O.cdk_train_step(xs[t], ys[t], towers, bufs, running, v, M, mu, lr, mom, max_norm, slope,
                                                    first_step=(t == 0))
        want = z[q + "loss"][t]
        assert abs(float(loss) - want[0]) < 1e-10 * max(1.0, abs(want[0]))
        assert abs(float(lop) - want[1]) < 1e-10 * abs(want[1]) and abs(float(lmet) - want[2]) < 1e-10 * abs(want[2])
        assert abs(float(total) - z[q + "total_norm"][t]) < 1e-10 * z[q + "total_norm"][t]
    for side, P, Bf, R in (("x", towers[0], bufs[0], running[0]), ("y", towers[1], bufs[1], running[1])):
        for k, n in CDK_KEYS.items():
            key = f"backbones.{side}.{n}"
            got = P[k].numpy()
            if name == "sb":
                assert abs(np.linalg.norm(got) - float(z[q + f"pnorm_{key}"])) < 1e-10 * max(1.0, float(z[q + f"pnorm_{key}"]))
                assert np.allclose(got.reshape(-1)[::5], z[q + f"param_{key}"], rtol=1e-9, atol=1e-12), key
                assert abs(np.linalg.norm(Bf[k].numpy()) - float(z[q + f"bufnorm_{key}"])) < 1e-9 * max(1e-3, float(z[q + f"bufnorm_{key}"]))
            else:
                assert np.allclose(got, z[q + f"param_{key}"], rtol=1e-9, atol=1e-12), key
                assert np.allclose(Bf[k].numpy(), z[q + f"buf_{key}"], rtol=1e-8, atol=1e-11), key
        for rk, n in (("rm1", "1.running_mean"), ("rv1", "1.running_var"), ("rm2", "4.running_mean"), ("rv2", "4.running_var")):
            got, want = R[rk].numpy(), z[q + f"param_backbones.{side}.{n}"]
            assert np.allclose(got.reshape(-1)[::5] if name == "sb" else got, want, rtol=1e-9, atol=1e-12), rk


# ------------------------------------------------------------------ the reference's AMP branch (float16 autocast + GradScaler)
TOWER_KEYS = {"W1": "0.weight", "b1": "0.bias", "g1": "1.weight", "be1": "1.bias", "W2": "3.weight", "b2": "3.bias",
              "g2": "4.weight", "be2": "4.bias"}


@pytest.mark.parametrize("mode", [True, "fused"])
def test_tower_float16_modes_against_the_references_autocast_run(mode):
    """tests/golden/amp.npz (make_golden.golden_amp): the reference's tower under torch.autocast(float16) on the CPU - its
    own modules, torch's own autocast placement of the float16 roundings. The oracle's two float16 modes must be at
    least as close to that run as that run is to float64 arithmetic (its own rounding error is the yardstick)."""
    z = G.load("amp")
    P = {k: torch.tensor(z["amp_tower_param0_" + n]).double() for k, n in TOWER_KEYS.items()}
    x, dz = torch.tensor(z["amp_tower_x"]).double(), torch.tensor(z["amp_tower_dz"]).double()
    zo, go, _ = O.tower_forward_backward(x, P, dz, 0.2, gemm_bf16=mode, half="f16")
    assert G.rel(zo.numpy(), z["amp_tower_f16_z"]) <= G.rel(z["amp_tower_f16_z"], z["amp_tower_f64_z"]) + 2e-4
    for k, n in TOWER_KEYS.items():
        if k in ("b1", "b2"):
            continue  # (a bias in front of a BatchNorm: zero gradient, rounding noise only)
        ref, exact = z["amp_tower_f16_grad_" + n], z["amp_tower_f64_grad_" + n]
        assert G.rel(go[k].numpy(), ref) <= G.rel(ref, exact) + 2e-4, (k, G.rel(go[k].numpy(), ref), G.rel(ref, exact))


@pytest.mark.parametrize("mode", [True, "fused"])
def test_cdk_step_float16_grad_scaler_against_the_references_amp_loop(mode):
    """amp.npz: eight iterations of the Sketchy loop body with its AMP branch on (main_sketchy.py:161,180-212; CPU float16
    autocast, torch.amp.GradScaler from 2^16 with growth_interval 2). The oracle's float16 + scaler step from the same
    weights and batches: the scale's trajectory exactly, the unscaled gradient norms to 2e-4, the losses to 3e-3 (under
    autocast the reference's loss itself is float16 arithmetic), every parameter's update to 1 % of its length."""
    import torch.nn as nn
    from neural_svd_amd.cdk import HeteroNetwork, get_mlp
    z = G.load("amp")
    B, d0, d1, d2, seed, nstep, T = [int(v) for v in z["amp_step_cfg"]]
    mu, lr, mom, max_norm, slope, init_scale, gi = [float(v) for v in z["amp_step_hyper"]]
    torch.manual_seed(seed)  # the reference's constructor calls in the reference's order: same initial weights
    sizes = [d0, d1, d2]
    model = HeteroNetwork([get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                           get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                          [nn.Identity(), nn.Identity()], mu=mu, regularize_mode="l2_ball").train()
    sd0 = {k: v.detach().double().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(77)
    xs, ys = torch.randn(nstep, B, d0, generator=g), torch.randn(nstep, B, d0, generator=g)
    v, M = O.cdk_masks(d2, False, 1, True)
    towers = [{k: sd0[f"backbones.{s}.{n}"].clone() for k, n in TOWER_KEYS.items()} for s in "xy"]
    bufs = [{k: torch.zeros_like(t_) for k, t_ in t.items()} for t in towers]
    running = [dict(rm1=sd0[f"backbones.{s}.1.running_mean"].clone(), rv1=sd0[f"backbones.{s}.1.running_var"].clone(),
                    rm2=sd0[f"backbones.{s}.4.running_mean"].clone(), rv2=sd0[f"backbones.{s}.4.running_var"].clone())
               for s in "xy"]
    sc = dict(scale=init_scale, growth_factor=2.0, backoff_factor=0.5, growth_interval=int(gi), growth_tracker=0,
              steps_ok=0, steps_skipped=0)
    for t in range(nstep):
        (loss, _, _), total = O.cdk_train_step(xs[t].double(), ys[t].double(), towers, bufs, running, v.double(),
                                               M.double(), mu, O.cosine_lr(lr, t, T), mom, max_norm, slope, False,
                                               gemm_bf16=mode, half="f16", scaler=sc)
        want = z["amp_step_rows"][t]
        assert abs(float(loss) - want[0]) < 3e-3 * abs(want[0]), (t, float(loss), want[0])
        assert abs(float(total) - want[1]) < 2e-4 * want[1], (t, float(total), want[1])
        assert sc["scale"] == want[2], (t, sc["scale"], want[2])
    assert sc["steps_skipped"] == 0
    for si, s in enumerate("xy"):
        for k, n in TOWER_KEYS.items():
            if k in ("b1", "b2"):
                continue
            ref = torch.tensor(z[f"amp_step_param_backbones.{s}.{n}"]).double()
            got = towers[si][k][..., ::3] if towers[si][k].dim() == 2 else towers[si][k]
            move = float(z[f"amp_step_move_backbones.{s}.{n}"])
            assert float((got - ref).norm()) < 1e-2 * move, (s, k, float((got - ref).norm()) / move)
