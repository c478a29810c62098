"""GPU parity tests: every C-ABI entry point vs the CPU oracle / the reference's golden vectors.

Tolerances (float32 path):
  * f, moments, loss-given-(f,Tf), gradients-given-df, optimiser: <= 2e-5 relative (L2) vs float64;
  * Tf (eps = 0.01 stencil, carried in even / odd form by every path), the end-to-end loss and every
    gradient: <= 1e-4 relative of the FLOAT64 stencil / the reference's float64 values (measured
    2e-7 .. 7e-6). The float32 reference's own error (1e-3 .. 4e-2 on Tf) is still evaluated by
    `check_tf` as a yardstick, but it is no longer the bar.
"""
import math

import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu

H = None


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    global H
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from neural_svd_amd import hip_ops
    H = hip_ops
    yield


DEV = "cuda:0"
PATHS = ["generic", "auto"]
# the MFMA path with its first layer in native float32 MFMAs, and with the three-way bf16 split (opt-in path)
FPATHS = ["auto", "bf16x3"]


def _path(name):
    return {"generic": H.PATH_GENERIC, "auto": H.PATH_AUTO, "bf16x3": H.PATH_FUSED_BF16X3}[name]


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def tf_noise_kappa(Tf, Tf64, f64, cfg):
    """FD-noise yardstick (DESIGN.md "Numerics"): a float32 evaluation of the eps-stencil carries
    an absolute error ~ kappa * op_scale * 2^-23 * |f| / eps^2 per element; kappa = the median of that ratio
    (medians: elements with f ~ 0 have unbounded kappa by construction). The float32 REFERENCE itself has a
    median kappa of 7..12 on the golden cases."""
    u = 2.0 ** -23
    s = cfg["operator_scale"] * u * np.abs(np.asarray(f64)) / cfg["laplacian_eps"] ** 2
    d = np.abs(torch.as_tensor(Tf).double().cpu().numpy() - np.asarray(Tf64))
    return float(np.median(d / np.maximum(s, 1e-300)))


def oracle32_kappa(x, p, prob, ref64, kcfg):
    """the same yardstick where no reference fixture exists: the oracle evaluated in float32 (the reference's arithmetic)
    on the same inputs, against its float64 self"""
    c32 = O.operator_forward(x.float(), p.to(torch.float32), prob)
    return tf_noise_kappa(c32.Tf, ref64.Tf.numpy() if hasattr(ref64, "Tf") else ref64["Tf"].numpy(),
                          ref64.f.numpy() if hasattr(ref64, "f") else ref64["f"].numpy(), kcfg)


def check_tf(Tf, z, case, cfg, step=0):
    """Tf of the HIP path against the float64 reference, held to what the float32 REFERENCE achieves on the SAME case:
    median noise factor kappa <= 1.5 x the reference's, relative L2 error <= 2 x the reference's (measured on MI355X:
    kappa ratio 0.69 .. 1.25, L2 ratio 0.63 .. 1.23 over the five cases and both paths)."""
    pre64, pre32 = f"{case}_f64_step{step}_", f"{case}_f32_step{step}_"
    k = tf_noise_kappa(Tf, z[pre64 + "Tf"], z[pre64 + "f"], cfg)
    k_ref = tf_noise_kappa(z[pre32 + "Tf"], z[pre64 + "Tf"], z[pre64 + "f"], cfg)
    assert k <= 1.5 * k_ref, (k, k_ref)
    ref_err = rel(z[pre32 + "Tf"], z[pre64 + "Tf"])
    got = rel(Tf, z[pre64 + "Tf"])
    assert got <= max(2 * ref_err, 1e-3), (got, ref_err)


def to_dev(p: O.Params):
    ws = [w.float().to(DEV).contiguous() for w in p.ws]
    bs = [b.float().to(DEV).contiguous() for b in p.bs]
    fB = p.fourier_B.float().to(DEV).contiguous()
    sc = None if p.scales is None else p.scales.float().to(DEV).contiguous()
    return ws, bs, fB, sc


def shape_of(p: O.Params):
    L, h0, F = p.ws[0].shape
    hidden = tuple(w.shape[1] for w in p.ws[:-1])
    return H.ModelShape(L=L, D=p.fourier_B.shape[0], m=F // 2, hidden=hidden, has_exp_mask=p.scales is not None)


def hip_problem(prob: O.Problem):
    return H.make_problem(prob.potential, prob.charge_or_k, prob.eps, prob.op_scale, prob.op_shift, prob.sigma,
                          prob.scale_kinetic, prob.hard_mul_const, prob.use_importance)


def run_hip(p: O.Params, prob: O.Problem, x, v, M, path, df_override=None, mask_kind=None):
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    gw = [torch.full_like(w, float("nan")) for w in ws_t]
    gb = [torch.full_like(b, float("nan")) for b in bs_t]
    gs = None if sc is None else torch.full_like(sc, float("nan"))
    grads = H.pack_params(shape, gw, gb, None, gs)
    hp = hip_problem(prob)
    xd = x.float().to(DEV).contiguous()
    B = xd.shape[0]
    ws = H.new_workspace(shape, B, DEV)
    f, Tf = H.operator_forward(shape, params, hp, xd, ws, path=path)
    vd, Md = v.float().to(DEV), M.float().to(DEV).contiguous()
    kind = H.MASK_CUSTOM if mask_kind is None else mask_kind
    mom = H.evd_moments(f, Tf, kind, vd if kind == H.MASK_CUSTOM else None)
    loss, df = H.evd_loss_grad(f, Tf, kind, vd if kind == H.MASK_CUSTOM else None,
                               Md if kind == H.MASK_CUSTOM else None, mom)
    dfin = df if df_override is None else df_override.float().to(DEV).contiguous()
    H.operator_backward(shape, params, hp, xd, dfin, grads, ws, path=path)
    torch.cuda.synchronize()
    g = gw + gb + ([gs] if gs is not None else [])
    return dict(f=f, Tf=Tf, loss=loss, df=df, mom=mom, grads=g, path=H.path_name(shape, B, path))


# ------------------------------------------------------------------------------ Fourier features
@pytest.mark.parametrize("B,D,m", [(7, 2, 5), (64, 2, 64), (33, 3, 17), (5, 1, 4)])
def test_fourier_features(B, D, m):
    g = torch.Generator().manual_seed(B * 100 + m)
    x = 16 * torch.randn(B, D, generator=g)
    fB = 2 * np.pi * 0.1 * torch.randn(D, m, generator=g)
    E = 1 + 2 * D
    out = H.fourier_features(x.to(DEV), fB.to(DEV), 0.01, E).cpu()
    pts = O.stencil_points(x.double(), 0.01)
    ref = torch.cat([O.fourier_features(p, fB.double()) for p in pts], dim=0).T  # (F, R)
    # |phi| <= 1; float32 projections of size ~50 rad carry ~4e-6 absolute argument error
    assert float((out.double() - ref).abs().max()) < 2e-5
    out1 = H.fourier_features(x.to(DEV), fB.to(DEV), 0.01, 1).cpu()
    assert torch.equal(out1, out[:, :B])


# ------------------------------------------------------------------------------ loss kernels vs reference goldens
@pytest.mark.parametrize("case", list("abcdef"))
def test_evd_loss_golden(case):
    z = G.load("evd_loss")
    B, L, seq, step = [int(v) for v in z[f"loss_{case}_cfg"]]
    f = torch.tensor(z[f"loss_{case}_f"]).float().to(DEV)
    Tf = torch.tensor(z[f"loss_{case}_Tf"]).float().to(DEV)
    v = torch.tensor(z[f"loss_{case}_v"]).float().to(DEV)
    M = torch.tensor(z[f"loss_{case}_M"]).float().to(DEV).contiguous()
    kinds = [H.MASK_CUSTOM]
    if seq:
        kinds.append(H.MASK_SEQUENTIAL)
    elif step == 1:
        kinds.append(H.MASK_JOINT)
    p = f"loss_{case}_f64_"
    for kind in kinds:
        cust = kind == H.MASK_CUSTOM
        mom = H.evd_moments(f, Tf, kind, v if cust else None)
        loss, df = H.evd_loss_grad(f, Tf, kind, v if cust else None, M if cust else None, mom)
        torch.cuda.synchronize()
        LL = L * L
        assert rel(mom[:LL].view(L, L), z[p + "lam1"]) < 2e-6
        if B > 1:
            assert rel(mom[LL:2 * LL].view(L, L), z[p + "lam2"]) < 2e-6
        assert abs(float(loss[0]) - float(z[p + "loss"])) < 2e-5 * max(1.0, abs(float(z[p + "loss"])))
        assert abs(float(loss[1]) + float(loss[2]) - float(loss[0])) < 1e-4 * max(1.0, abs(float(loss[0])))
        assert rel(df, z[p + "grad_f"]) < 2e-6
    # one-call variant (single launch at these sizes)
    mom_f = torch.empty_like(mom)
    loss_f = torch.empty(3, device=DEV)
    df_f = torch.empty_like(f)
    H.evd_loss_fused(f, Tf, H.MASK_CUSTOM, v, M, mom_f, loss_f, df_f, H.evd_scratch(B, L, DEV))
    assert rel(df_f, z[p + "grad_f"]) < 2e-6 and rel(mom_f[:2 * L * L], mom[:2 * L * L]) < 1e-6
    assert abs(float(loss_f[0]) - float(z[p + "loss"])) < 2e-5 * max(1.0, abs(float(z[p + "loss"])))
    # grad_scale and loss-only call
    loss2, none = H.evd_loss_grad(f, Tf, H.MASK_CUSTOM, v, M, mom, want_grad=False)
    assert none is None and float(loss2[0]) == float(loss[0])
    _, df3 = H.evd_loss_grad(f, Tf, H.MASK_CUSTOM, v, M, mom, grad_scale=0.25)
    assert rel(df3 * 4, z[p + "grad_f"]) < 2e-6


@pytest.mark.parametrize("case", list("abcdef"))
def test_evd_loss_function_with_independent_f1_f2(case):
    """the lower seam (reference methods/nestedlora.py:70-111, :84 "f1 and f2 must be independent", the call at
    :239-244): NestedLoRALossFunctionEVD.apply DIRECTLY with f1, f2 from other batches (any row counts; cases a, e pass
    f itself as f1) against the reference's own run of the same call (tests/golden/evd_loss_indep.npz, grad_output
    1.5): the loss and separate gradients to f, f1, f2"""
    from neural_svd_amd.nested_lowrank import NestedLoRALossFunctionEVD
    z = G.load("evd_loss_indep")
    B, B1, B2, L, seq, step, f1_is_f = [int(t) for t in z[f"indep_{case}_cfg"]]
    f = torch.tensor(z[f"indep_{case}_f"]).float().to(DEV).requires_grad_(True)
    Tf = torch.tensor(z[f"indep_{case}_Tf"]).float().to(DEV).requires_grad_(True)
    f1 = f if f1_is_f else torch.tensor(z[f"indep_{case}_f1"]).float().to(DEV).requires_grad_(True)
    f2 = torch.tensor(z[f"indep_{case}_f2"]).float().to(DEV).requires_grad_(True)
    v = torch.tensor(z[f"indep_{case}_v"]).float()   # (host tensors, as NestedLoRA holds them)
    M = torch.tensor(z[f"indep_{case}_M"]).float()
    loss = NestedLoRALossFunctionEVD.apply(f, Tf, f1, f2, v, M)
    (loss * 1.5).backward()
    torch.cuda.synchronize()
    p = f"indep_{case}_f64_"
    assert abs(float(loss) - float(z[p + "loss"])) < 2e-5 * max(1.0, abs(float(z[p + "loss"])))
    assert Tf.grad is None
    assert rel(f.grad, z[p + "grad_f"]) < 2e-6
    if not f1_is_f:
        assert rel(f1.grad, z[p + "grad_f1"]) < 2e-6
    assert rel(f2.grad, z[p + "grad_f2"]) < 2e-6
    # and the oracle's restatement says the same
    lo, g, g1, g2 = O.evd_loss_independent(f.detach().double().cpu(), Tf.detach().double().cpu(),
                                           f1.detach().double().cpu(), f2.detach().double().cpu(), v.double(),
                                           M.double(), 1.5)
    assert abs(float(loss) - float(lo)) < 2e-5 * max(1.0, abs(float(lo)))
    assert rel(f2.grad, g2) < 2e-6


def test_evd_loss_large_L_and_B():
    g = torch.Generator().manual_seed(5)
    B, L = 8192, 64
    f = torch.randn(B, L, generator=g)
    Tf = torch.randn(B, L, generator=g)
    v, M = O.joint_nesting_masks(L, 1)
    loss, lam1, lam2, _, _ = O.evd_loss_forward(f.double(), Tf.double(), v.double(), M.double())
    gref = O.evd_loss_backward(f.double(), Tf.double(), v.double(), M.double(), lam1, lam2)
    fd, Td = f.to(DEV), Tf.to(DEV)
    mom = H.evd_moments(fd, Td, H.MASK_JOINT, None)
    l, df = H.evd_loss_grad(fd, Td, H.MASK_JOINT, None, None, mom)
    assert abs(float(l[0]) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(df, gref) < 5e-6
    # the one-call API falls back to the multi-launch pipeline at this size
    mom2, l2, df2 = torch.empty_like(mom), torch.empty(3, device=DEV), torch.empty_like(fd)
    H.evd_loss_fused(fd, Td, H.MASK_JOINT, None, None, mom2, l2, df2, H.evd_scratch(B, L, DEV))
    assert torch.equal(df2, df) and torch.equal(l2, l)


# ------------------------------------------------------------------------------ operator fwd/bwd on the goldens
SMALL = ["hyd_small", "osc_small", "hyd_ragged"]


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("case", SMALL)
def test_operator_forward_backward_small(case, path):
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    ref = O.loss_and_grads(x.double(), p.to(torch.float64), prob, v, M)
    r = run_hip(p, prob, x, v, M, _path(path), df_override=ref["df"])
    pre64, pre32 = f"{case}_f64_step0_", f"{case}_f32_step0_"
    assert rel(r["f"], z[pre64 + "f"]) < 2e-5
    check_tf(r["Tf"], z, case, cfg)
    assert rel(r["Tf"], z[pre64 + "Tf"]) < 1e-4, rel(r["Tf"], z[pre64 + "Tf"])  # even / odd stencil form: every path
    # gradients given the SAME df (isolates the backward kernels)
    names = G.trainable_names(z, case)
    for n, g, gr in zip(names, r["grads"], ref["grads"]):
        assert torch.isfinite(g).all(), n
        assert rel(g.view(-1), gr.reshape(-1)) < 3e-5, (n, rel(g.view(-1), gr.reshape(-1)))
    # loss given the path's own (f, Tf): tight
    l_given, *_ = O.evd_loss_forward(r["f"].double().cpu(), r["Tf"].double().cpu(), v.double(), M.double())
    assert abs(float(r["loss"][0]) - float(l_given)) < 1e-5 * abs(float(l_given))
    # end to end (df from the HIP loss kernels): rounds 1-3 could only band this (a point-wise float32 stencil does not
    # average its noise over 24-32 rows: the float32 reference's own gradients are ref_err = 1e-2 .. 1e-1 away); with the
    # stencil in even / odd form it is north_star's 1e-4 against the reference's float64 loss and gradients
    r2 = run_hip(p, prob, x, v, M, _path(path))
    assert abs(float(r2["loss"][0]) - float(z[pre64 + "loss"])) <= 1e-4 * abs(float(z[pre64 + "loss"]))
    for n, g in zip(names, r2["grads"]):
        g64 = z[pre64 + "grad_" + n]
        assert rel(g.view(-1), g64.reshape(-1)) < 1e-4, (n, rel(g.view(-1), g64.reshape(-1)),
                                                         rel(z[pre32 + "grad_" + n], g64))


@pytest.mark.parametrize("path", PATHS + ["bf16x3"])
@pytest.mark.parametrize("case", ["hyd_med", "cfg1"])
def test_operator_headline_shapes(case, path):
    """H=128x3 shapes (the fused-kernel shapes) with weights regenerated from the seed recipe."""
    z = G.load("model_headline")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_seed(cfg)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    r = run_hip(p, prob, x, v, M, _path(path))
    pre64, pre32 = f"{case}_f64_step0_", f"{case}_f32_step0_"
    assert rel(r["f"], z[pre64 + "f"]) < 2e-5
    check_tf(r["Tf"], z, case, cfg)
    l_given, *_ = O.evd_loss_forward(r["f"].double().cpu(), r["Tf"].double().cpu(), v.double(), M.double())
    assert abs(float(r["loss"][0]) - float(l_given)) < 1e-5 * abs(float(l_given))
    # every path carries the stencil in even / odd form: Tf - and with it the loss - agrees with the
    # FLOAT64 stencil to north_star's 1e-4, where the reference's own float32 arithmetic is a few per cent away
    assert r["path"] == ("generic" if path == "generic" else "fused_mfma"), r["path"]
    assert rel(r["Tf"], z[pre64 + "Tf"]) < 1e-4, rel(r["Tf"], z[pre64 + "Tf"])
    assert abs(float(r["loss"][0]) - float(z[pre64 + "loss"])) < 1e-4 * abs(float(z[pre64 + "loss"]))
    loss_ref_err = abs(float(z[pre32 + "loss"]) - float(z[pre64 + "loss"])) / abs(float(z[pre64 + "loss"]))
    assert abs(float(r["loss"][0]) - float(z[pre64 + "loss"])) <= max(4 * loss_ref_err, 2e-2) * abs(
        float(z[pre64 + "loss"]))
    names = G.trainable_names(z, case)
    stride = 997 if case == "hyd_med" else 9973
    # pooled yardstick: the float32 reference's own error over every sampled gradient element
    num = sum(float(np.sum((z[pre32 + "gradsample_" + n] - z[pre64 + "gradsample_" + n]) ** 2)) for n in names)
    den = sum(float(np.sum(z[pre64 + "gradsample_" + n] ** 2)) for n in names)
    pooled = (num / den) ** 0.5
    got_num = 0.0
    for n, g in zip(names, r["grads"]):
        gs64 = z[pre64 + "gradsample_" + n]
        got = g.reshape(-1)[::stride].double().cpu().numpy()
        got_num += float(np.sum((got - gs64) ** 2))
        gn = float(z[pre64 + "gradnorm_" + n])
        ref_gn_err = abs(float(z[pre32 + "gradnorm_" + n]) - gn) / gn
        assert abs(float(g.double().norm()) - gn) < max(4 * ref_gn_err, 4 * pooled, 2e-3) * gn, n
        if gs64.size >= 64:  # per-tensor element check only where the sample is statistically meaningful
            ref_err = rel(z[pre32 + "gradsample_" + n], gs64)
            assert rel(got, gs64) < max(4 * ref_err, 4 * pooled, 2e-3), (n, rel(got, gs64), ref_err)
    assert (got_num / den) ** 0.5 < max(4 * pooled, 2e-3), ((got_num / den) ** 0.5, pooled)
    # end to end in the scripts' default mode: with Tf at 1e-6 of the float64 stencil, d loss / d f and every sampled
    # gradient element follow - the pooled error against the FLOAT64 reference is at north_star's 1e-4 (the float32
    # reference's own pooled error on these cases: `pooled`, 1e-3 .. 1e-2)
    assert (got_num / den) ** 0.5 < 1e-4, ((got_num / den) ** 0.5, pooled)


@pytest.mark.parametrize("path", PATHS)
def test_backward_given_df_headline(path):
    """Backward kernels in isolation at H=128x3: same df in, float64 oracle gradients out."""
    z = G.load("model_headline")
    cfg = G.cfg_of(z, "hyd_med")
    prob = G.problem_of(cfg)
    p = G.params_from_seed(cfg)
    v, M = G.masks_of(z, "hyd_med")
    x = torch.tensor(z["hyd_med_x"][0])
    ref = O.loss_and_grads(x.double(), p.to(torch.float64), prob, v, M)
    r = run_hip(p, prob, x, v, M, _path(path), df_override=ref["df"])
    for i, (g, gr) in enumerate(zip(r["grads"], ref["grads"])):
        assert rel(g.view(-1), gr.reshape(-1)) < 3e-5, (i, rel(g.view(-1), gr.reshape(-1)))


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("case,fn", [("osc_small", "model_small"), ("hyd_ragged", "model_small"),
                                     ("hyd_med", "model_headline")])
def test_backward_evd_matches_two_step(case, fn, path):
    """nsvd_operator_backward_evd (loss gradient evaluated inside the backward kernels, from partial or
    reduced moments) == nsvd_evd_loss_grad followed by nsvd_operator_backward."""
    z = G.load(fn)
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case) if fn == "model_small" else G.params_from_seed(cfg)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    two = run_hip(p, prob, x, v, M, _path(path))
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    hp = hip_problem(prob)
    xd = x.float().to(DEV).contiguous()
    B, L = xd.shape[0], shape.L
    vd, Md = v.float().to(DEV), M.float().to(DEV).contiguous()
    fused = H.path_name(shape, B, _path(path)) == "fused_mfma"
    for mode in ("partial", "reduced", "direct"):
        reduced = mode == "reduced"
        gw = [torch.full_like(w, float("nan")) for w in ws_t]
        gb = [torch.full_like(b, float("nan")) for b in bs_t]
        gs = None if sc is None else torch.full_like(sc, float("nan"))
        grads = H.pack_params(shape, gw, gb, None, gs)
        ws = H.new_workspace(shape, B, DEV)
        f, Tf = H.operator_forward(shape, params, hp, xd, ws, path=_path(path))
        scratch = H.evd_scratch(B, L, DEV)
        mom = torch.full((2 * L * L + 1,), float("nan"), device=DEV)
        loss = torch.empty(3, device=DEV)
        if mode == "direct":
            # no moment kernel at all: the backward takes each head's moments from f (MFMA path only)
            if not fused:
                with pytest.raises(H.NsvdError):
                    H.operator_backward_evd(shape, params, hp, xd, f, Tf, H.MASK_CUSTOM, vd, Md, None, False, None, None,
                                            grads, ws, 1.0, _path(path))
                continue
            H.operator_backward_evd(shape, params, hp, xd, f, Tf, H.MASK_CUSTOM, vd, Md, None, False, None, None, grads,
                                    ws, 1.0, _path(path))
        else:
            if reduced:
                H.evd_moments(f, Tf, H.MASK_CUSTOM, vd, mom, scratch)
            else:
                H.evd_partial(f, Tf, H.MASK_CUSTOM, vd, scratch)
            H.operator_backward_evd(shape, params, hp, xd, f, Tf, H.MASK_CUSTOM, vd, Md, mom, reduced, scratch, loss,
                                    grads, ws, 1.0, _path(path))
        torch.cuda.synchronize()
        if mode != "direct":
            assert rel(mom, two["mom"]) < 1e-6
            assert abs(float(loss[0]) - float(two["loss"][0])) <= 1e-5 * abs(float(two["loss"][0]))
        for i, (g, g2) in enumerate(zip(gw + gb + ([gs] if gs is not None else []), two["grads"])):
            assert torch.isfinite(g).all()
            assert rel(g, g2) < 2e-5, (i, mode, rel(g, g2))


@pytest.mark.parametrize("B,hidden", [(65536, (128,)), (4099, (16,))])
def test_device_sampler(B, hidden):
    """nsvd_operator_sample_features: x ~ N(0, sigma^2) from the counter-based generator (fused into the feature
    kernel on the MFMA path, stand-alone on the generic path): moments of the draw, reproducibility in
    (seed, offset), and the features it leaves in the workspace equal those of nsvd_operator_features(x)."""
    D, m, sigma = 2, 64, 16.0
    shape = H.ModelShape(L=1, D=D, m=m, hidden=hidden)
    p = O.init_params(1, D, m, hidden, 0.1, seed=1)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, sigma)
    ws = H.new_workspace(shape, B, DEV)
    x = torch.empty(B, D, device=DEV)
    H.operator_sample_features(shape, params, prob, 1234, 7, x, ws)
    f1, Tf1 = H.operator_forward(shape, params, prob, x, ws, features_ready=True)
    xs = x.double().cpu()
    n = xs.numel()
    assert abs(float(xs.mean())) < 5 * sigma / np.sqrt(n)
    assert abs(float(xs.std()) / sigma - 1) < 5 / np.sqrt(2 * n)
    z = (xs / sigma).flatten()
    assert abs(float((z ** 3).mean())) < 5 * np.sqrt(15.0 / n)          # skewness
    assert abs(float((z ** 4).mean()) - 3.0) < 5 * np.sqrt(96.0 / n)    # kurtosis
    assert abs(float((xs[:, 0] * xs[:, 1]).mean())) / sigma ** 2 < 5 / np.sqrt(B)   # the two coordinates
    assert abs(float((xs[1:, 0] * xs[:-1, 0]).mean())) / sigma ** 2 < 5 / np.sqrt(B)  # neighbouring samples
    # Kolmogorov-Smirnov against the normal CDF
    from scipy import stats
    assert stats.kstest(z.numpy()[::max(1, n // 20000)], "norm").pvalue > 1e-3
    # pure function of (seed, offset); different offsets / seeds are different draws
    x2 = torch.empty_like(x)
    H.operator_sample_features(shape, params, prob, 1234, 7, x2, ws)
    assert torch.equal(x, x2)
    H.operator_sample_features(shape, params, prob, 1234, 8, x2, ws)
    assert not torch.equal(x, x2) and abs(float((x2.double().cpu() * xs).mean())) / sigma ** 2 < 5 / np.sqrt(n)
    H.operator_sample_features(shape, params, prob, 1235, 7, x2, ws)
    assert not torch.equal(x, x2)
    # the features written next to the draw are the features of that x
    f2, Tf2 = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, B, DEV))
    assert torch.equal(f1, f2) and torch.equal(Tf1, Tf2)


def test_device_sampler_never_draws_the_origin():
    """Regression: with 24 random bits + 1/2 the Box-Muller uniform rounded to exactly 1 once in 2^24 draws, the radius
    was exactly 0 and the hydrogen potential of that sample -inf (first seen at batch 4080, row 217 of a configs[1]
    run with sample key 1: every parameter was NaN one step later). The radius is now bounded below by
    sigma * sqrt(-2 log(1 - 2^-24)) = 3.45e-4 sigma; that very draw is replayed here."""
    D, m, sigma, B = 2, 64, 16.0, 512
    shape = H.ModelShape(L=1, D=D, m=m, hidden=(128,))
    p = O.init_params(1, D, m, (128,), 0.1, seed=1)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, sigma)
    ws = H.new_workspace(shape, B, DEV)
    x = torch.empty(B, D, device=DEV)
    rmin = sigma * float(np.sqrt(-2.0 * np.log1p(-2.0 ** -24)))
    H.operator_sample_features(shape, params, prob, 1, 4080, x, ws)
    r = x.double().norm(dim=1)
    assert float(r.min()) >= 0.99 * rmin and float(r[217]) < 100 * rmin, (float(r.min()), float(r[217]))
    f, Tf = H.operator_forward(shape, params, prob, x, ws, features_ready=True)
    assert torch.isfinite(f).all() and torch.isfinite(Tf).all()
    for off in range(4000, 4200):
        H.operator_sample_features(shape, params, prob, 1, off, x, ws)
        assert float(x.double().norm(dim=1).min()) >= 0.99 * rmin, off


@pytest.mark.parametrize("L,B,mask", [(2, 1024, True), (1, 2048, False)])
def test_backward_split_k_matches_oracle(L, B, mask):
    """Few heads on many rows (what a head-parallel rank sees): the weight-gradient kernel splits the batch
    contraction into slices (partial tiles + the reduce pass). Same df in, float64 oracle gradients out, and the
    MFMA path agrees with the generic FMA path."""
    D, m, hidden = 2, 128, (128, 128)
    p = O.init_params(L, D, m, hidden, 0.1, exp_mask_init=10.0 if mask else None, seed=11)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=4.0)
    v, M = O.joint_nesting_masks(L, 1)
    x = 4.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(B), dtype=torch.float64)
    x = x.float().double()
    ref = O.loss_and_grads(x, p.to(torch.float64), prob, v, M)
    fused = run_hip(p, prob, x, v, M, H.PATH_AUTO, df_override=ref["df"])
    assert fused["path"] == "fused_mfma"
    generic = run_hip(p, prob, x, v, M, H.PATH_GENERIC, df_override=ref["df"])
    for i, (g, g2, gr) in enumerate(zip(fused["grads"], generic["grads"], ref["grads"])):
        assert torch.isfinite(g).all()
        assert rel(g.view(-1), gr.reshape(-1)) < 3e-5, (i, rel(g.view(-1), gr.reshape(-1)))
        assert rel(g, g2) < 3e-5, i


@pytest.mark.parametrize("world", [2, 4])
def test_head_parallel_matches_single_rank(world):
    """Head-parallel sharding simulated on one GPU: each "rank" owns L/world heads of the same model, runs
    the forward on the whole batch, the (B, L/world) blocks are concatenated by hand (what the all-gather
    does), and nsvd_operator_backward_evd(L_total, l_offset) must give the head slices of the single-rank
    gradients and the same loss."""
    z = G.load("model_headline")
    cfg = G.cfg_of(z, "hyd_med")  # L = 4, H = 128 x 3: fused kernels
    prob = G.problem_of(cfg)
    p = G.params_from_seed(cfg)
    v, M = G.masks_of(z, "hyd_med")
    x = torch.tensor(z["hyd_med_x"][0])
    full = run_hip(p, prob, x, v, M, H.PATH_AUTO)
    L = p.ws[0].shape[0]
    Ll = L // world
    hp = hip_problem(prob)
    xd = x.float().to(DEV).contiguous()
    B = xd.shape[0]
    vd, Md = v.float().to(DEV), M.float().to(DEV).contiguous()
    ranks = []
    for r in range(world):
        pr = O.Params([w[r * Ll:(r + 1) * Ll] for w in p.ws], [b[r * Ll:(r + 1) * Ll] for b in p.bs], p.fourier_B)
        shape = shape_of(pr)
        ws_t, bs_t, fB, sc = to_dev(pr)
        params = H.pack_params(shape, ws_t, bs_t, fB, sc)
        ws = H.new_workspace(shape, B, DEV)
        f, Tf = H.operator_forward(shape, params, hp, xd, ws)
        ranks.append(dict(shape=shape, params=params, ws=ws, f=f, Tf=Tf, ws_t=ws_t, bs_t=bs_t))
    f_g = torch.cat([r["f"] for r in ranks], dim=1).contiguous()
    Tf_g = torch.cat([r["Tf"] for r in ranks], dim=1).contiguous()
    assert rel(f_g, full["f"]) < 1e-6
    for r, R in enumerate(ranks):
        gw = [torch.full_like(w, float("nan")) for w in R["ws_t"]]
        gb = [torch.full_like(b, float("nan")) for b in R["bs_t"]]
        grads = H.pack_params(R["shape"], gw, gb, None, None)
        scratch = H.evd_scratch(B, L, DEV)
        H.evd_partial(f_g, Tf_g, H.MASK_CUSTOM, vd, scratch)
        mom = torch.empty(2 * L * L + 1, device=DEV)
        loss = torch.empty(3, device=DEV)
        H.operator_backward_evd(R["shape"], R["params"], hp, xd, f_g, Tf_g, H.MASK_CUSTOM, vd, Md, mom, False,
                                scratch, loss, grads, R["ws"], 1.0, H.PATH_AUTO, l_offset=r * Ll)
        torch.cuda.synchronize()
        assert abs(float(loss[0]) - float(full["loss"][0])) < 1e-5 * abs(float(full["loss"][0]))
        nl = len(gw)
        for i in range(nl):
            assert rel(gw[i], full["grads"][i][r * Ll:(r + 1) * Ll]) < 2e-5, (r, i)
            assert rel(gb[i], full["grads"][nl + i][r * Ll:(r + 1) * Ll]) < 2e-5, (r, i)


def test_edge_cases_clamp_and_origin():
    """sqrt(p) clamp (far-out samples), x exactly at the origin (hydrogen potential singular -> the
    same inf/nan pattern as the oracle), B = 2 (one row per half)."""
    L, D, m, hidden = 3, 2, 4, (8,)
    p = O.init_params(L, D, m, hidden, 0.1, seed=11)
    prob = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, sigma=1.0)
    x = torch.tensor([[8.0, 7.0], [0.0, 0.0], [0.3, -0.2], [30.0, 1.0]])  # sigma=1: rows 0 and 3 hit the clamp
    c = O.operator_forward(x.double(), p.to(torch.float64), prob)
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    ws = H.new_workspace(shape, 4, DEV)
    f, Tf = H.operator_forward(shape, params, hip_problem(prob), x.to(DEV), ws)
    f, Tf = f.cpu(), Tf.cpu()
    assert rel(f[[0, 2]], c.f[[0, 2]]) < 1e-4
    assert torch.equal(torch.isfinite(Tf), torch.isfinite(c.Tf.float()))
    assert float(c.spc0[0]) == 1e-5  # the clamp really is active in this test
    x2 = x[2:4].contiguous()
    v, M = O.sequential_nesting_masks(L)
    r = run_hip(p, O.Problem(potential=O.POT_HARMONIC, eps=0.01, sigma=4.0, op_shift=16.0), x2, v, M, H.PATH_AUTO)
    ref = O.loss_and_grads(x2.double(), p.to(torch.float64), O.Problem(potential=O.POT_HARMONIC, eps=0.01, sigma=4.0,
                                                                        op_shift=16.0), v, M)
    assert rel(r["f"], ref["f"]) < 2e-5


def test_invalid_arguments_are_rejected():
    from neural_svd_amd._lib import NsvdError
    shape = H.ModelShape(L=2, D=2, m=4, hidden=(8,))
    p = O.init_params(2, 2, 4, (8,), 0.1, seed=0)
    ws_t, bs_t, fB, _ = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, None)
    x = torch.zeros(4, 2, device=DEV)
    small = torch.empty(256, dtype=torch.uint8, device=DEV)
    with pytest.raises(NsvdError):
        H.operator_forward(shape, params, H.make_problem(0, 1.0, 0.01, 1.0, 0.0, 1.0), x, small)  # workspace too small
    ws = H.new_workspace(shape, 4, DEV)
    with pytest.raises(NsvdError):
        H.operator_forward(shape, params, H.make_problem(0, 1.0, 0.0, 1.0, 0.0, 1.0), x, ws)  # eps == 0
    with pytest.raises(NsvdError):
        H.operator_forward(shape, params, H.make_problem(7, 1.0, 0.01, 1.0, 0.0, 1.0), x, ws)  # unknown potential
    with pytest.raises(NsvdError):
        H.pack_params(shape, ws_t[::-1], bs_t, fB, None)  # wrong shapes


# ------------------------------------------------------------------------------ model forward (eigenfunction eval)
@pytest.mark.parametrize("case", ["hyd_small", "osc_small"])
def test_model_forward(case):
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    p = G.params_from_golden(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    p64 = p.to(torch.float64)
    base = O.mlp_forward(O.fourier_features(x.double(), p64.fourier_B), p64)
    mk = O.boundary_mask(x.double(), p64)
    ref = base if mk is None else base * mk
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    out = H.model_forward(shape, params, x.to(DEV), 1.0, H.new_workspace(shape, x.shape[0], DEV))
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize("fpath", FPATHS)
@pytest.mark.parametrize("case", ["hyd_exact", "osc_exact"])
def test_exact_laplacian_mode(case, fpath):
    """laplacian_eps = 0: forward-mode jets in the fused kernel against the reference's double-autograd exact mode
    (goldens model_exact.npz). No finite differences, so Tf is held to the tolerance of f: 2e-5 relative (the
    float32 reference itself is 1e-6 .. 1e-5 from its float64 values here); loss and gradients follow."""
    z = G.load("model_exact")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    assert prob.eps == 0.0
    p = G.params_from_golden(z, case)
    v, M = G.masks_of(z, case)
    x = torch.tensor(z[f"{case}_x"][0])
    r = run_hip(p, prob, x, v, M, _path(fpath))
    assert r["path"] == "fused_mfma"
    pre = f"{case}_f64_step0_"
    assert rel(r["f"], z[pre + "f"]) < 2e-5
    ref32 = rel(z[f"{case}_f32_step0_Tf"], z[pre + "Tf"])
    assert rel(r["Tf"], z[pre + "Tf"]) < max(2e-5, 3 * ref32), (rel(r["Tf"], z[pre + "Tf"]), ref32)
    assert abs(float(r["loss"][0]) - float(z[pre + "loss"])) < 1e-4 * abs(float(z[pre + "loss"]))
    for n, g in zip(G.trainable_names(z, case), r["grads"]):
        gs = g.reshape(-1).double().cpu().numpy()
        assert abs(np.linalg.norm(gs) - float(z[pre + f"gradnorm_{n}"])) < 1e-4 * float(z[pre + f"gradnorm_{n}"]), n
        assert G.rel(gs[::13], z[pre + f"gradsample_{n}"]) < 1e-4, n
    # shapes the MFMA path does not take have no exact mode
    zs = "osc_exact_small"
    ps = G.params_from_golden(z, zs)
    with pytest.raises(Exception):
        run_hip(ps, G.problem_of(G.cfg_of(z, zs)), torch.tensor(z[f"{zs}_x"][0]), *G.masks_of(z, zs), H.PATH_AUTO)


@pytest.mark.parametrize("fpath", FPATHS)
@pytest.mark.parametrize("eps", [0.01, 0.0])
def test_fused_path_one_dimensional(eps, fpath):
    """D = 1 on the MFMA path (E = 3 stencil instance, and the 3-stream jet instance for eps = 0) against the
    float64 oracle: f to 2e-5; Tf to 1e-4 in exact mode, to the finite-difference noise level otherwise; gradients
    given the oracle's d loss / d f to 3e-5."""
    L, D, m, hidden, B = 3, 1, 64, (128, 128), 64
    p = O.init_params(L, D, m, hidden, 0.3, exp_mask_init=5.0, seed=21)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=eps, op_scale=1.0, op_shift=4.0, sigma=2.0)
    v, M = O.joint_nesting_masks(L, 1)
    x = (2.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(4), dtype=torch.float64)).float().double()
    ref = O.loss_and_grads(x, p.to(torch.float64), prob, v, M)
    r = run_hip(p, prob, x, v, M, _path(fpath), df_override=ref["df"])
    assert r["path"] == "fused_mfma"
    assert rel(r["f"], ref["f"]) < 2e-5
    if eps > 0:
        kcfg = dict(operator_scale=1.0, laplacian_eps=eps)
        k = tf_noise_kappa(r["Tf"], ref["Tf"].numpy(), ref["f"].numpy(), kcfg)
        k_ref = oracle32_kappa(x, p, prob, ref, kcfg)
        assert k <= 1.5 * k_ref, (k, k_ref)
    else:
        assert rel(r["Tf"], ref["Tf"]) < 1e-4
    for i, (g, gr) in enumerate(zip(r["grads"], ref["grads"])):
        assert rel(g.view(-1), gr.reshape(-1)) < 3e-5, i


@pytest.mark.parametrize("fpath", FPATHS)
def test_exact_mode_three_dimensional(fpath):
    """D = 3 in exact mode (5 jet streams in one workgroup): f, Tf and the gradients against the float64 oracle,
    hydrogen potential with the exponential mask; the stencil mode of the same model (7 columns per sample) runs too."""
    L, D, m, hidden, B = 2, 3, 64, (128, 128, 128), 64
    p = O.init_params(L, D, m, hidden, 0.2, exp_mask_init=4.0, seed=33)
    prob = O.Problem(potential=O.POT_HYDROGEN, eps=0.0, op_scale=10.0, op_shift=1.0, sigma=3.0, hard_mul_const=0.9)
    v, M = O.sequential_nesting_masks(L)
    x = (3.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(8), dtype=torch.float64)).float().double()
    ref = O.loss_and_grads(x, p.to(torch.float64), prob, v, M)
    r = run_hip(p, prob, x, v, M, _path(fpath), df_override=ref["df"])
    assert rel(r["f"], ref["f"]) < 2e-5
    assert rel(r["Tf"], ref["Tf"]) < 1e-4
    for i, (g, gr) in enumerate(zip(r["grads"], ref["grads"])):
        assert rel(g.view(-1), gr.reshape(-1)) < 3e-5, i
    # ... and the stencil mode of the same model (since round 3 on the MFMA kernels too: split-stencil form below)
    prob_fd = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=10.0, op_shift=1.0, sigma=3.0, hard_mul_const=0.9)
    r2 = run_hip(p, prob_fd, x, v, M, H.PATH_AUTO)
    assert rel(r2["f"], ref["f"]) < 2e-5


@pytest.mark.parametrize("hidden,B,D,L,m", [
    ((64, 64, 64), 512, 2, 3, 256),   # 64 x 64 tiles (M <= 64), every operand form vectorised; K groups (K = 512, 3 tiles)
    ((256, 256), 512, 2, 2, 256),     # 128 x 128 tiles where the launch has >= 1024 workgroups, smaller ones elsewhere
    ((96, 96), 128, 2, 5, 128),       # M = 96: clamped rows of the second 64-row tile
    ((40, 24), 100, 2, 3, 34),        # K = 68 / 40 / 24 / 100 (K tails by select), N = 500, M = 40 / 24 (ragged everywhere)
    ((50,), 101, 3, 2, 33),           # nothing 16-byte aligned: the scalar kernel (gemm_generic2) takes every launch
    ((320, 64), 64, 1, 2, 64),        # M = 320 (three 128-row tiles, the last ragged), D = 1 (three stencil blocks)
    ((64, 48), 272, 2, 2, 40),        # K groups with K = 272: 17 K steps over four groups (5, 5, 5, 2), M = 48 ragged
])
@pytest.mark.parametrize("minwg", [None, "1", "1000000"])
def test_generic_path_at_other_hidden_widths(hidden, B, D, L, m, minwg, monkeypatch):
    """Hidden widths the fused MFMA kernels do not take (the reference accepts any --mlp_hidden_dims,
    examples/models/mlp.py:187-221) run the generic contractions (gemm_generic.hip: round 6's vectorised kernel with its
    four tile shapes, clamped edges and K tails, the scalar kernel for unaligned launches, the in-place even / odd
    softplus pass between layers - vectorised and scalar): f, Tf and every gradient against the float64 oracle at the
    same tolerances as the fused path. minwg: the workgroup count the tile choice asks for (NSVD_G3_MINWG, read per
    launch) - "1" gives every launch the LARGEST tile its M allows (128 x 128 / 64 x 256: ragged edges inside big tiles),
    "1000000" the smallest (64 x 64), None the default rule."""
    if minwg is None:
        monkeypatch.delenv("NSVD_G3_MINWG", raising=False)
    else:
        monkeypatch.setenv("NSVD_G3_MINWG", minwg)
    p = O.init_params(L, D, m, hidden, 0.2, exp_mask_init=4.0, seed=7)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=3.0)
    v, M = O.sequential_nesting_masks(L)
    x = (3.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(11), dtype=torch.float64)).float().double()
    assert H.path_name(shape_of(p), B, H.PATH_AUTO, hip_problem(prob)) == "generic"
    ref = O.loss_and_grads(x, p.to(torch.float64), prob, v, M)
    r = run_hip(p, prob, x, v, M, H.PATH_AUTO, df_override=ref["df"])
    assert rel(r["f"], ref["f"]) < 2e-5
    assert rel(r["Tf"], ref["Tf"]) < 1e-4, rel(r["Tf"], ref["Tf"])
    for i, (a, b) in enumerate(zip(r["grads"], ref["grads"])):
        assert torch.isfinite(a).all(), i
        assert rel(a.view(-1), b.reshape(-1)) < 3e-5, (i, rel(a.view(-1), b.reshape(-1)))



@pytest.mark.parametrize("D,L,B,m", [(3, 3, 96, 64), (2, 16, 128, 64), (2, 16, 128, 256), (2, 4, 64, 128)])
def test_split_stencil_form(D, L, B, m):
    """The split-stencil form of the fused forward (one direction's two shifted points + the centre per workgroup, raw
    head outputs combined by the generic FD epilogue): the only way the 7 stencil columns of a 3-D problem fit the
    MFMA kernels, and what small batches (configs[0]: 64 workgroups of the plain form on 256 CUs) take to fill the chip.
    f, the FD-noise yardstick on Tf, and every gradient against the float64 oracle; f and the saved state bit-identical
    to ... nothing else computes them, so: the generic kernels as a second witness at float32 level.
    The last two cases (m >= 128, at most 64 plain workgroups: configs[0]'s situation) also cut layer 0 into two K slices
    of one workgroup each, added by a second launch that runs the rest of the network (pmlp_common.h: fwd_kslices)."""
    hidden = (128, 128, 128)
    p = O.init_params(L, D, m, hidden, 0.2, exp_mask_init=4.0, seed=44)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=3.0)
    v, M = O.sequential_nesting_masks(L)
    x = (3.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(9), dtype=torch.float64)).float().double()
    shape = shape_of(p)
    assert H.path_name(shape, B, H.PATH_AUTO, hip_problem(prob)) == "fused_mfma"
    ref = O.loss_and_grads(x, p.to(torch.float64), prob, v, M)
    r = run_hip(p, prob, x, v, M, H.PATH_FUSED, df_override=ref["df"])
    g = run_hip(p, prob, x, v, M, H.PATH_GENERIC, df_override=ref["df"])
    assert rel(r["f"], ref["f"]) < 2e-5 and rel(g["f"], ref["f"]) < 2e-5
    # Tf: float32 finite differences - the MFMA path no further from float64 than 2 x the generic kernels are
    assert rel(r["Tf"], ref["Tf"]) < max(2.0 * rel(g["Tf"], ref["Tf"]), 1e-3)
    for i, (a, b) in enumerate(zip(r["grads"], ref["grads"])):
        assert rel(a.view(-1), b.reshape(-1)) < 3e-5, i


def test_k_split_forward_agrees_with_the_unsplit_kernel():
    """configs[0]'s K-split (layer 0 of a 128-row batch cut into four K slices, pmlp_common.h: fwd_kslices) against the
    plain kernel: the same 128 rows evaluated as the head of a 512-row batch (256 plain workgroups: no K-split). The
    rows never see their neighbours, so f and Tf agree up to the summation order of layer 0."""
    D, L, m, hidden = 2, 16, 1024, (128, 128, 128)
    p = O.init_params(L, D, m, hidden, 0.1, exp_mask_init=None, seed=5)
    prob = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    hp = hip_problem(prob)
    x = (16.0 * torch.randn(512, D, generator=torch.Generator().manual_seed(2))).float().to(DEV)
    f_big, Tf_big = H.operator_forward(shape, params, hp, x, H.new_workspace(shape, 512, DEV), path=H.PATH_FUSED)
    xs = x[:128].contiguous()
    f_ks, Tf_ks = H.operator_forward(shape, params, hp, xs, H.new_workspace(shape, 128, DEV), path=H.PATH_FUSED)
    assert rel(f_ks, f_big[:128]) < 2e-6
    assert rel(Tf_ks, Tf_big[:128]) < 2e-5  # (the stencil amplifies the last bits of layer 0 by 1 / eps^2 x eps-sized terms)


@pytest.mark.parametrize("L,B,mask,m", [(16, 128, False, 1024), (8, 128, True, 1024), (4, 64, True, 192), (16, 96, False, 256)])
def test_k_split_epilogue_fold_is_bit_identical(L, B, mask, m, monkeypatch):
    """K-split forwards (configs[0]: four K slices + the split form; smaller grids: two slices): the direction group that
    arrives LAST at a (head, sample block) forms f, Tf inside the second launch (arrival tickets, one agent-scope
    release / acquire pair: pmlp_fwd.hip) - against the same launches followed by the separate finite-difference
    epilogue kernel (NSVD_KSPLIT_FOLD=0): f, Tf and the saved Jacobian factors bit for bit, repeatedly (whichever
    group arrives last, the arithmetic is the same), and the backward that consumes them."""
    D, hidden = 2, (128, 128, 128)  # (m = 192: two K slices in split form; the others: four in plain form)
    p = O.init_params(L, D, m, hidden, 0.1, exp_mask_init=10.0 if mask else None, seed=L + B)
    prob = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    hp = hip_problem(prob)
    x = (16.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(B))).float().to(DEV)
    df = torch.randn(B, L, generator=torch.Generator().manual_seed(L)).to(DEV) / B

    def run():
        ws = H.new_workspace(shape, B, DEV)
        f, Tf = H.operator_forward(shape, params, hp, x, ws, True, H.PATH_FUSED)
        gw = [torch.zeros_like(w) for w in ws_t]
        gb = [torch.zeros_like(b) for b in bs_t]
        gs = torch.zeros_like(sc) if sc is not None else None
        H.operator_backward(shape, params, hp, x, df, H.pack_params(shape, gw, gb, None, gs), ws, H.PATH_FUSED)
        torch.cuda.synchronize()
        return [f.clone(), Tf.clone()] + gw + gb + ([gs] if gs is not None else [])
    monkeypatch.setenv("NSVD_KSPLIT_FOLD", "0")
    want = run()
    monkeypatch.delenv("NSVD_KSPLIT_FOLD")
    for _ in range(5):
        got = run()
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), i
    assert bool(torch.isfinite(want[0]).all()) and float(want[1].abs().max()) > 0


@pytest.mark.parametrize("D,L,B,mask", [(16, 3, 64, False), (2, 4, 96, True), (40, 1, 32, True), (16, 64, 1024, False),
                                        (5, 32, 2048, True)])
def test_model_forward_backward_mfma(D, L, B, mask):
    """Plain model evaluation with 128-wide hidden layers takes the plain-tile instances of the fused MFMA forward (any
    input dimension up to 64; one 32-sample tile per workgroup, or - the last two cases: at least 512 workgroups of
    128 samples - four) and the fused backward: c * model(x) and the parameter gradients of sum(dout * out) against the
    float64 oracle."""
    m, hidden, c = 64, (128, 128), 0.7
    p = O.init_params(L, D, m, hidden, 0.05, exp_mask_init=6.0 if mask else None, seed=D)
    p64 = p.to(torch.float64)
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, D, generator=g, dtype=torch.float64).float().double()
    dout = torch.randn(B, L, generator=g, dtype=torch.float64).float().double()
    phi = O.fourier_features(x, p64.fourier_B)
    base = O.mlp_forward(phi, p64)
    mk = O.boundary_mask(x, p64)
    ref = c * (base if mk is None else base * mk)
    # float64 gradients through autograd on the oracle's own forward
    leaves = [t.clone().requires_grad_(True) for t in p64.trainable()]
    nl = len(p64.ws)
    q = O.Params(leaves[:nl], leaves[nl:2 * nl], p64.fourier_B, leaves[2 * nl] if mask else None)
    outq = O.mlp_forward(phi, q)
    mq = O.boundary_mask(x, q)
    (c * (outq if mq is None else outq * mq) * dout).sum().backward()
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    gw = [torch.full_like(w, float("nan")) for w in ws_t]
    gb = [torch.full_like(b, float("nan")) for b in bs_t]
    gs = None if sc is None else torch.full_like(sc, float("nan"))
    grads = H.pack_params(shape, gw, gb, None, gs)
    ws = H.model_workspace(shape, B, DEV)
    xd = x.float().to(DEV).contiguous()
    out = H.model_forward(shape, params, xd, c, ws, save_for_backward=True)
    H.model_backward(shape, params, xd, dout.float().to(DEV).contiguous(), grads, ws)
    torch.cuda.synchronize()
    assert rel(out, ref) < 1e-5
    for i, (got, leaf) in enumerate(zip(gw + gb + ([gs] if gs is not None else []), leaves)):
        assert torch.isfinite(got).all()
        assert rel(got, leaf.grad) < 3e-5, (i, rel(got, leaf.grad))


# ------------------------------------------------------------------------------ optimiser
@pytest.mark.parametrize("n", [1, 3, 1000, 4099, 1 << 20])
def test_rmsprop_ema(n):
    g = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g) * 10
    sq = torch.rand(n, generator=g)
    em = torch.randn(n, generator=g)
    pr, sr, er = [p.double().clone()], [sq.double().clone()], [em.double().clone()]
    pd, gd, sd, ed = p.to(DEV), gr.to(DEV), sq.to(DEV), em.to(DEV)
    nup = 0
    for it in range(3):
        lr = O.cosine_lr(1e-3, it, 10)
        O.rmsprop_step(pr, [gr.double()], sr, lr, alpha=0.999, eps=1e-10)
        d = min(0.995, (1 + nup + 1) / (10 + nup + 1))
        nup = O.ema_update(er, pr, 0.995, nup)
        H.rmsprop_ema_step(pd, gd, sd, ed, lr, 0.999, 1e-10, d)
    torch.cuda.synchronize()
    assert rel(pd, pr[0]) < 1e-6 and rel(sd, sr[0]) < 1e-6 and rel(ed, er[0]) < 1e-6
    # no-EMA variant and grad_scale
    p2 = p.to(DEV)
    s2 = sq.to(DEV)
    H.rmsprop_ema_step(p2, gd * 4, s2, None, 1e-3, 0.999, 1e-10, 0.0, grad_scale=0.25)
    p3, s3 = [p.double().clone()], [sq.double().clone()]
    O.rmsprop_step(p3, [gr.double()], s3, 1e-3)
    assert rel(p2, p3[0]) < 1e-6


def test_rmsprop_golden_two_steps():
    """the reference's own RMSprop + cosine schedule trajectory (float32 golden)."""
    z = G.load("model_small")
    case = "hyd_small"
    cfg = G.cfg_of(z, case)
    names = G.trainable_names(z, case)
    for n in names:
        p = torch.tensor(z[f"{case}_param0_{n}"]).float().to(DEV).contiguous()
        sq = torch.zeros_like(p)
        for it in range(2):
            g = torch.tensor(z[f"{case}_f32_step{it}_grad_{n}"]).float().to(DEV).contiguous()
            H.rmsprop_ema_step(p.view(-1), g.view(-1), sq.view(-1), None, O.cosine_lr(cfg["lr"], it, cfg["num_iters"]),
                               cfg["rmsprop_decay"], 1e-10, 0.0)
            assert rel(p, z[f"{case}_f32_step{it}_param_{n}"]) < 1e-6, (n, it)


# ------------------------------------------------------------------------------ spectrum
@pytest.mark.parametrize("case", SMALL)
def test_spectrum_matches_reference(case):
    z = G.load("model_small")
    cfg = G.cfg_of(z, case)
    prob = G.problem_of(cfg)
    p = G.params_from_golden(z, case, prefix="f64_step1_param_")
    grid = torch.tensor(z[f"{case}_val_data"])
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    L = shape.L
    cov = torch.zeros(L, L, device=DEV)
    quad = torch.zeros(L, L, device=DEV)
    chunk = 150  # ragged chunks on purpose
    for i in range(0, grid.shape[0], chunk):
        xb = grid[i:i + chunk].to(DEV).contiguous()
        ws = H.new_workspace(shape, xb.shape[0], DEV)
        f, Tf = H.operator_forward(shape, params, hip_problem(prob), xb, ws, save_for_backward=False)
        H.spectrum_accumulate(f, Tf, xb, prob.sigma, True, cfg["lim"], cov, quad)
    n = grid.shape[0]
    cov, quad = cov.cpu().double() / n, quad.cpu().double() / n
    eig = torch.diag(quad) / torch.diag(cov)
    assert rel(torch.diag(cov), z[f"{case}_f64_spec_norms"]) < 1e-4
    e64, e32 = z[f"{case}_f64_spec_eigvals"], z[f"{case}_f32_spec_eigvals"]
    ref_err = float(np.max(np.abs(e32 - e64) / np.abs(e64)))
    got_err = float(np.max(np.abs(eig.numpy() - e64) / np.abs(e64)))
    assert got_err < max(3 * ref_err, 1e-4), (got_err, ref_err)


@pytest.mark.parametrize("fpath", FPATHS)
def test_ragged_evaluation_batch_goes_through_the_mfma_kernels(fpath):
    """An evaluation batch (no backward layout) of ANY size on a model the MFMA kernels take - the validation grids of
    the reference's scripts are not multiples of 32 rows - is padded onto them and the padding dropped: the rows are
    bit for bit the rows of the padded call (and Tf within 1e-4 of the float64 stencil). Also a batch beyond 8192 rows
    (pieces), and the generic kernels asked for by name."""
    L, D, m, hidden = 4, 2, 64, (128, 128, 128)
    p = O.init_params(L, D, m, hidden, 0.15, seed=3)
    prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
    shape, prob = shape_of(p), hip_problem(prob_o)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    path = _path(fpath)
    for B in (1060, 7, 8192 + 45):
        x = (16.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(B))).to(DEV)
        f, Tf = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, B, DEV), False, path)
        assert f.shape == (B, L) and Tf.shape == (B, L)
        lo = (B - 1) // 8192 * 8192          # the last piece, padded by hand
        xp = torch.cat([x[lo:], x[-1:].expand(-B % 32, -1)]).contiguous()
        fp, Tfp = H.operator_forward(shape, params, prob, xp, H.new_workspace(shape, xp.shape[0], DEV), False, path)
        assert torch.equal(f[lo:], fp[:B - lo]) and torch.equal(Tf[lo:], Tfp[:B - lo])
        rows = torch.arange(0, B, max(1, B // 64))
        ref = O.operator_forward(x[rows.to(DEV)].double().cpu(), p.to(torch.float64), prob_o)
        assert rel(f[rows.to(DEV)], ref.f) < 2e-5
        assert rel(Tf[rows.to(DEV)], ref.Tf) < 1e-4, (B, rel(Tf[rows.to(DEV)], ref.Tf))
    # the generic kernels, asked for by name, take the ragged batch as it is (the same even / odd stencil, FMA GEMMs)
    x = (16.0 * torch.randn(100, D, generator=torch.Generator().manual_seed(1))).to(DEV)
    fg, Tfg = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, 100, DEV), False, H.PATH_GENERIC)
    f, Tf = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, 100, DEV), False, path)
    assert rel(fg, f.double().cpu()) < 1e-5 and rel(Tfg, Tf.double().cpu()) < 1e-4


@pytest.mark.parametrize("path", ["generic", "auto", "bf16x3"])
@pytest.mark.parametrize("eps,wscale", [(0.3, 1.0), (0.01, 30.0), (0.05, 4.0), (0.5, 20.0)])
def test_wide_stencils_and_large_perturbations(eps, wscale, path):
    """The even / odd form expands the softplus around the centre value - valid for small perturbations, which is what
    eps = 0.01 gives; laplacian_eps is the caller's, though (and so are the weights): beyond |perturbation| = 0.25 the
    kernels take the plain differences of softplus values, which are accurate THERE. Wide stencils and inflated first-
    layer weights, every path, against the float64 stencil of the same eps."""
    L, D, m, hidden = 4, 2, 64, (128, 128, 128)
    p = O.init_params(L, D, m, hidden, 1.0, exp_mask_init=10.0, seed=11)
    p.ws[0] = p.ws[0] * wscale
    prob_o = O.Problem(potential=O.POT_HARMONIC, eps=eps, op_scale=1.0, op_shift=16.0, sigma=4.0)
    shape, prob = shape_of(p), hip_problem(prob_o)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    B = 96
    x = (4.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(12))).to(DEV)
    f, Tf = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, B, DEV), False, _path(path))
    ref = O.operator_forward(x.double().cpu(), p.to(torch.float64), prob_o)
    assert rel(f, ref.f) < 2e-5
    assert rel(Tf, ref.Tf) < 1e-4, rel(Tf, ref.Tf)


def test_spectrum_accumulators_float64():
    """nsvd_spectrum_accumulate_f64 (the evaluation paths' accumulators) against numpy float64 on the same float32
    inputs, and the float32 accumulators (the reference's) beside it."""
    g = torch.Generator().manual_seed(2)
    B, L, D = 5000, 16, 2
    f = torch.randn(B, L, generator=g).to(DEV)
    Tf = (100.0 * torch.randn(B, L, generator=g)).to(DEV)
    x = (10.0 * torch.randn(B, D, generator=g)).to(DEV)
    x[17] = 0.0  # a row at the origin: its Tphi is zeroed (methods/spectrum.py:73)
    c64 = torch.zeros(L, L, dtype=torch.float64, device=DEV)
    q64 = torch.zeros_like(c64)
    c32 = torch.zeros(L, L, device=DEV)
    q32 = torch.zeros_like(c32)
    for i in range(0, B, 1700):
        H.spectrum_accumulate(f[i:i + 1700].contiguous(), Tf[i:i + 1700].contiguous(), x[i:i + 1700].contiguous(), 16.0, True, 50.0, c64, q64)
        H.spectrum_accumulate(f[i:i + 1700].contiguous(), Tf[i:i + 1700].contiguous(), x[i:i + 1700].contiguous(), 16.0, True, 50.0, c32, q32)
    xd = x.double().cpu()
    sp = torch.exp(-(xd ** 2).sum(1) / (4 * 16.0 ** 2)) / math.sqrt(2 * math.pi * 16.0 ** 2)  # sqrt of the D = 2 Gaussian pdf
    w = (sp * math.sqrt(100.0 ** 2)).unsqueeze(1)   # / sqrt(p_val), p_val = 1 / (2 lim)^D
    ph, tp = w * f.double().cpu(), w * Tf.double().cpu()
    tp[17] = 0.0
    assert rel(c64, ph.T @ ph) < 1e-6 and rel(q64, ph.T @ tp) < 1e-6  # (the weights are float32: 1e-7 each)
    assert rel(c32, ph.T @ ph) < 1e-5 and rel(q32, ph.T @ tp) < 1e-4
    with pytest.raises(H.NsvdError):
        H.spectrum_accumulate(f, Tf, x, 16.0, True, 50.0, c64, q32)


# ------------------------------------------------------------------------------ full-size properties (cfg2)
@pytest.mark.parametrize("fpath", FPATHS)
@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_headline_size_properties(cfg, fpath):
    """BASELINE.json configs[1] (hydrogen, L = 16, B = 512, m = 1024, H = 128 x 3) and configs[2] per GPU
    (oscillator, L = 32, B = 512, m = 256, exponential mask) at full size, where a float64 oracle run is out of reach
    for a unit test: size-independent properties of the operator.
      * run-to-run bit reproducibility (no atomics anywhere on the path);
      * row equivariance: permuting the batch permutes f and Tf bit for bit (a sample never sees its neighbours);
      * homogeneity in the last layer: scaling W_last, b_last by c scales f and Tf by c;
      * a sampled float64 check: 8 rows x all heads against the oracle."""
    if cfg == "cfg2":
        L, D, m, hidden, B = 16, 2, 1024, (128, 128, 128), 512
        p = O.init_params(L, D, m, hidden, 0.1, seed=0)
        prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
        kcfg = dict(operator_scale=100.0, laplacian_eps=0.01)
    else:
        L, D, m, hidden, B = 32, 2, 256, (128, 128, 128), 512
        p = O.init_params(L, D, m, hidden, 1.0, exp_mask_init=10.0, seed=0)
        prob_o = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=4.0)
        kcfg = dict(operator_scale=1.0, laplacian_eps=0.01)
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    prob = hip_problem(prob_o)
    x = (prob_o.sigma * torch.randn(B, D, generator=torch.Generator().manual_seed(5))).to(DEV)
    ws = H.new_workspace(shape, B, DEV)
    path = _path(fpath)
    f, Tf = H.operator_forward(shape, params, prob, x, ws, path=path)
    f2, Tf2 = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, B, DEV), path=path)
    assert H.path_name(shape, B, H.PATH_AUTO, prob) == "fused_mfma"
    assert torch.equal(f, f2) and torch.equal(Tf, Tf2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(6)).to(DEV)
    fp, Tfp = H.operator_forward(shape, params, prob, x[perm].contiguous(), ws, path=path)
    assert torch.equal(fp, f[perm]) and torch.equal(Tfp, Tf[perm])
    c = 2.0  # a power of two: the scaled run is the same arithmetic with shifted exponents
    ws_c = list(ws_t[:-1]) + [ws_t[-1] * c]
    bs_c = list(bs_t[:-1]) + [bs_t[-1] * c]
    fc, Tfc = H.operator_forward(shape, H.pack_params(shape, ws_c, bs_c, fB, sc), prob, x, ws, path=path)
    assert torch.equal(fc, c * f) and torch.equal(Tfc, c * Tf)
    rows = torch.tensor([0, 1, 63, 64, 255, 256, 300, 511])
    ref = O.operator_forward(x[rows.to(DEV)].double().cpu(), p.to(torch.float64), prob_o)
    assert rel(f[rows.to(DEV)], ref.f) < 2e-5
    # even / odd stencil form: Tf against the FLOAT64 stencil at north_star's tolerance (measured at configs[1]:
    # 1.5e-6 native, 6.9e-6 bf16x3; the float32 oracle - the reference's arithmetic - is at 4e-2)
    assert rel(Tf[rows.to(DEV)], ref.Tf) < 1e-4, rel(Tf[rows.to(DEV)], ref.Tf)
    k = tf_noise_kappa(Tf[rows.to(DEV)], ref.Tf.numpy(), ref.f.numpy(), kcfg)
    k_ref = oracle32_kappa(x[rows.to(DEV)].double().cpu(), p, prob_o, ref, kcfg)
    assert k <= 2.0 * k_ref, (k, k_ref)  # (medians over 8 rows x L elements only: a looser factor than the fixtures')


def test_backward_headline_size_sampled_heads():
    """configs[1] at full size: the backward kernels with a given d loss / d f; the float64 oracle is evaluated for
    two of the 16 heads only (a head's gradients depend on its own parameters and its own column of df)."""
    L, D, m, hidden, B = 16, 2, 1024, (128, 128, 128), 512
    p = O.init_params(L, D, m, hidden, 0.1, seed=0)
    prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
    gen = torch.Generator().manual_seed(9)
    x = (16.0 * torch.randn(B, D, generator=gen)).double()
    df = torch.randn(B, L, generator=gen, dtype=torch.float64) / B
    v, M = O.joint_nesting_masks(L, 1)
    r = run_hip(p, prob_o, x, v, M, H.PATH_AUTO, df_override=df)
    assert r["path"] == "fused_mfma"
    nl = len(p.ws)
    for l in (0, 15):
        ph = O.Params([w[l:l + 1] for w in p.ws], [b[l:l + 1] for b in p.bs], p.fourier_B, None).to(torch.float64)
        c = O.operator_forward(x, ph, prob_o)
        gref = O.operator_backward(c, ph, prob_o, df[:, l:l + 1])
        for i in range(nl):
            assert rel(r["grads"][i][l], gref[i][0]) < 3e-5, (l, i, rel(r["grads"][i][l], gref[i][0]))
            assert rel(r["grads"][nl + i][l], gref[nl + i][0]) < 3e-5, (l, i)


def test_configs2_at_its_global_batch_on_one_gpu():
    """BASELINE.json configs[2] at its OWN batch on one GPU - 2D harmonic oscillator, L = 32, B = 4096, sequential
    nesting, exponential mask, m = 256 (scripts/exps/pde/oscillator.sh:12-53) - the N = 1 end of the 1 -> 8 curve and a
    different code path from the 512-row per-GPU shape: 16 rounds of chain workgroups, partial-moment backward (batches
    beyond 1024 rows), split weight-gradient tiles.
      * forward: bit reproducibility, row equivariance, homogeneity, 8 sampled rows x all heads against float64;
      * loss: moments, loss and d loss / d f from (f, Tf) against the float64 formulas (methods/nestedlora.py:70-111);
      * backward (nsvd_operator_backward_evd on the partial moments): every gradient of two sampled heads, all 4096
        rows, against the float64 oracle's backward with the float64 d loss / d f of the same (f, Tf)."""
    L, D, m, hidden, B = 32, 2, 256, (128, 128, 128), 4096
    p = O.init_params(L, D, m, hidden, 1.0, exp_mask_init=10.0, seed=0)
    prob_o = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=4.0)
    kcfg = dict(operator_scale=1.0, laplacian_eps=0.01)
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    prob = hip_problem(prob_o)
    x = (prob_o.sigma * torch.randn(B, D, generator=torch.Generator().manual_seed(5))).to(DEV)
    ws = H.new_workspace(shape, B, DEV)
    assert H.path_name(shape, B, H.PATH_AUTO, prob) == "fused_mfma"
    f, Tf = H.operator_forward(shape, params, prob, x, ws)
    f2, Tf2 = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, B, DEV))
    assert torch.equal(f, f2) and torch.equal(Tf, Tf2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(6)).to(DEV)
    fp, Tfp = H.operator_forward(shape, params, prob, x[perm].contiguous(), H.new_workspace(shape, B, DEV))
    assert torch.equal(fp, f[perm]) and torch.equal(Tfp, Tf[perm])
    c = 2.0
    fc, Tfc = H.operator_forward(shape, H.pack_params(shape, list(ws_t[:-1]) + [ws_t[-1] * c],
                                                      list(bs_t[:-1]) + [bs_t[-1] * c], fB, sc), prob, x,
                                 H.new_workspace(shape, B, DEV))
    assert torch.equal(fc, c * f) and torch.equal(Tfc, c * Tf)
    rows = torch.tensor([0, 31, 32, 1023, 1024, 2047, 2048, 4095])
    p64 = p.to(torch.float64)
    ref = O.operator_forward(x[rows.to(DEV)].double().cpu(), p64, prob_o)
    assert rel(f[rows.to(DEV)], ref.f) < 2e-5
    # even / odd stencil form: Tf against the FLOAT64 stencil at north_star's tolerance (measured at configs[1]:
    # 1.5e-6 native, 6.9e-6 bf16x3; the float32 oracle - the reference's arithmetic - is at 4e-2)
    assert rel(Tf[rows.to(DEV)], ref.Tf) < 1e-4, rel(Tf[rows.to(DEV)], ref.Tf)
    k = tf_noise_kappa(Tf[rows.to(DEV)], ref.Tf.numpy(), ref.f.numpy(), kcfg)
    k_ref = oracle32_kappa(x[rows.to(DEV)].double().cpu(), p, prob_o, ref, kcfg)
    assert k <= 2.0 * k_ref, (k, k_ref)
    # loss + backward through the C call the trainer makes for batches beyond 1024 rows
    v, M = O.sequential_nesting_masks(L)
    scratch = H.evd_scratch(B, L, DEV)
    H.evd_partial(f, Tf, H.MASK_SEQUENTIAL, None, scratch)
    gw = [torch.full_like(w, float("nan")) for w in ws_t]
    gb = [torch.full_like(b, float("nan")) for b in bs_t]
    gs = torch.full_like(sc, float("nan"))
    grads = H.pack_params(shape, gw, gb, None, gs)
    mom = torch.empty(2 * L * L + 1, device=DEV)
    loss = torch.empty(3, device=DEV)
    H.operator_backward_evd(shape, params, prob, x, f, Tf, H.MASK_SEQUENTIAL, None, None, mom, False, scratch, loss,
                            grads, ws)
    torch.cuda.synchronize()
    f64, Tf64 = f.double().cpu(), Tf.double().cpu()
    l64, lam1, lam2 = O.evd_loss_forward(f64, Tf64, v.double(), M.double())[:3]
    assert abs(float(loss[0]) - float(l64)) < 2e-5 * max(abs(float(loss[1])), abs(float(loss[2])))
    assert rel(mom[:L * L], lam1.reshape(-1)) < 2e-6 and rel(mom[L * L:2 * L * L], lam2.reshape(-1)) < 2e-6
    df64 = O.evd_loss_backward(f64, Tf64, v.double(), M.double(), lam1, lam2)
    xc = x.double().cpu()
    nl = len(p.ws)
    for l in (0, 31):
        ph = O.Params([w[l:l + 1] for w in p.ws], [b[l:l + 1] for b in p.bs], p.fourier_B,
                      p.scales[l:l + 1]).to(torch.float64)
        ch = O.operator_forward(xc, ph, prob_o)
        assert rel(f[:, l], ch.f[:, 0]) < 2e-5  # the head's whole column, all 4096 rows
        gref = O.operator_backward(ch, ph, prob_o, df64[:, l:l + 1])
        for i in range(nl):
            assert rel(gw[i][l], gref[i][0]) < 3e-5, (l, i, rel(gw[i][l], gref[i][0]))
            assert rel(gb[i][l], gref[nl + i][0]) < 3e-5, (l, i, rel(gb[i][l], gref[nl + i][0]))
        assert rel(gs[l], gref[2 * nl][0]) < 3e-5, (l, rel(gs[l], gref[2 * nl][0]))


@pytest.mark.parametrize("case", ["hydrogen_L36", "oscillator_L55"])
def test_reference_scripts_head_counts_at_full_size(case):
    """The head counts the reference's own PDE scripts run (--neigs 36: scripts/exps/pde/hydrogen.sh:28, 128,128,128 /
    m = 1024 / sigma 16 / operator_scale 100; --neigs 55: oscillator.sh:27, exponential mask, m = 256, sigma 4, shift
    16) at B = 512 on the fused MFMA path - 576 and 880 workgroups on 256 CUs, odd head counts, head windows that do
    not divide - against the float64 oracle:
      * forward: bit reproducibility; 8 sampled rows x ALL heads of f and Tf;
      * loss: moments, loss from (f, Tf) against the float64 formulas (methods/nestedlora.py:70-111), joint (hydrogen)
        and sequential (oscillator) nesting;
      * backward (the fused step's loss-gradient-inside-the-backward call): every gradient of the first, a middle and
        the LAST head, all 512 rows, against the float64 oracle's backward."""
    if case == "hydrogen_L36":
        L, D, m, hidden, B = 36, 2, 1024, (128, 128, 128), 512
        p = O.init_params(L, D, m, hidden, 0.1, seed=0)
        prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
        v, M = O.joint_nesting_masks(L, 1)
        kind = H.MASK_JOINT
    else:
        L, D, m, hidden, B = 55, 2, 256, (128, 128, 128), 512
        p = O.init_params(L, D, m, hidden, 1.0, exp_mask_init=10.0, seed=0)
        prob_o = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=4.0)
        v, M = O.sequential_nesting_masks(L)
        kind = H.MASK_SEQUENTIAL
    shape = shape_of(p)
    ws_t, bs_t, fB, sc = to_dev(p)
    params = H.pack_params(shape, ws_t, bs_t, fB, sc)
    prob = hip_problem(prob_o)
    x = (prob_o.sigma * torch.randn(B, D, generator=torch.Generator().manual_seed(5))).to(DEV)
    ws = H.new_workspace(shape, B, DEV)
    assert H.path_name(shape, B, H.PATH_AUTO, prob) == "fused_mfma"
    f, Tf = H.operator_forward(shape, params, prob, x, ws)
    f2, Tf2 = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, B, DEV))
    assert torch.equal(f, f2) and torch.equal(Tf, Tf2)
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 300, 511])
    p64 = p.to(torch.float64)
    ref = O.operator_forward(x[rows.to(DEV)].double().cpu(), p64, prob_o)
    assert rel(f[rows.to(DEV)], ref.f) < 2e-5
    assert rel(Tf[rows.to(DEV)], ref.Tf) < 1e-4, rel(Tf[rows.to(DEV)], ref.Tf)
    # loss + backward: the direct form (moments from f inside the backward kernels, B <= 1024) the trainer takes
    gw = [torch.full_like(w, float("nan")) for w in ws_t]
    gb = [torch.full_like(b, float("nan")) for b in bs_t]
    gs = torch.full_like(sc, float("nan")) if sc is not None else None
    grads = H.pack_params(shape, gw, gb, None, gs)
    loss = torch.zeros(3, device=DEV)
    H.operator_backward_evd(shape, params, prob, x, f, Tf, kind, None, None, None, False, None, loss, grads, ws)
    mom = H.evd_moments(f, Tf, kind, None)
    loss2, _ = H.evd_loss_grad(f, Tf, kind, None, None, mom, want_grad=False)
    torch.cuda.synchronize()
    f64, Tf64 = f.double().cpu(), Tf.double().cpu()
    l64, lam1, lam2 = O.evd_loss_forward(f64, Tf64, v.double(), M.double())[:3]
    assert abs(float(loss2[0]) - float(l64)) < 2e-5 * max(abs(float(loss2[1])), abs(float(loss2[2])))
    assert abs(float(loss[0]) - float(l64)) < 2e-5 * max(abs(float(loss2[1])), abs(float(loss2[2])))
    assert rel(mom[:L * L], lam1.reshape(-1)) < 2e-6 and rel(mom[L * L:2 * L * L], lam2.reshape(-1)) < 2e-6
    df64 = O.evd_loss_backward(f64, Tf64, v.double(), M.double(), lam1, lam2)
    xc = x.double().cpu()
    nl = len(p.ws)
    for l in (0, L // 2, L - 1):
        ph = O.Params([w[l:l + 1] for w in p.ws], [b[l:l + 1] for b in p.bs], p.fourier_B,
                      None if p.scales is None else p.scales[l:l + 1]).to(torch.float64)
        ch = O.operator_forward(xc, ph, prob_o)
        assert rel(f[:, l], ch.f[:, 0]) < 2e-5 and rel(Tf[:, l], ch.Tf[:, 0]) < 1e-4
        gref = O.operator_backward(ch, ph, prob_o, df64[:, l:l + 1])
        for i in range(nl):
            assert rel(gw[i][l], gref[i][0]) < 3e-5, (l, i, rel(gw[i][l], gref[i][0]))
            assert rel(gb[i][l], gref[nl + i][0]) < 3e-5, (l, i, rel(gb[i][l], gref[nl + i][0]))
        if gs is not None:
            assert rel(gs[l], gref[2 * nl][0]) < 3e-5, (l, rel(gs[l], gref[2 * nl][0]))
    for g in gw + gb:
        assert bool(torch.isfinite(g).all())


@pytest.mark.parametrize("world,rank", [(8, 7), (4, 1)])
def test_head_sharded_backward_at_the_multi_gpu_rank_shape(world, rank):
    """What one rank of an N-GPU head-sharded run of configs[1] executes (L / N heads of 16 on the global batch of
    512 N rows, partial-moment backward with a head offset), at FULL size on one GPU: f, Tf of the rank's heads are the
    unsharded run's columns bit for bit, and its gradients (EVD loss of the global (B, 16) arrays, moments included)
    are the unsharded backward's head slices - the same kernels on the same columns; the freedom is float32 summation
    order (the rank's weight gradients take their split-K form over the 512 N rows): 5e-6."""
    L, D, m, hidden = 16, 2, 1024, (128, 128, 128)
    B, Ll = 512 * world, L // world
    l0 = rank * Ll
    p = O.init_params(L, D, m, hidden, 0.1, seed=0)
    prob = hip_problem(O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0))
    x = (16.0 * torch.randn(B, D, generator=torch.Generator().manual_seed(world))).to(DEV)

    def run(pp, f_g=None, Tf_g=None, l_offset=0):
        shape = shape_of(pp)
        ws_t, bs_t, fB, sc = to_dev(pp)
        params = H.pack_params(shape, ws_t, bs_t, fB, sc)
        gw, gb = [torch.zeros_like(w) for w in ws_t], [torch.zeros_like(b) for b in bs_t]
        grads = H.pack_params(shape, gw, gb, None, None)
        ws = H.new_workspace(shape, B, DEV)
        f, Tf = H.operator_forward(shape, params, prob, x, ws, True, H.PATH_FUSED)
        if f_g is None:
            f_g, Tf_g = f, Tf
        scratch = H.evd_scratch(B, L, DEV)
        moments = torch.empty(2 * L * L + 1, device=DEV)
        loss = torch.zeros(3, device=DEV)
        H.evd_partial(f_g, Tf_g, H.MASK_JOINT, None, scratch)
        H.operator_backward_evd(shape, params, prob, x, f_g, Tf_g, H.MASK_JOINT, None, None, moments, False, scratch, loss,
                                grads, ws, 1.0, H.PATH_FUSED, l_offset=l_offset)
        torch.cuda.synchronize()
        return f, Tf, gw + gb, loss.clone()

    f, Tf, g_full, loss_full = run(p)
    sl = slice(l0, l0 + Ll)
    p_loc = O.Params([w[sl] for w in p.ws], [b[sl] for b in p.bs], p.fourier_B, None)
    f_l, Tf_l, g_loc, loss_loc = run(p_loc, f, Tf, l0)
    assert torch.equal(f_l, f[:, sl]) and torch.equal(Tf_l, Tf[:, sl])
    assert rel(loss_loc, loss_full) < 1e-6
    for i, (a, b) in enumerate(zip(g_loc, g_full)):
        assert torch.isfinite(a).all() and rel(a, b[sl]) < 5e-6, (i, rel(a, b[sl]))


@pytest.mark.parametrize("W,B,L,kind", [(8, 512, 36, "joint"), (8, 4096, 55, "seq"), (3, 96, 5, "custom"), (4, 1280, 7, "seq"),
                                        (8, 64, 8, "joint"), (2, 64, 8, "seq")])
def test_gather_head_blocks_with_uneven_head_counts(W, B, L, kind):
    """nsvd_evd_gather_head_blocks: rank w's block of 2 B ceil(L / W) floats begins with its packed f (B, n_w) | Tf (B,
    n_w), n_w = L // W + (w < L % W) (the scripts' --neigs 36 / 55 on 8 ranks: 5 / 4 and 7 / 6 heads); the tail of a
    short block is poisoned with NaN and must never be read. f, Tf bit for bit the concatenation, the partial
    moments bit for bit those of nsvd_evd_partial on them."""
    from neural_svd_amd.parallel import head_block, head_range
    g = torch.Generator().manual_seed(W * B + L)
    Lb = head_block(L, W)
    gath = torch.full((W, 2 * B * Lb), float("nan"))
    fs, Ts = [], []
    for w in range(W):
        _, n = head_range(L, w, W)
        fw, tw = torch.randn(B, n, generator=g), torch.randn(B, n, generator=g)
        gath[w, :B * n], gath[w, B * n:2 * B * n] = fw.reshape(-1), tw.reshape(-1)
        fs.append(fw), Ts.append(tw)
    gath = gath.to(DEV)
    want_f, want_T = torch.cat(fs, 1).contiguous().to(DEV), torch.cat(Ts, 1).contiguous().to(DEV)
    v, M = (O.sequential_nesting_masks(L) if kind != "joint" else O.joint_nesting_masks(L, 1))
    mk = {"seq": H.MASK_SEQUENTIAL, "joint": H.MASK_JOINT, "custom": H.MASK_CUSTOM}[kind]
    vd = v.float().to(DEV) if kind == "custom" else None
    f, Tf = torch.empty(B, L, device=DEV), torch.empty(B, L, device=DEV)
    s1, s2 = H.evd_scratch(B, L, DEV), H.evd_scratch(B, L, DEV)
    s1.zero_(); s2.zero_()
    H.evd_gather_head_blocks(gath, L, f, Tf, mk, vd, s1)
    H.evd_partial(want_f, want_T, mk, vd, s2)
    torch.cuda.synchronize()
    assert torch.equal(f, want_f) and torch.equal(Tf, want_T)
    assert torch.equal(s1, s2)
    f2, Tf2 = torch.zeros(B, L, device=DEV), torch.zeros(B, L, device=DEV)
    H.evd_gather_head_blocks(gath, L, f2, Tf2, mk, vd, None)  # copy only
    assert torch.equal(f2, want_f) and torch.equal(Tf2, want_T)
    if L % W == 0:  # the even layout IS the (W, 2, B, L / W) array of nsvd_evd_gather_heads
        f3, Tf3 = torch.zeros(B, L, device=DEV), torch.zeros(B, L, device=DEV)
        H.evd_gather_heads(gath.view(W, 2, B, L // W), f3, Tf3, mk, vd, None)
        assert torch.equal(f3, want_f) and torch.equal(Tf3, want_T)


@pytest.mark.parametrize("W,B,Ll,kind", [(4, 96, 3, "seq"), (8, 4096, 2, "joint"), (2, 1280, 5, "custom"), (1, 64, 4, "seq")])
def test_gather_heads_is_the_permuting_copy_plus_evd_partial(W, B, Ll, kind):
    """nsvd_evd_gather_heads: the all-gathered (world, 2, B, L_local) blocks -> f, Tf (B, L) bit for bit what a permuting
    copy gives, and its partial moments bit for bit those of nsvd_evd_partial on that f, Tf."""
    L = W * Ll
    g = torch.Generator().manual_seed(W * B)
    gath = torch.randn(W, 2, B, Ll, generator=g).to(DEV)
    want = gath.permute(1, 2, 0, 3).reshape(2, B, L).contiguous()
    v, M = (O.sequential_nesting_masks(L) if kind != "joint" else O.joint_nesting_masks(L, 1))
    mk = {"seq": H.MASK_SEQUENTIAL, "joint": H.MASK_JOINT, "custom": H.MASK_CUSTOM}[kind]
    vd = v.float().to(DEV) if kind == "custom" else None
    f, Tf = torch.empty(B, L, device=DEV), torch.empty(B, L, device=DEV)
    s1, s2 = H.evd_scratch(B, L, DEV), H.evd_scratch(B, L, DEV)
    s1.zero_(); s2.zero_()
    H.evd_gather_heads(gath, f, Tf, mk, vd, s1)
    H.evd_partial(want[0], want[1], mk, vd, s2)
    torch.cuda.synchronize()
    assert torch.equal(f, want[0]) and torch.equal(Tf, want[1])
    assert torch.equal(s1, s2)
    f2, Tf2 = torch.zeros(B, L, device=DEV), torch.zeros(B, L, device=DEV)
    H.evd_gather_heads(gath, f2, Tf2, mk, vd, None)  # copy only
    assert torch.equal(f2, want[0]) and torch.equal(Tf2, want[1])
