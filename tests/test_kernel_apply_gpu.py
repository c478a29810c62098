"""GPU tests of the dense kernel-operator row (nsvd_kernel_apply, neural_svd_amd/kernel_ops.py). The reference ships
no kernel operator (parity unpinned, see oracle/nsvd_oracle.py:kernel_apply): the checks are against the float64
restatement of the build's own definition, Kf = K[rows][:, cols] @ f / len(cols). Tolerance 3e-5 relative (L2):
float32 MFMA contraction over up to 10 000 terms."""
import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = torch.as_tensor(a).double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("N,B1,B2,L", [(300, 64, 64, 8), (1000, 193, 70, 33), (130, 5, 400, 1), (2048, 512, 512, 64),
                                       (130, 401, 90, 5), (300, 1024, 1024, 8)])
def test_kernel_apply_matches_float64(N, B1, B2, L):
    """the last two cases have more output rows than points: every row of K is multiplied once and the output rows
    are gathered from the product (head-sharded runs: the batch grows with the world size, the point set does not)"""
    from neural_svd_amd import hip_ops as H
    g = torch.Generator().manual_seed(N + B1)
    A = torch.randn(N, 32, generator=g, dtype=torch.float64)
    K = (A @ A.T / 32 + 1e-3 * torch.eye(N, dtype=torch.float64)).float().double()
    rows = torch.randint(N, (B1,), generator=g)
    cols = torch.randint(N, (B2,), generator=g)  # with replacement: duplicates are summed
    f = torch.randn(B2, L, generator=g, dtype=torch.float64).float().double()
    want = O.kernel_apply(K, rows, cols, f)
    ld = (N + 63) // 64 * 64
    Kd = torch.full((N, ld), 7.0, device=DEV)  # finite garbage in the padding must not matter
    Kd[:, :N] = K.float().to(DEV)
    got = H.kernel_apply(Kd, N, rows.to(DEV), cols.to(DEV), f.float().to(DEV), 1.0 / B2)
    assert rel(got, want) < 3e-5
    # out-of-range indices: zero rows / no contribution
    rows2 = rows.clone()
    rows2[0] = -1
    cols2 = cols.clone()
    cols2[0] = N + 5
    got2 = H.kernel_apply(Kd, N, rows2.to(DEV), cols2.to(DEV), f.float().to(DEV), 1.0 / B2)
    f2 = f.clone()
    f2[0] = 0
    want2 = O.kernel_apply(K, rows, cols, f2)
    want2[0] = 0
    assert rel(got2, want2) < 3e-5


def test_kernel_apply_full_size_sampled_rows():
    """cfg4 size (N = 10 000, B = 8192, L = 64): 96 randomly chosen output rows against float64 on the CPU."""
    from neural_svd_amd.kernel_ops import synthetic_psd_kernel
    from neural_svd_amd import hip_ops as H
    op = synthetic_psd_kernel(10000, 256, 16, 0, DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    idx = op.sample_indices(8192, g)
    f = torch.randn(8192, 64, device=DEV, generator=g)
    got = H.kernel_apply(op.K, op.N, idx, idx, f, 1.0 / 8192)
    torch.cuda.synchronize()
    pick = torch.randperm(8192, generator=torch.Generator().manual_seed(1))[:96]
    K64 = op.K[:, :op.N].double().cpu()
    ic = idx.cpu()
    want = O.kernel_apply(K64, ic[pick], ic, f.double().cpu())
    assert rel(got[pick.to(DEV)], want) < 3e-5
    # K symmetric PSD => the batch quadratic form is non-negative
    assert float((f * got).sum()) > 0


def test_compute_loss_kernel_with_dense_operator():
    """NestedLoRA.compute_loss_kernel (both split_batch modes) on a DenseKernelOperator with a 16-dimensional input
    model (plain model evaluation takes any input dimension): loss and parameter gradients vs the float64 oracle."""
    from types import SimpleNamespace as NS
    from neural_svd_amd.kernel_ops import synthetic_psd_kernel
    from neural_svd_amd.models import get_wavefunctions
    from neural_svd_amd.nested_lowrank import get_evd_method
    N, D, L, B = 500, 16, 6, 96
    op = synthetic_psd_kernel(N, 32, D, 5, DEV)
    args = NS(ndim=D, n_particles=1, use_fourier_feature=True, fourier_mapping_size=12, fourier_scale=0.05,
              fourier_deterministic=False, fourier_append_raw=False, mlp_hidden_dims="24,16", neigs=L, parallel=1,
              nonlinearity="softplus", apply_exp_mask=0, exp_mask_init_scale=1.0, hard_mul_const=1.0, apply_boundary=0,
              sort=0, loss=NS(neuralsvd=NS(step=1, sequential=False)))
    torch.manual_seed(4)
    net = get_wavefunctions(args).to(DEV)
    method = get_evd_method(args, "neuralsvd", op.index_model(net)).to(DEV)
    sd = {k: v.detach().double().cpu() for k, v in method.state_dict().items()}
    nl = 3
    pre = "model.net.base."
    p64 = O.Params([sd[f"{pre}ws.{i}"] for i in range(nl)], [sd[f"{pre}bs.{i}"] for i in range(nl)],
                   sd[f"{pre}feature_map._B"], None)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=0.01, use_importance=False)
    idx = op.sample_indices(B, torch.Generator(device=DEV).manual_seed(9))
    z = op.points.double().cpu()
    K64 = op.K[:, :N].double().cpu()
    ic = idx.cpu()
    v, M = method.vector_mask.double(), method.matrix_mask.double()
    names = [f"{pre}ws.{i}" for i in range(nl)] + [f"{pre}bs.{i}" for i in range(nl)]
    got = dict(method.named_parameters())

    # split_batch = False
    loss, aux = method.compute_loss_kernel(op.get_approx_kernel_op, idx, None, split_batch=False)
    loss.backward()
    c = O.operator_forward(z[ic], p64, prob)
    Kf = O.kernel_apply(K64, ic, ic, c.f)
    l64, lam1, lam2, _, _ = O.evd_loss_forward(c.f, Kf, v, M)
    gref = O.operator_backward(c, p64, prob, O.evd_loss_backward(c.f, Kf, v, M, lam1, lam2))
    assert abs(float(loss.detach()) - float(l64)) < 1e-4 * abs(float(l64))
    assert rel(aux["Tf"], Kf) < 1e-4
    for n, gr in zip(names, gref):
        assert rel(got[n].grad, gr) < 1e-4, n

    # split_batch = True
    method.zero_grad()
    loss, aux = method.compute_loss_kernel(op.get_approx_kernel_op, idx, None, split_batch=True)
    loss.backward()
    B1 = (B + 1) // 2
    c1 = O.operator_forward(z[ic[:B1]], p64, prob)
    c2 = O.operator_forward(z[ic[B1:]], p64, prob)
    Kf1 = O.kernel_apply(K64, ic[:B1], ic[B1:], c2.f)
    lam1, lam2 = c1.f.T @ c1.f / B1, c2.f.T @ c2.f / (B - B1)
    l64 = -2.0 * ((c1.f * Kf1) @ v).mean() + (M * lam1 * lam2).sum()
    df1 = -(4.0 / B1) * Kf1 * v + (2.0 / B1) * c1.f @ (M * lam2)
    df2 = (2.0 / (B - B1)) * c2.f @ (M * lam1)
    gref = [a + b for a, b in zip(O.operator_backward(c1, p64, prob, df1), O.operator_backward(c2, p64, prob, df2))]
    assert abs(float(loss.detach()) - float(l64)) < 1e-4 * abs(float(l64))
    for n, gr in zip(names, gref):
        assert rel(got[n].grad, gr) < 1e-4, n


@pytest.mark.parametrize("case", ["ka", "kb", "kc"])
@pytest.mark.parametrize("split", [False, True])
def test_compute_loss_kernel_matches_reference_golden(case, split):
    """NestedLoRA.compute_loss_kernel on the HIP path (differentiable model evaluation + EVD loss kernels) against the
    REFERENCE's own compute_loss_kernel (methods/nestedlora.py:230-252; tests/golden/kernel_loss.npz: its
    WaveFunctions model, a toy Gaussian-kernel `get_approx_kernel_op`, both split_batch modes): loss, f, Kf and
    every parameter gradient vs the float64 reference values, held to what the float32 reference itself achieves.
    kb takes the fused MFMA model kernels (128-wide hidden layers, exponential mask), ka / kc the generic ones."""
    from tests import _golden as G
    from tests.test_dropin_gpu import make_args
    from neural_svd_amd.models import get_wavefunctions
    from neural_svd_amd.nested_lowrank import get_evd_method
    z = G.load("kernel_loss")
    cfg = G.cfg_of(z, case)
    args = make_args(cfg)
    torch.manual_seed(cfg["seed"])  # the reference's draw order: same initial weights bit for bit
    method = get_evd_method(args, "neuralsvd", get_wavefunctions(args)).to(DEV)
    if f"{case}_param0_model.base.ws.0" in z.files:
        for n, t in method.named_parameters():
            assert torch.equal(t.detach().cpu(), torch.tensor(z[f"{case}_param0_{n}"])), n
    assert torch.equal(method.vector_mask.cpu(), torch.tensor(z[f"{case}_v"]))
    assert torch.equal(method.matrix_mask.cpu(), torch.tensor(z[f"{case}_M"]))
    ell = float(z[f"{case}_ell"])
    x = torch.tensor(z[f"{case}_x"]).to(DEV)

    def get_approx_kernel_op(x_ref):  # the user's operator: arbitrary Python around method(x)
        def op(m, xe, importance=None):
            f = m(xe)
            with torch.no_grad():
                Kmat = torch.exp(-torch.cdist(xe.double(), x_ref.double()) ** 2 / (2.0 * ell ** 2))
                Kf = (Kmat @ m(x_ref).double() / x_ref.shape[0]).float()
            return Kf, f
        return op

    loss, aux = method.compute_loss_kernel(get_approx_kernel_op, x, None, split_batch=split)
    loss.backward()
    q64, q32 = f"{case}_f64_split{int(split)}_", f"{case}_f32_split{int(split)}_"

    def tol(key, floor):  # a few times the float32 reference's own distance from its float64 self
        return max(4.0 * G.rel(z[q32 + key], z[q64 + key]), floor)
    assert abs(float(loss) - float(z[q64 + "loss"])) < max(4 * abs(float(z[q32 + "loss"]) - float(z[q64 + "loss"])),
                                                           2e-6 * abs(float(z[q64 + "loss"])))
    assert rel(aux["f"].detach(), torch.tensor(z[q64 + "f"])) < tol("f", 2e-6)
    assert rel(aux["Tf"].detach(), torch.tensor(z[q64 + "Kf"])) < tol("Kf", 2e-6)
    for n, t in method.named_parameters():
        if t.grad is None:
            continue
        if q64 + f"grad_{n}" in z.files:
            assert rel(t.grad.reshape(z[q64 + f"grad_{n}"].shape), torch.tensor(z[q64 + f"grad_{n}"])) < \
                tol(f"grad_{n}", 5e-6), n
        else:
            want = float(z[q64 + f"gradnorm_{n}"])
            assert abs(float(t.grad.double().norm()) - want) < 1e-5 * want, n
            assert rel(t.grad.reshape(-1)[::61], torch.tensor(z[q64 + f"gradsample_{n}"])) < \
                tol(f"gradsample_{n}", 5e-6), n


def test_fused_kernel_trainer_matches_the_module_loop():
    """FusedKernelTrainer (model evaluation, kernel_apply, loss, backward and RMSprop inside the C calls) against the
    reference-style loop on the same index batches: NestedLoRA.compute_loss_kernel on the same DenseKernelOperator,
    loss.backward(), torch.optim.RMSprop - three steps, parameters and losses."""
    from types import SimpleNamespace as NS
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.kernel_ops import FusedKernelTrainer, synthetic_psd_kernel
    from neural_svd_amd.models import get_wavefunctions
    from neural_svd_amd.nested_lowrank import get_evd_method
    N, D, L, B, m = 1000, 16, 8, 256, 64
    op = synthetic_psd_kernel(N, 64, D, 3, DEV)
    fk = FusedKernelTrainer(op, L=L, m=m, hidden=(128, 128), batch_size=B, sequential=False, lr=1e-3, rmsprop_decay=0.99,
                            rmsprop_eps=1e-8, fourier_scale=0.05, seed=11)
    args = NS(ndim=D, n_particles=1, use_fourier_feature=True, fourier_mapping_size=m, fourier_scale=0.05,
              fourier_deterministic=False, fourier_append_raw=False, mlp_hidden_dims="128,128", neigs=L, parallel=1,
              nonlinearity="softplus", apply_exp_mask=0, exp_mask_init_scale=1.0, hard_mul_const=1.0, apply_boundary=0,
              sort=0, loss=NS(neuralsvd=NS(step=1, sequential=False)))
    net = get_wavefunctions(args).to(DEV)
    method = get_evd_method(args, "neuralsvd", op.index_model(net)).to(DEV)
    # same initial weights: the trainer's flat buffers -> the module
    sd = fk.P.state_dict()
    with torch.no_grad():
        for n, p in net.named_parameters():
            key = "model." + n
            p.copy_(sd[key].reshape(p.shape))
        net.base.feature_map._B.copy_(sd["model.base.feature_map._B"])
    opt = torch.optim.RMSprop(method.parameters(), lr=1e-3, alpha=0.99, eps=1e-8)
    g = torch.Generator(device=DEV).manual_seed(5)
    for t in range(3):
        idx = op.sample_indices(B, g)
        la = fk.step(idx).clone()
        opt.zero_grad()
        lb, _ = method.compute_loss_kernel(op.get_approx_kernel_op, idx, None, split_batch=False)
        lb.backward()
        opt.step()
        assert abs(float(la[0]) - float(lb)) < 2e-5 * max(1.0, abs(float(lb))), (t, float(la[0]), float(lb))
    sd = fk.P.state_dict()
    for n, p in net.named_parameters():
        got = sd["model." + n].reshape(p.shape)
        d = float((got - p).double().norm() / p.double().norm().clamp_min(1e-30))
        # biases start at zero: after three sign-like RMSprop steps they are ~3e-3 and float32 noise in a gradient of
        # the same magnitude on both sides shows at 1e-5..1e-4 of that
        assert d < (2e-4 if ".bs." in n else 2e-5), (n, d)


def test_configs3_full_size_fused_step():
    """BASELINE.json configs[3] at its OWN size - dense PSD kernel operator on N = 10 000 points, L = 64, B = 8192
    indices, joint nesting - the COMPOSITE step FusedKernelTrainer takes there (reduced moments from 128 chunk partials,
    split weight-gradient tiles, the optimiser in the weight-gradient epilogue), against the float64 oracle of
    NestedLoRA.compute_loss_kernel (reference methods/nestedlora.py:230-252; the operator itself has no reference
    implementation: K f in float64 from the same K):
      * f = model(z[idx]) (all 64 heads) and Kf = K[idx][:, idx] f / B against float64;
      * moments, loss against the float64 formulas on the float64 (f, Kf) - end to end;
      * every gradient of two sampled heads (first and last) against the float64 backward with the float64 d loss / d f;
      * the step itself: its loss is that loss, its parameters are RMSprop(those gradients) bit for bit."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.kernel_ops import FusedKernelTrainer, synthetic_psd_kernel
    from oracle import nsvd_oracle as O
    N, D, L, B, m = 10000, 16, 64, 8192, 64
    op = synthetic_psd_kernel(N, 256, D, 0, DEV)
    lr, alpha, eps = 1e-4, 0.99, 1e-8
    fk = FusedKernelTrainer(op, L=L, m=m, hidden=(128, 128), batch_size=B, sequential=False, lr=lr, rmsprop_decay=alpha,
                            rmsprop_eps=eps, fourier_scale=0.05, seed=0)
    assert fk.mask_kind == H.MASK_JOINT
    idx = op.sample_indices(B, torch.Generator(device=DEV).manual_seed(3))
    x = op.points.index_select(0, idx)
    # the trainer's own call sequence with the gradients written instead of consumed
    H.model_forward(fk.shape, fk._params, x, 1.0, fk.ws, save_for_backward=True, out=fk.f_loc)
    H.kernel_apply(op.K, op.N, idx, idx, fk.f_loc, 1.0 / B, ws=fk.ka_ws, out=fk.Kf_loc)
    H.evd_moments(fk.f, fk.Kf, fk.mask_kind, None, fk.moments, fk.scratch)
    fk.P.grad.fill_(float("nan"))
    loss = torch.zeros(3, device=DEV)
    H.model_backward_evd_step(fk.shape, fk._params, x, fk.f, fk.Kf, fk.mask_kind, None, None, fk.moments, True, None,
                              loss, fk.P.pack(fk.P.grad, False), None, fk.ws)
    torch.cuda.synchronize()
    f_hip, Kf_hip, mom = fk.f.clone(), fk.Kf.clone(), fk.moments.clone()
    grads = [g.clone() for g in fk.P.views(fk.P.grad)]
    assert all(bool(torch.isfinite(g).all()) for g in grads)
    # ---- float64 oracle
    sd = {k: v.double().cpu() for k, v in fk.P.state_dict().items()}
    nl = 3
    p64 = O.Params([sd[f"model.base.ws.{i}"] for i in range(nl)], [sd[f"model.base.bs.{i}"] for i in range(nl)],
                   sd["model.base.feature_map._B"])
    x64 = x.double().cpu()
    phi = O.fourier_features(x64, p64.fourier_B)
    f64 = torch.cat([O.mlp_forward(phi, O.Params([w[l0:l0 + 8] for w in p64.ws], [b[l0:l0 + 8] for b in p64.bs],
                                                 p64.fourier_B)) for l0 in range(0, L, 8)], dim=1)
    assert rel(f_hip, f64) < 2e-5
    ic = idx.cpu().numpy()
    Ksub = op.K.cpu().numpy()[ic][:, ic].astype(np.float64)
    Kf64 = torch.from_numpy(Ksub @ f64.numpy() / B)
    assert rel(Kf_hip, Kf64) < 2e-5
    v, M = O.joint_nesting_masks(L, 1)
    l64, lam1, lam2 = O.evd_loss_forward(f64, Kf64, v.double(), M.double())[:3]
    assert rel(mom[:L * L], lam1.reshape(-1)) < 2e-5 and rel(mom[L * L:2 * L * L], lam2.reshape(-1)) < 2e-5
    assert abs(float(loss[0]) - float(l64)) < 1e-4 * max(abs(float(loss[1])), abs(float(loss[2])))
    df64 = O.evd_loss_backward(f64, Kf64, v.double(), M.double(), lam1, lam2)
    plain = O.Problem(potential=O.POT_HARMONIC, eps=0.01, use_importance=False, hard_mul_const=1.0)
    one = torch.ones(B, 1, dtype=torch.float64)
    for l in (0, L - 1):
        ph = O.Params([w[l:l + 1] for w in p64.ws], [b[l:l + 1] for b in p64.bs], p64.fourier_B)
        out, zs = O.mlp_forward(phi, ph, keep=True)
        c = O.OperatorCache(x64, phi, zs, out, None, one, one, out, None)
        gref = O.operator_backward(c, ph, plain, df64[:, l:l + 1])
        for i in range(nl):
            assert rel(grads[i][l], gref[i][0]) < 5e-5, (l, i, rel(grads[i][l], gref[i][0]))
            assert rel(grads[nl + i][l], gref[nl + i][0]) < 5e-5, (l, i, rel(grads[nl + i][l], gref[nl + i][0]))
    # ---- the fused step on the same batch: same loss, RMSprop of exactly those gradients
    p0 = fk.P.flat.clone()
    want = p0.clone()
    H.rmsprop_ema_step(want, fk.P.grad.clone(), torch.zeros_like(p0), None, lr, alpha, eps, 0.0)
    got_loss = fk.step(idx).clone()
    torch.cuda.synchronize()
    assert torch.equal(got_loss, loss)
    for a, b in zip(fk.P.views(fk.P.flat), fk.P.views(want)):
        assert torch.equal(a, b)
    assert not torch.equal(fk.P.flat, p0)
