"""GPU tests of the host-side mirror of the reference API (get_problem / get_wavefunctions /
get_evd_method / compute_loss_operator / train_operator / compute_spectrum_evd) and of the fused
trainer, against the oracle and the reference's golden vectors."""
import argparse

import numpy as np
import pytest
import torch

from oracle import nsvd_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().numpy()
    b = np.asarray(torch.as_tensor(b).double().cpu().numpy())
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def make_args(cfg):
    from neural_svd_amd.nested_lowrank import get_evd_method  # noqa: F401
    a = argparse.Namespace(**{k: v for k, v in cfg.items() if k not in ("sequential", "step")})
    a.loss = argparse.Namespace(name="neuralsvd",
                                neuralsvd=argparse.Namespace(step=cfg["step"], sequential=cfg["sequential"]))
    a.adam_eps = 1e-7
    a.use_lr_scheduler = True
    a.ema_decay = 0.995
    a.print_freq = 10 ** 9
    a.eval_freq = 10 ** 9
    a.log_dir = None
    return a


def build(case, z):
    from neural_svd_amd.models import get_wavefunctions
    from neural_svd_amd.nested_lowrank import get_evd_method
    from neural_svd_amd.operators import get_dataloader, get_problem
    cfg = G.cfg_of(z, case)
    args = make_args(cfg)
    torch.manual_seed(cfg["seed"])
    operator, gt = get_problem(args, DEV)
    model = get_wavefunctions(args)
    loaders = get_dataloader(args, DEV)
    method = get_evd_method(args, "neuralsvd", model).to(DEV)
    return cfg, args, operator, gt, method, loaders


@pytest.mark.parametrize("case", ["hyd_small", "osc_small"])
def test_reference_style_pipeline(case):
    z = G.load("model_small")
    cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build(case, z)
    # same names / values as the reference objects
    names = [n for n, _ in method.named_parameters()]
    assert names == [str(n) for n in z[f"{case}_param_names"]]
    assert np.allclose(gt[:cfg["neigs"]], z[f"{case}_gt"][:cfg["neigs"]])
    assert torch.equal(method.vector_mask, torch.tensor(z[f"{case}_v"]))
    assert torch.equal(method.matrix_mask, torch.tensor(z[f"{case}_M"]))
    # the seeded construction draws the reference's initial weights bit for bit
    sd = method.state_dict()
    for n in names:
        assert torch.equal(sd[n].cpu(), torch.tensor(z[f"{case}_param0_{n}"])), n
    x = torch.tensor(z[f"{case}_x"][0]).to(DEV)
    method.train()
    loss, aux = method.compute_loss_operator(operator, x, importance=imp_train)
    assert loss.dim() == 0 and loss.requires_grad and aux["eigvals"] is None
    loss.backward()
    pre64, pre32 = f"{case}_f64_step0_", f"{case}_f32_step0_"
    assert rel(aux["f"], z[pre64 + "f"]) < 2e-5
    # end to end through the reference's API, against the reference's FLOAT64 loss and gradients at north_star's 1e-4 (the
    # stencil in even / odd form; its own float32 run is ref_err = 1e-2 .. 1e-1 away on these 24-32 rows)
    assert abs(float(loss.detach()) - float(z[pre64 + "loss"])) < 1e-4 * abs(float(z[pre64 + "loss"]))
    for n, p in method.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        g64 = z[pre64 + "grad_" + n]
        assert p.grad is not None and rel(p.grad, g64) < 1e-4, (n, rel(p.grad, g64), rel(z[pre32 + "grad_" + n], g64))
    # forward(x): eigenfunction values
    p64 = G.params_from_golden(z, case).to(torch.float64)
    base = O.mlp_forward(O.fourier_features(x.double().cpu(), p64.fourier_B), p64)
    mk = O.boundary_mask(x.double().cpu(), p64)
    assert rel(method(x), base if mk is None else base * mk) < 1e-5
    # evaluation with the reference's dataloader protocol
    from neural_svd_amd.spectrum import compute_spectrum_evd
    method.eval()
    out = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator, importance_train=imp_train,
                               importance_val=imp_val, normalize=True, device=DEV)
    assert out["eigvals"].shape == (cfg["neigs"],) and out["eigfuncs"].shape == (len(val_data), cfg["neigs"])
    assert np.allclose(np.diag(out["cov"]), 1.0, atol=1e-5)


class _PlainHarmonicOperator:
    """A foreign operator with the reference's contract operator(model, x, importance=None) -> (Tf, f)
    (examples/__init__.py:7-9): the harmonic-oscillator Hamiltonian with the point-wise central-difference stencil
    and the Gaussian re-weighting (diff_ops.py:9-52, schrodinger/__init__.py:16-22), in plain torch around model(x)."""

    def __init__(self, eps, scale, shift, k=1.0):
        self.eps, self.scale, self.shift, self.k = eps, scale, shift, k

    def __call__(self, model, x, importance=None):
        x = x.reshape(x.shape[0], -1)
        D = x.shape[1]
        g = (lambda z: importance(z).sqrt() * model(z)) if importance is not None else model
        gs = g(x)
        lap = -2 * D * gs
        for i in range(D):
            e = torch.zeros((1, D), device=x.device)
            e[0, i] = self.eps
            lap = lap + g(x + e) + g(x - e)
        lap = lap / self.eps ** 2
        sw = torch.clamp(importance(x).sqrt(), min=1e-5) if importance is not None else 1.0
        lap, fs = lap / sw, gs / sw
        V = (self.k * x.norm(dim=1, p=2) ** 2).reshape(-1, 1)
        Tf = -(-lap + V * fs)
        return self.scale * Tf + self.shift * fs, fs


def test_foreign_operator_falls_back():
    """methods/nestedlora.py:254-267 takes ANY callable operator(model, x, importance): a plain-torch restatement of the
    harmonic Hamiltonian goes through NestedLoRA.compute_loss_operator (model evaluations and the loss on the HIP
    kernels, the operator's algebra in torch) and gives the fused path's loss and gradients. eps = 0.1 keeps the
    point-wise float32 stencil's rounding noise (~ 1e-7 |g| (2D + 2) / eps^2) below the 1e-4 asserted."""
    z = G.load("model_small")
    res = {}
    for kind in ("fused", "foreign"):
        cfg, args, operator, gt, method, loaders = build("osc_small", z)
        operator.operator.laplacian_eps = 0.1
        imp = loaders[3]
        op = operator if kind == "fused" else _PlainHarmonicOperator(0.1, operator.scale, operator.shift)
        x = torch.tensor(z["osc_small_x"][0]).to(DEV)
        method.train()
        loss, aux = method.compute_loss_operator(op, x, importance=imp)
        loss.backward()
        res[kind] = (float(loss.detach()), aux["f"].detach().clone(), aux["Tf"].detach().clone(),
                     {n: p.grad.detach().clone() for n, p in method.named_parameters() if p.requires_grad})
    assert rel(res["foreign"][1], res["fused"][1]) < 1e-6
    assert rel(res["foreign"][2], res["fused"][2]) < 1e-4
    assert abs(res["foreign"][0] - res["fused"][0]) < 1e-4 * abs(res["fused"][0])
    for n, g in res["fused"][3].items():
        assert rel(res["foreign"][3][n], g) < 1e-4, n
    # evaluation takes the same callable (methods/spectrum.py:62)
    from neural_svd_amd.spectrum import compute_spectrum_evd
    cfg, args, operator, gt, method, (mb, val_data, batch_ftn_val, imp, imp_val) = build("osc_small", z)
    operator.operator.laplacian_eps = 0.1
    method.eval()
    a = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator, importance_train=imp,
                             importance_val=imp_val, device=DEV)
    b = compute_spectrum_evd(method, dataloader=batch_ftn_val(),
                             operator=_PlainHarmonicOperator(0.1, operator.scale, operator.shift),
                             importance_train=imp, importance_val=imp_val, device=DEV)
    # (an untrained model: some quotients are differences of near-equal terms - absolute tolerance at their scale)
    assert np.allclose(a["eigvals"], b["eigvals"], rtol=1e-4, atol=1e-3)
    with pytest.raises(Exception):
        method.compute_loss_operator(lambda m, xx, importance=None: (xx, xx), torch.randn(8, 2, device=DEV),
                                     importance=imp)  # (B, D) is not (B, L): refused with a shape message


@pytest.mark.parametrize("mode", ["laplacian", "uniform"])
def test_laplace_and_uniform_samplers(mode):
    """--sampling_mode laplacian / uniform (main_pde.py:101-118): the sampler, its importance density, and a step
    through the reference's stencil around HIP model evaluations (OperatorWrapper.apply_stencil), against the float64
    oracle model evaluated at the same stencil points with the same density."""
    from neural_svd_amd.models import get_wavefunctions
    from neural_svd_amd.nested_lowrank import get_evd_method
    from neural_svd_amd.operators import get_dataloader, get_problem
    z = G.load("model_small")
    cfg = dict(G.cfg_of(z, "osc_small"), sampling_mode=mode, laplacian_eps=0.1, batch_size=64)
    args = make_args(cfg)
    torch.manual_seed(3)
    operator, gt = get_problem(args, DEV)
    method = get_evd_method(args, "neuralsvd", get_wavefunctions(args)).to(DEV)
    make_batch, val_data, batch_ftn_val, imp, imp_val = get_dataloader(args, DEV)
    x = make_batch()
    assert x.shape == (64, 1, 2) and x.device.type == "cuda"
    s = cfg["sampling_scale"]
    x2 = x.reshape(64, 2).double().cpu()
    if mode == "uniform":
        assert float(x.abs().max()) <= s
        p64 = torch.full((64, 1), 1.0 / (2 * s) ** 2, dtype=torch.float64)
        dens = lambda q: torch.full((q.shape[0], 1), 1.0 / (2 * s) ** 2, dtype=torch.float64)  # noqa: E731
    else:
        dens = lambda q: torch.exp(-(q.abs() / s).sum(-1, keepdim=True)) / (2 * s) ** 2  # noqa: E731
        p64 = dens(x2)
    assert rel(imp(x), p64) < 1e-6
    method.train()
    loss, aux = method.compute_loss_operator(operator, x, importance=imp)
    loss.backward()
    # float64 restatement with the oracle's model
    names = [n for n, _ in method.named_parameters()]
    sd = {k: v.detach().double().cpu() for k, v in method.state_dict().items()}
    nl = len([n for n in names if ".ws." in n])
    p = O.Params(fourier_B=sd["model.base.feature_map._B"], ws=[sd[f"model.base.ws.{i}"] for i in range(nl)],
                 bs=[sd[f"model.base.bs.{i}"] for i in range(nl)], scales=sd["model.boundary_mask.scales"])

    def model64(q):
        return O.mlp_forward(O.fourier_features(q, p.fourier_B), p) * O.boundary_mask(q, p)
    g = lambda q: dens(q).sqrt() * model64(q)  # noqa: E731
    gs = g(x2)
    lap = -4 * gs
    for i in range(2):
        e = torch.zeros(1, 2, dtype=torch.float64)
        e[0, i] = 0.1
        lap = lap + g(x2 + e) + g(x2 - e)
    sw = torch.clamp(dens(x2).sqrt(), min=1e-5)
    lap, fs = lap / 0.01 / sw, gs / sw
    Tf = -(-lap + (x2.norm(dim=1) ** 2).reshape(-1, 1) * fs)
    Tf = cfg["operator_scale"] * Tf + cfg["operator_shift"] * fs
    assert rel(aux["f"], fs) < 1e-5
    assert rel(aux["Tf"], Tf) < 2e-4
    v, M = O.joint_nesting_masks(cfg["neigs"], 1)
    l64, *_ = O.evd_loss_forward(fs, Tf, v.double(), M.double())
    assert abs(float(loss.detach()) - float(l64)) < 2e-4 * abs(float(l64))
    for n, q in method.named_parameters():
        assert (q.grad is not None) == q.requires_grad
    # the evaluation path weights the rows itself for these densities
    from neural_svd_amd.spectrum import compute_spectrum_evd
    method.eval()
    out = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator, importance_train=imp,
                               importance_val=imp_val, device=DEV)
    assert out["eigvals"].shape == (cfg["neigs"],) and np.isfinite(out["eigvals"]).all()


def test_spectrum_first_mode_const_and_post_align():
    """compute_spectrum_evd(set_first_mode_const=True, post_align=True) (methods/spectrum.py:68-70,98-101,161-169): the
    constant-one column in front of the weighted phi / Tphi and the whitened re-diagonalisation, against numpy on the
    HIP path's own f, Tf."""
    from neural_svd_amd.spectrum import compute_spectrum_evd
    z = G.load("model_small")
    cfg, args, operator, gt, method, (mb, val_data, batch_ftn_val, imp, imp_val) = build("osc_small", z)
    method.eval()
    L = cfg["neigs"]
    out = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator, importance_train=imp,
                               importance_val=imp_val, set_first_mode_const=True, device=DEV)
    plain = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator, importance_train=imp,
                                 importance_val=imp_val, post_align=True, device=DEV)
    assert out["cov"].shape == (L + 1, L + 1) and np.allclose(out["cov"][1:, 1:], plain["cov"], rtol=1e-6, atol=1e-9)
    assert np.allclose(out["quad"][1:, 1:], plain["quad"], rtol=1e-6, atol=1e-9)
    # the padded column by hand
    with torch.no_grad():
        Tphi, phi = operator(method, val_data, importance=imp)
    w = (imp(val_data).sqrt() / imp_val(val_data).sqrt()).double().cpu().numpy()
    ph = np.nan_to_num(w * phi.double().cpu().numpy())
    tp = np.nan_to_num(w * Tphi.double().cpu().numpy())
    nz = ~np.all(np.isclose(val_data.cpu().numpy(), 0.0), axis=1)
    n = len(val_data)
    assert np.isclose(out["cov"][0, 0], 1.0) and np.isclose(out["quad"][0, 0], nz.sum() / n)
    assert np.allclose(out["cov"][0, 1:], ph.sum(0) / n, rtol=1e-5, atol=1e-8)
    assert np.allclose(out["quad"][0, 1:], (tp * nz[:, None]).sum(0) / n, rtol=1e-5, atol=1e-7)
    assert np.allclose(out["quad"][1:, 0], (ph * nz[:, None]).sum(0) / n, rtol=1e-5, atol=1e-8)
    # post_align: orthonormal aligned functions under the grid measure, eigenvalues = sqrt of the whitened quad's
    ef, ev, I = plain["eigfuncs_aligned"], plain["eigvals_aligned"], plain["cov_aligned"]
    assert ef.shape == plain["eigfuncs"].shape and ev.shape == (L,) and np.array_equal(I, np.eye(L))
    from scipy.linalg import eigh
    ec, vc = eigh(plain["cov"].astype(np.float64))
    wh = vc @ np.diag(ec ** -0.5) @ vc.T
    want = eigh(wh @ plain["quad"].astype(np.float64) @ wh)[0][::-1]  # descending, as the reference flips them
    # (an untrained model: negative whitened eigenvalues - the reference's np.sqrt gives NaN there, and so does this)
    assert np.array_equal(np.isnan(ev), want < 0)
    assert np.allclose(ev[want > 0] ** 2, want[want > 0], rtol=1e-4)


@pytest.mark.parametrize("iters,lr,tol", [(1, 1e-3, 2e-6), (12, 1e-5, 2e-3)])
@pytest.mark.parametrize("case", ["hyd_small", "osc_small"])
def test_train_operator_fused_loop_matches_plain_loop(case, iters, lr, tol):
    """train_operator's two loop bodies (FusedTrainer on the model's weights vs torch.optim + autograd) from the same
    seed and batches: parameters, EMA shadow, RMSprop state, learning rate and EMA counter agree - to rounding after
    one step; after 12 steps to what the sign-like early RMSprop updates make of that rounding (an element whose
    gradient is near zero can flip the sign of its +-lr/sqrt(1-alpha) update). The caller's random streams are not
    disturbed by the fused trainer's construction (the batches are the same)."""
    import neural_svd_amd.drop_in as DI
    z = G.load("model_small")
    res = {}
    for fused in (True, False):
        cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build(case, z)
        args.optimizer, args.lr, args.rmsprop_decay, args.momentum = "rmsprop", lr, 0.999, 0.0
        args.num_iters, args.print_freq, args.eval_freq, args.fused_loop = iters, 10 ** 9, iters, fused
        captured = {}
        orig_opt, orig_ema = DI.get_optimizer, DI.ExponentialMovingAverage

        def cap_opt(a, m, _o=orig_opt):
            captured["opt"] = _o(a, m)
            return captured["opt"]

        class CapEma(orig_ema):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                captured["ema"] = self

        DI.get_optimizer, DI.ExponentialMovingAverage = cap_opt, CapEma
        try:
            torch.manual_seed(123)
            eig, norms = DI.train_operator(args, method, operator, make_batch, val_data, batch_ftn_val, None, None, DEV,
                                           imp_train, imp_val)
        finally:
            DI.get_optimizer, DI.ExponentialMovingAverage = orig_opt, orig_ema
        opt, ema = captured["opt"], captured["ema"]
        train = [(n, p) for n, p in method.named_parameters() if p.requires_grad]
        res[fused] = dict(params={n: p.detach().clone() for n, p in train},
                          sq={n: opt.state[p]["square_avg"].clone() for n, p in train},
                          sh={n: s.clone() for (n, p), s in zip(train, ema.shadow_params)},
                          lr=opt.param_groups[0]["lr"], n=ema.num_updates, eig=eig[-1])
    a, b = res[True], res[False]
    assert a["n"] == b["n"] == iters and abs(a["lr"] - b["lr"]) < 1e-12 * max(1.0, abs(b["lr"]))
    upd = iters * lr / np.sqrt(1.0 - 0.999)  # the largest total movement of an element (biases start at zero)

    def close(x, y, t):
        x, y = x.double(), y.double()
        return float((x - y).norm()) <= t * (float(y.norm()) + upd * np.sqrt(y.numel()))

    for n in a["params"]:
        assert close(a["params"][n], b["params"][n], tol), n
        assert close(a["sh"][n], b["sh"][n], tol), n
        assert rel(a["sq"][n], b["sq"][n]) < 20 * tol, n
    # the eigenvalue metric differences Tf on the float32 stencil: 1e-9 of weight difference shows at 1e-3 there
    assert np.allclose(a["eig"], b["eig"], rtol=1e-2)


@pytest.mark.parametrize("case", ["hyd_small", "osc_small"])
def test_train_operator_plain_loop_replayed_from_a_graph(case):
    """the plain loop body (compute_loss_operator + loss.backward() + optimiser / scheduler / EMA) captured into a HIP
    graph after three eager iterations (drop_in.CapturedPlainStep) against the eager plain loop with torch.optim.RMSprop,
    CosineAnnealingLR and the foreach EMA on the same batches: after ONE step the parameters, square averages and EMA
    shadow agree to rounding (two implementations of the same float32 update); after 9 steps (3 eager + capture + 5
    replays, a moving cosine schedule of 9 steps) to what the sign-like early RMSprop updates make of that rounding; the
    torch objects' counters are in step (checkpoints written from them are the reference's)."""
    import neural_svd_amd.drop_in as DI
    z = G.load("model_small")
    for iters, tol in ((1, 2e-6), (9, 2e-3)):
        res = {}
        for graph in (True, False):
            cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build(case, z)
            args.optimizer, args.lr, args.rmsprop_decay, args.momentum = "rmsprop", 1e-5, 0.999, 0.0
            args.num_iters, args.print_freq, args.eval_freq = iters, 10 ** 9, iters
            args.fused_loop, args.graph_loop = False, graph
            box = {}
            orig_opt, orig_ema, orig_cap = DI.get_optimizer, DI.ExponentialMovingAverage, DI.CapturedPlainStep

            def cap_opt(a, m, _o=orig_opt):
                box["opt"] = _o(a, m)
                return box["opt"]

            class CapEma(orig_ema):
                def __init__(self, *a, **k):
                    super().__init__(*a, **k)
                    box["ema"] = self

            class CapStep(orig_cap):
                def __init__(self, *a, **k):
                    super().__init__(*a, **k)
                    box["step"] = self

            DI.get_optimizer, DI.ExponentialMovingAverage, DI.CapturedPlainStep = cap_opt, CapEma, CapStep
            try:
                torch.manual_seed(123)
                eig, _ = DI.train_operator(args, method, operator, make_batch, val_data, batch_ftn_val, None, None, DEV,
                                           imp_train, imp_val)
            finally:
                DI.get_optimizer, DI.ExponentialMovingAverage, DI.CapturedPlainStep = orig_opt, orig_ema, orig_cap
            opt, ema = box["opt"], box["ema"]
            assert ("step" in box) == graph
            if graph:
                assert box["step"].steps == iters and (box["step"].graph is not None) == (iters > 3)
                assert box["step"].state.read().step == iters
            train = [(n, p) for n, p in method.named_parameters() if p.requires_grad]
            res[graph] = dict(params={n: p.detach().clone() for n, p in train},
                              sq={n: opt.state[p]["square_avg"].clone() for n, p in train},
                              sh={n: s.clone() for (n, p), s in zip(train, ema.shadow_params)},
                              lr=opt.param_groups[0]["lr"], n=ema.num_updates, eig=eig[-1],
                              ostep=float(opt.state[train[0][1]]["step"]))
        a, b = res[True], res[False]
        assert a["n"] == b["n"] == iters == int(a["ostep"]) == int(b["ostep"])
        assert abs(a["lr"] - b["lr"]) < 1e-12 * max(1.0, abs(b["lr"]))
        upd = iters * 1e-5 / np.sqrt(1.0 - 0.999)

        def close(x, y, t):
            x, y = x.double(), y.double()
            return float((x - y).norm()) <= t * (float(y.norm()) + upd * np.sqrt(y.numel()))

        for n in a["params"]:
            assert close(a["params"][n], b["params"][n], tol), (iters, n)
            assert close(a["sh"][n], b["sh"][n], tol), (iters, n)
            assert rel(a["sq"][n], b["sq"][n]) < 20 * tol, (iters, n)
        assert np.allclose(a["eig"], b["eig"], rtol=1e-2)


def test_train_operator_smoke():
    """the reference-signature loop: a few iterations with eval + checkpoint dict, parameters move and stay finite."""
    from neural_svd_amd.drop_in import train_operator
    z = G.load("model_small")
    cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build("osc_small", z)
    args.num_iters, args.eval_freq, args.print_freq = 6, 3, 2
    before = {n: p.detach().clone() for n, p in method.named_parameters()}
    eigs, norms = train_operator(args, method, operator, make_batch, val_data, batch_ftn_val, None, None, DEV,
                                 imp_train, imp_val, gt)
    assert len(eigs) == 2 and eigs[0].shape == (cfg["neigs"],)
    for n, p in method.named_parameters():
        assert torch.isfinite(p).all()
        if p.requires_grad:
            assert not torch.equal(p, before[n]), n


@pytest.mark.parametrize("batch", [None, 2048])
def test_train_operator_writes_the_reference_csv_columns(batch, tmp_path):
    """main_pde.py:195-198 / examples/utils.py:40-45 hand train_operator a csv.DictWriter with exactly the fields iter,
    train_loss, avg_train_loss, time (extrasaction='raise'): every printed row must carry those keys and no other -
    also where the loss is only sampled (batches above 1024 rows, several ranks: loss_stride > 1)."""
    import csv
    from neural_svd_amd.drop_in import train_operator
    z = G.load("model_headline")
    cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build("hyd_med", z)
    if batch is not None:
        from neural_svd_amd.operators import get_dataloader
        args.batch_size = batch
        make_batch, val_data, batch_ftn_val, imp_train, imp_val = get_dataloader(args, DEV)
    args.num_iters, args.print_freq = 64, 32
    path = tmp_path / "log.csv"
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["iter", "train_loss", "avg_train_loss", "time"])
        w.writeheader()
        train_operator(args, method, operator, make_batch, val_data, batch_ftn_val, w, fh, DEV, imp_train, imp_val, gt)
    rows = list(csv.DictReader(open(path)))
    assert [int(r["iter"]) for r in rows] == [32, 64]
    for r in rows:
        assert set(r) == {"iter", "train_loss", "avg_train_loss", "time"}
        assert np.isfinite(float(r["train_loss"])) and np.isfinite(float(r["avg_train_loss"]))


def test_train_operator_use_amp_selects_the_split_bf16_forward():
    """args.use_amp - the reference's autocast + GradScaler switch (examples/operator/__init__.py:37-38,62-72) - selects
    this package's mixed-precision forward (NSVD_PATH_FUSED_BF16X3: bf16 MFMA on three-way split operands, float32
    accumulation) where the MFMA kernels take the model, and is refused elsewhere. Same accuracy class as float32: after
    a few steps the two runs' parameters agree to 1e-4 of their updates' scale."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.drop_in import train_operator
    runs = {}
    for amp in (False, True):
        z = G.load("model_headline")
        cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build("hyd_med", z)
        args.num_iters, args.use_amp = 5, amp
        train_operator(args, method, operator, make_batch, val_data, batch_ftn_val, None, None, DEV, imp_train, imp_val, gt)
        assert method.path == (H.PATH_FUSED_BF16X3 if amp else H.PATH_AUTO)
        runs[amp] = {n: p.detach().clone() for n, p in method.named_parameters() if p.requires_grad}
    z = G.load("model_headline")
    p0 = {n: p.detach().clone() for n, p in build("hyd_med", z)[4].named_parameters() if p.requires_grad}
    for n in runs[True]:
        upd = (runs[False][n] - p0[n]).double().norm()
        assert torch.isfinite(runs[True][n]).all() and float(upd) > 0
        # RMSprop's first steps are sign-like: elements whose gradient sits at the rounding level may flip; the bulk agrees
        assert float((runs[True][n] - runs[False][n]).double().norm()) < 0.05 * float(upd), n
    z = G.load("model_small")  # hidden layers the MFMA kernels do not take: no such forward
    cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build("hyd_small", z)
    args.num_iters, args.use_amp = 2, True
    with pytest.raises(NotImplementedError):
        train_operator(args, method, operator, make_batch, val_data, batch_ftn_val, None, None, DEV, imp_train, imp_val, gt)


@pytest.mark.parametrize("hidden,m,B", [((32, 32), 16, 24), ((128, 128, 128), 64, 64), ((128, 128), 64, 160),
                                        ((128, 128), 128, 96)])
def test_fused_trainer_step_matches_oracle(hidden, m, B):
    """one full optimiser step of FusedTrainer (generic and fused-MFMA shapes) vs the float64 oracle
    on the same batch: loss, gradient, RMSprop update, EMA."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    L, D = 4, 2
    shape = H.ModelShape(L=L, D=D, m=m, hidden=hidden)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    tr = FusedTrainer(shape, prob, B, sequential=False, lr=1e-4, num_iters=100, sampling_scale=16.0,
                      fourier_scale=0.1, seed=5, device=DEV, keep_grads=True)
    p = O.init_params(L, D, m, hidden, 0.1, seed=5)
    for got, want in zip(tr.P.views(tr.P.flat), p.trainable()):
        assert torch.equal(got.cpu(), want)
    x = tr.sample().clone()
    assert abs(float(x.std()) - 16.0) < 16.0 * 5.0 / np.sqrt(x.numel())
    prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
    v, M = O.joint_nesting_masks(L, 1)
    p64 = p.to(torch.float64)
    ref = O.loss_and_grads(x.double().cpu(), p64, prob_o, v, M)
    tr.step(x)
    torch.cuda.synchronize()
    assert abs(float(tr.loss[0]) - float(ref["loss"])) < 0.1 * abs(float(ref["loss"]))
    gflat = torch.cat([g.reshape(-1) for g in ref["grads"]])
    got = torch.cat([g.reshape(-1) for g in tr.P.views(tr.P.grad)])
    assert rel(got, gflat) < 0.1
    # RMSprop's first step is -lr/sqrt(1-alpha) * sign(g) (+-3.16e-3): compare the update where the
    # gradient is not at the noise floor
    sq = [torch.zeros_like(t) for t in p64.trainable()]
    before = [t.clone() for t in p64.trainable()]
    O.rmsprop_step(p64.trainable(), ref["grads"], sq, O.cosine_lr(1e-4, 0, 100), alpha=0.999, eps=1e-10)
    upd_ref = torch.cat([(a - b).reshape(-1) for a, b in zip(p64.trainable(), before)])
    upd_got = torch.cat([(a.double().cpu() - b).reshape(-1) for a, b in zip(tr.P.views(tr.P.flat), before)])
    strong = gflat.abs() > 0.05 * gflat.abs().mean()
    agree = (torch.sign(upd_ref[strong]) == torch.sign(upd_got[strong])).double().mean()
    assert agree > 0.97, float(agree)
    # every update (away from the eps = 1e-10 regime of vanishing gradients) has the RMSprop first-step size
    assert float((upd_got[strong].abs() - 1e-4 / np.sqrt(1e-3)).abs().max()) < 1e-6
    # EMA after one update: shadow = p0 - (1 - 2/11) (p0 - p1)
    ema = torch.cat([t.reshape(-1) for t in tr.P.views(tr.P.ema)]).double().cpu()
    p0 = torch.cat([t.reshape(-1) for t in before])
    p1 = torch.cat([t.reshape(-1) for t in tr.P.views(tr.P.flat)]).double().cpu()
    assert rel(ema, p0 - (1 - 2 / 11) * (p0 - p1)) < 1e-6
    assert tr.t == 1 and tr.num_updates == 1


@pytest.mark.parametrize("hidden,m,B,mask", [((32, 32), 16, 24, False), ((128, 128, 128), 128, 64, True),
                                              ((128, 128), 64, 96, False), ((128, 128), 128, 1024, True),
                                              ((128, 128), 64, 160, False)])
def test_optimiser_step_fused_into_backward_is_bit_identical(hidden, m, B, mask):
    """nsvd_operator_backward_evd_step (RMSprop + EMA inside the weight-gradient kernel, gradients never stored)
    vs nsvd_operator_backward_evd + nsvd_rmsprop_ema_step: same parameters, square averages and EMA shadows bit
    for bit after several steps, on the generic path, the fused-MFMA path and its split-K form (last case)."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=4, D=2, m=m, hidden=hidden, has_exp_mask=mask)
    prob = H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0)
    kw = dict(sequential=False, lr=1e-3, num_iters=50, sampling_scale=4.0, fourier_scale=0.15,
              exp_mask_init=10.0 if mask else None, seed=2, device=DEV)
    a = FusedTrainer(shape, prob, B, fused_step=True, **kw)
    b = FusedTrainer(shape, prob, B, fused_step=False, **kw)
    c = FusedTrainer(shape, prob, B, fused_step=True, keep_grads=True, **kw)
    assert a.fused_step and not b.fused_step
    for _ in range(4):
        x = b.sample().clone()
        a.step(x)
        b.step(x)
        c.step(x)
    torch.cuda.synchronize()
    for name in ("flat", "sq", "ema"):
        assert torch.equal(getattr(a.P, name), getattr(b.P, name)), name
        assert torch.equal(getattr(c.P, name), getattr(b.P, name)), name
    assert torch.equal(c.P.grad, b.P.grad)
    if H.path_name(shape, B) == "fused_mfma":
        assert float(a.P.grad.abs().max()) == 0.0  # never written
    assert torch.equal(a.loss, b.loss) and a.t == b.t == 4 and a.num_updates == b.num_updates == 4


def test_fused_trainer_learns_oscillator():
    """convergence smoke: tiny oscillator model, a few thousand steps, Rayleigh quotients
    approach the analytic spectrum [14, 12, 12, 10, 10, 10] (shift 16 - (2n + 2))."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=6, D=2, m=64, hidden=(64, 64, 64), has_exp_mask=True)
    prob = H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0)
    steps = 4000
    tr = FusedTrainer(shape, prob, 256, sequential=True, lr=1e-3, num_iters=steps, sampling_scale=4.0,
                      fourier_scale=0.15, exp_mask_init=10.0, seed=0, device=DEV)
    for _ in range(steps):
        tr.step()
    out = tr.spectrum(lim=5.0, val_eps=0.1)
    gt = np.array([14.0, 12, 12, 10, 10, 10])
    err = np.abs(out["eigvals"].numpy() - gt) / gt
    assert np.isfinite(err).all() and err.mean() < 2e-2, (out["eigvals"], err)


@pytest.mark.parametrize("eps", [0.01, 0.0])
def test_fused_trainer_learns_oscillator_mfma_path(eps):
    """the same convergence smoke on the MFMA path (128-wide hidden layers), with the finite-difference stencil
    (eps = 0.01) and with the exact-Laplacian jets (eps = 0): optimiser step inside the weight-gradient kernel,
    batches drawn inside the feature kernel."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=6, D=2, m=64, hidden=(128, 128), has_exp_mask=True)
    prob = H.make_problem(H.POT_HARMONIC, 1.0, eps, 1.0, 16.0, 4.0)
    steps = 4000
    tr = FusedTrainer(shape, prob, 256, sequential=True, lr=1e-3, num_iters=steps, sampling_scale=4.0,
                      fourier_scale=0.15, exp_mask_init=10.0, seed=0, device=DEV)
    assert H.path_name(shape, 256) == "fused_mfma" and tr.fused_step and tr.device_sampler
    for _ in range(steps):
        tr.step()
    out = tr.spectrum(lim=5.0, val_eps=0.1)
    gt = np.array([14.0, 12, 12, 10, 10, 10])
    err = np.abs(out["eigvals"].numpy() - gt) / gt
    assert np.isfinite(err).all() and err.mean() < 2e-2, (out["eigvals"], err)


@pytest.mark.parametrize("path", ["auto", "bf16x3"])
def test_headline_config_soak(path):
    """configs[1] with its own sampler, 6000 optimiser steps (1.6 s): parameters, EMA shadow and optimiser state stay
    finite and the loss has dropped (median over the first 1000 steps' samples against the last 2000 steps'). (A sample drawn exactly at the origin - one in 2^24 with the first version of the
    sampler - made every parameter NaN at step 4081 of this very run.)"""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    tr = FusedTrainer(shape, prob, 512, sequential=False, step=1, lr=1e-4, num_iters=500000, seed=0, device=DEV,
                      path=H.PATH_FUSED_BF16X3 if path == "bf16x3" else H.PATH_AUTO)
    losses = []
    for i in range(6000):
        tr.step()
        if i % 25 == 24:
            losses.append(tr.loss[0].clone())
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    assert all(np.isfinite(losses)), losses
    for t in (tr.P.flat, tr.P.ema, tr.P.sq, tr.f, tr.Tf):
        assert bool(torch.isfinite(t).all())
    # single-batch losses of the hydrogen problem scatter over +-10^4 (the -Z/r potential: one sample near the origin
    # moves a batch's operator term by thousands), so the trend is read from medians over 40 / 80 batches, not from
    # a handful of values
    assert np.median(losses[-80:]) < np.median(losses[:40]) - 500.0, (np.median(losses[:40]), np.median(losses[-80:]))


def test_model_autograd_and_compute_loss_kernel():
    """method(x) is differentiable (nsvd_model_forward/_backward), and compute_loss_kernel
    (methods/nestedlora.py:230-252, both split_batch modes) runs a user kernel operator built on it."""
    z = G.load("model_small")
    case = "osc_small"
    cfg, args, operator, gt, method, loaders = build(case, z)
    p64 = G.params_from_golden(z, case).to(torch.float64)
    x = torch.tensor(z[f"{case}_x"][0]).to(DEV)
    xc = x.double().cpu()
    B = x.shape[0]

    # --- plain autograd through method(x)
    wgt = torch.randn(B, cfg["neigs"], generator=torch.Generator().manual_seed(3))
    out = method(x)
    assert out.requires_grad
    (out * wgt.to(DEV)).sum().backward()
    prob_plain = O.Problem(potential=O.POT_HARMONIC, eps=0.01, use_importance=False)
    c = O.operator_forward(xc, p64, prob_plain)          # without importance, f == model(x)
    assert rel(out, c.f) < 1e-5
    gref = O.operator_backward(c, p64, prob_plain, wgt.double())
    names = G.trainable_names(z, case)
    got = dict(method.named_parameters())
    for n, g in zip(names, gref):
        assert rel(got[n].grad, g) < 3e-5, n
    method.zero_grad()

    # --- a Gaussian kernel operator: (K f)(x_i) = mean_j k(x_i, x_j) f(x_j)
    def get_approx_kernel_op(xk):
        K = torch.exp(-0.02 * torch.cdist(xk, xk) ** 2)

        def op(model, xx, importance=None):
            f = model(xx)
            return K @ f / xx.shape[0], f
        return op

    loss, aux = method.compute_loss_kernel(get_approx_kernel_op, x, None, split_batch=False)
    loss.backward()
    K64 = torch.exp(-0.02 * torch.cdist(xc, xc) ** 2)
    Kf64 = K64 @ c.f / B
    v, M = method.vector_mask.double(), method.matrix_mask.double()
    l64, lam1, lam2, _, _ = O.evd_loss_forward(c.f, Kf64, v, M)
    df = O.evd_loss_backward(c.f, Kf64, v, M, lam1, lam2)
    gref = O.operator_backward(c, p64, prob_plain, df)
    assert abs(float(loss.detach()) - float(l64)) < 1e-4 * abs(float(l64))
    for n, g in zip(names, gref):
        assert rel(got[n].grad, g) < 1e-4, n

    # --- split_batch=True: operator term on x1 with the kernel anchored at x2, f2 = model(x2) independent
    method.zero_grad()

    def get_cross_kernel_op(xk):
        def op(model, xx, importance=None):
            K = torch.exp(-0.02 * torch.cdist(xx, xk) ** 2)
            return K @ model(xk) / xk.shape[0], model(xx)
        return op

    loss, aux = method.compute_loss_kernel(get_cross_kernel_op, x, None, split_batch=True)
    loss.backward()
    B1 = (B + 1) // 2
    c1 = O.operator_forward(xc[:B1], p64, prob_plain)
    c2 = O.operator_forward(xc[B1:], p64, prob_plain)
    Kf1 = torch.exp(-0.02 * torch.cdist(xc[:B1], xc[B1:]) ** 2) @ c2.f / (B - B1)
    lam1, lam2 = c1.f.T @ c1.f / B1, c2.f.T @ c2.f / (B - B1)
    l64 = -2.0 * ((c1.f * Kf1) @ v).mean() + (M * lam1 * lam2).sum()
    df1 = -(4.0 / B1) * Kf1 * v + (2.0 / B1) * c1.f @ (M * lam2)
    df2 = (2.0 / (B - B1)) * c2.f @ (M * lam1)
    gref = [a + b for a, b in zip(O.operator_backward(c1, p64, prob_plain, df1),
                                  O.operator_backward(c2, p64, prob_plain, df2))]
    assert abs(float(loss.detach()) - float(l64)) < 1e-4 * abs(float(l64))
    assert aux["f"].shape[0] == B1 and rel(aux["Tf"], Kf1) < 1e-4
    for n, g in zip(names, gref):
        assert rel(got[n].grad, g) < 1e-4, n


@pytest.mark.gpu
def test_fused_step_bit_identical_at_headline_size():
    """configs[1]'s size (every CU holds a dW_0 tile, the optimiser state of half of each tile is prefetched under the
    K loop): the fused step, the fused step that also stores gradients and backward + separate optimiser kernel stay
    bit-identical over ten steps of the internal sampler."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    kw = dict(sequential=True, seed=0, device=DEV)
    a = FusedTrainer(shape, prob, 512, fused_step=True, **kw)
    b = FusedTrainer(shape, prob, 512, fused_step=False, **kw)
    c = FusedTrainer(shape, prob, 512, fused_step=True, keep_grads=True, **kw)
    for it in range(10):
        a.step(); b.step(); c.step()
    torch.cuda.synchronize()
    assert torch.equal(a.x, b.x)
    for t in (a, c):
        assert torch.equal(t.P.flat, b.P.flat) and torch.equal(t.P.ema, b.P.ema) and torch.equal(t.P.sq, b.P.sq)
    assert torch.equal(c.P.grad, b.P.grad)


def test_three_step_trajectory_at_the_headline_shape_against_the_oracle():
    """configs[1]'s shape (hydrogen, L = 16, B = 512, m = 1024, 128 x 3, joint nesting): three consecutive
    FusedTrainer.step() calls - device sampler, forward, loss, backward, RMSprop with a moving cosine learning rate,
    EMA with its warm-up - each checked against the float64 oracle (loss_and_grads' pieces + rmsprop_step + ema_update,
    i.e. the loop body of examples/operator/__init__.py:55-74) on two sampled heads, all 512 rows.
    The oracle is re-started from the trainer's own state at the beginning of every step: RMSprop's first steps are
    sign-like (|update| = lr / sqrt(1 - alpha) whatever |g|), so free-running float32 and float64 trajectories separate
    at the elements whose gradient is at the rounding level - what is compared is every step of the trajectory, with
    the state (parameters, square averages, EMA shadow, schedule position) carried by the trainer."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    L, D, m, hidden, B, T = 16, 2, 1024, (128, 128, 128), 512, 10
    shape = H.ModelShape(L=L, D=D, m=m, hidden=hidden)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    tr = FusedTrainer(shape, prob, B, sequential=False, lr=1e-4, num_iters=T, seed=0, device=DEV)
    assert tr.fused_step and tr.guest_features and tr.direct_moments
    prob_o = O.Problem(potential=O.POT_HYDROGEN, eps=0.01, op_scale=100.0, op_shift=0.0, sigma=16.0)
    v, M = O.joint_nesting_masks(L, 1)
    v, M = v.double(), M.double()
    nl = len(hidden) + 1
    heads = (3, 12)
    for t in range(3):
        before = {k: [u.clone() for u in tr.P.views(getattr(tr.P, k))] for k in ("flat", "sq", "ema")}
        tr.step()
        torch.cuda.synchronize()
        x, f64, Tf64 = tr.x.double().cpu(), tr.f.double().cpu(), tr.Tf.double().cpu()
        loss64, lam1, lam2, _, _ = O.evd_loss_forward(f64, Tf64, v, M)
        assert abs(float(tr.loss[0]) - float(loss64)) < 2e-5 * float(tr.loss[1:].abs().max())
        df64 = O.evd_loss_backward(f64, Tf64, v, M, lam1, lam2)
        lr = O.cosine_lr(1e-4, t, T)
        after = {k: tr.P.views(getattr(tr.P, k)) for k in ("flat", "sq", "ema")}
        for l in heads:
            ph = O.Params([before["flat"][i][l:l + 1].double().cpu() for i in range(nl)],
                          [before["flat"][nl + i][l:l + 1].double().cpu() for i in range(nl)],
                          tr.P.fourier_B.double().cpu(), None)
            c = O.operator_forward(x, ph, prob_o)
            assert rel(tr.f[:, l], c.f[:, 0]) < 2e-5
            g = O.operator_backward(c, ph, prob_o, df64[:, l:l + 1])
            ps = ph.trainable()
            sq = [before["sq"][i][l:l + 1].double().cpu() for i in range(2 * nl)]
            sh = [before["ema"][i][l:l + 1].double().cpu() for i in range(2 * nl)]
            p0 = [q.clone() for q in ps]
            O.rmsprop_step(ps, g, sq, lr, alpha=0.999, eps=1e-10)
            assert O.ema_update(sh, ps, 0.995, t) == t + 1
            for i in range(2 * nl):
                got_p, got_sq, got_e = (after[k][i][l:l + 1].double().cpu() for k in ("flat", "sq", "ema"))
                assert rel(got_sq, sq[i]) < 2e-4, (t, l, i, rel(got_sq, sq[i]))
                strong = g[i].abs() > 0.05 * g[i].abs().mean()
                upd_ref, upd_got = (ps[i] - p0[i])[strong], (got_p - p0[i])[strong]
                assert float((upd_got - upd_ref).norm() / upd_ref.norm()) < 2e-3, (t, l, i)
                # EMA: shadow - (1 - d)(shadow - p) with d = min(0.995, (1 + n) / (10 + n)), n = t + 1
                d = min(0.995, (2 + t) / (11 + t))
                want_e = before["ema"][i][l:l + 1].double().cpu()
                want_e = want_e - (1 - d) * (want_e - got_p)
                assert rel(got_e, want_e) < 1e-6, (t, l, i)
                assert float((got_e - sh[i])[strong].norm()) <= 2e-3 * float((ps[i] - p0[i])[strong].norm()) + 1e-12
    assert tr.t == tr.num_updates == 3


@pytest.mark.parametrize("D,L,m,B", [(2, 4, 64, 64), (1, 2, 128, 32), (2, 16, 1024, 512)])
def test_next_batch_by_guest_workgroups_is_bit_identical(D, L, m, B):
    """nsvd_operator_backward_evd_step_next: the next batch drawn and its features written by guest workgroups of the
    backward's chain kernel give, bit for bit, the run that launches nsvd_operator_sample_features per step - same
    batches (the sampler is counter-based), same features, same parameters; one batch is prepared ahead."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    shape = H.ModelShape(L=L, D=D, m=m, hidden=(128, 128, 128), has_exp_mask=True)
    prob = H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0)
    kw = dict(sequential=True, seed=3, device=DEV, sampling_scale=4.0, fourier_scale=0.5, exp_mask_init=10.0, lr=1e-3)
    a = FusedTrainer(shape, prob, B, **kw)
    b = FusedTrainer(shape, prob, B, overlap=False, **kw)
    assert a.guest_features and not b.guest_features and a.fused_step and b.fused_step
    for it in range(7):
        a.step()
        b.step()
        if it == 3:  # an externally supplied batch in between: the prepared one is kept for the next internal step
            xe = torch.full((B, D), 0.5, device=DEV)
            a.step(xe)
            b.step(xe)
    torch.cuda.synchronize()
    assert a.batches_drawn == b.batches_drawn + 1 == 8
    assert torch.equal(a.x, b.x) and torch.equal(a.f, b.f) and torch.equal(a.Tf, b.Tf)
    assert torch.equal(a.P.flat, b.P.flat) and torch.equal(a.P.ema, b.P.ema) and torch.equal(a.P.sq, b.P.sq)
    assert torch.equal(a.loss, b.loss)


def test_register_eigvals_permutation_on_the_operator_path():
    """reference methods/nestedlora.py:195-210: after register_eigvals() the training forward hands out the model's
    columns sorted by eigenvalue (descending), so the operator's Tf, f - and with them the nesting order of the loss -
    are permuted. Check against the oracle: loss of the permuted columns, and parameter gradients == the oracle's
    backward of d loss / d f routed back through the inverse permutation."""
    z = G.load("model_small")
    case = "osc_small"
    cfg, args, operator, gt, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = build(case, z)
    x = torch.tensor(z[f"{case}_x"][0]).to(DEV)
    method.train()
    L = cfg["neigs"]
    eig = torch.linspace(1.0, 2.0, L)[torch.randperm(L, generator=torch.Generator().manual_seed(1))]
    method.register_eigvals(eig.numpy())
    idx = method.sort_indices
    assert not torch.equal(idx, torch.arange(L))
    loss, aux = method.compute_loss_operator(operator, x, importance=imp_train)
    loss.backward()
    got = {n: p.grad.clone() for n, p in method.named_parameters() if p.grad is not None}
    # the same thing by hand: unpermuted outputs, permuted columns, oracle loss / gradient
    method.reset_eigvals()
    for p in method.parameters():
        p.grad = None
    loss0, aux0 = method.compute_loss_operator(operator, x, importance=imp_train)
    f0, Tf0 = aux0["f"].detach(), aux0["Tf"].detach()
    assert torch.equal(aux["f"].detach(), f0[:, idx.to(DEV)]) and torch.equal(aux["Tf"].detach(), Tf0[:, idx.to(DEV)])
    v, M = method.vector_mask.double(), method.matrix_mask.double()
    fp, Tfp = f0[:, idx.to(DEV)].double().cpu(), Tf0[:, idx.to(DEV)].double().cpu()
    want_loss, lam1, lam2, _, _ = O.evd_loss_forward(fp, Tfp, v, M)
    assert abs(float(loss.detach()) - float(want_loss)) < 2e-5 * abs(float(want_loss))
    dfp = O.evd_loss_backward(fp, Tfp, v, M, lam1, lam2)
    df = torch.zeros_like(dfp)
    df[:, idx] = dfp                                   # back through f[:, idx]
    (f0.double().cpu() * df).sum()                     # (shape check)
    # gradients of sum(f * df) through the HIP backward == what the permuted loss produced
    for p in method.parameters():
        p.grad = None
    Tf1, f1 = method.apply_operator(operator, x, imp_train)
    (f1 * df.float().to(DEV)).sum().backward()
    for n, p in method.named_parameters():
        if p.grad is None:
            continue
        assert rel(got[n], p.grad) < 2e-5, n
