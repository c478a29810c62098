"""nsvd_cdk_step (the whole Sketchy training step in one C call, behind cdk.FusedCdkStep) against

  * the REFERENCE's own step (tests/golden/cdk_step.npz, case sb: 128 -> 256 -> 128 towers, batch 128, three steps of
    HeteroNetwork + NestedLoRAForCDK + clip_grad_norm_ + SGD momentum + CosineAnnealingLR): losses, total gradient
    norms, parameters, momentum buffers, running statistics - float64 truth with the reference's float32 run as the
    yardstick;
  * this package's module-by-module path (torch autograd around the same HIP stages + torch's clip and SGD) at
    BASELINE configs[4]'s size, where only the optimiser arithmetic differs (fp32 rounding)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


KEYS = {"W1": "0.weight", "b1": "0.bias", "g1": "1.weight", "be1": "1.bias", "W2": "3.weight", "b2": "3.bias",
        "g2": "4.weight", "be2": "4.bias"}


def _build(sizes, mu, seed):
    from neural_svd_amd.cdk import HeteroNetwork, NestedLoRAForCDK, get_mlp
    torch.manual_seed(seed)  # the reference's constructor calls in the reference's order: same initial weights
    model = HeteroNetwork([get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                           get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                          [nn.Identity(), nn.Identity()], mu=mu, regularize_mode="l2_ball").to(DEV).train()
    method = NestedLoRAForCDK(model, neigs=sizes[-1], step=1, sequential=False, set_first_mode_const=True).to(DEV)
    return model, method


def test_cdk_step_matches_reference_golden():
    from neural_svd_amd.cdk import FusedCdkStep
    z = G.load("cdk_step")
    name = "sb"
    B, d0, d1, d2, seed, nstep, T = [int(v) for v in z[f"{name}_cfg"]]
    mu, lr0, mom, max_norm, slope = [float(v) for v in z[f"{name}_hyper"]]
    model, method = _build([d0, d1, d2], mu, seed)
    assert np.array_equal(method.vector_mask.cpu().numpy().astype(np.float64), z[f"{name}_v"].astype(np.float64))
    g = torch.Generator().manual_seed(3000 + seed)
    xs = torch.randn(nstep, B, d0, generator=g, dtype=torch.float64).float().to(DEV)
    ys = torch.randn(nstep, B, d0, generator=g, dtype=torch.float64).float().to(DEV)
    fs = FusedCdkStep(method, lr=lr0, momentum=mom, max_grad_norm=max_norm, t_max=T, batch_size=B)
    q64, q32 = f"{name}_f64_", f"{name}_f32_"
    for t in range(nstep):
        out = fs.step(xs[t], ys[t]).cpu().double().numpy()
        want, ref32 = z[q64 + "loss"][t], z[q32 + "loss"][t]
        for i in range(3):
            tol = max(4 * abs(ref32[i] - want[i]), 3e-6 * max(1.0, abs(want[i])))
            assert abs(out[i] - want[i]) < tol, (t, i, out[i], want[i])
        wn, rn = z[q64 + "total_norm"][t], z[q32 + "total_norm"][t]
        assert abs(out[3] - wn) < max(4 * abs(rn - wn), 3e-6 * wn), (t, out[3], wn)
    fs.flush_counters()
    torch.cuda.synchronize()
    sd = model.state_dict()
    for side in "xy":
        for k, n in KEYS.items():
            key = f"backbones.{side}.{n}"
            got = sd[key].double().cpu().numpy()
            want_n, ref32_n = float(z[q64 + f"pnorm_{key}"]), float(z[q32 + f"pnorm_{key}"])
            assert abs(np.linalg.norm(got) - want_n) < max(4 * abs(ref32_n - want_n), 2e-6 * max(1.0, want_n)), key
            samp = z[q64 + f"param_{key}"]
            err = np.linalg.norm(got.reshape(-1)[::5] - samp) / max(np.linalg.norm(samp), 1e-30)
            assert err < (2e-4 if k in ("b1", "b2") else 5e-6), (key, err)  # biases in front of a BatchNorm: ~0 gradients
            bgot = fs.bufs["xy".index(side)][k].double().cpu().numpy()
            bw, b32 = float(z[q64 + f"bufnorm_{key}"]), float(z[q32 + f"bufnorm_{key}"])
            assert abs(np.linalg.norm(bgot) - bw) < max(6 * abs(b32 - bw), 1e-5 * max(bw, 1e-3)), (key, np.linalg.norm(bgot), bw)
        for n in ("1.running_mean", "1.running_var", "4.running_mean", "4.running_var"):
            key = f"backbones.{side}.{n}"
            samp = z[q64 + f"param_{key}"]
            got = sd[key].double().cpu().numpy().reshape(-1)[::5]
            assert np.linalg.norm(got - samp) / np.linalg.norm(samp) < 5e-6, key
        assert int(sd[f"backbones.{side}.1.num_batches_tracked"]) == nstep


def test_cdk_step_matches_module_path_at_headline_size():
    """configs[4]'s shapes (1024 x 512 -> 8192 -> 512, L = 512): three fused steps against three steps of the module-by-
    module path on an identically initialised copy (same HIP forward / backward stages; torch's clip_grad_norm_ and SGD)"""
    from neural_svd_amd.cdk import FusedCdkStep
    sizes, B, mu = [512, 8192, 512], 1024, 16.0
    ma, meth_a = _build(sizes, mu, 77)
    mb, meth_b = _build(sizes, mu, 77)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa, pb)
    g = torch.Generator().manual_seed(78)
    xs = torch.randn(3, B, sizes[0], generator=g).to(DEV)
    ys = torch.randn(3, B, sizes[0], generator=g).to(DEV)
    ok, why = FusedCdkStep.supported(meth_a, B)
    assert ok, why
    fs = FusedCdkStep(meth_a, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=100, batch_size=B)
    opt = torch.optim.SGD(mb.parameters(), lr=5e-3, momentum=0.9)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 100)
    for t in range(3):
        la = fs.step(xs[t], ys[t]).clone()
        opt.zero_grad()
        _, fx, _, fy = meth_b(xs[t], ys[t])
        lb = meth_b.compute_loss(fx, fy)
        lb[0].backward()
        nb = nn.utils.clip_grad_norm_(mb.parameters(), max_norm=1.0)
        opt.step()
        sched.step()
        assert abs(float(la[0]) - float(lb[0])) < 2e-5 * max(1.0, abs(float(lb[0]))), (t, float(la[0]), float(lb[0]))
        assert abs(float(la[3]) - float(nb)) < 2e-5 * float(nb), (t, float(la[3]), float(nb))
        assert abs(fs.current_lr() - sched.get_last_lr()[0]) < 1e-12
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        d = float((pa - pb).double().norm() / pb.double().norm())
        assert d < (1e-3 if n.endswith(("0.bias", "3.bias")) else 5e-6), (n, d)
    for k in ("backbones.x.1.running_var", "backbones.y.4.running_mean"):
        a, b = ma.state_dict()[k], mb.state_dict()[k]
        assert float((a - b).double().norm() / b.double().norm()) < 2e-6
    # bit reproducibility of the fused step
    mc, meth_c = _build(sizes, mu, 77)
    fc = FusedCdkStep(meth_c, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=100, batch_size=B)
    for t in range(3):
        fc.step(xs[t], ys[t])
    for pa, pc in zip(ma.parameters(), mc.parameters()):
        assert torch.equal(pa, pc)


def test_cdk_step_refuses_what_it_does_not_implement():
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.cdk import FusedCdkStep
    model, method = _build([128, 256, 128], 16.0, 5)
    assert not FusedCdkStep.supported(method, 100)[0]           # batch not a multiple of 128
    with pytest.raises(H.NsvdError):
        FusedCdkStep(method, lr=1e-3, batch_size=100)
    fs = FusedCdkStep(method, lr=1e-3, batch_size=128)
    model.eval()
    with pytest.raises(H.NsvdError):
        fs.step(torch.zeros(128, 128, device=DEV), torch.zeros(128, 128, device=DEV))


@pytest.mark.parametrize("form", ["fused", "strips"])
def test_cdk_step_mixed_precision_against_the_oracle_with_the_same_rounding(form, monkeypatch):
    """(form: the wide layer with BatchNorm inside the contraction - csrc/tower_col.h, the default - or the contraction +
    strip kernels, NSVD_TOWER16_FUSED=0; each against the oracle mode that restates its roundings.)
    FusedCdkStep(use_amp=True): three training steps in the mixed-precision mode (bfloat16 operands and wide
    activations, both towers through every launch together, the bfloat16 weight copies of steps 2 and 3 written by the
    previous step's optimiser kernel), against oracle.cdk_train_step(gemm_bf16=True) in float64 with the same roundings
    from the same initial weights and batches - losses, total gradient norms, parameters and momentum buffers - and the
    float32 mode beside it (close, not equal)."""
    from oracle import nsvd_oracle as O
    from neural_svd_amd.cdk import FusedCdkStep
    sizes, B, mu, lr, mom, max_norm, slope = [128, 256, 256], 256, 16.0, 5e-3, 0.9, 1.0, 0.2
    from neural_svd_amd import hip_ops as H
    if form == "strips":
        monkeypatch.setenv("NSVD_TOWER16_FUSED", "0")
    assert H.tower_mixed_fused(B, *sizes, slope) == (form == "fused")
    omode = "fused" if form == "fused" else True
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(3, B, sizes[0], generator=g)
    ys = torch.randn(3, B, sizes[0], generator=g)
    runs = {}
    for amp in (True, False):
        model, method = _build(sizes, mu, 11)
        if amp:
            sd0 = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
        fs = FusedCdkStep(method, lr=lr, momentum=mom, max_grad_norm=max_norm, t_max=0, batch_size=B, use_amp=amp)
        outs = [fs.step(xs[t].to(DEV), ys[t].to(DEV)).cpu().double().clone() for t in range(3)]
        fs.flush_counters()
        runs[amp] = (outs, {k: v.detach().double().cpu() for k, v in model.state_dict().items()},
                     [{k: b.double().cpu() for k, b in bb.items()} for bb in fs.bufs], method)
    # the float64 oracle with the same operand rounding
    method = runs[True][3]
    towers = [{k: sd0[f"backbones.{s}.{n}"].clone() for k, n in KEYS.items()} for s in "xy"]
    bufs = [{k: torch.zeros_like(v) for k, v in t.items()} for t in towers]
    running = [dict(rm1=sd0[f"backbones.{s}.1.running_mean"].clone(), rv1=sd0[f"backbones.{s}.1.running_var"].clone(),
                    rm2=sd0[f"backbones.{s}.4.running_mean"].clone(), rv2=sd0[f"backbones.{s}.4.running_var"].clone())
               for s in "xy"]
    v, M = method.vector_mask.double().cpu(), method.matrix_mask.double().cpu()
    for t in range(3):
        (loss, lop, lmet), total = O.cdk_train_step(xs[t].double(), ys[t].double(), towers, bufs, running, v, M, mu, lr,
                                                    mom, max_norm, slope, t == 0, gemm_bf16=omode)
        got = runs[True][0][t]
        assert abs(float(got[0]) - float(loss)) < 2e-4 * max(1.0, abs(float(loss))), (t, float(got[0]), float(loss))
        assert abs(float(got[3]) - float(total)) < 2e-3 * float(total), (t, float(got[3]), float(total))
    sd = runs[True][1]
    for si, s in enumerate("xy"):
        for k, n in KEYS.items():
            got, want = sd[f"backbones.{s}.{n}"], towers[si][k]
            move = float((want - sd0[f"backbones.{s}.{n}"]).norm())
            assert float((got - want).norm()) < 5e-3 * move + 1e-6 * float(want.norm()), (s, k)
    # float32 mode: the same trajectory up to the bfloat16 operand rounding
    l_amp, l_32 = float(runs[True][0][2][0]), float(runs[False][0][2][0])
    assert l_amp != l_32 and abs(l_amp - l_32) < 2e-2 * max(1.0, abs(l_32)), (l_amp, l_32)


@pytest.mark.parametrize("form", ["fused", "strips"])
def test_cdk_step_float16_with_the_grad_scaler_against_the_oracle(form, monkeypatch):
    """FusedCdkStep(use_amp=True, amp_dtype="float16"): the reference's own half type and its GradScaler (main_sketchy.py:
    161,194-208) - eight steps from a loss scale of 2^24 with growth_interval = 2 and CosineAnnealingLR(T_max = 6):
    the scaled float16 gradients overflow at 2^24 and fit at 2^23 (margins of 25 % and 50 %: found from the oracle, not
    a knife edge), so the run SKIPS steps 0, 3 and 6 (nothing may change in them but the scale), takes the other five
    with the scale doubling every second clean step, and the schedule advances on EVERY iteration as the script's does
    (main_sketchy.py:205-206). Against
    oracle.cdk_train_step(half="f16", scaler=...) in float64 with the same roundings and the same scaler arithmetic:
    every step's loss, unscaled gradient norm (or its non-finiteness), the scaler's trajectory, and the final
    parameters."""
    from oracle import nsvd_oracle as O
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.cdk import FusedCdkStep
    sizes, B, mu, lr, mom, max_norm, slope, T = [128, 256, 256], 256, 16.0, 5e-3, 0.9, 1.0, 0.2, 6
    if form == "strips":
        monkeypatch.setenv("NSVD_TOWER16_FUSED", "0")
    assert H.tower_mixed_fused(B, *sizes, slope) == (form == "fused")
    omode = "fused" if form == "fused" else True
    nstep = 8
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(nstep, B, sizes[0], generator=g)
    ys = torch.randn(nstep, B, sizes[0], generator=g)
    model, method = _build(sizes, mu, 11)
    sd0 = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
    fs = FusedCdkStep(method, lr=lr, momentum=mom, max_grad_norm=max_norm, t_max=T, batch_size=B, use_amp=True,
                      amp_dtype="float16", init_scale=2.0 ** 24, growth_interval=2)
    assert fs.scaler is not None and fs.scaler_state()["scale"] == 2.0 ** 24
    towers = [{k: sd0[f"backbones.{s}.{n}"].clone() for k, n in KEYS.items()} for s in "xy"]
    bufs = [{k: torch.zeros_like(v) for k, v in t.items()} for t in towers]
    running = [dict(rm1=sd0[f"backbones.{s}.1.running_mean"].clone(), rv1=sd0[f"backbones.{s}.1.running_var"].clone(),
                    rm2=sd0[f"backbones.{s}.4.running_mean"].clone(), rv2=sd0[f"backbones.{s}.4.running_var"].clone())
               for s in "xy"]
    v, M = method.vector_mask.double().cpu(), method.matrix_mask.double().cpu()
    sc = dict(scale=2.0 ** 24, growth_factor=2.0, backoff_factor=0.5, growth_interval=2, growth_tracker=0, steps_ok=0,
              steps_skipped=0)
    skipped = []
    for t in range(nstep):
        before = {k: p.detach().clone() for k, p in model.state_dict().items() if "running" not in k and "num_batches" not in k}
        got = fs.step(xs[t].to(DEV), ys[t].to(DEV)).cpu().double().clone()
        st = fs.scaler_state()
        (loss, lop, lmet), total = O.cdk_train_step(xs[t].double(), ys[t].double(), towers, bufs, running, v, M, mu,
                                                    O.cosine_lr(lr, t, T), mom, max_norm, slope, False, gemm_bf16=omode,
                                                    half="f16", scaler=sc)
        assert abs(float(got[0]) - float(loss)) < 2e-4 * max(1.0, abs(float(loss))), (t, float(got[0]), float(loss))
        assert (st["scale"], st["growth_tracker"], st["steps_ok"], st["steps_skipped"]) == \
            (sc["scale"], sc["growth_tracker"], sc["steps_ok"], sc["steps_skipped"]), (t, st, sc)
        if not bool(torch.isfinite(total)):
            skipped.append(t)
            assert st["last_found_inf"] == 1 and not np.isfinite(float(got[3]))
            after = model.state_dict()
            for k, p in before.items():  # scaler.step() skipped optimizer.step(): no parameter moved
                assert torch.equal(after[k], p), (t, k)
        else:
            assert st["last_found_inf"] == 0
            assert abs(float(got[3]) - float(total)) < 2e-3 * float(total), (t, float(got[3]), float(total))
    assert skipped == [0, 3, 6], skipped
    assert sc["steps_ok"] == 5 and sc["scale"] == 2.0 ** 23
    fs.flush_counters()
    sd = {k: p.detach().double().cpu() for k, p in model.state_dict().items()}
    for si, s in enumerate("xy"):
        for k, n in KEYS.items():
            got, want = sd[f"backbones.{s}.{n}"], towers[si][k]
            move = float((want - sd0[f"backbones.{s}.{n}"]).norm())
            assert float((got - want).norm()) < 5e-3 * move + 1e-6 * float(want.norm()), (s, k)
        for i, buf in fs.bufs[si].items():
            if i in ("b1", "b2"):  # a bias in front of a BatchNorm has a zero gradient: rounding noise on both sides
                assert float(buf.abs().max()) < 1e-5 and float(bufs[si][i].abs().max()) < 1e-5
                continue
            assert rel(buf, bufs[si][i]) < 5e-3, (s, i)


@pytest.mark.parametrize("form", ["fused", "strips"])
def test_cdk_step_float16_against_the_references_amp_loop(form, monkeypatch):
    """tests/golden/amp.npz: eight iterations of the REFERENCE's Sketchy loop body with its AMP branch on
    (main_sketchy.py:161,180-212: float16 autocast, GradScaler from 2^16 with growth_interval 2, clip_grad_norm_, SGD
    momentum, CosineAnnealingLR stepped every iteration) - run on the CPU where the fixture was made. FusedCdkStep(
    use_amp=True, amp_dtype="float16") from the same weights and batches: the scale's trajectory exactly, the unscaled
    gradient norms to 2e-4, the losses to 3e-3 (under autocast the reference's loss itself is float16 arithmetic; here
    it is float32), every parameter's update to 1 % of its length."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.cdk import FusedCdkStep
    if form == "strips":
        monkeypatch.setenv("NSVD_TOWER16_FUSED", "0")
    z = G.load("amp")
    B, d0, d1, d2, seed, nstep, T = [int(v) for v in z["amp_step_cfg"]]
    mu, lr, mom, max_norm, slope, init_scale, gi = [float(v) for v in z["amp_step_hyper"]]
    assert H.tower_mixed_fused(B, d0, d1, d2, slope) == (form == "fused")
    model, method = _build([d0, d1, d2], mu, seed)
    g = torch.Generator().manual_seed(77)
    xs, ys = torch.randn(nstep, B, d0, generator=g), torch.randn(nstep, B, d0, generator=g)
    fs = FusedCdkStep(method, lr=lr, momentum=mom, max_grad_norm=max_norm, t_max=T, batch_size=B, use_amp=True,
                      amp_dtype="float16", init_scale=init_scale, growth_interval=int(gi))
    for t in range(nstep):
        got = fs.step(xs[t].to(DEV), ys[t].to(DEV)).cpu().double().clone()
        want = z["amp_step_rows"][t]
        assert abs(float(got[0]) - want[0]) < 3e-3 * abs(want[0]), (t, float(got[0]), want[0])
        assert abs(float(got[3]) - want[1]) < 2e-4 * want[1], (t, float(got[3]), want[1])
        assert fs.scaler_state()["scale"] == want[2], (t, fs.scaler_state(), want[2])
    assert fs.scaler_state()["steps_skipped"] == 0 and fs.scaler_state()["steps_ok"] == nstep
    sd = {k: p.detach().double().cpu() for k, p in model.state_dict().items()}
    for s in "xy":
        for k, n in KEYS.items():
            if k in ("b1", "b2"):
                continue  # (a bias in front of a BatchNorm: zero gradient, rounding noise only)
            ref = torch.tensor(z[f"amp_step_param_backbones.{s}.{n}"]).double()
            got = sd[f"backbones.{s}.{n}"]
            got = got[..., ::3] if got.dim() == 2 else got
            move = float(z[f"amp_step_move_backbones.{s}.{n}"])
            assert float((got - ref).norm()) < 1e-2 * move, (s, k, float((got - ref).norm()) / move)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("sizes,B", [([128, 256, 256], 256), ([256, 512, 256], 512)])
def test_mixed_step_is_reproducible_run_to_run(sizes, B, dtype):
    """The same three mixed-precision steps from the same weights, 25 times in one process, bit for bit (losses,
    gradient norms, every parameter and running statistic). No atomics and fixed summation orders make the step
    reproducible by construction; what this test is for is a HAZARD - round 6 found an inline-asm global_store_dwordx4
    without the wait state hipcc does not add behind it (the next instruction overwrote the store's data registers):
    a few elements of the stored gradient wrong in about every second run of exactly this shape, every single run inside
    the other tests' tolerances."""
    import copy
    from neural_svd_amd.cdk import FusedCdkStep, NestedLoRAForCDK
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(3, B, sizes[0], generator=g).to(DEV)
    ys = torch.randn(3, B, sizes[0], generator=g).to(DEV)
    model, _ = _build(sizes, 16.0, 11)
    sd0 = copy.deepcopy(model.state_dict())
    ref = None
    for rep in range(25):
        model.load_state_dict(sd0)
        method = NestedLoRAForCDK(model, neigs=sizes[-1], step=1, sequential=False, set_first_mode_const=True).to(DEV)
        fs = FusedCdkStep(method, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=B, use_amp=True,
                          amp_dtype=dtype, grad_scaler=False)
        if rep % 3 == 1:
            torch.empty(1 << 22, device=DEV).normal_()  # (another cache / timing state in front of the steps)
        outs = [fs.step(xs[t], ys[t]).clone() for t in range(3)]
        torch.cuda.synchronize()
        cur = [o.cpu() for o in outs] + [v.detach().cpu().clone() for k, v in model.state_dict().items()
                                         if "num_batches" not in k]
        if ref is None:
            ref = cur
            continue
        for i, (a, b) in enumerate(zip(cur, ref)):
            assert torch.equal(a, b), (rep, i, float((a.double() - b.double()).abs().max()))


def test_float16_mode_refuses_what_it_cannot_do():
    """a GradScaler needs the mixed-precision step; amp_dtype is one of two names"""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.cdk import FusedCdkStep
    model, method = _build([128, 256, 256], 16.0, 3)
    with pytest.raises(H.NsvdError):
        FusedCdkStep(method, lr=1e-3, batch_size=256, use_amp=False, grad_scaler=True)
    with pytest.raises(H.NsvdError):
        FusedCdkStep(method, lr=1e-3, batch_size=256, use_amp=True, amp_dtype="float8")
    fs = FusedCdkStep(method, lr=1e-3, batch_size=256, use_amp=True, amp_dtype="bfloat16")
    assert fs.scaler is None and fs.scaler_state() is None


@pytest.mark.parametrize("amp", [False, True])
def test_cdk_step_at_headline_size_against_the_oracle(amp):
    """BASELINE.json configs[4] at its own size (B = 1024, towers 512 -> 8192 -> 512, L = 512 + constant mode): ONE fused
    nsvd_cdk_step against oracle.cdk_train_step (reference main_sketchy.py:180-212; float64; amp: the same roundings) from
    the same weights and batch: loss terms, total gradient norm, and the update of every parameter tensor on sampled
    rows / columns (the oracle's 16.8 M-parameter step runs once, in float64, on the CPU: about half a minute)."""
    from oracle import nsvd_oracle as O
    from neural_svd_amd.cdk import FusedCdkStep
    sizes, B, mu, lr, mom, max_norm, slope = [512, 8192, 512], 1024, 16.0, 5e-3, 0.9, 1.0, 0.2
    g = torch.Generator().manual_seed(123)
    x, y = torch.randn(B, sizes[0], generator=g), torch.randn(B, sizes[0], generator=g)
    model, method = _build(sizes, mu, 5)
    sd0 = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
    fs = FusedCdkStep(method, lr=lr, momentum=mom, max_grad_norm=max_norm, t_max=0, batch_size=B, use_amp=amp)
    out = fs.step(x.to(DEV), y.to(DEV)).cpu().double().clone()
    fs.flush_counters()
    sd = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    towers = [{k: sd0[f"backbones.{s}.{n}"].clone() for k, n in KEYS.items()} for s in "xy"]
    bufs = [{k: torch.zeros_like(v) for k, v in t.items()} for t in towers]
    running = [dict(rm1=sd0[f"backbones.{s}.1.running_mean"].clone(), rv1=sd0[f"backbones.{s}.1.running_var"].clone(),
                    rm2=sd0[f"backbones.{s}.4.running_mean"].clone(), rv2=sd0[f"backbones.{s}.4.running_var"].clone())
               for s in "xy"]
    v, M = method.vector_mask.double().cpu(), method.matrix_mask.double().cpu()
    from neural_svd_amd import hip_ops as H
    omode = ("fused" if H.tower_mixed_fused(B, *sizes, slope) else True) if amp else False
    (loss, lop, lmet), total = O.cdk_train_step(x.double(), y.double(), towers, bufs, running, v, M, mu, lr, mom,
                                                max_norm, slope, True, gemm_bf16=omode)
    # (mixed precision, measured: loss 8e-7, norm 1.4e-8, updates <= 1.6e-3 of their length - scripts/dev/
    # cdk_headline_margins.py; the bounds below leave a factor of 3 to 25 and are the float32 mode's where that is looser)
    tl = 2e-5
    assert abs(float(out[0]) - float(loss)) < tl * max(1.0, abs(float(loss))), (float(out[0]), float(loss))
    assert abs(float(out[1]) - float(lop)) < tl * max(1.0, abs(float(lop)))
    assert abs(float(out[2]) - float(lmet)) < tl * max(1.0, abs(float(lmet)))
    assert abs(float(out[3]) - float(total)) < 1e-4 * float(total), (float(out[3]), float(total))
    rows = torch.tensor([0, 1, 127, 128, 255, 256, 300, 511])
    for si, s in enumerate("xy"):
        for k, n in KEYS.items():
            got, want, start = sd[f"backbones.{s}.{n}"], towers[si][k], sd0[f"backbones.{s}.{n}"]
            if got.dim() == 2:  # sampled rows and a stride of columns of the weight matrices
                r = rows[rows < got.shape[0]]
                got, want, start = got[r][:, ::7], want[r][:, ::7], start[r][:, ::7]
            move = float((want - start).norm())
            d = float((got - want).norm())
            # the UPDATE (lr x clipped gradient) against the oracle's; float32 storage of the parameter itself: 6e-8 |p|
            assert d < (5e-3 if amp else 2e-4) * move + 2e-7 * float(want.norm()), (s, k, d, move)
        for tag in ("1", "4"):
            for stat in ("running_mean", "running_var"):
                got = sd[f"backbones.{s}.{tag}.{stat}"]
                want = running[si][("rm" if stat == "running_mean" else "rv") + ("1" if tag == "1" else "2")]
                assert rel(got, want) < (2e-3 if amp else 1e-5), (s, tag, stat, rel(got, want))
