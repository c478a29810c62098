"""HIP-graph capture of the training step ("graph-replayed steps"; the loop body being
replayed is examples/operator/__init__.py:55-74): the per-step scalars - CosineAnnealingLR's learning rate, torch_ema's
warmed-up decay, the sampler's batch counter - live in a device-resident nsvd_step_state, so a captured step replays
ALONG the schedule. Every comparison here is bit for bit."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _trainer(L, m, B, device_schedule, num_iters=60, seed=4, potential="hydrogen", **kw):
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import FusedTrainer
    if potential == "hydrogen":
        shape = H.ModelShape(L=L, D=2, m=m, hidden=(128, 128, 128))
        prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
        extra = dict(sampling_scale=16.0, fourier_scale=0.1)
    else:
        shape = H.ModelShape(L=L, D=2, m=m, hidden=(128, 128, 128), has_exp_mask=True)
        prob = H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0)
        extra = dict(sampling_scale=4.0, fourier_scale=1.0, exp_mask_init=10.0)
    return FusedTrainer(shape, prob, B, sequential=False, lr=1e-3, num_iters=num_iters, seed=seed, device=DEV,
                        device_schedule=device_schedule, **extra, **kw)


def _same(a, b):
    for name in ("flat", "sq", "ema"):
        assert torch.equal(getattr(a.P, name), getattr(b.P, name)), name
    assert torch.equal(a.x, b.x) and torch.equal(a.f, b.f) and torch.equal(a.Tf, b.Tf)
    assert torch.equal(a.loss, b.loss)


def test_device_schedule_values_match_the_host_schedule():
    """state.cur after nsvd_step_state_init(step = t) against the host expressions of FusedTrainer._advance_schedule /
    trainer.cosine_lr rounded where nsvd_make_hyper rounds: CosineAnnealingLR's closed form and torch_ema's warm-up
    (examples/operator/__init__.py:35-36,71-73). The device cosine may differ from libm's in the last bit of the
    double: at most one float32 ulp on the learning rate, and rarely (asserted: < 1 % of the sampled steps)."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import cosine_lr
    T, lr0, decay = 500000, 1e-4, 0.995
    st = H.StepState(DEV, lr0, T, 0.999, 1e-10, decay)
    rng = np.random.default_rng(0)
    ts = [0, 1, 2, 8, 9, 10, 1789, 1790, 1791, T // 2, T - 1, T] + [int(t) for t in rng.integers(0, T, 300)]
    off = 0
    for t in ts:
        st.reset(t)
        h = st.read()
        assert h.step == t and h.T_max == T
        want_lr = np.float32(cosine_lr(lr0, t, T))
        n = t + 1
        want_omd = np.float32(1.0 - min(decay, (1 + n) / (10 + n)))
        got_lr = np.float32(h.cur.lr)
        assert np.float32(h.cur.one_minus_decay) == want_omd, t
        assert np.float32(h.cur.alpha) == np.float32(0.999) and np.float32(h.cur.one_minus_alpha) == np.float32(1.0 - 0.999)
        assert np.float32(h.cur.eps) == np.float32(1e-10) and h.cur.grad_scale == 1.0
        if got_lr != want_lr:
            off += 1
            assert abs(float(got_lr) - float(want_lr)) <= float(np.spacing(want_lr)), (t, got_lr, want_lr)
    assert off <= 0.01 * len(ts), off
    # no scheduler: constant learning rate
    st2 = H.StepState(DEV, 3e-4, 0, 0.99, 1e-8, 0.9, step=12345)
    h = st2.read()
    assert np.float32(h.cur.lr) == np.float32(3e-4) and np.float32(h.cur.one_minus_decay) == np.float32(1.0 - 0.9)


@pytest.mark.parametrize("L,m,B,potential", [(4, 64, 64, "hydrogen"), (8, 128, 128, "oscillator")])
def test_device_schedule_step_is_the_host_schedule_step(L, m, B, potential):
    """the same trainer with the schedule on the device (eager launches, no graph) against the plain one: identical
    batches, parameters, square averages, EMA shadows and losses after 40 steps of a 60-step cosine schedule - and the
    device counter has advanced by itself."""
    a = _trainer(L, m, B, False, potential=potential)
    b = _trainer(L, m, B, True, potential=potential)
    assert b.state is not None and a.state is None and a.guest_features and b.guest_features
    for _ in range(40):
        a.step()
        b.step()
    torch.cuda.synchronize()
    _same(a, b)
    h = b.state.read()
    assert h.step == 40 == b.t == a.t
    # without guest workgroups (one feature launch per step, its batch counter read on the device)
    c = _trainer(L, m, B, True, potential=potential, overlap=False)
    assert not c.guest_features
    for _ in range(40):
        c.step()
    torch.cuda.synchronize()
    _same(a, c)


@pytest.mark.parametrize("L,m,B,steps", [(4, 64, 64, 2), (4, 64, 64, 4), (16, 1024, 512, 2)])
def test_graph_replay_is_bit_identical_to_eager_steps(L, m, B, steps):
    """torch.cuda.CUDAGraph capture of `steps` FusedTrainer.step() calls, replayed 25 times, against the same number
    of eager steps (host schedule AND device schedule): every buffer bit-identical, the schedule has moved (the last
    learning rate differs from the first), the host counters follow. The last case is configs[1]'s shape."""
    from neural_svd_amd.trainer import cosine_lr
    n_replay = 25
    g = _trainer(L, m, B, True)
    gs = g.capture_graph(steps)
    pre = g.t  # one eager step was taken to prepare the first batch
    assert pre == 1
    gs.replay(n_replay)
    torch.cuda.synchronize()
    total = pre + n_replay * steps
    assert g.t == g.num_updates == total and g.state.read().step == total
    e_host = _trainer(L, m, B, False)
    e_dev = _trainer(L, m, B, True)
    for _ in range(total):
        e_host.step()
        e_dev.step()
    torch.cuda.synchronize()
    _same(e_dev, g)
    _same(e_host, g)
    # the schedule moved inside the graph: the values of the LAST step taken
    h = g.state.read()
    assert np.float32(h.cur.lr) == np.float32(cosine_lr(1e-3, total - 1, 60)) != np.float32(1e-3)
    # eager steps continue seamlessly after a replay, and a replay after them
    g.step(); e_host.step()
    g.step(); e_host.step()
    gs.replay()
    for _ in range(steps):
        e_host.step()
    torch.cuda.synchronize()
    _same(e_host, g)


def test_graph_of_the_plain_optimiser_launch():
    """the capturable form of optimizer.step() + scheduler.step() + ema.update() for loop bodies that keep their own
    backward (examples/operator/__init__.py:69-73): nsvd_step_state_begin + nsvd_rmsprop_ema_step_dev captured once,
    replayed along the schedule, against nsvd_rmsprop_ema_step with host-scheduled scalars."""
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd.trainer import cosine_lr
    n, T = 100003, 30
    gen = torch.Generator(device=DEV).manual_seed(1)
    p0 = torch.randn(n, device=DEV, generator=gen)
    grads = [torch.randn(n, device=DEV, generator=gen) for _ in range(12)]
    pa, sqa, ema_a = p0.clone(), torch.zeros_like(p0), p0.clone()
    pb, sqb, ema_b = p0.clone(), torch.zeros_like(p0), p0.clone()
    for t, g in enumerate(grads):
        nup = t + 1
        H.rmsprop_ema_step(pa, g, sqa, ema_a, cosine_lr(1e-3, t, T), 0.999, 1e-10, min(0.995, (1 + nup) / (10 + nup)))
    st = H.StepState(DEV, 1e-3, T, 0.999, 1e-10, 0.995)
    gbuf = torch.empty_like(p0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            st.begin()
            H.rmsprop_ema_step_dev(pb, gbuf, sqb, ema_b, st, 1.0, True)
    torch.cuda.current_stream().wait_stream(side)
    for g in grads:
        gbuf.copy_(g)
        graph.replay()
    torch.cuda.synchronize()
    assert st.read().step == len(grads)
    assert torch.equal(pa, pb) and torch.equal(sqa, sqb) and torch.equal(ema_a, ema_b)


def test_loss_value_of_every_step_comes_from_the_step_s_own_kernels():
    """single GPU, batch <= 1024 rows: the backward takes its moments straight from f (no moment kernel) and ALSO
    leaves the loss value - the reference computes it every step (methods/nestedlora.py:92-94) - as per-head partial
    sums the weight-gradient kernel adds; against the stand-alone loss kernels on the same f, Tf and the float64
    oracle formula."""
    from neural_svd_amd import hip_ops as H
    from oracle import nsvd_oracle as O
    for L, m, B, seq in ((16, 1024, 512, False), (16, 256, 1024, True), (4, 64, 96, False), (32, 128, 512, True)):
        from neural_svd_amd.trainer import FusedTrainer
        shape = H.ModelShape(L=L, D=2, m=m, hidden=(128, 128, 128))
        prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
        tr = FusedTrainer(shape, prob, B, sequential=seq, seed=1, device=DEV)
        assert tr.direct_moments and tr._direct_loss
        for _ in range(3):
            tr.step()
        got = tr._loss.clone()
        assert not tr._loss_stale
        ref = torch.empty(3, device=DEV)
        mom = torch.empty(2 * L * L + 1, device=DEV)
        H.evd_loss_fused(tr.f, tr.Tf, tr.mask_kind, None, None, mom, ref, None, tr.scratch)
        v, M = (O.sequential_nesting_masks(L) if seq else O.joint_nesting_masks(L, 1))
        l64, *_ = O.evd_loss_forward(tr.f.double().cpu(), tr.Tf.double().cpu(), v.double(), M.double())
        torch.cuda.synchronize()
        scale = float(ref[1:].abs().max())
        assert float((got - ref).abs().max()) < 2e-6 * scale, (L, B, got, ref)
        assert abs(float(got[0]) - float(l64)) < 2e-5 * scale
        assert torch.equal(tr.moments, mom)  # on demand, and the reported loss stays the step's own
        assert torch.equal(tr.loss, got)


def test_torch_binding_runs_the_same_step_bit_for_bit():
    """NSVD_BINDING=torch: the hot-path calls go through the tensor-level torch extension (csrc/torch_binding.cpp:
    tensors checked and the current stream taken in C++) instead of ctypes - the same C ABI underneath, so the same bits:
    five trainer steps, the stand-alone loss pieces, the optimiser, and the refusal of a non-contiguous tensor."""
    import os
    from neural_svd_amd import hip_ops as H
    old = os.environ.get("NSVD_BINDING")
    try:
        res = {}
        for binding in ("ctypes", "torch"):
            os.environ["NSVD_BINDING"] = binding
            assert (H.torch_binding() is not None) == (binding == "torch")
            tr = _trainer(4, 64, 64, False)
            for _ in range(5):
                tr.step()
            mom = H.evd_moments(tr.f, tr.Tf, tr.mask_kind, None)
            loss, df = H.evd_loss_grad(tr.f, tr.Tf, tr.mask_kind, None, None, mom)
            torch.cuda.synchronize()
            res[binding] = (tr.P.flat.clone(), tr.P.ema.clone(), tr.f.clone(), tr.Tf.clone(), mom, loss, df)
        for a, b in zip(res["ctypes"], res["torch"]):
            assert torch.equal(a, b)
        os.environ["NSVD_BINDING"] = "torch"
        f = torch.zeros(8, 4, device=DEV)
        with pytest.raises(RuntimeError, match="contiguous"):
            H.evd_moments(f.t(), f.t(), H.MASK_SEQUENTIAL, None, torch.zeros(2 * 8 * 8 + 1, device=DEV))
    finally:
        if old is None:
            os.environ.pop("NSVD_BINDING", None)
        else:
            os.environ["NSVD_BINDING"] = old


@pytest.mark.parametrize("L,m,B,potential,overlap", [(4, 64, 64, "hydrogen", True), (8, 128, 128, "oscillator", True),
                                                     (4, 64, 96, "hydrogen", False), (16, 1024, 512, "hydrogen", True)])
def test_bf16x3_step_leaves_the_planes_of_the_updated_weights(L, m, B, potential, overlap):
    """NSVD_PATH_FUSED_BF16X3: the fused step's weight-gradient epilogue writes the three bf16 planes of every updated
    element of W_0 / W_1 .. where the NEXT forward reads them (the other workspace set with guest features, the same one
    without), and that forward skips its split launch (NSVD_W_PLANES_READY). Same rounding, same bits: 30 steps against a
    trainer whose forwards always split (and against one whose parameters are reloaded half way: the planes are tied to
    FlatParams.version), eager and replayed from a HIP graph; the last case is configs[1]'s shape."""
    from neural_svd_amd import hip_ops as H
    # (B = 96: also with the gradients stored beside the fused step - the weight-gradient kernel's plain epilogue)
    kw = dict(keep_grads=True) if B == 96 else {}
    a = _trainer(L, m, B, False, potential=potential, overlap=overlap, **kw)
    b = _trainer(L, m, B, False, potential=potential, overlap=overlap, **kw)
    a.path = b.path = H.PATH_FUSED_BF16X3
    assert H.step_emits_planes(a.shape, B, a.path) and not H.step_emits_planes(a.shape, B, H.PATH_AUTO)
    b._note_planes = lambda ws: None  # never claims the planes: every forward of b splits the weights itself
    used = 0
    for t in range(30):
        a.step()
        b.step()
        used += a._planes_ws is not None
        if t == 14:  # a reload between two steps: the planes in the workspace are stale, the version says so
            sd = a.P.state_dict()
            a.P.load_state_dict(sd, reset_optimizer=False)
            b.P.load_state_dict(sd, reset_optimizer=False)
            assert a._planes_version != a.P.version
    torch.cuda.synchronize()
    assert used == 30 and b._planes_ws is None
    _same(a, b)
    # replayed from a graph (device schedule): the captured forwards carry the flag
    g = _trainer(L, m, B, True, potential=potential, overlap=overlap)
    e = _trainer(L, m, B, True, potential=potential, overlap=overlap)
    g.path = e.path = H.PATH_FUSED_BF16X3
    e._note_planes = lambda ws: None
    if overlap:
        gs = g.capture_graph(2)
        pre = g.t
        gs.replay(10)
        for _ in range(pre + 20):
            e.step()
        torch.cuda.synchronize()
        _same(e, g)


def test_graph_replay_refuses_weights_changed_behind_its_back():
    """A captured step froze host-side decisions (bf16x3: whether the forward may read the planes of the weights): after
    P.load / load_state_dict (P.version moves) or a host-side edit of the counters, replay() must refuse, not run on."""
    from neural_svd_amd import hip_ops as H
    g = _trainer(4, 64, 64, True)
    gs = g.capture_graph(2)
    gs.replay(2)
    sd = g.P.state_dict()
    g.P.load_state_dict(sd, reset_optimizer=False)
    with pytest.raises(H.NsvdError, match="capture the steps again"):
        gs.replay()
    g2 = _trainer(4, 64, 64, True)
    gs2 = g2.capture_graph(2)
    gs2.replay()
    g2.num_updates = 0
    with pytest.raises(H.NsvdError, match="counters"):
        gs2.replay()
    gs3 = g2.capture_graph(2)  # a fresh capture takes the new offsets
    gs3.replay()
    torch.cuda.synchronize()


@pytest.mark.parametrize("L,m,B,potential", [(4, 64, 64, "hydrogen"), (16, 1024, 512, "hydrogen"), (8, 256, 256, "oscillator")])
def test_two_window_backward_is_bit_identical(L, m, B, potential):
    """FusedTrainer(backward_windows=2): ONE fused step whose backward runs as two head windows on two streams
    (nsvd_operator_backward_evd_step_window: window 0 hosts the next batch's guest workgroups, window 1 advances the
    device-resident schedule and adds up the loss) - every buffer and the loss bit-identical to the one-call step,
    eagerly (host and device schedule) and replayed from a HIP graph. (Measured slower than one window at configs[1]:
    scripts/experiments/README.md - kept as an option, not a default.)"""
    a = _trainer(L, m, B, False, potential=potential)
    for dsched in (False, True):
        b = _trainer(L, m, B, dsched, potential=potential)
        b.backward_windows = 2
        a2 = _trainer(L, m, B, False, potential=potential)
        for _ in range(12):
            a2.step()
            b.step()
        torch.cuda.synchronize()
        assert b.backward_windows == 2 and b._side_stream is not None
        _same(a2, b)
        assert torch.equal(a2.loss, b.loss)
    g = _trainer(L, m, B, True, potential=potential)
    g.backward_windows = 2
    gs = g.capture_graph(2)
    gs.replay(5)
    for _ in range(g.t):
        a.step()
    torch.cuda.synchronize()
    _same(a, g)
