"""Data-parallel exchange logic (neural_svd_amd/parallel.py) on CPU: world_size 2, gloo, the oracle
injected as the compute backend. Checks the identity the N>1 path relies on:
sharding the global batch arranged as [f1_0, f1_1, f2_0, f2_1] over 2 ranks, all-reducing (mean) the
2L^2+1 moment floats and then (sum, scaled 1/world) the gradients reproduces the single-process
loss and gradient of the reference formulation exactly."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import nsvd_oracle as O


class OracleBackend:
    """The compute-backend protocol of parallel.dp_step / hp_step (the one trainer.FusedTrainer implements on HIP
    kernels) on the CPU oracle (test only). dp: owns the full model on its rows; hp: owns heads
    [l_off, l_off + L_loc) on the whole batch."""

    def __init__(self, p, prob, v, M, l_off=0, world=1, n_buckets=3):
        self.p, self.prob, self.v, self.M = p, prob, v.double(), M.double()
        self.l_off, self.world, self.n_buckets = l_off, world, n_buckets
        self.applied, self.calls = None, []

    # -- protocol
    def forward(self, x):
        self.calls.append("forward")
        self.ctx = O.operator_forward(x, self.p, self.prob)
        self.f_loc, self.Tf_loc = self.ctx.f, self.ctx.Tf
        self.f, self.Tf = self.f_loc, self.Tf_loc

    def local_moments(self):
        self.calls.append("local_moments")
        _, lam1, lam2, loss_op, _ = O.evd_loss_forward(self.f, self.Tf, self.v, self.M)
        self.mom = torch.cat([lam1.reshape(-1), lam2.reshape(-1), (loss_op / -2.0).reshape(1)])
        return self.mom

    def backward(self, reduced_moments, take_step):
        self.calls.append(f"backward(reduced={reduced_moments is not None}, take_step={take_step})")
        mom = reduced_moments if reduced_moments is not None else self.local_moments()
        L = self.f.shape[1]
        lam1, lam2 = mom[:L * L].view(L, L), mom[L * L:2 * L * L].view(L, L)
        self.loss = -2.0 * mom[2 * L * L] + (self.M * lam1 * lam2).sum()
        df = O.evd_loss_backward(self.f, self.Tf, self.v, self.M, lam1, lam2)
        Ll = self.ctx.f.shape[1]
        df = df[:, self.l_off:self.l_off + Ll].contiguous()
        self.grad = torch.cat([g.reshape(-1) for g in O.operator_backward(self.ctx, self.p, self.prob, df)])
        self.applied = torch.full_like(self.grad, float("nan"))
        if take_step:
            self.begin_apply()
            self.apply(0, self.grad.numel(), 1.0)

    def grad_buffer(self):
        return self.grad

    def grad_buckets(self):
        n = self.grad.numel()
        # every bucket a multiple of the world size (the reduce-scatter exchange needs equal slices); the tail that
        # does not divide rides... nowhere: _setup() sizes the model so that it does
        q = n // (self.n_buckets * self.world) * self.world
        cuts = [q * i for i in range(self.n_buckets)] + [n]
        return list(zip(cuts[:-1], cuts[1:]))

    def backward_staged(self, reduced_moments):
        """the oracle has no head windows: one backward, then the buckets (the HIP backend's windows are covered by
        tests/test_multirank_gpu.py); what this exercises is dp_step's use of the generator"""
        self.backward(reduced_moments, False)
        self.params = torch.arange(self.grad.numel(), dtype=self.grad.dtype) * 1e-3  # stand-in parameter buffer
        self.params0 = self.params.clone()
        for b in self.grad_buckets():
            self.calls.append("bucket")
            yield b

    def grad_shard(self, lo, hi):
        return torch.empty((hi - lo) // self.world, dtype=self.grad.dtype)

    def apply_shard(self, lo, hi, g, scale):
        self.calls.append("apply_shard")
        self.applied[lo:hi] = g * scale
        self.params[lo:hi] -= 0.5 * g * scale  # a step only the owner of [lo, hi) takes

    def param_buffer(self):
        return self.params

    def a2a_buffers(self, lo, hi):
        q = (hi - lo) // self.world
        return torch.empty((self.world, q), dtype=self.grad.dtype), torch.empty((self.world, q), dtype=self.grad.dtype)

    def sum_slices(self, recv, out):
        self.calls.append("sum_slices")
        torch.sum(recv, dim=0, out=out)

    def begin_apply(self):
        self.calls.append("begin_apply")

    def apply(self, lo, hi, scale):
        self.calls.append("apply")
        self.applied[lo:hi] = self.grad[lo:hi] * scale

    def gather_buffers(self):
        """the protocol's block layout (parallel.py): 2 B head_block(L, world) values per rank, beginning with this
        rank's packed f (B, n_r) | Tf (B, n_r)"""
        from neural_svd_amd.parallel import head_block
        B, Ll = self.f_loc.shape
        n = 2 * B * head_block(self.v.numel(), self.world)
        self.gath = torch.empty((self.world, n), dtype=self.f_loc.dtype)
        blk = torch.full((n,), float("nan"), dtype=self.f_loc.dtype)  # (a short rank's tail must never be read)
        blk[:2 * B * Ll] = torch.stack([self.f_loc, self.Tf_loc]).reshape(-1)
        return self.gath, blk

    def after_gather(self):
        """what nsvd_evd_gather_head_blocks does on the device: rank w's n_w = head_range(L, w, world) columns"""
        from neural_svd_amd.parallel import head_range
        L, B = self.v.numel(), self.f_loc.shape[0]
        f, Tf = [], []
        for w in range(self.world):
            _, n = head_range(L, w, self.world)
            f.append(self.gath[w, :B * n].view(B, n))
            Tf.append(self.gath[w, B * n:2 * B * n].view(B, n))
        self.f, self.Tf = torch.cat(f, dim=1).contiguous(), torch.cat(Tf, dim=1).contiguous()
        assert self.f.shape == (B, L) and bool(torch.isfinite(self.f).all())

    def prefetch(self):
        self.calls.append("prefetch")


def slice_heads(p, lo, hi):
    return O.Params([w[lo:hi] for w in p.ws], [b[lo:hi] for b in p.bs], p.fourier_B,
                    None if p.scales is None else p.scales[lo:hi])


def _setup():
    # 3 * (12*10 + 10 + 10*8 + 8 + 8 + 1) + 3 = 684 gradient elements: divisible by 2, 3, 4 and 6
    L, D, m, hidden, Bg = 3, 2, 6, (10, 8), 16
    # (the eight-rank topology test: 32 heads, 64 rows - through the environment, which the spawned ranks inherit)
    L, Bg = int(os.environ.get("NSVD_TEST_L", L)), int(os.environ.get("NSVD_TEST_B", Bg))
    p = O.init_params(L, D, m, hidden, 0.1, exp_mask_init=10.0, seed=3).to(torch.float64)
    prob = O.Problem(potential=O.POT_HARMONIC, eps=0.01, op_scale=1.0, op_shift=16.0, sigma=4.0)
    v, M = O.joint_nesting_masks(L, 2)
    g = torch.Generator().manual_seed(7)
    x = 4.0 * torch.randn(Bg, D, generator=g, dtype=torch.float64)
    return p, prob, v, M, x


def _worker(rank, world, port, tmp, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from neural_svd_amd import parallel
    comm = parallel.Communicator.from_env(device=None, backend="gloo")
    assert comm.rank == rank and comm.world == world and comm.count_ranks() == world
    p, prob, v, M, x = _setup()
    Bg = x.shape[0]
    h = Bg // 2
    q = h // world
    # global arrangement [f1_0, f1_1, f2_0, f2_1]: rank r owns rows r*q..(r+1)*q of each half
    x_local = torch.cat([x[rank * q:(rank + 1) * q], x[h + rank * q:h + (rank + 1) * q]])
    be = OracleBackend(p, prob, v, M, world=world)
    probe = parallel.CommProbe(None)
    parallel.dp_step(be, comm, x_local, exchange=exchange, probe=probe)
    assert abs(comm.max_float(float(rank)) - (world - 1)) < 1e-12
    comm.barrier()
    # the exchange sequence: moments before the backward, each bucket's collective issued as the backend hands it
    # over and before the prefetch, one schedule advance, then the optimiser bucket by bucket
    head = ["forward", "local_moments", "backward(reduced=True, take_step=False)"] + ["bucket"] * 3 + \
        ["prefetch", "begin_apply"]
    spans = list(probe.summary())
    if exchange == "allreduce":
        assert be.calls == head + ["apply"] * 3, be.calls
        assert spans == ["moments_allreduce"] + [f"grad_bucket{k}_allreduce_wait" for k in range(3)], spans
        extra = {}
    else:
        if exchange == "a2a":
            assert be.calls == head + ["sum_slices", "apply_shard"] * 3, be.calls
            assert spans == ["moments_allreduce"] + [f"grad_bucket{k}_all_to_all_wait" for k in range(3)] + \
                [f"param_bucket{k}_all_to_all_wait" for k in range(3)], spans
        else:
            assert be.calls == head + ["apply_shard"] * 3, be.calls
            assert spans == ["moments_allreduce"] + [f"grad_bucket{k}_reduce_scatter_wait" for k in range(3)] + \
                [f"param_bucket{k}_all_gather_wait" for k in range(3)], spans
        # what this rank applied: its own slice of every bucket, nothing else; the parameters came back complete
        own = torch.zeros(be.grad.numel(), dtype=torch.bool)
        for lo, hi in be.grad_buckets():
            slo, shi = parallel.shard_range(lo, hi, rank, world)
            own[slo:shi] = True
        assert bool(torch.isnan(be.applied[~own]).all()) and not bool(torch.isnan(be.applied[own]).any())
        extra = dict(own=own, params=be.params, params0=be.params0)
    assert probe.steps == 1
    # compute-only mode: the same call sequence with every collective skipped
    comm.stub = True
    be2 = OracleBackend(p, prob, v, M, world=world)
    parallel.dp_step(be2, comm, x_local, exchange=exchange)
    comm.stub = False
    assert be2.calls == be.calls
    t = torch.ones(3, dtype=torch.float64) * (rank + 1)
    comm.broadcast(t, 0)
    assert float(t[0]) == 1.0
    torch.save(dict(loss=be.loss, grad=be.applied, mom=be.mom, **extra), os.path.join(tmp, f"r{rank}.pt"))
    comm.close()


def _worker_hp(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from neural_svd_amd import parallel
    comm = parallel.Communicator.from_env(device=None, backend="gloo")
    p, prob, v, M, x = _setup()
    L = p.ws[0].shape[0]
    l_off, Ll = parallel.head_range(L, rank, world)  # any L >= world: the first L % world ranks own one head more
    be = OracleBackend(slice_heads(p, l_off, l_off + Ll), prob, v, M, l_off=l_off, world=world)
    parallel.hp_step(be, comm, x)
    assert be.calls[:3] == ["forward", "prefetch", "backward(reduced=False, take_step=True)"], be.calls
    assert comm.count_ranks() == world
    torch.save(dict(loss=be.loss, grad=be.applied, f=be.f), os.path.join(tmp, f"h{rank}.pt"))
    comm.close()


def _worker_forced(rank, world, port, tmp):
    """a world of ONE with Communicator.force_exchange: every exchange sequence, asynchronous and blocking, must run
    its collectives (probe spans present) and reproduce the plain single-process step"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from neural_svd_amd import parallel
    comm = parallel.Communicator.from_env(device=None, backend="gloo")
    assert comm.world == 1 and not comm.multi
    comm.force_exchange = True
    assert comm.multi
    p, prob, v, M, x = _setup()
    ref = O.loss_and_grads(x, p, prob, v, M)
    gref = torch.cat([g.reshape(-1) for g in ref["grads"]])
    for exchange in parallel.DP_EXCHANGES:
        for sync in (False, True):
            be = OracleBackend(p, prob, v, M, world=1)
            probe = parallel.CommProbe(None)
            parallel.dp_step(be, comm, x, exchange=exchange, probe=probe, sync=sync)
            spans = list(probe.summary())
            assert spans[0] == "moments_allreduce" and len(spans) == (4 if exchange == "allreduce" else 7), spans
            assert "backward(reduced=True, take_step=False)" in be.calls
            assert abs(float(be.loss) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
            assert float((be.applied - gref).norm() / gref.norm()) < 1e-12, (exchange, sync)
    for sync in (False, True):
        be = OracleBackend(p, prob, v, M, world=1)
        probe = parallel.CommProbe(None)
        parallel.hp_step(be, comm, x, probe=probe, sync=sync)
        assert list(probe.summary()) == ["f_Tf_all_gather_wait"]
        assert ("prefetch" in be.calls) == (not sync)
        assert float((be.applied - gref).norm() / gref.norm()) < 1e-12
    torch.save(dict(ok=True), os.path.join(tmp, "forced.pt"))
    comm.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_dp_two_ranks_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p, prob, v, M, x = _setup()
    ref = O.loss_and_grads(x, p, prob, v, M)
    gref = torch.cat([g.reshape(-1) for g in ref["grads"]])
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]
    for o in outs:
        assert abs(float(o["loss"]) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
        assert float((o["grad"] - gref).norm() / gref.norm()) < 1e-12
        L = 3
        assert torch.allclose(o["mom"][:L * L].view(L, L), ref["lam1"], rtol=1e-13, atol=1e-15)
        assert torch.allclose(o["mom"][L * L:2 * L * L].view(L, L), ref["lam2"], rtol=1e-13, atol=1e-15)
    assert torch.equal(outs[0]["grad"], outs[1]["grad"])  # identical update on every rank


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,exchange", [(2, "rs_ag"), (4, "rs_ag"), (2, "a2a"), (4, "a2a")])
def test_dp_reduce_scatter_all_gather_equals_single_process(tmp_path, world, exchange):
    """exchanges "rs_ag" and "a2a": every rank receives the summed gradient of ITS slice of each bucket (== the
    single-process gradient there), steps only that slice, and the gather leaves every rank with the same, complete
    parameters."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), exchange), nprocs=world, join=True)
    p, prob, v, M, x = _setup()
    ref = O.loss_and_grads(x, p, prob, v, M)
    gref = torch.cat([g.reshape(-1) for g in ref["grads"]])
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]
    covered = torch.zeros_like(outs[0]["own"])
    for o in outs:
        assert abs(float(o["loss"]) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
        own = o["own"]
        assert float((o["grad"][own] - gref[own]).norm() / gref[own].norm()) < 1e-12
        assert not bool((covered & own).any())
        covered |= own
        assert torch.equal(o["params"], outs[0]["params"])                    # complete and identical everywhere
        want = o["params0"] - 0.5 * gref
        assert float((o["params"] - want).norm() / want.norm()) < 1e-12       # every element stepped exactly once
    assert bool(covered.all())


@pytest.mark.timeout(300)
def test_forced_exchange_in_a_world_of_one(tmp_path):
    mp.spawn(_worker_forced, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert torch.load(os.path.join(str(tmp_path), "forced.pt"))["ok"]


def test_dp_step_single_process_is_plain_step():
    from neural_svd_amd import parallel
    p, prob, v, M, x = _setup()
    be = OracleBackend(p, prob, v, M)
    parallel.dp_step(be, None, x)
    ref = O.loss_and_grads(x, p, prob, v, M)
    assert abs(float(be.loss) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
    gref = torch.cat([g.reshape(-1) for g in ref["grads"]])
    assert float((be.applied - gref).norm() / gref.norm()) < 1e-12   # one rank: the step is applied unscaled
    be2 = OracleBackend(p, prob, v, M)
    parallel.hp_step(be2, None, x)                                  # a world of one: same thing
    assert torch.equal(be2.applied, be.applied)


@pytest.mark.timeout(300)
def test_hp_three_ranks_equals_single_process(tmp_path):
    """head-parallel: 3 heads over 3 ranks, each on the whole batch; gathered f, local gradients == the
    corresponding head slices of the single-process gradient."""
    world = 3
    mp.spawn(_worker_hp, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p, prob, v, M, x = _setup()
    ref = O.loss_and_grads(x, p, prob, v, M)
    for r in range(world):
        o = torch.load(os.path.join(str(tmp_path), f"h{r}.pt"))
        assert abs(float(o["loss"]) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
        assert torch.allclose(o["f"], ref["f"], rtol=1e-13, atol=1e-15)
        want = torch.cat([(g[r:r + 1]).reshape(-1) for g in ref["grads"]])
        assert float((o["grad"] - want).norm() / want.norm()) < 1e-12


@pytest.mark.timeout(900)
@pytest.mark.parametrize("L,world", [(36, 8), (55, 8), (5, 3), (7, 4)])
def test_hp_uneven_head_counts(tmp_path, L, world):
    """heads sharded with L not a multiple of the world size - the reference scripts' own head counts (--neigs 36:
    scripts/exps/pde/hydrogen.sh:28, --neigs 55: oscillator.sh:27) on EIGHT ranks (4 x 5 + 4 x 4 heads; 7 x 7 + 6), and
    two small worlds: L // W heads per rank, the first L % W ranks one more (parallel.head_range), equally long
    all-gather blocks whose tails are never read. Every rank ends with the single-process loss, the whole (B, L) f,
    and its own heads' slices of the single-process gradient."""
    from neural_svd_amd.parallel import head_range
    os.environ["NSVD_TEST_L"], os.environ["NSVD_TEST_B"] = str(L), "16"
    try:
        mp.spawn(_worker_hp, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
        p, prob, v, M, x = _setup()
    finally:
        del os.environ["NSVD_TEST_L"], os.environ["NSVD_TEST_B"]
    assert p.ws[0].shape[0] == L
    ref = O.loss_and_grads(x, p, prob, v, M)
    seen = 0
    for r in range(world):
        o = torch.load(os.path.join(str(tmp_path), f"h{r}.pt"))
        lo, n = head_range(L, r, world)
        assert lo == seen and n in (L // world, L // world + 1)
        seen += n
        assert abs(float(o["loss"]) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
        assert torch.allclose(o["f"], ref["f"], rtol=1e-13, atol=1e-15)
        want = torch.cat([(g[lo:lo + n]).reshape(-1) for g in ref["grads"]])
        assert o["grad"].numel() == want.numel()
        assert float((o["grad"] - want).norm() / want.norm()) < 1e-12
    assert seen == L


def test_head_range_partitions_the_heads():
    from neural_svd_amd.parallel import head_block, head_range
    for L, W in [(36, 8), (55, 8), (16, 8), (32, 8), (8, 8), (9, 8), (64, 3), (5, 5)]:
        rs = [head_range(L, r, W) for r in range(W)]
        assert rs[0][0] == 0 and all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(W - 1))
        assert rs[-1][0] + rs[-1][1] == L and max(n for _, n in rs) == head_block(L, W)
        assert max(n for _, n in rs) - min(n for _, n in rs) <= 1
    with pytest.raises(ValueError):
        head_range(7, 0, 8)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("exchange", ["allreduce", "rs_ag", "a2a", "hp"])
def test_world_of_eight_at_the_target_topology(tmp_path, exchange):
    """BASELINE.json configs[2]'s topology - EIGHT ranks, L = 32 heads (4 per rank when heads are sharded; 7296 gradient
    elements in 3 buckets of 8 equal slices), a global batch of 64 rows (8 per rank: the exchange logic, not the
    arithmetic, is under test) - for every gradient exchange of dp_step and for hp_step: every rank ends with the
    single-process result."""
    world = 8
    os.environ["NSVD_TEST_L"], os.environ["NSVD_TEST_B"] = "32", "64"
    try:
        if exchange == "hp":
            mp.spawn(_worker_hp, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
        else:
            mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), exchange), nprocs=world, join=True)
        p, prob, v, M, x = _setup()
    finally:
        del os.environ["NSVD_TEST_L"], os.environ["NSVD_TEST_B"]
    L = p.ws[0].shape[0]
    assert L == 32 and x.shape[0] == 64
    ref = O.loss_and_grads(x, p, prob, v, M)
    gref = torch.cat([g.reshape(-1) for g in ref["grads"]])
    if exchange == "hp":
        Ll = L // world
        for r in range(world):
            o = torch.load(os.path.join(str(tmp_path), f"h{r}.pt"))
            assert abs(float(o["loss"]) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
            assert torch.allclose(o["f"], ref["f"], rtol=1e-13, atol=1e-15)
            want = torch.cat([(g[r * Ll:(r + 1) * Ll]).reshape(-1) for g in ref["grads"]])
            assert float((o["grad"] - want).norm() / want.norm()) < 1e-12
        return
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]
    covered = torch.zeros(gref.numel(), dtype=torch.bool)
    for o in outs:
        assert abs(float(o["loss"]) - float(ref["loss"])) < 1e-12 * abs(float(ref["loss"]))
        assert torch.allclose(o["mom"][:L * L].view(L, L), ref["lam1"], rtol=1e-13, atol=1e-15)
        if exchange == "allreduce":
            assert float((o["grad"] - gref).norm() / gref.norm()) < 1e-12
            assert torch.equal(o["grad"], outs[0]["grad"])
        else:
            own = o["own"]
            assert float((o["grad"][own] - gref[own]).norm() / gref[own].norm()) < 1e-12
            assert not bool((covered & own).any())
            covered |= own
            assert torch.equal(o["params"], outs[0]["params"])
            want = o["params0"] - 0.5 * gref
            assert float((o["params"] - want).norm() / want.norm()) < 1e-12
    assert exchange == "allreduce" or bool(covered.all())
