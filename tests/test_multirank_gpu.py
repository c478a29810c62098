"""The multi-rank PRODUCT path on the GPU: trainer.FusedTrainer as the HIP backend of parallel.dp_step / hp_step,
two ranks on one device over gloo. What must hold:
  dp  sharding the global batch arranged [f1_0, f1_1, f2_0, f2_1] over 2 ranks, averaging the 2 L^2 + 1 moments and
      summing the bucketed gradients reproduces the single-process step on the global batch (float32 summation
      order aside), and both replicas end bit-identical;
  hp  each rank's L/2 heads on the whole batch + one all-gather of f, Tf reproduces the single-process step's head
      slices;
  overlap  preparing the next batch under the collective changes nothing, bit for bit; unseeded replicas start from
      rank 0's weights.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_multirank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(mode, world, out_dir, backend="gloo"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NSVD_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   NSVD_DIST_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(out_dir)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-4000:]}"
    return [torch.load(os.path.join(str(out_dir), f"{mode}_r{r}.pt"), weights_only=False) for r in range(world)]


def single_process(world, b_local=None, L=4):
    """the same steps by ONE trainer on the global batches (separate optimiser kernel so that the gradient is kept)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _multirank_worker as W
    W.CASE["B_local"] = b_local or 64  # (module state: set on every call)
    W.CASE["L"] = L
    from neural_svd_amd.trainer import FusedTrainer
    dev = torch.device("cuda:0")
    tr = FusedTrainer(W.make_shape(), W.make_problem(), W.CASE["B_local"] * world, seed=5, device=dev,
                      keep_grads=True, **W.trainer_kw())
    out = {}
    for i, xg in enumerate(W.global_batches(world)):
        tr.step(xg.to(dev))
        if i == 0:
            out["grad0"], out["loss0"], out["mom0"] = tr.P.grad.clone().cpu(), tr.loss.clone().cpu(), \
                tr.moments.clone().cpu()
    torch.cuda.synchronize()
    out.update(flat=tr.P.flat.cpu(), ema=tr.P.ema.cpu(), sq=tr.P.sq.cpu())
    return out


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@pytest.mark.timeout(900)
def test_dp_two_ranks_match_single_process(tmp_path):
    world = 2
    rs = run_ranks("dp", world, tmp_path)
    ref = single_process(world)
    for r in rs:
        assert r["t"] == 3 and not r["fused_step"] and len(r["buckets"]) == 3
        assert r["buckets"][0][0] == 0 and r["buckets"][-1][1] == ref["flat"].numel()
        # exchange 1: the averaged moments ARE the global batch's moments; loss of the global batch
        assert rel(r["mom0"], ref["mom0"]) < 2e-6
        assert rel(r["loss0"], ref["loss0"]) < 2e-6
        # exchange 2: summed bucket by bucket, scaled 1/world in the optimiser = the global batch's gradient
        assert rel(r["grad0"] / world, ref["grad0"]) < 1e-5
        assert rel(r["sq"], ref["sq"]) < 1e-4
    # identical replicas
    for name in ("flat", "ema", "sq", "grad0"):
        assert torch.equal(rs[0][name], rs[1][name]), name
    # net parameter movement against the single process: the few elements whose gradient is at the float32 noise
    # floor may step the other way (RMSprop's first steps are +-lr/sqrt(1-alpha) whatever |g| is)
    from neural_svd_amd.trainer import FusedTrainer
    import _multirank_worker as W
    tr0 = FusedTrainer(W.make_shape(), W.make_problem(), 64, seed=5, device="cuda:0", **W.trainer_kw())
    p0 = tr0.P.flat.cpu()
    upd, upd_ref = rs[0]["flat"] - p0, ref["flat"] - p0
    assert float(upd_ref.norm()) > 0
    assert rel(upd, upd_ref) < 2e-2
    assert float(((upd - upd_ref).abs() > 1e-4).double().mean()) < 1e-3


@pytest.mark.timeout(1200)
def test_dp_head_windows_and_reduce_scatter_match_plain_dp(tmp_path):
    """The backward cut into two head windows (bucket k on the wire while window k + 1 is computed) gives the plain dp
    run's gradients and parameters BIT FOR BIT (same tiles, same summation order; with two ranks a + b is
    commutative); so do the reduce-scatter / sharded-optimiser / all-gather exchange and its all-to-all spelling once
    their sharded RMSprop and EMA state has been gathered."""
    world = 2
    plain = run_ranks("dp", world, tmp_path)
    win = run_ranks("dp_win", world, tmp_path)
    rsag = run_ranks("dp_rsag", world, tmp_path)
    a2a = run_ranks("dp_a2a", world, tmp_path)
    for r in range(world):
        assert not plain[r]["sharded0"] and not win[r]["sharded0"] and rsag[r]["sharded0"] and a2a[r]["sharded0"]
        assert len(win[r]["buckets"]) == 3 and win[r]["buckets"][0][0] == 0
        for name in ("mom0", "loss0", "flat", "ema", "sq"):
            assert torch.equal(win[r][name], plain[r][name]), ("windows", name)
            assert torch.equal(rsag[r][name], plain[r][name]), ("rs_ag", name)
            assert torch.equal(a2a[r][name], plain[r][name]), ("a2a", name)
        assert torch.equal(win[r]["grad0"], plain[r]["grad0"])  # the all-reduced gradient itself
    for name in ("flat", "ema", "sq"):
        assert torch.equal(rsag[0][name], rsag[1][name]), name
        assert torch.equal(a2a[0][name], a2a[1][name]), name


def _views(flat, L):
    """per-tensor views of a flat buffer of an L-head model (the layout rule of trainer.FlatParams)"""
    import math
    import _multirank_worker as W
    from neural_svd_amd import hip_ops as H
    shapes = H.ModelShape(L=L, D=W.CASE["D"], m=W.CASE["m"], hidden=W.CASE["hidden"], has_exp_mask=True).param_shapes()
    views, off = [], 0
    for s in shapes:
        n = math.prod(s)
        views.append(flat[off:off + n].view(s))
        off += (n + 63) // 64 * 64
    return views


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,big", [(2, False), (4, False), (2, True)])
def test_hp_ranks_match_single_process(tmp_path, world, big):
    """world 4: ONE head per rank on four times the local batch - the extreme of the head split; big: a global batch of
    1280 rows, beyond the 1024 up to which the backward takes its moments from f itself - the partial-sum form with a
    head offset, which is what every rank of an 8-GPU run of configs[1] (4096 rows) executes"""
    rs = run_ranks("hp_big" if big else "hp", world, tmp_path)
    ref = single_process(world, 640 if big else 64)
    import _multirank_worker as W
    L = W.CASE["L"]
    Ll = L // world
    # FusedTrainer.state_dict(): the WHOLE model in the reference's layout, the same on every rank
    for k in rs[0]["sd"]:
        assert all(torch.equal(r["sd"][k], rs[0]["sd"][k]) and torch.equal(r["sd_ema"][k], rs[0]["sd_ema"][k])
                   for r in rs), k
        if k.endswith("_B"):
            continue
        assert rs[0]["sd"][k].shape[0] == L
        for rank, r in enumerate(rs):
            i = [n for n in rs[0]["sd"] if not n.endswith("_B")].index(k)
            assert torch.equal(rs[0]["sd"][k][rank * Ll:(rank + 1) * Ll], _views(r["flat"], Ll)[i]), k
    for rank, r in enumerate(rs):
        assert r["t"] == 3 and r["fused_step"] and r["l_off"] == rank * Ll
        assert rel(r["loss0"], ref["loss0"]) < 2e-6 and rel(r["mom0"], ref["mom0"]) < 2e-6
        # this rank's tensors are the head slices [l_off, l_off + Ll) of the single-process tensors
        for name, tol in (("grad0", 1e-5), ("sq", 1e-4)):
            want = torch.cat([t[rank * Ll:(rank + 1) * Ll].reshape(-1) for t in _views(ref[name], L)])
            got = torch.cat([t.reshape(-1) for t in _views(r[name], Ll)])
            assert rel(got, want) < tol, (name, rel(got, want))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,L", [(2, 5), (4, 7), (3, 4)])
def test_hp_with_heads_that_do_not_divide_over_the_ranks(tmp_path, world, L):
    """heads sharded for ANY L >= world (the reference scripts run --neigs 36 / 55: scripts/exps/pde/hydrogen.sh:28,
    oscillator.sh:27): L // world heads per rank, the first L % world ranks one more (parallel.head_range) - 3 + 2,
    2 + 2 + 2 + 1 and 2 + 1 + 1 heads here; equally long all-gather blocks unpacked by nsvd_evd_gather_head_blocks.
    Every rank's tensors are its head slices of the single-process run's, and state_dict() is the whole model."""
    from neural_svd_amd.parallel import head_range
    rs = run_ranks(f"hp_L{L}", world, tmp_path)
    ref = single_process(world, 64, L)
    names = [n for n in rs[0]["sd"] if not n.endswith("_B")]
    for k in rs[0]["sd"]:
        assert all(torch.equal(r["sd"][k], rs[0]["sd"][k]) and torch.equal(r["sd_ema"][k], rs[0]["sd_ema"][k])
                   for r in rs), k
        if not k.endswith("_B"):
            assert rs[0]["sd"][k].shape[0] == L
    for rank, r in enumerate(rs):
        lo, n = head_range(L, rank, world)
        assert r["t"] == 3 and r["fused_step"] and r["l_off"] == lo
        assert rel(r["loss0"], ref["loss0"]) < 2e-6 and rel(r["mom0"], ref["mom0"]) < 2e-6
        for i, k in enumerate(names):
            assert torch.equal(rs[0]["sd"][k][lo:lo + n], _views(r["flat"], n)[i]), k
        for name, tol in (("grad0", 1e-5), ("sq", 1e-4), ("flat", 1e-5)):
            want = torch.cat([t[lo:lo + n].reshape(-1) for t in _views(ref[name], L)])
            got = torch.cat([t.reshape(-1) for t in _views(r[name], n)])
            assert rel(got, want) < tol, (name, rank, rel(got, want))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["dp_overlap", "hp_overlap"])
def test_overlapped_prefetch_is_bit_identical_and_replicas_agree(tmp_path, mode):
    world = 2
    rs = run_ranks(mode, world, tmp_path)
    for r in rs:
        ov, plain, blocking = r["runs"]
        for k in ("init", "flat", "ema", "fB", "loss"):
            assert torch.equal(ov[k], plain[k]) and torch.equal(blocking[k], plain[k]), k
        assert torch.equal(ov["x"], plain["x"])
        assert ov["drawn"] == plain["drawn"] + 1 == 7  # one batch prepared ahead
        assert bool(torch.isfinite(ov["flat"]).all())
    a, b = rs[0]["runs"][0], rs[1]["runs"][0]
    assert torch.equal(a["fB"], b["fB"])              # the frozen Fourier matrix is shared in both shardings
    if mode == "dp_overlap":
        assert torch.equal(a["init"], b["init"]) and torch.equal(a["flat"], b["flat"]) and torch.equal(a["ema"], b["ema"])
        assert not torch.equal(a["x"], b["x"])        # every rank its own rows
    else:
        assert torch.equal(a["x"], b["x"])            # every rank the same global batch


def test_every_exchange_on_rccl_in_a_world_of_one(tmp_path):
    """The box has one GPU and RCCL refuses two ranks per device, so the two-rank tests above run over gloo. This one
    runs the SAME exchange sequences on the real library: one rank, backend "nccl", Communicator.force_exchange - every
    collective of dp (all-reduce buckets with and without head windows, reduce-scatter / all-gather, all-to-all;
    asynchronous with late waits, and blocking) and of hp (all-gather of f, Tf) is an RCCL call, ordered against the
    HIP kernels by the work handles / the stream. With one rank every sum has one term and the 1/world scale is 1, so:
    every dp variant must give the SAME BITS as every other (a missing stream dependency would show up right here),
    and those agree with the plain single-GPU trainer up to the summation order of the moments (the plain step takes
    them inside the backward kernel, dp from the moment kernel's vector); hp reproduces the plain trainer's parameters, square averages and EMA bit for bit
    (and its loss value up to the order of a float32 sum: the plain step sums it per 32-row block in its own kernels)."""
    r, = run_ranks("rccl1", 1, tmp_path, backend="nccl")
    assert r["rccl_ranks"] == 1
    plain = r["plain"]
    assert not plain["multi"] and plain["fused_step"]
    dp = ("allreduce", "allreduce_windows", "rs_ag", "a2a", "allreduce_blocking", "rs_ag_blocking")
    for name in dp + ("hp", "hp_blocking"):
        v = r[name]
        hp = name.startswith("hp")
        assert v["multi"] and v["hp"] == hp and v["fused_step"] == hp, name
        assert v["windows"] == (1 if name in ("allreduce", "allreduce_blocking") or hp else 2), name
        for k in ("flat", "ema", "sq", "loss"):
            if hp and k == "loss":
                # the plain step sums the loss in its own kernels (per 32-row block), hp from the gathered moments:
                # the same number up to the order of a float32 sum
                assert torch.equal(v[k], r["hp"][k]) and rel(v[k], plain[k]) < 1e-6, (name, k, v[k], plain[k])
            elif hp:
                assert torch.equal(v[k], plain[k]), (name, k)
            else:
                assert torch.equal(v[k], r["allreduce"][k]), (name, k)
                assert rel(v[k], plain[k]) < (1e-3 if k == "sq" else 1e-4), (name, k, rel(v[k], plain[k]))
    w = r["allreduce"]["waits"]
    assert "moments_allreduce" in w and sum(k.endswith("_allreduce_wait") for k in w) == 3
    assert set(r["allreduce_blocking"]["waits"]) == set(w)
    w = r["rs_ag"]["waits"]
    assert sum(k.endswith("_reduce_scatter_wait") for k in w) == 3 and sum(k.endswith("_all_gather_wait") for k in w) == 3
    w = r["a2a"]["waits"]
    assert sum(k.endswith("_all_to_all_wait") for k in w) == 6
    assert list(r["hp"]["waits"]) == ["f_Tf_all_gather_wait"]
    # the internal device sampler: next batch prepared under the collectives (dp, hp) / riding in the backward (blocking)
    for name in ("dp_internal", "hp_internal", "hp_internal_blocking"):
        assert r[name]["overlap"] == (not name.endswith("blocking")) and r[name]["multi"]
        for k in ("flat", "ema", "sq"):
            if name.startswith("hp"):
                assert torch.equal(r[name][k], r["plain_internal"][k]), (name, k)
            else:
                assert rel(r[name][k], r["plain_internal"][k]) < (1e-3 if k == "sq" else 1e-4), (name, k)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_kernel_operator_step_with_heads_sharded_matches_single_process(tmp_path, world):
    """BASELINE configs[3]'s step (kernel_ops.FusedKernelTrainer) with the heads sharded over `world` ranks: each rank
    evaluates its L / world heads and applies K to its own columns of f on the whole batch, one all-gather of [f | Kf];
    after three steps every rank's weights are the head slices of the single-process trainer's."""
    rs = run_ranks("ko_hp", world, tmp_path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _multirank_worker as W
    from neural_svd_amd.kernel_ops import FusedKernelTrainer
    dev = torch.device("cuda:0")
    op, kw, batches = W.ko_case(dev, world)
    fk = FusedKernelTrainer(op, **kw)
    ref = {}
    for i, idx in enumerate(batches):
        loss = fk.step(idx.to(dev))
        if i == 0:
            ref.update(loss0=loss.clone().cpu(), mom0=fk.moments.clone().cpu(), f0=fk.f.clone().cpu(),
                       Kf0=fk.Kf.clone().cpu())
    views, sq = [v.cpu() for v in fk.P.views(fk.P.flat)], [v.cpu() for v in fk.P.views(fk.P.sq)]
    Ll = kw["L"] // world
    for rank, r in enumerate(rs):
        assert r["t"] == 3 and r["l_off"] == rank * Ll
        # the gathered (B, L) arrays are the single-process ones: same kernels on the same columns
        assert torch.equal(r["f0"], ref["f0"]) and rel(r["Kf0"], ref["Kf0"]) < 1e-6
        assert rel(r["mom0"], ref["mom0"]) < 2e-6 and rel(r["loss0"], ref["loss0"]) < 2e-6
        for got, want, s_got, s_want in zip(r["views"], views, r["sq"], sq):
            sl = slice(rank * Ll, (rank + 1) * Ll)
            assert rel(got, want[sl]) < 1e-5 and rel(s_got, s_want[sl]) < 1e-4


@pytest.mark.timeout(900)
def test_reference_style_loop_on_two_ranks(tmp_path):
    """drop_in.train_operator - the reference's train_operator signature - started on two ranks by a launcher (RANK /
    WORLD_SIZE in the environment, nothing else changed): heads sharded, every rank steps on the global batch of
    2 x batch_size rows (the sampler called twice per step) and ends with the WHOLE model in `method`, equal on both
    ranks and equal to one process training on 64-row batches made of the same blocks. Samples sharded
    (args.parallelism = "dp"): replicas identical, finite, and moved."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _multirank_worker as W
    import neural_svd_amd.drop_in as DI
    hp = run_ranks("dropin_hp", 2, tmp_path)
    dev = torch.device("cuda:0")
    args, operator, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = W.dropin_case(64, dev)
    init = {k: v.detach().cpu().clone() for k, v in method.state_dict().items()}
    blocks = iter(W.dropin_blocks())
    eig, _ = DI.train_operator(args, method, operator, lambda: torch.cat([next(blocks), next(blocks)]), val_data,
                               batch_ftn_val, None, None, dev, imp_train, imp_val)
    ref = {k: v.detach().cpu() for k, v in method.state_dict().items()}
    moved = 0.0
    for k in ref:
        assert torch.equal(hp[0]["sd"][k], hp[1]["sd"][k]), k
        d = float((hp[0]["sd"][k].double() - ref[k].double()).norm())
        step = float((ref[k].double() - init[k].double()).norm())
        moved += step
        assert d <= 2e-2 * step + 1e-6 * float(ref[k].double().norm()), (k, d, step)
    assert moved > 0
    assert np.allclose(hp[0]["eig"], eig[-1], rtol=2e-2) and np.allclose(hp[0]["eig"], hp[1]["eig"])
    dp = run_ranks("dropin_dp", 2, tmp_path)
    for k in ref:
        assert torch.equal(dp[0]["sd"][k], dp[1]["sd"][k]), k
        assert bool(torch.isfinite(dp[0]["sd"][k]).all())
    assert any(not torch.equal(dp[0]["sd"][k], init[k]) for k in ref)
    # ranks that seeded themselves differently before building the model (a common DDP habit): rank 0's weights and
    # Fourier matrix are broadcast before the first step, so the runs are the equally seeded ones, bit for bit
    for mode, same in (("dropin_hp_perrank", hp), ("dropin_dp_perrank", dp)):
        pr = run_ranks(mode, 2, tmp_path)
        for k in ref:
            assert torch.equal(pr[0]["sd"][k], pr[1]["sd"][k]), (mode, k)
            assert torch.equal(pr[0]["sd"][k], same[0]["sd"][k]), (mode, k)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,amp", [(2, False), (4, False), (2, True)])
def test_cdk_step_with_the_hidden_width_sharded_matches_single_process(tmp_path, world, amp):
    """cdk.ShardedCdkStep: both towers' hidden width split over `world` ranks (BatchNorm statistics are per column: no
    exchange for them), one all-reduce of the partial second-layer products and one of a scalar per step - three
    training steps reproduce the single-process FusedCdkStep on the same batches: losses, gradient norms and, after
    gather_into_model(), every parameter and running statistic, identical on all ranks. amp: the mixed-precision mode."""
    rs = run_ranks("cdk_tp_amp" if amp else "cdk_tp", world, tmp_path)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _multirank_worker as W
    from neural_svd_amd.cdk import FusedCdkStep
    dev = torch.device("cuda:0")
    model, method, xs, ys = W.cdk_case(dev, amp)
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    fs = FusedCdkStep(method, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=xs[0].shape[0], use_amp=amp)
    losses = [fs.step(xs[t], ys[t]).clone().cpu() for t in range(3)]
    fs.flush_counters()
    ref = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tol = 2e-3 if amp else 2e-5  # (mixed: a partial sum that differs in the last bit moves a bfloat16 rounding now and then)
    for r in rs:
        assert r["d1_local"] == 512 // world
        for t in range(3):
            assert rel(r["losses"][t][:3], losses[t][:3]) < tol and rel(r["losses"][t][3:], losses[t][3:]) < tol
        for k in ref:
            assert torch.equal(r["sd"][k], rs[0]["sd"][k]), k
            if ref[k].dtype.is_floating_point:
                moved = float((ref[k].double() - init[k].double()).norm())
                d = float((r["sd"][k].double() - ref[k].double()).norm())
                assert d <= 10 * tol * moved + 1e-6 * float(ref[k].double().norm()), (k, d, moved)
            else:
                assert torch.equal(r["sd"][k], ref[k]), k
