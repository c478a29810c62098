"""Child process of tests/test_multirank_gpu.py: one rank of a FusedTrainer run over gloo with every rank on ONE
device (NSVD_FORCE_DEVICE - the GPU box has a single GPU; RCCL refuses two ranks per device, gloo does not care).
Started as a fresh interpreter (RANK / WORLD_SIZE / MASTER_* in the environment), writes its results to argv[2]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from neural_svd_amd import hip_ops as H  # noqa: E402
from neural_svd_amd import parallel  # noqa: E402
from neural_svd_amd.trainer import FusedTrainer  # noqa: E402

CASE = dict(L=4, D=2, m=64, hidden=(128, 128, 128), B_local=64, steps=3)


def make_problem():
    return H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0)


def make_shape():
    return H.ModelShape(L=CASE["L"], D=CASE["D"], m=CASE["m"], hidden=CASE["hidden"], has_exp_mask=True)


def trainer_kw():
    return dict(sequential=False, lr=1e-3, num_iters=50, sampling_scale=4.0, fourier_scale=0.15, exp_mask_init=10.0)


def global_batches(world):
    g = torch.Generator().manual_seed(11)
    return [4.0 * torch.randn(CASE["B_local"] * world, CASE["D"], generator=g) for _ in range(CASE["steps"])]


def dp_rows(xg, rank, world):
    """global batch arranged [f1_0 .. f1_{W-1}, f2_0 .. f2_{W-1}]: rank r's rows of each half"""
    h = xg.shape[0] // 2
    q = h // world
    return torch.cat([xg[rank * q:(rank + 1) * q], xg[h + rank * q:h + (rank + 1) * q]]).contiguous()


def ko_case(dev, world):
    """a small dense kernel operator, the trainer keywords and three index batches (the same on every rank)"""
    from neural_svd_amd.kernel_ops import synthetic_psd_kernel
    # 300 points: the batch of two ranks (256) is smaller, that of four ranks (512) larger than the point set (the
    # latter takes nsvd_kernel_apply's multiply-every-row-once form)
    op = synthetic_psd_kernel(N=300, rank=48, dim=5, seed=3, device=dev)
    kw = dict(L=8, m=64, hidden=(128, 128), batch_size=128 * world, sequential=True, lr=1e-3, seed=2)
    g = torch.Generator().manual_seed(21)
    batches = [torch.randint(300, (128 * world,), generator=g) for _ in range(3)]
    return op, kw, batches


def dropin_case(world_rows, dev, seed_offset=0):
    """the reference-style objects of a small oscillator problem on the MFMA kernels (128-wide layers), built through
    this package's mirrors of the reference API exactly as main_pde.py builds them; batch_size = world_rows"""
    import argparse
    from tests import _golden as G
    from neural_svd_amd.models import get_wavefunctions
    from neural_svd_amd.nested_lowrank import get_evd_method
    from neural_svd_amd.operators import get_dataloader, get_problem
    cfg = dict(G.cfg_of(G.load("model_small"), "osc_small"), mlp_hidden_dims="128,128,128", fourier_mapping_size=64,
               neigs=4, batch_size=world_rows, num_iters=4, lr=1e-3)
    a = argparse.Namespace(**{k: v for k, v in cfg.items() if k not in ("sequential", "step")})
    a.loss = argparse.Namespace(name="neuralsvd",
                                neuralsvd=argparse.Namespace(step=cfg["step"], sequential=cfg["sequential"]))
    a.adam_eps, a.use_lr_scheduler, a.ema_decay, a.log_dir = 1e-7, True, 0.995, None
    a.print_freq, a.eval_freq = 10 ** 9, cfg["num_iters"]
    torch.manual_seed(cfg["seed"] + seed_offset)  # seed_offset: a rank that seeds itself differently (a DDP habit)
    operator, _ = get_problem(a, dev)
    model = get_wavefunctions(a)
    loaders = get_dataloader(a, dev)
    method = get_evd_method(a, "neuralsvd", model).to(dev)
    return a, operator, method, loaders


def dropin_blocks():
    """32-row blocks of coordinates, the same in every process: a step of `world` ranks consumes `world` of them"""
    g = torch.Generator().manual_seed(31)
    return [4.0 * torch.randn(32, 2, generator=g) for _ in range(16)]


def cdk_case(dev, amp=False):
    """the Sketchy-style CDK objects (two towers 128 -> 512 -> 128, l2_ball, NestedLoRAForCDK) and three (x, y) batches
    of 128 rows; amp: 128 -> 512 -> 256 and 256 rows (the mixed-precision kernels take multiples of 256)"""
    import torch.nn as nn
    from neural_svd_amd.cdk import HeteroNetwork, NestedLoRAForCDK, get_mlp
    torch.manual_seed(17)
    sizes, B = ([128, 512, 256], 256) if amp else ([128, 512, 128], 128)
    model = HeteroNetwork([get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                           get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                          [nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(dev).train()
    method = NestedLoRAForCDK(model, neigs=sizes[-1], step=1, sequential=False, set_first_mode_const=True).to(dev)
    g = torch.Generator().manual_seed(18)
    xs = [torch.randn(B, 128, generator=g).to(dev) for _ in range(3)]
    ys = [torch.randn(B, 128, generator=g).to(dev) for _ in range(3)]
    return model, method, xs, ys


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    if "_L" in mode:  # hp_L5: another head count (heads that do not divide over the ranks)
        mode, l = mode.rsplit("_L", 1)
        CASE["L"] = int(l)
    if mode.endswith("_big"):  # batches beyond 1024 rows: the backward takes partial moment sums instead of f itself
        CASE["B_local"] = 640
        mode = mode[:-4]
    dev = torch.device("cuda", int(os.environ["NSVD_FORCE_DEVICE"]))
    torch.cuda.set_device(dev)
    comm = parallel.Communicator.from_env(dev, backend=os.environ.get("NSVD_DIST_BACKEND", "gloo"))
    rank, world = comm.rank, comm.world
    shape, prob, kw = make_shape(), make_problem(), trainer_kw()
    res = {}
    if mode in ("dp", "hp", "dp_win", "dp_rsag", "dp_a2a"):
        # external batches: comparable with a single-process run on the global batch.
        # dp_win: the backward cut into two head windows, each window's bucket all-reduced as it is enqueued;
        # dp_rsag: the same windows with reduce-scatter -> optimiser on this rank's slices -> all-gather
        # dp_a2a: the same two phases as all-to-alls
        extra = {}
        if mode != "dp" and mode != "hp":
            extra = dict(grad_windows=2, dp_exchange={"dp_rsag": "rs_ag", "dp_a2a": "a2a"}.get(mode, "allreduce"))
        tr = FusedTrainer(shape, prob, CASE["B_local"], seed=5, device=dev, comm=comm, parallelism=mode[:2],
                          keep_grads=True, grad_buckets=3, **extra, **kw)
        assert tr.hp == (mode == "hp") and tr.world == world
        if extra:
            assert len(tr._windows) == 2 and len(tr.grad_buckets()) == 3
        for i, xg in enumerate(global_batches(world)):
            x = (xg if mode == "hp" else dp_rows(xg, rank, world)).to(dev)
            tr.step(x)
            if i == 0:
                res["sharded0"] = tr._state_sharded
                res["grad0"] = tr.P.grad.clone().cpu()
                res["loss0"] = tr.loss.clone().cpu()
                res["mom0"] = tr.moments.clone().cpu()
        tr.gather_optimizer_state()
        res.update(flat=tr.P.flat.cpu(), ema=tr.P.ema.cpu(), sq=tr.P.sq.cpu(), t=tr.t, l_off=tr.l_off,
                   buckets=tr.grad_buckets(), fused_step=tr.fused_step,
                   sd={k: v.cpu() for k, v in tr.state_dict().items()},
                   sd_ema={k: v.cpu() for k, v in tr.state_dict(ema=True).items()})
    elif mode in ("dp_overlap", "hp_overlap"):
        # internal device sampler: the batch prepared under the collective vs the plain ordering, bit for bit;
        # seed=None: every rank draws different initial weights, rank 0's must win
        par = mode[:2]
        runs = []
        for ov, sy in ((True, False), (False, False), (True, True)):
            # (the third ordering: blocking collectives - nothing prepared under them, hp's next batch rides in the backward)
            torch.manual_seed(1234)  # same "unseeded" stream for every ordering of this rank
            torch.randn(rank + 1)    # ... but a different one per rank
            tr = FusedTrainer(shape, prob, CASE["B_local"], seed=None, sample_seed=9, device=dev, comm=comm,
                              parallelism=par, overlap=ov, sync_collectives=sy, **kw)
            assert tr.overlap == (ov and not sy) and tr.guest_features == (ov and par == "hp")
            init = tr.P.flat.clone().cpu()
            for _ in range(6):
                tr.step()
            torch.cuda.synchronize()
            runs.append(dict(init=init, flat=tr.P.flat.cpu(), ema=tr.P.ema.cpu(), fB=tr.P.fourier_B.cpu(),
                             loss=tr.loss.cpu(), drawn=tr.batches_drawn, x=tr.x.cpu()))
        res["runs"] = runs
    elif mode == "ko_hp":
        # the kernel-operator step (kernel_ops.FusedKernelTrainer) with heads sharded: every rank the same index batches
        from neural_svd_amd.kernel_ops import FusedKernelTrainer
        op, kw_ko, batches = ko_case(dev, world)
        fk = FusedKernelTrainer(op, comm=comm, **kw_ko)
        assert fk.world == world and fk.shape.L == kw_ko["L"] // world and fk.l_off == rank * fk.shape.L
        for i, idx in enumerate(batches):
            loss = fk.step(idx.to(dev))
            if i == 0:
                res["loss0"], res["mom0"] = loss.clone().cpu(), fk.moments.clone().cpu()
                res["f0"], res["Kf0"] = fk.f.clone().cpu(), fk.Kf.clone().cpu()
        res.update(views=[v.cpu() for v in fk.P.views(fk.P.flat)], sq=[v.cpu() for v in fk.P.views(fk.P.sq)],
                   l_off=fk.l_off, t=fk.t)
    elif mode in ("dropin_hp", "dropin_dp", "dropin_hp_perrank", "dropin_dp_perrank"):
        # the reference-signature training loop (drop_in.train_operator) started on several ranks by a launcher: it
        # finds RANK / WORLD_SIZE itself, shards heads (or samples) and leaves the WHOLE model in `method` on every rank
        import neural_svd_amd.drop_in as DI
        comm.close()  # train_operator makes its own communicator from the environment
        os.environ["NSVD_DIST_BACKEND"] = "gloo"
        # *_perrank: every rank builds its model from its own seed - rank 0's weights and Fourier matrix must win
        perrank = mode.endswith("_perrank")
        args, operator, method, (make_batch, val_data, batch_ftn_val, imp_train, imp_val) = \
            dropin_case(32, dev, seed_offset=100 * rank if perrank else 0)
        mode = mode.replace("_perrank", "")
        args.parallelism = mode[-2:]
        blocks = iter(dropin_blocks())
        if mode == "dropin_dp":  # every rank its own rows: rank r takes block world * step + r
            allb = dropin_blocks()
            blocks = iter(allb[rank::world])
        eig, norms = DI.train_operator(args, method, operator, lambda: next(blocks), val_data, batch_ftn_val, None, None,
                                       dev, imp_train, imp_val)
        res.update(sd={k: v.detach().cpu() for k, v in method.state_dict().items()}, eig=eig[-1], norms=norms[-1])
        torch.save(res, os.path.join(out_dir, f"{sys.argv[1]}_r{rank}.pt"))
        return
    elif mode in ("cdk_tp", "cdk_tp_amp"):
        # the CDK training step with the towers' hidden width sharded over the ranks (cdk.ShardedCdkStep)
        from neural_svd_amd.cdk import ShardedCdkStep
        amp = mode.endswith("amp")
        model, method, xs, ys = cdk_case(dev, amp)
        st = ShardedCdkStep(method, comm, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=xs[0].shape[0],
                            use_amp=amp)
        res["losses"] = [st.step(xs[t], ys[t]).clone().cpu() for t in range(3)]
        st.gather_into_model()
        res["sd"] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        res["d1_local"] = st.d1
    elif mode == "rccl1":
        # ONE rank on the real collective library (backend "nccl" = RCCL): the exchange sequences forced on in a world
        # of one, so that every RCCL call of the product path - argument views, in-place gathers, AVG, async work
        # handles and their stream ordering against the HIP kernels - runs on the one GPU this box has
        assert world == 1 and comm.backend == "nccl"
        comm.force_exchange = True
        res["rccl_ranks"] = comm.count_ranks()
        batches = [x.to(dev) for x in global_batches(1)]

        def run(c, internal=False, probe=False, **kw2):
            tr = FusedTrainer(shape, prob, CASE["B_local"], seed=5, sample_seed=9, device=dev, comm=c,
                              keep_grads=not internal, grad_buckets=3, **kw2, **kw)
            if probe:
                tr.probe = parallel.CommProbe(dev)
            for i in range(CASE["steps"]):
                tr.step(None if internal else batches[i])
            tr.gather_optimizer_state()
            torch.cuda.synchronize()
            out = dict(flat=tr.P.flat.cpu(), ema=tr.P.ema.cpu(), sq=tr.P.sq.cpu(), loss=tr.loss.cpu(), multi=tr.multi,
                       windows=len(tr._windows), fused_step=tr.fused_step, hp=tr.hp, overlap=tr.overlap)
            if probe:
                out["waits"] = dict(tr.probe.summary())
            return out

        res["plain"] = run(None)
        res["allreduce"] = run(comm, grad_windows=1, probe=True)
        res["allreduce_windows"] = run(comm, grad_windows=2)
        res["rs_ag"] = run(comm, grad_windows=2, dp_exchange="rs_ag", probe=True)
        res["a2a"] = run(comm, grad_windows=2, dp_exchange="a2a", probe=True)
        res["hp"] = run(comm, parallelism="hp", probe=True)
        res["allreduce_blocking"] = run(comm, grad_windows=1, sync_collectives=True, probe=True)
        res["rs_ag_blocking"] = run(comm, grad_windows=2, dp_exchange="rs_ag", sync_collectives=True)
        res["hp_blocking"] = run(comm, parallelism="hp", sync_collectives=True)
        res["plain_internal"] = run(None, internal=True)
        res["dp_internal"] = run(comm, internal=True, dp_exchange="rs_ag")
        res["hp_internal"] = run(comm, internal=True, parallelism="hp")
        res["hp_internal_blocking"] = run(comm, internal=True, parallelism="hp", sync_collectives=True)
    else:
        raise SystemExit(f"unknown mode {mode}")
    torch.cuda.synchronize()
    comm.barrier()
    torch.save(res, os.path.join(out_dir, f"{sys.argv[1]}_r{rank}.pt"))
    comm.close()


if __name__ == "__main__":
    main()
