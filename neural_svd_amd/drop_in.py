"""train_operator with the reference's signature (examples/operator/__init__.py:20-153) for
main_pde.py-style drivers. Two loop bodies with the same semantics (RMSprop, cosine schedule, torch_ema-style EMA,
evaluation under the EMA weights, checkpoints with the reference's keys):
  * the fused one (default whenever the configuration is the one the PDE scripts use: NestedLoRA over WaveFunctions,
    OperatorWrapper, Gaussian importance, RMSprop without momentum): the steps are taken by trainer.FusedTrainer on the
    model's own weights - four kernel launches per step - and the nn.Module, the EMA object and the optimiser state are
    refreshed from it whenever something looks at them (evaluation, checkpoint, return);
  * the plain one (``args.fused_loop = False``, or any other configuration): torch autograd around the HIP loss /
    operator Functions; with RMSprop (no momentum) the whole body - backward, optimiser, scheduler, EMA - is captured
    into a HIP graph after three eager iterations and replayed (CapturedPlainStep; ``args.graph_loop = False`` keeps it
    eager: torch.optim + the foreach EMA, host-bound).
Plotting and the local-energy monitor are left out (they are not part of the hot path)."""
from __future__ import annotations

import contextlib
import os
import time

import torch

from .spectrum import compute_spectrum_evd


class ExponentialMovingAverage:
    """torch_ema.ExponentialMovingAverage's documented behaviour (the package is not in this image: parity
    unpinned): shadow = clone(params); on update n += 1, d = min(decay, (1 + n) / (10 + n)),
    shadow -= (1 - d) (shadow - param); store / restore / copy_to / average_parameters; state_dict with the keys
    decay, num_updates, shadow_params, collected_params (what torch_ema's load_state_dict reads)."""

    def __init__(self, parameters, decay: float, use_num_updates: bool = True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.params = [p for p in parameters if p.requires_grad]
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in self.params]
        self.collected_params = None

    def _resolve(self, parameters):
        if parameters is None:
            return self.params
        ps = [p for p in parameters if p.requires_grad]
        if len(ps) != len(self.shadow_params):
            raise ValueError("Number of parameters passed as argument is different from number of shadow parameters "
                             "maintained by this ExponentialMovingAverage")
        return ps

    @torch.no_grad()
    def update(self, parameters=None):
        params = self._resolve(parameters)
        d = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            d = min(d, (1 + self.num_updates) / (10 + self.num_updates))
        diffs = torch._foreach_sub(self.shadow_params, [p.detach() for p in params])
        torch._foreach_mul_(diffs, 1.0 - d)
        torch._foreach_sub_(self.shadow_params, diffs)

    @torch.no_grad()
    def copy_to(self, parameters=None):
        for p, s in zip(self._resolve(parameters), self.shadow_params):
            p.copy_(s)

    @torch.no_grad()
    def store(self, parameters=None):
        self.collected_params = [p.detach().clone() for p in self._resolve(parameters)]

    @torch.no_grad()
    def restore(self, parameters=None):
        if self.collected_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        for p, c in zip(self._resolve(parameters), self.collected_params):
            p.copy_(c)

    @contextlib.contextmanager
    def average_parameters(self, parameters=None):
        params = self._resolve(parameters)
        self.store(params)
        self.copy_to(params)
        try:
            yield
        finally:
            self.restore(params)

    def to(self, device=None, dtype=None):
        self.shadow_params = [s.to(device=device, dtype=dtype if s.is_floating_point() else None)
                              for s in self.shadow_params]
        if self.collected_params is not None:
            self.collected_params = [c.to(device=device, dtype=dtype if c.is_floating_point() else None)
                                     for c in self.collected_params]

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params,
                    collected_params=self.collected_params)

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        self.decay = sd["decay"]
        if not 0.0 <= self.decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.num_updates = sd["num_updates"]
        shadow = sd["shadow_params"]
        if len(shadow) != len(self.params):
            raise ValueError("shadow_params does not match the parameters of this ExponentialMovingAverage")
        self.shadow_params = [s.detach().clone().to(device=p.device, dtype=p.dtype) for s, p in zip(shadow, self.params)]
        coll = sd.get("collected_params")
        self.collected_params = None if coll is None else \
            [c.detach().clone().to(device=p.device, dtype=p.dtype) for c, p in zip(coll, self.params)]


def get_optimizer(args, model):
    """examples/utils.py:48-72 (rmsprop branch is what the PDE scripts use)."""
    if args.optimizer == "rmsprop":
        return torch.optim.RMSprop(model.parameters(), lr=args.lr, alpha=args.rmsprop_decay, eps=1e-10,
                                   weight_decay=0, momentum=args.momentum)
    if args.optimizer == "adam":
        return torch.optim.Adam(model.parameters(), lr=args.lr, eps=args.adam_eps)
    if args.optimizer == "sgd":
        return torch.optim.SGD(model.parameters(), lr=args.lr, momentum=args.momentum)
    raise NotImplementedError


def _comm_from_env(device):
    """torch.distributed.run (or any launcher that sets RANK / WORLD_SIZE / MASTER_*) started this script on several
    GPUs: the Communicator of this rank, else None. The reference has no live distributed path (tools/generic.py:65-180
    is never imported): with this package the same main_pde.py scales by `torchrun --nproc-per-node N main_pde.py ...`."""
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or torch.device(device).type != "cuda":
        return None
    from .parallel import Communicator
    return Communicator.from_env(torch.device(device), backend=os.environ.get("NSVD_DIST_BACKEND"))


def _fused_loop_trainer(args, method, operator, importance_train, device, comm=None):
    """A FusedTrainer over the SAME weights when the configuration is one it implements, else None."""
    from .models import WaveFunctions
    from .nested_lowrank import NestedLoRA, nesting_masks
    from .operators import GaussianImportance, OperatorWrapper, fused_problem_of
    from .trainer import FusedTrainer
    if not getattr(args, "fused_loop", True):
        return None
    if args.optimizer != "rmsprop" or float(getattr(args, "momentum", 0.0)) != 0.0:
        return None
    if not (isinstance(method, NestedLoRA) and isinstance(method.model, WaveFunctions)) or method.sort_indices is not None:
        return None
    if not isinstance(operator, OperatorWrapper) or not isinstance(importance_train, GaussianImportance):
        return None
    if torch.device(device).type != "cuda":
        return None
    try:
        step = int(args.loss.neuralsvd.step)
    except AttributeError:
        return None
    v, M, _ = nesting_masks(method.neigs, bool(method.sequential), step)
    if not (torch.equal(v.float(), method.vector_mask.float().cpu()) and torch.equal(M.float(), method.matrix_mask.float().cpu())):
        return None
    model = method.model
    dev = torch.device(device)
    # the trainer's constructor draws (and here discards) initial weights: keep the caller's random streams untouched
    par = "dp"
    if comm is not None:
        # several ranks: heads sharded (no gradient traffic, one all-gather of f, Tf per step: parallel.py) for any
        # head count with at least one head per rank - the scripts' --neigs 36 / 55 on 8 GPUs give ranks of 5 / 4 and
        # 7 / 6 heads (parallel.head_range) -, samples sharded otherwise; args.parallelism overrides
        par = getattr(args, "parallelism", None) or ("hp" if model.shape.L >= comm.world else "dp")
    with torch.random.fork_rng(devices=[dev.index if dev.index is not None else torch.cuda.current_device()]):
        tr = _make_fused(FusedTrainer, args, method, model, operator, importance_train, step, device,
                         fused_problem_of, comm, par)
    if comm is not None:
        # every rank must train the SAME model: rank 0's weights (and frozen Fourier matrix) win, whatever the ranks'
        # random streams were when the script built the module (per-rank seeds are a common DDP habit; unseeded runs
        # differ anyway). Without this the replicas of a sample-sharded run start apart and a head-sharded run gathers
        # f / Tf columns computed with different Fourier matrices.
        with torch.no_grad():
            for t in [model.base.feature_map._B] + list(model.base.ws) + list(model.base.bs) + \
                    ([model.boundary_mask.scales] if model.has_exp_mask else []):
                comm.broadcast(t.data, 0)
    sl = slice(tr.l_off, tr.l_off + tr.shape.L)  # this rank's heads (all of them unless heads are sharded)
    tr.P.load(model.base.feature_map._B.data, [w.data[sl] for w in model.base.ws], [b.data[sl] for b in model.base.bs],
              model.boundary_mask.scales.data[sl] if model.has_exp_mask else None)
    return tr


def _make_fused(FusedTrainer, args, method, model, operator, importance_train, step, device, fused_problem_of,
                comm=None, parallelism="dp"):
    return FusedTrainer(model.shape, fused_problem_of(operator, importance_train, model), args.batch_size,
                        sequential=bool(method.sequential), step=step, lr=args.lr, rmsprop_decay=args.rmsprop_decay,
                        rmsprop_eps=1e-10, ema_decay=args.ema_decay, num_iters=args.num_iters,
                        use_lr_scheduler=bool(args.use_lr_scheduler), sampling_scale=importance_train.sigma, seed=0,
                        device=device, path=method.path, device_sampler=False, comm=comm, parallelism=parallelism,
                        sync_collectives=True,
                        exp_mask_init=1.0 if model.has_exp_mask else None)  # initial values: replaced by the caller


class CaptureUnavailable(RuntimeError):
    """CapturedPlainStep cannot take this step (and has changed nothing): run the eager plain loop instead."""


class CapturedPlainStep:
    """The PLAIN loop body of the reference (examples/operator/__init__.py:55-74) - zero_grad, compute_loss_operator,
    loss.backward(), optimizer.step(), scheduler.step(), ema.update() - captured once into a HIP graph and replayed:
    torch autograd around the HIP Functions as in the eager plain loop, but the optimiser / scheduler / EMA triple is
    ONE capturable launch per parameter tensor (nsvd_rmsprop_ema_step_dev) that reads the step's learning rate and EMA
    decay from a device-resident nsvd_step_state and advances it, so nothing host-side changes between replays.
    The torch objects stay the owners of the state: the RMSprop square averages are optimizer.state[p]['square_avg'],
    the EMA shadow is ema.shadow_params, the gradients are p.grad - all updated in place; their Python counters
    (optimizer step, scheduler.last_epoch, param_group lr, ema.num_updates) are brought up to date by sync_counters().
    The first WARMUP calls run the same body eagerly (torch needs a few eager iterations before capturing autograd);
    eager and replayed steps are the same launches with the same arguments."""
    WARMUP = 3

    def __init__(self, args, method, operator, importance_train, optimizer, scheduler, ema, device):
        from . import hip_ops as H
        self.H = H
        self.args, self.method, self.operator, self.importance = args, method, operator, importance_train
        self.optimizer, self.scheduler, self.ema = optimizer, scheduler, ema
        self.device = torch.device(device)
        self.params = [p for p in method.parameters() if p.requires_grad]
        assert len(self.params) == len(ema.shadow_params)
        self.state = H.StepState(self.device, args.lr, args.num_iters if args.use_lr_scheduler else 0,
                                 args.rmsprop_decay, 1e-10, args.ema_decay)
        for p in self.params:  # torch.optim.RMSprop's state layout, created up front (it is lazy)
            st = optimizer.state[p]
            if "square_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        self.sq = [optimizer.state[p]["square_avg"] for p in self.params]
        self.x = None
        self.graph = None
        self.loss = None
        self.total = torch.zeros((), dtype=torch.float64, device=self.device)
        self.steps = 0

    def _body(self):
        H = self.H
        self.state.begin()
        self.method.train()
        self.optimizer.zero_grad(set_to_none=True)
        loss, _aux = self.method.compute_loss_operator(self.operator, self.x, importance=self.importance)
        loss.backward()
        n = len(self.params)
        if any(p.grad is None for p in self.params):
            # a trainable parameter the loss does not reach (torch's RMSprop would skip it): nothing has been updated
            # yet - the caller takes this and every later step with the eager plain loop
            raise CaptureUnavailable("a trainable parameter received no gradient")
        for i, (p, sq, sh) in enumerate(zip(self.params, self.sq, self.ema.shadow_params)):
            H.rmsprop_ema_step_dev(p.data.view(-1), p.grad.view(-1), sq.view(-1), sh.view(-1), self.state, 1.0,
                                   advance=(i == n - 1))
        self.total += loss.detach()
        return loss.detach()

    def step(self, x: torch.Tensor) -> torch.Tensor:
        """one optimiser step on the batch x; returns the (device) loss of that step"""
        if self.x is None:
            self.x = torch.empty_like(x)
        self.x.copy_(x)
        with torch.cuda.device(self.device):
            if self.steps < self.WARMUP:
                self.loss = self._body()
            else:
                if self.graph is None:
                    torch.cuda.synchronize()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    graph = torch.cuda.CUDAGraph()
                    try:
                        with torch.cuda.stream(side):
                            with torch.cuda.graph(graph, stream=side):
                                self.loss = self._body()
                    except CaptureUnavailable:
                        raise
                    except Exception as e:  # capture refused (an op that syncs, an allocation the pool cannot serve ..):
                        # captured launches never ran, so the state is that of the last eager step
                        torch.cuda.current_stream().wait_stream(side)
                        raise CaptureUnavailable(f"HIP-graph capture of the plain loop body failed: {e!r}") from e
                    torch.cuda.current_stream().wait_stream(side)
                    self.graph = graph
                self.graph.replay()
        self.steps += 1
        return self.loss

    def sync_counters(self) -> None:
        """the torch objects' Python-side counters <- the steps taken (their tensors were updated in place)"""
        from .trainer import cosine_lr
        for p in self.params:
            self.optimizer.state[p]["step"] = torch.tensor(float(self.steps))
        self.ema.num_updates = self.steps
        if self.args.use_lr_scheduler:
            self.scheduler.last_epoch = self.steps
            for g in self.optimizer.param_groups:
                g["lr"] = cosine_lr(self.args.lr, self.steps, self.args.num_iters)


def _captured_plain_step(args, method, operator, importance_train, optimizer, scheduler, ema, device):
    """A CapturedPlainStep when the plain loop can be captured (RMSprop without momentum, this package's NestedLoRA /
    OperatorWrapper with a generated nesting mask, GPU, float32), else None: the eager plain loop takes over."""
    from . import hip_ops as H
    from .nested_lowrank import NestedLoRA, _mask_kind
    from .operators import OperatorWrapper
    if not getattr(args, "graph_loop", True) or torch.device(device).type != "cuda":
        return None
    if args.optimizer != "rmsprop" or float(getattr(args, "momentum", 0.0)) != 0.0:
        return None
    if not isinstance(method, NestedLoRA) or not isinstance(operator, OperatorWrapper):
        return None
    if _mask_kind(method.vector_mask, method.matrix_mask) == H.MASK_CUSTOM:
        return None  # custom masks are copied to the device per call: a host copy cannot be captured
    if any(p.dtype != torch.float32 or not p.is_contiguous() for p in method.parameters()):
        return None
    return CapturedPlainStep(args, method, operator, importance_train, optimizer, scheduler, ema, device)


@torch.no_grad()
def _refresh_from_trainer(tr, method, ema, optimizer, scheduler):
    """nn.Module parameters, EMA shadow, RMSprop state and schedule position <- the fused trainer's buffers."""
    named = dict(method.named_parameters())
    shadow = {id(p): s for p, s in zip(ema.params, ema.shadow_params)}
    tr.gather_optimizer_state()  # (samples sharded with a sharded optimiser: make the state whole first)

    whole = tr.gather_heads_tensor  # heads sharded: every rank's slice of a tensor -> the (L, ...) tensor, on every rank

    for n, w, e, q in zip(tr.P.names, tr.P.views(tr.P.flat), tr.P.views(tr.P.ema), tr.P.views(tr.P.sq)):
        w, e, q = whole(w), whole(e), whole(q)
        p = named[n]
        p.data.copy_(w.view_as(p))
        shadow[id(p)].copy_(e.view_as(p))
        st = optimizer.state[p]
        st["step"] = torch.tensor(float(tr.t))
        st["square_avg"] = q.view_as(p).clone()
    ema.num_updates = tr.num_updates
    if tr.use_sched:
        from .trainer import cosine_lr
        scheduler.last_epoch = tr.t
        for g in optimizer.param_groups:
            g["lr"] = cosine_lr(tr.lr, tr.t, tr.num_iters)


def train_operator(args, method, operator, make_batch_ftn_train, val_data, batch_ftn_val, log_writer, log_file,
                   device, importance_train, importance_val, ground_truth_spectrum=None):
    if getattr(args, "use_amp", False):
        # The reference's mixed-precision switch wraps the step in autocast + GradScaler (examples/operator/__init__.py:
        # 37-38,62-72): half-precision matmuls, loss scaling against their underflow. Here it selects this package's
        # mixed-precision forward, NSVD_PATH_FUSED_BF16X3 - every layer on the bf16 MFMA with operands split into three
        # bfloat16 planes, float32 accumulation: the speed-up the flag asks for (2.2 x on the forward) with f and Tf as
        # close to float64 as the float32 path's, so there is nothing for a GradScaler to do.
        # DIFFERENT arithmetic from the reference's autocast, not bit-comparable to it (nor is the float32 path). Models
        # the MFMA kernels do not take have no such forward: refuse rather than ignore the flag.
        from . import hip_ops as H
        from .models import WaveFunctions
        from .nested_lowrank import NestedLoRA
        from .operators import OperatorWrapper
        model = getattr(method, "model", None)
        # gated on the batch the training forward really runs (no padding applies there: rows must come in 32s)
        ok = isinstance(method, NestedLoRA) and isinstance(model, WaveFunctions) and isinstance(operator, OperatorWrapper) \
            and int(args.batch_size) % 32 == 0 \
            and H.path_name(model.shape, int(args.batch_size), H.PATH_FUSED_BF16X3) == "fused_mfma"
        if not ok:
            raise NotImplementedError("use_amp: the mixed-precision forward (NSVD_PATH_FUSED_BF16X3) exists for models "
                                      "the MFMA kernels take only (128-wide hidden layers, D <= 3) and batches that "
                                      "are a multiple of 32 rows")
        method.path = H.PATH_FUSED_BF16X3
        import warnings
        warnings.warn("train_operator(use_amp=True): this package runs the forward as split-bfloat16 products with "
                      "float32 accumulation (NSVD_PATH_FUSED_BF16X3) - NOT the reference's float16 autocast + GradScaler "
                      "(examples/operator/__init__.py:37-38,62-72): no loss scaling, no skipped steps, different "
                      "rounding; the evaluation forward (compute_spectrum_evd) runs on the same path", stacklevel=2)
    optimizer = get_optimizer(args, method)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, args.num_iters)
    ema = ExponentialMovingAverage(method.parameters(), decay=args.ema_decay)
    comm = _comm_from_env(device)
    fused = _fused_loop_trainer(args, method, operator, importance_train, device, comm)
    captured = None
    if fused is None and comm is None:
        captured = _captured_plain_step(args, method, operator, importance_train, optimizer, scheduler, ema, device)
    if comm is not None and fused is None:
        raise NotImplementedError("several ranks (WORLD_SIZE > 1) need the fused loop: this configuration is not one "
                                  "it implements (rmsprop without momentum, NestedLoRA on WaveFunctions, GPU)")
    rank0 = comm is None or comm.rank == 0
    # per-GPU batch = args.batch_size (weak scaling). Heads sharded: every rank steps on the SAME global batch of
    # world x batch_size rows - the sampler is called world times per step, and equally seeded ranks draw equal rows.
    # Samples sharded: every rank needs its OWN rows - its generator is re-seeded by rank once the model is built.
    draws = comm.world if (comm is not None and fused.hp) else 1
    if comm is not None and not fused.hp:
        torch.manual_seed((torch.initial_seed() * 1000003 + 7919 * comm.rank + 1) % (1 << 62))
    all_eigvals, all_norms = [], []
    start = time.time()
    # the reference adds loss.item() to a host total on EVERY step (operator/__init__.py:74,99: a device sync per
    # step); here the running total lives on the device and is read at print time only. The fused loop's backward
    # kernels leave the loss scalars themselves on one GPU with batches of <= 1024 rows; otherwise (heads sharded, larger
    # batches) the loss is evaluated (one extra launch) on every `loss_stride`-th step and at print time only: the
    # `avg_train_loss` column (the reference's key: main_pde.py:195-198 builds a csv.DictWriter with exactly
    # iter / train_loss / avg_train_loss / time, which raises on any other key) is then the mean over those sampled
    # steps - said once on stdout and in the printed row's `avg_train_loss_over`, never in the CSV row -
    # args.loss_every_step = True restores the every-step mean at one launch per step
    total_loss = torch.zeros((), dtype=torch.float64, device=device)
    n_loss = 0
    # (single GPU, batches of <= 1024 rows: the step's own kernels leave the loss value every step - no extra launch -
    # and the every-step mean is the default, as in the reference's log column)
    every = getattr(args, "loss_every_step", None)
    if every is None:
        every = fused is not None and fused._direct_loss
    loss_stride = 1 if every else max(1, int(args.print_freq) // 16)
    sampled_mean = loss_stride > 1 and fused is not None
    if sampled_mean and rank0:
        print(f"train_operator: avg_train_loss is the mean over every {loss_stride}-th step's loss (and the printed "
              f"steps'), not over every step: set args.loss_every_step = True for the reference's every-step mean")
    for it in range(args.num_iters):
        x = make_batch_ftn_train() if draws == 1 else torch.cat([make_batch_ftn_train() for _ in range(draws)])
        x = x.to(device)
        x = x.reshape(x.shape[0], -1)
        if fused is not None and it == 0 and x.shape[0] != fused.B:
            if comm is not None:
                raise ValueError(f"the sampler returns {x.shape[0] // draws} rows, args.batch_size says "
                                 f"{args.batch_size}: several ranks need them equal")
            fused = None  # the sampler does not produce args.batch_size rows: the plain loop takes any batch
            captured = _captured_plain_step(args, method, operator, importance_train, optimizer, scheduler, ema, device)
        if fused is not None and it == 0 and comm is not None and fused.hp:
            # heads sharded: every rank must step on the SAME global batch (equally seeded samplers): compare the first
            # one across the ranks instead of trusting the script's seeding
            probe = x.float().reshape(-1)[:256].contiguous()
            allp = torch.empty((comm.world, probe.numel()), dtype=torch.float32, device=probe.device)
            comm.all_gather(allp, probe)
            if not bool((allp == allp[0:1]).all()):
                raise RuntimeError("heads sharded over several ranks: the ranks' samplers draw different batches "
                                   "(seed every rank identically, or pass args.parallelism = 'dp')")
        if fused is not None:
            fused.step(x.float().contiguous())
            if (it + 1) % loss_stride == 0 or (it + 1) % args.print_freq == 0:
                loss = fused.loss[0]  # evaluated on the device from this step's f, Tf (no sync)
                total_loss += loss
                n_loss += 1
        if fused is None and captured is not None:
            # the plain loop body replayed from a HIP graph (CapturedPlainStep): same launches, no host work per step
            try:
                loss = captured.step(x.float().contiguous())
                n_loss += 1
            except CaptureUnavailable as e:
                # nothing was updated by the refused step: the eager body below takes it, and every later one
                print(f"train_operator: {e}; continuing with the eager plain loop")
                captured.sync_counters()
                total_loss = total_loss + captured.total
                captured = None
        if fused is None and captured is None:
            method.train()
            optimizer.zero_grad()
            loss, _aux = method.compute_loss_operator(operator, x, importance=importance_train)
            loss.backward()
            optimizer.step()
            if args.use_lr_scheduler:
                scheduler.step()
            ema.update()
            total_loss += loss.detach()
            n_loss += 1
        if (it + 1) % args.print_freq == 0:
            # the only host sync, and only at print time (the reference syncs every step)
            if captured is not None:
                total_loss = captured.total
            assert n_loss > 0
            row = {"iter": it + 1, "train_loss": float(loss), "avg_train_loss": float(total_loss) / n_loss,
                   "time": time.time() - start}
            if rank0:
                print(dict(row, avg_train_loss_over=f"every {loss_stride}-th step") if sampled_mean else row)
            if log_writer is not None and rank0:
                # the reference's writer (examples/utils.py:40-45) takes exactly these four keys
                names = getattr(log_writer, "fieldnames", None)
                log_writer.writerow(row if names is None else {k: v for k, v in row.items() if k in names})
                if log_file is not None:
                    log_file.flush()
        if (it + 1) % args.eval_freq == 0:
            if fused is not None:
                _refresh_from_trainer(fused, method, ema, optimizer, scheduler)
            if captured is not None:
                captured.sync_counters()
            method.eval()
            with ema.average_parameters():
                if batch_ftn_val is not None:
                    outputs = compute_spectrum_evd(method, dataloader=batch_ftn_val(), operator=operator,
                                                   importance_train=importance_train, importance_val=importance_val,
                                                   normalize=True, set_first_mode_const=False, device=device)
                    if rank0:
                        print(f"it{it + 1} eigvals: {outputs['eigvals']}")
                        print(f"it{it + 1} norms: {outputs['norms']}")
                    all_eigvals.append(outputs["eigvals"])
                    all_norms.append(outputs["norms"])
            if getattr(args, "log_dir", None) and rank0:
                os.makedirs(args.log_dir, exist_ok=True)
                torch.save(dict(args=args, method=method.state_dict(), ema=ema.state_dict(),
                                optimizer=optimizer.state_dict()), os.path.join(args.log_dir, f"{it + 1}.pth"))
    if fused is not None:
        _refresh_from_trainer(fused, method, ema, optimizer, scheduler)
    if captured is not None:
        captured.sync_counters()
    return all_eigvals, all_norms
