"""The NestedLoRA PDE training step as ONE object: sample -> operator forward -> EVD loss -> backward
-> RMSprop (+ cosine LR) -> EMA, every stage a HIP kernel launch on the current stream.

Mirrors the body of the reference's ``train_operator`` loop (examples/operator/__init__.py:55-74)
with the optimiser of examples/utils.py:50-57 and the sampler of main_pde.py:92-93; parameters,
gradients and optimiser state live in flat float32 buffers (the gradient exchange of a sample-sharded
run is a few bucketed all-reduces over contiguous ranges) with per-tensor views in the reference's
state_dict layout.

Multi-GPU (one process per GPU): the exchange sequences live in parallel.py (dp_step / hp_step); this
class is their HIP compute backend, so a single-GPU step is parallel.dp_step with a world of one.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import os

import torch

from . import hip_ops as H


def cosine_lr(base_lr: float, t: int, T: int, eta_min: float = 0.0) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingLR after t scheduler steps (closed form)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / T)) / 2.0


def reference_init(shape: H.ModelShape, fourier_scale: float, exp_mask_init: Optional[float], seed: Optional[int]):
    """Draw weights on the CPU generator in the reference's order (Fourier _B, then W_i; b_i = 0) so
    that ``seed`` reproduces the reference's initial weights bit for bit
    (examples/utils.py:116-119, examples/models/mlp.py:185-189, pde/boundary.py:43)."""
    if seed is not None:
        torch.manual_seed(seed)
    fB = 2 * torch.pi * fourier_scale * torch.randn((shape.D, shape.m)).float()
    ws, bs, prev = [], [], 2 * shape.m
    for h in shape.dims:
        ws.append(math.sqrt(2.0 / prev) * torch.randn(shape.L, h, prev))
        bs.append(torch.zeros(shape.L, h, 1))
        prev = h
    scales = exp_mask_init * torch.ones(shape.L) if shape.has_exp_mask else None
    return fB, ws, bs, scales


class FlatParams:
    """All trainable tensors of one model in one contiguous float32 buffer (+ same-layout gradient,
    RMSprop square-average and EMA shadow buffers). Names follow the reference's state_dict:
    model.base.ws.{i}, model.base.bs.{i}, model.boundary_mask.scales, model.base.feature_map._B."""

    def __init__(self, shape: H.ModelShape, device, with_state: bool = True):
        self.shape = shape
        self.device = torch.device(device)
        self.version = 0  # bumped by every change of the parameters that is not a fused training step (derived copies
                          # of the weights - the bf16x3 path's planes - are tied to it); bump it after writing `flat`
        shapes = shape.param_shapes()
        # every tensor starts on a 256-B boundary so kernels may use 16-B accesses per tensor
        self.offsets, off = [], 0
        for s in shapes:
            self.offsets.append(off)
            n = 1
            for d in s:
                n *= d
            off += (n + 63) // 64 * 64
        self.numel = off
        self.n_trainable = sum(math.prod(s) for s in shapes)
        self.shapes = shapes
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.flat)
        self.sq = torch.zeros_like(self.flat) if with_state else None
        self.ema = torch.zeros_like(self.flat) if with_state else None
        self.fourier_B = torch.zeros((shape.D, shape.m), dtype=torch.float32, device=self.device)
        self.state_sharded = False  # set by the trainer while sq / ema are valid on this rank's shards only
        nl = len(shape.dims)
        self.names = [f"model.base.ws.{i}" for i in range(nl)] + [f"model.base.bs.{i}" for i in range(nl)]
        if shape.has_exp_mask:
            self.names.append("model.boundary_mask.scales")

    def views(self, buf: torch.Tensor) -> List[torch.Tensor]:
        return [buf[o:o + math.prod(s)].view(s) for o, s in zip(self.offsets, self.shapes)]

    def pack(self, buf: torch.Tensor, with_fourier: bool) -> H.Params:
        v = self.views(buf)
        nl = len(self.shape.dims)
        return H.pack_params(self.shape, v[:nl], v[nl:2 * nl], self.fourier_B if with_fourier else None,
                             v[2 * nl] if self.shape.has_exp_mask else None)

    def load(self, fB, ws, bs, scales=None) -> None:
        self.version += 1
        self.fourier_B.copy_(fB)
        tensors = list(ws) + list(bs) + ([scales] if self.shape.has_exp_mask else [])
        for dst, src in zip(self.views(self.flat), tensors):
            dst.copy_(src.reshape(dst.shape))
        if self.ema is not None:
            self.ema.copy_(self.flat)  # torch_ema: shadow = clone of the parameters at construction

    def state_dict(self, ema: bool = False) -> Dict[str, torch.Tensor]:
        if ema and self.state_sharded:
            raise RuntimeError("the EMA shadow is valid on this rank's optimiser shards only (dp_exchange rs_ag / a2a): "
                               "call FusedTrainer.gather_optimizer_state() on EVERY rank first")
        src = self.ema if ema else self.flat
        d = {n: t.clone() for n, t in zip(self.names, self.views(src))}
        d["model.base.feature_map._B"] = self.fourier_B.clone()
        return d

    def load_state_dict(self, sd: Dict[str, torch.Tensor], ema_sd: Optional[Dict[str, torch.Tensor]] = None,
                        reset_optimizer: bool = True) -> None:
        """Weights (and frozen Fourier matrix) from a reference-layout state_dict. The EMA shadow is taken from
        ``ema_sd`` (same keys) or restarted from the loaded weights (what constructing torch_ema on them does); the
        RMSprop square averages restart from zero unless ``reset_optimizer`` is False."""
        self.version += 1
        for n, t in zip(self.names, self.views(self.flat)):
            t.copy_(sd[n].reshape(t.shape))
        self.fourier_B.copy_(sd["model.base.feature_map._B"])
        if self.ema is not None:
            if ema_sd is not None:
                for n, t in zip(self.names, self.views(self.ema)):
                    t.copy_(ema_sd[n].reshape(t.shape))
            else:
                self.ema.copy_(self.flat)
        if self.sq is not None and reset_optimizer:
            self.sq.zero_()


class FusedTrainer:
    """The step as one object, and the HIP compute backend of parallel.dp_step / parallel.hp_step (every protocol
    method below enqueues kernels on the current stream of ``device`` and returns)."""

    def __init__(self, shape: H.ModelShape, problem: H.Problem, batch_size: int, sequential: bool, step: int = 1,
                 lr: float = 1e-4, rmsprop_decay: float = 0.999, rmsprop_eps: float = 1e-10, ema_decay: float = 0.995,
                 num_iters: int = 500000, use_lr_scheduler: bool = True, sampling_scale: float = 16.0,
                 fourier_scale: float = 0.1, exp_mask_init: Optional[float] = None, seed: Optional[int] = 0,
                 device="cuda:0", path: int = H.PATH_AUTO, comm=None, sample_seed: Optional[int] = None,
                 parallelism: str = "dp", fused_step: bool = True, keep_grads: bool = False,
                 device_sampler: bool = True, overlap: bool = True, grad_buckets: int = 4,
                 dp_exchange: str = "allreduce", grad_windows: Optional[int] = None, sync_collectives: bool = False,
                 device_schedule: bool = False, backward_windows: int = 1):
        """batch_size is the per-GPU batch. parallelism (only with comm.world > 1): "dp" = every rank draws its
        own batch_size rows (moments + gradient all-reduce); "hp" = every rank owns L/world heads and evaluates
        them on the same global batch of batch_size * world rows (one all-gather of f, Tf; see parallel.py).
        fused_step: take the RMSprop + EMA step inside the weight-gradient kernel (nsvd_operator_backward_evd_step)
        whenever no gradient exchange sits between backward and optimiser (single GPU, hp); keep_grads then
        also stores the gradients (P.grad), which the fused step otherwise never writes.
        device_sampler: draw the batch inside the feature kernel (nsvd_operator_sample_features, counter-based
        Philox) instead of torch's generator + a separate feature launch.
        overlap (world > 1, device sampler): the next batch and its features (they depend on no weight) are produced
        into a second workspace while the step's collective is in flight, instead of leaving the GPU idle.
        grad_buckets (dp): number of gradient exchange buckets (cut on head boundaries of W_0) when the backward is one
        window.
        dp_exchange (dp, world > 1): "allreduce" (bucketed all-reduce, identical optimiser pass on every rank) or
        "rs_ag" (reduce-scatter, optimiser on this rank's 1/world of each bucket, all-gather of the parameters; the
        RMSprop square averages and the EMA shadow then live sharded - gather_optimizer_state() before reading them)
        or "a2a" (the same two phases as point-to-point all-to-alls; parallel.py).
        grad_windows (dp, world > 1): head windows the backward is cut into so that a window's gradients go on the
        wire while the next window is computed (None: as many of 4 / 2 / 1 as still give every CU a dW_0 tile).
        sync_collectives (world > 1): blocking collectives on the compute stream instead of asynchronous ones with late
        waits (parallel.dp_step); nothing is then prepared under a collective (hp: the next batch rides in the backward).
        device_schedule (fused step on the MFMA path): the scheduled learning rate, the warmed-up EMA decay and the
        sampler's batch counter live in DEVICE memory (hip_ops.StepState; the backward's first kernel derives the step's
        values, its last one advances the counter) instead of travelling as launch arguments - what makes a step
        replayable from a captured HIP graph (capture_graph). The values are the host path's except that the device
        cosine may differ from libm's in the last bit of the double (include/nsvd.h)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise H.NsvdError(f"FusedTrainer needs a GPU device (got {self.device}); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.path = path
        self.comm = comm  # parallel.Communicator or None
        world = comm.world if comm is not None else 1
        rank = comm.rank if comm is not None else 0
        self.world = world
        # the step goes through an exchange sequence (always with world > 1; a world of one only on request:
        # Communicator.force_exchange)
        self.multi = multi = comm is not None and comm.multi
        self.sync_collectives = bool(sync_collectives) and multi
        self.hp = parallelism == "hp" and multi
        if parallelism not in ("dp", "hp"):
            raise ValueError("parallelism must be 'dp' or 'hp'")
        self.full_shape = shape
        self.Lg, self.l_off = shape.L, 0
        if self.hp:
            # any L >= world: L // world heads each, the first L % world ranks one more (parallel.head_range)
            from .parallel import head_range
            self.l_off, Ll = head_range(shape.L, rank, world)
            shape = H.ModelShape(L=Ll, D=shape.D, m=shape.m, hidden=shape.hidden, has_exp_mask=shape.has_exp_mask)
            batch_size = int(batch_size) * world
        self.shape, self.problem, self.B = shape, problem, int(batch_size)
        self.lr, self.alpha, self.eps = lr, rmsprop_decay, rmsprop_eps
        self.ema_decay, self.num_iters, self.use_sched = ema_decay, num_iters, use_lr_scheduler
        self.sigma = sampling_scale
        self.device_schedule = bool(device_schedule)
        # backward_windows = 2 (single process, fused step on the MFMA kernels): the backward of a step as two head
        # windows on two streams (nsvd_operator_backward_evd_step_window): window 1's latency-bound chain and window 0's
        # HBM-bound optimiser epilogue sit under the other window's MFMA loops. Bit-identical to one window.
        self.backward_windows = int(backward_windows)
        self._side_stream = None
        with torch.cuda.device(self.device):
            self._build(shape, problem, fourier_scale, exp_mask_init, seed, sample_seed, sequential, step, fused_step,
                        keep_grads, device_sampler, overlap, grad_buckets, world, rank, dp_exchange, grad_windows)

    def _build(self, shape, problem, fourier_scale, exp_mask_init, seed, sample_seed, sequential, step, fused_step,
               keep_grads, device_sampler, overlap, grad_buckets, world, rank, dp_exchange, grad_windows):
        path, multi = self.path, self.multi
        self.P = FlatParams(shape, self.device)
        fB0, ws0, bs0, sc0 = reference_init(self.full_shape, fourier_scale, exp_mask_init, seed)
        if self.hp:  # this rank's heads of the (identically seeded) full model
            sl = slice(self.l_off, self.l_off + shape.L)
            ws0, bs0 = [w[sl] for w in ws0], [b[sl] for b in bs0]
            sc0 = sc0[sl] if sc0 is not None else None
        self.P.load(fB0, ws0, bs0, sc0)
        if multi:
            # replicas must start from identical weights (dp) / share the frozen Fourier matrix (hp) whatever the
            # ranks' random streams were (seed=None): rank 0's values win
            self.comm.broadcast(self.P.fourier_B, 0)
            if not self.hp:
                self.comm.broadcast(self.P.flat, 0)
                self.P.ema.copy_(self.P.flat)
        self._params = self.P.pack(self.P.flat, True)
        self._grads = self.P.pack(self.P.grad, False)
        self._ema_params = self.P.pack(self.P.ema, True)
        self._sq_params = self.P.pack(self.P.sq, False)
        self.fused_step = bool(fused_step) and (not multi or self.hp)
        # the generic (non-MFMA) path takes the step with per-tensor optimiser launches and needs the gradients
        self.keep_grads = bool(keep_grads) or H.path_name(shape, self.B, path, problem) != "fused_mfma"
        # nesting masks (methods/nestedlora.py:183-192)
        from .nested_lowrank import nesting_masks
        self.vector_mask, self.matrix_mask, self.mask_kind = nesting_masks(self.Lg, sequential, step)
        self.v_dev = self.vector_mask.to(self.device)
        self.M_dev = self.matrix_mask.to(self.device).contiguous()
        L, Lg = shape.L, self.Lg
        self.ws = H.new_workspace(shape, self.B, self.device)
        self.device_sampler = bool(device_sampler)
        # overlap: two (workspace, x) sets used alternately; set k holds the features of batch k
        self.overlap = bool(overlap) and multi and self.device_sampler and not self.sync_collectives
        # single GPU / head-parallel with the fused step: the NEXT batch is drawn and its features written by guest
        # workgroups of the backward's chain kernel (nsvd_operator_backward_evd_step_next) - no feature launch per step
        self.guest_features = bool(overlap) and self.device_sampler and self.fused_step and not self.keep_grads and \
            H.path_name(shape, self.B, path, problem) == "fused_mfma" and shape.D <= 3
        two_sets = self.overlap or self.guest_features
        self._ws_other = H.new_workspace(shape, self.B, self.device) if two_sets else None
        self._x_other = torch.empty((self.B, shape.D), dtype=torch.float32, device=self.device) \
            if two_sets else None
        self._partials_ready = False  # the scratch already holds the partial moments of the current f_g, Tf_g
        self._next_ready = False     # the other set already holds the next batch and its features
        self._features_ready = False  # the current set holds the features of the batch being stepped on
        self._planes_ws, self._planes_version = None, -1  # bf16x3: which workspace holds the current weights' planes
        self._emits_planes_cached = None
        self._own_batch = False      # the batch being stepped on came from the internal device sampler
        # local (B, L_local) outputs of the forward, packed [f | Tf] so that one all-gather moves both
        # (heads sharded: the all-gather block - as long as the largest rank's, parallel.head_block - begins with it)
        from .parallel import head_block
        self._Lb = head_block(Lg, world) if self.hp else L
        self._fTf_blk = torch.empty(2 * self.B * self._Lb, dtype=torch.float32, device=self.device)
        self.fTf_loc = self._fTf_blk[:2 * self.B * L].view(2, self.B, L)
        self.f, self.Tf = self.fTf_loc[0], self.fTf_loc[1]
        if self.hp:
            if self._Lb != L:
                self._fTf_blk[2 * self.B * L:].zero_()  # (never read; keeps what goes on the wire defined)
            self.gath = torch.empty((world, 2 * self.B * self._Lb), dtype=torch.float32, device=self.device)
            self.fTf_g = torch.empty((2, self.B, Lg), dtype=torch.float32, device=self.device)
            self.f_g, self.Tf_g = self.fTf_g[0], self.fTf_g[1]
        else:
            self.f_g, self.Tf_g = self.f, self.Tf
        self._moments = torch.empty(2 * Lg * Lg + 1, dtype=torch.float32, device=self.device)
        self._loss = torch.zeros(3, dtype=torch.float32, device=self.device)
        # "direct" moments: with no exchange between forward and backward and a batch of <= 1024 rows, the backward
        # kernel takes the 2 L moments each head needs from f itself and no moment kernel runs; the loss scalars
        # (logging only) and the moment vector are then evaluated on demand (properties below)
        self.direct_moments = (not multi or self.hp) and self.B <= 1024 and \
            H.path_name(shape, self.B, path, problem) == "fused_mfma"
        self._loss_stale = False
        self._moments_stale = False
        # direct moments AND every head local: the backward's kernels also leave the loss value of the step
        # (include/nsvd.h, nsvd_operator_backward_evd: per-head partial sums added in the weight-gradient kernel)
        self._direct_loss = self.direct_moments and not self.hp and self.B <= 1024
        self.scratch = H.evd_scratch(self.B, Lg, self.device)
        self.x = torch.empty((self.B, shape.D), dtype=torch.float32, device=self.device)
        self._x_cur = self.x
        self.gen = torch.Generator(device=self.device)
        # dp: every rank its own stream of samples; hp: all ranks draw the SAME global batch
        srank = 0 if self.hp else rank
        self.sample_key = (sample_seed if sample_seed is not None else (seed or 0)) * 1000003 + 7919 * srank + 1
        self.gen.manual_seed(self.sample_key)
        self.batches_drawn = 0
        if self.hp:
            # every rank must draw the SAME global batch: check once that equally seeded generators agree
            probe = torch.empty(64, dtype=torch.float32, device=self.device).normal_(generator=self.gen)
            allp = torch.empty((world, 64), dtype=torch.float32, device=self.device)
            self.comm.all_gather(allp, probe)
            if not bool((allp == allp[0:1]).all()):
                raise RuntimeError("head-parallel sharding: ranks draw different samples from equal seeds")
        self.t = 0            # optimiser / scheduler steps taken
        self.num_updates = 0  # torch_ema counter
        self._lr_now, self._decay_now = self.lr, self.ema_decay
        self.state = None     # hip_ops.StepState: the schedule on the device
        if self.device_schedule:
            if not (self.fused_step and H.path_name(shape, self.B, path, problem) == "fused_mfma"):
                raise H.NsvdError("device_schedule needs the fused optimiser step on the MFMA path (single GPU or "
                                  "heads sharded, 128-wide hidden layers)")
            self.state = H.StepState(self.device, self.lr, self.num_iters if self.use_sched else 0, self.alpha,
                                     self.eps, self.ema_decay)
        # dp gradient buckets: W_0 (89 % of the bytes, first in the flat buffer) cut on head boundaries, the small
        # tensors ride with the last cut
        nb = max(1, min(int(grad_buckets), shape.L))
        w0 = shape.dims[0] * 2 * shape.m
        cuts = [shape.L * i // nb * w0 for i in range(nb)] + [self.P.numel]
        self._buckets = [(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]
        # dp head windows of the backward (parallel.py): window k's W_0 gradients are a contiguous range of the flat
        # buffer and go on the wire while window k + 1 is computed; the small tensors (complete after the last
        # window only) are the last bucket
        from . import parallel
        if dp_exchange not in parallel.DP_EXCHANGES:
            raise ValueError(f"dp_exchange must be one of {parallel.DP_EXCHANGES}")
        self.dp_exchange = dp_exchange
        self.probe = None            # parallel.CommProbe while bench.py measures the exposed waits
        self._windows = [(0, shape.L)]
        if multi and not self.hp and H.path_name(shape, self.B, path, problem) == "fused_mfma":
            for G in ((4, 2) if grad_windows is None else (int(grad_windows),)):
                if G <= 1 or shape.L % G != 0:
                    continue
                nA128 = (2 * shape.m // 128) * (shape.L // G)
                tiles = nA128 * (2 if nA128 <= 128 else 1)  # pmlp_common.h wgrad_tile_width: 64-wide tiles then
                if grad_windows is None and tiles < 256:
                    continue  # the window's dW_0 tiles would leave CUs idle
                if H.backward_head_window_ok(shape, problem, self.B, path, shape.L // G):
                    self._windows = [(g * (shape.L // G), shape.L // G) for g in range(G)]
                    break
        if len(self._windows) > 1:
            self._stage_buckets = [(l0 * w0, (l0 + lc) * w0) for l0, lc in self._windows] + \
                [(shape.L * w0, self.P.numel)]
        else:
            self._stage_buckets = self._buckets
        self._state_sharded = False  # rs_ag: sq / ema valid on this rank's shards only
        self._gshard = None
        if multi and not self.hp and dp_exchange in ("rs_ag", "a2a"):
            for lo, hi in self._stage_buckets:
                if (hi - lo) % world != 0:
                    raise ValueError(f"dp_exchange={dp_exchange!r}: bucket of {hi - lo} elements does not split over "
                                     f"{world} ranks")
            self._gshard = torch.empty(self.P.numel // world + 64, dtype=torch.float32, device=self.device)
            if dp_exchange == "a2a":  # staging of the two all-to-alls: received slices, replicated updated slice
                self._a2a_recv = torch.empty(self.P.numel, dtype=torch.float32, device=self.device)
                self._a2a_send = torch.empty(self.P.numel, dtype=torch.float32, device=self.device)

    # -- stages -------------------------------------------------------------------------------
    def sample(self) -> torch.Tensor:
        """x = sigma * randn(B, D) on the device (reference: host randn + H2D copy, main_pde.py:92-93)."""
        self.x.normal_(0.0, self.sigma, generator=self.gen)  # one kernel (randn + scale)
        return self.x

    def _masks(self):
        cust = self.mask_kind == H.MASK_CUSTOM
        return (self.v_dev, self.M_dev) if cust else (None, None)

    # -- compute-backend protocol of parallel.dp_step / parallel.hp_step ------------------------
    def forward(self, x: torch.Tensor) -> None:
        self._x_cur = x
        # bf16x3: the planes of the current weights are in this workspace if the step that produced them put them there
        planes = self._planes_ws is not None and self._planes_ws == self.ws.data_ptr() and \
            self.path == H.PATH_FUSED_BF16X3 and self._planes_version == self.P.version
        H.operator_forward(self.shape, self._params, self.problem, x, self.ws, True, self.path, out=(self.f, self.Tf),
                           features_ready=self._features_ready, planes_ready=planes)

    def local_moments(self) -> torch.Tensor:
        v, _ = self._masks()
        H.evd_moments(self.f, self.Tf, self.mask_kind, v, self._moments, self.scratch)
        return self._moments

    def backward(self, reduced_moments: Optional[torch.Tensor], take_step: bool) -> None:
        """loss + d loss / d f are evaluated inside the backward kernels from the moments."""
        v, M = self._masks()
        moments, scratch, loss, reduced = self._moments, self.scratch, self._loss, reduced_moments is not None
        if reduced:
            assert reduced_moments is self._moments
        elif self.direct_moments:
            moments = scratch = None
            loss = self._loss if self._direct_loss else None
            self._loss_stale = loss is None
            self._moments_stale = True
        elif self._partials_ready:
            self._partials_ready = False  # left by after_gather (heads sharded), for this f, Tf
        else:
            H.evd_partial(self.f_g, self.Tf_g, self.mask_kind, v, self.scratch)
        x = self._x_cur
        if self.fused_step and take_step:
            t_before = self.t
            lr, decay = self._advance_schedule()
            # (device schedule: lr / decay are read on the device, the host values only keep the counters in step)
            opt = H.rmsprop_state(self._sq_params, self._ema_params, lr, self.alpha, self.eps, decay, self.state)
            if self._two_windows():
                ride = self.guest_features and self._own_batch and not self._next_ready
                self._backward_two_windows(x, v, M, moments, reduced, scratch, loss, opt, ride,
                                           self.batches_drawn - (t_before if self.state is not None else 0))
                if ride:
                    self.batches_drawn += 1
                    self._next_ready = True
                self._note_planes(None)
                return
            if self.guest_features and self._own_batch and not self._next_ready and \
                    (not self.multi or not self.overlap):
                # the next batch rides in this backward's first launch (hp with overlap prepares it under its
                # all-gather instead: parallel.hp_step -> prefetch)
                H.operator_backward_evd_step_next(self.shape, self._params, self.problem, x, self.f_g, self.Tf_g,
                                                  self.mask_kind, v, M, moments, reduced, scratch, loss, None, opt,
                                                  self.ws, self.sample_key,
                                                  self.batches_drawn - (t_before if self.state is not None else 0),
                                                  self._x_other, self._ws_other, 1.0, self.path, l_offset=self.l_off)
                self.batches_drawn += 1
                self._next_ready = True
                self._note_planes(self._ws_other)
                return
            H.operator_backward_evd_step(self.shape, self._params, self.problem, x, self.f_g, self.Tf_g,
                                         self.mask_kind, v, M, moments, reduced, scratch, loss,
                                         self._grads if self.keep_grads else None, opt, self.ws, 1.0, self.path,
                                         l_offset=self.l_off)
            self._note_planes(self.ws)
            return
        H.operator_backward_evd(self.shape, self._params, self.problem, x, self.f_g, self.Tf_g, self.mask_kind, v, M,
                                moments, reduced, scratch, loss, self._grads, self.ws, 1.0, self.path,
                                l_offset=self.l_off)
        if take_step and (not self.multi or self.hp):  # no exchange between backward and optimiser
            self.begin_apply()
            self.apply(0, self.P.numel, 1.0)

    def _two_windows(self) -> bool:
        if self.backward_windows != 2 or self.multi or self.keep_grads or self.shape.L % 2:
            return False
        if self._side_stream is None:
            ok = H.path_name(self.shape, self.B, self.path, self.problem) == "fused_mfma" and \
                self.path != H.PATH_FUSED_BF16X3 and \
                H.backward_head_window_ok(self.shape, self.problem, self.B, self.path, self.shape.L // 2)
            if not ok:
                self.backward_windows = 1
                return False
            self._side_stream = torch.cuda.Stream(device=self.device)
            self._ev_chain0 = torch.cuda.Event()
            self._ev_done1 = torch.cuda.Event()
        return True

    def _backward_two_windows(self, x, v, M, moments, reduced, scratch, loss, opt, ride, next_offset) -> None:
        half = self.shape.L // 2
        cur = torch.cuda.current_stream(self.device)
        kw = dict(grad_scale=1.0, path=self.path, l_offset=self.l_off)
        # window 0 on the current stream (the next batch rides in its chain launch); the event sits between its launches
        H.operator_backward_evd_step_window(self.shape, self._params, self.problem, x, self.f_g, self.Tf_g,
                                            self.mask_kind, v, M, moments, reduced, scratch, loss, opt, self.ws, 0, half,
                                            False, self._ev_chain0, self.sample_key, next_offset,
                                            self._x_other if ride else None, self._ws_other if ride else None, **kw)
        # window 1 (the LAST: schedule advance, loss) on the side stream, behind window 0's chain
        self._side_stream.wait_event(self._ev_chain0)
        with torch.cuda.stream(self._side_stream):
            H.operator_backward_evd_step_window(self.shape, self._params, self.problem, x, self.f_g, self.Tf_g,
                                                self.mask_kind, v, M, moments, reduced, scratch, loss, opt, self.ws,
                                                half, self.shape.L - half, True, None, **kw)
            self._ev_done1.record(self._side_stream)
        cur.wait_event(self._ev_done1)

    def _note_planes(self, ws: torch.Tensor) -> None:
        """a fused bf16x3 step has left the planes of the weights it updated in `ws` (include/nsvd.h:
        nsvd_step_emits_planes); they stay valid until the parameters change by any other route (P.version)"""
        if self.path == H.PATH_FUSED_BF16X3 and not self.multi and self.l_off == 0 and self._emits_planes():
            self._planes_ws, self._planes_version = ws.data_ptr(), self.P.version
        else:
            self._planes_ws = None

    def _emits_planes(self) -> bool:
        if self._emits_planes_cached is None:  # (one library call, not one per step)
            self._emits_planes_cached = H.step_emits_planes(self.shape, self.B, H.PATH_FUSED_BF16X3)
        return self._emits_planes_cached

    def grad_buffer(self) -> torch.Tensor:
        return self.P.grad

    def grad_buckets(self):
        return self._stage_buckets

    def backward_staged(self, reduced_moments: torch.Tensor):
        """the backward, window by window: yields each bucket of the flat gradient as soon as the launches that
        complete it are enqueued (parallel.dp_step issues its collective right there)"""
        if len(self._windows) == 1:
            self.backward(reduced_moments, False)
            yield from self._stage_buckets
            return
        assert reduced_moments is self._moments
        v, M = self._masks()
        for k, (l0, lc) in enumerate(self._windows):
            H.operator_backward_evd_heads(self.shape, self._params, self.problem, self._x_cur, self.f_g, self.Tf_g,
                                          self.mask_kind, v, M, self._moments, True, self.scratch, self._loss,
                                          self._grads, self.ws, l0, lc, 1.0, self.path, l_offset=self.l_off)
            yield self._stage_buckets[k]
        yield self._stage_buckets[-1]

    def grad_shard(self, lo: int, hi: int) -> torch.Tensor:
        q = (hi - lo) // self.world
        off = lo // self.world  # buckets are disjoint and each a multiple of world: the slices do not overlap
        return self._gshard[off:off + q]

    def apply_shard(self, lo: int, hi: int, g: torch.Tensor, grad_scale: float) -> None:
        self._state_sharded = self.P.state_sharded = True
        H.rmsprop_ema_step(self.P.flat[lo:hi], g, self.P.sq[lo:hi], self.P.ema[lo:hi], self._lr_now, self.alpha,
                           self.eps, self._decay_now, grad_scale)

    def param_buffer(self) -> torch.Tensor:
        return self.P.flat

    def a2a_buffers(self, lo: int, hi: int):
        q = (hi - lo) // self.world
        return self._a2a_recv[lo:hi].view(self.world, q), self._a2a_send[lo:hi].view(self.world, q)

    def sum_slices(self, recv: torch.Tensor, out: torch.Tensor) -> None:
        torch.sum(recv, dim=0, out=out)

    def gather_optimizer_state(self) -> None:
        """rs_ag keeps the RMSprop square averages and the EMA shadow on the rank that updates them (1/world of each
        bucket): make both complete on every rank (before an evaluation under EMA weights, a checkpoint)."""
        if not self._state_sharded or self.comm is None:
            return
        from .parallel import shard_range
        for buf in (self.P.ema, self.P.sq):
            for lo, hi in self._stage_buckets:
                slo, shi = shard_range(lo, hi, self.comm.rank, self.world)
                self.comm.all_gather_flat(buf[lo:hi], buf[slo:shi])
        self._state_sharded = self.P.state_sharded = False

    def state_dict(self, ema: bool = False, gather: bool = True) -> Dict[str, torch.Tensor]:
        """The model in the reference's state_dict layout (keys model.base.ws.{i}, ...). ema=True: the EMA weights (what
        the reference evaluates and checkpoints).
        COLLECTIVES ARE EXPLICIT: with heads sharded (hp) the whole model is gathered - call on EVERY rank
        (`if rank == 0: tr.state_dict()` would wait for the other ranks forever) - or pass gather=False for this rank's
        (L / world, ...) slices only; with a sharded optimiser state (dp, rs_ag / a2a) ema=True raises until
        gather_optimizer_state() has been called (on every rank: it is the collective)."""
        sd = self.P.state_dict(ema)
        if not self.hp or not gather:
            return sd
        for n in self.P.names:
            sd[n] = self.gather_heads_tensor(sd[n])
        return sd

    def gather_heads_tensor(self, v: torch.Tensor) -> torch.Tensor:
        """heads sharded: every rank's (n_r, ...) slice of a per-head tensor -> the (L, ...) tensor, on every rank (a
        collective: call on every rank). Ranks own L // world or L // world + 1 heads (parallel.head_range): the
        all-gather moves equally long blocks, a rank with fewer heads pads its block."""
        if not self.hp:
            return v
        from .parallel import head_range
        v = v.contiguous()
        blk = v
        if v.shape[0] != self._Lb:
            blk = torch.zeros((self._Lb,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            blk[:v.shape[0]] = v
        out = torch.empty((self.world,) + tuple(blk.shape), dtype=v.dtype, device=v.device)
        self.comm.all_gather(out, blk)
        if self.Lg % self.world == 0:
            return out.view((self.Lg,) + tuple(v.shape[1:]))
        return torch.cat([out[r, :head_range(self.Lg, r, self.world)[1]] for r in range(self.world)])

    def begin_apply(self) -> None:
        self._lr_now, self._decay_now = self._advance_schedule()

    def apply(self, lo: int, hi: int, grad_scale: float) -> None:
        self.P.version += 1
        H.rmsprop_ema_step(self.P.flat[lo:hi], self.P.grad[lo:hi], self.P.sq[lo:hi], self.P.ema[lo:hi], self._lr_now,
                           self.alpha, self.eps, self._decay_now, grad_scale)

    def gather_buffers(self):
        return self.gath, self._fTf_blk

    def after_gather(self) -> None:
        """the ranks' gathered [f | Tf] blocks -> the (B, L) arrays the backward reads; beyond 1024 rows (where the
        backward wants per-chunk partial moments) those leave the same launch (nsvd_evd_gather_head_blocks)"""
        v, _ = self._masks()
        want_partials = not self.direct_moments
        H.evd_gather_head_blocks(self.gath, self.Lg, self.f_g, self.Tf_g, self.mask_kind, v,
                                 self.scratch if want_partials else None)
        self._partials_ready = want_partials

    def prefetch(self) -> None:
        """Draw batch t+1 and write its features into the other (workspace, x) set - no weight is involved."""
        if not (self.overlap and self._own_batch):
            return
        H.operator_sample_features(self.shape, self._params, self.problem, self.sample_key, self.batches_drawn,
                                   self._x_other, self._ws_other, True, self.path)
        self.batches_drawn += 1
        self._next_ready = True

    # -- the step -----------------------------------------------------------------------------
    def forward_backward(self, x: torch.Tensor, features_ready: bool = False, take_step: bool = True) -> None:
        """forward + exchange + loss + backward (+ the optimiser step). take_step=False: gradients only - with
        samples sharded over several ranks these are THIS RANK'S gradients of the global loss (the moments are
        exchanged, the gradient buffer is not: sum P.grad over the ranks for the global gradient)."""
        from . import parallel
        self._features_ready = bool(features_ready)
        with torch.cuda.device(self.device):
            if self.hp:
                parallel.hp_step(self, self.comm, x, take_step, probe=self.probe, sync=self.sync_collectives)
            else:
                parallel.dp_step(self, self.comm, x, take_step, exchange=self.dp_exchange, probe=self.probe,
                                 sync=self.sync_collectives)

    def _refresh_loss(self, want_moments: bool = False) -> None:
        # direct-moment steps do not produce the moment vector (and, with heads sharded, not the loss either):
        # evaluate for the last batch now
        if self._loss_stale or (want_moments and self._moments_stale):
            v, M = self._masks()
            keep = self._loss.clone() if not self._loss_stale else None
            H.evd_loss_fused(self.f_g, self.Tf_g, self.mask_kind, v, M, self._moments, self._loss, None, self.scratch)
            if keep is not None:
                self._loss.copy_(keep)  # the value the step's own kernels left stays the one reported
            self._loss_stale = self._moments_stale = False

    @property
    def loss(self) -> torch.Tensor:
        """{loss, operator term, metric term} of the last step's batch (single GPU, batches of <= 1024 rows: written by
        the step's own kernels every step, as the reference computes it every step: methods/nestedlora.py:92-94)."""
        self._refresh_loss()
        return self._loss

    @property
    def moments(self) -> torch.Tensor:
        """(2 L^2 + 1) moment vector of the last step's batch."""
        self._refresh_loss(want_moments=True)
        return self._moments

    def _advance_schedule(self):
        """(lr, ema decay) of the step being taken; advances the scheduler / torch_ema counters."""
        lr = cosine_lr(self.lr, self.t, self.num_iters) if self.use_sched else self.lr
        self.num_updates += 1
        decay = min(self.ema_decay, (1 + self.num_updates) / (10 + self.num_updates))
        self.t += 1
        return lr, decay

    def step(self, x: Optional[torch.Tensor] = None) -> None:
        """One optimiser step on the internal sampler's next batch, or on ``x`` (tests, external samplers)."""
        self._own_batch = x is None and self.device_sampler
        if x is not None:
            # an externally supplied batch; a batch prepared ahead (overlap) stays in the other set and is consumed by
            # the next internal step, so that the sampler's stream is the same with and without overlap
            self.forward_backward(x)
        elif self.device_sampler:
            # one launch draws the batch and writes its features; the forward then skips the feature stage
            if self._next_ready:  # produced under the previous step's collective: switch sets
                self.ws, self._ws_other = self._ws_other, self.ws
                self.x, self._x_other = self._x_other, self.x
                self._next_ready = False
            else:
                with torch.cuda.device(self.device):
                    if self.state is not None:  # the batch counter is read on the device: base + state.step
                        H.operator_sample_features_dev(self.shape, self._params, self.problem, self.sample_key,
                                                       self.batches_drawn - self.t, self.state, self.x, self.ws, True,
                                                       self.path)
                    else:
                        H.operator_sample_features(self.shape, self._params, self.problem, self.sample_key,
                                                   self.batches_drawn, self.x, self.ws, True, self.path)
                self.batches_drawn += 1
            self.forward_backward(self.x, features_ready=True)
        else:
            self.forward_backward(self.sample())

    # -- HIP-graph replay -------------------------------------------------------------------------
    def capture_graph(self, steps: int = 2) -> "GraphedSteps":
        """Capture `steps` consecutive step() calls (internal device sampler) into one HIP graph and return the
        replayable object. Needs device_schedule=True (otherwise the captured launches would carry one step's learning
        rate, EMA decay and batch counter as frozen arguments) and a single process (no collective inside). With the
        next batch prepared by the backward's guest workgroups the two (workspace, x) sets alternate, so `steps` must be
        even; one eager step is taken first when no batch is prepared yet."""
        return GraphedSteps(self, steps)

    # -- evaluation (methods/spectrum.py:29-102 under EMA weights, operator/__init__.py:108) ----
    @torch.no_grad()
    def spectrum(self, lim: float, val_eps: float, use_ema: bool = True, chunk: int = 8192):
        """Rayleigh-quotient spectrum on the uniform grid arange(-lim, lim, val_eps)^D (main_pde.py:121-130), over ALL
        heads of the model (heads sharded: a collective - call on every rank)."""
        if use_ema and self._state_sharded:
            raise RuntimeError("spectrum(use_ema=True): the EMA shadow is sharded over the ranks (dp_exchange rs_ag / "
                               "a2a); call gather_optimizer_state() on every rank first")
        if self.hp:
            # heads sharded: the metric needs every head (cov / quad are (L, L) over ALL heads) - gather the model (a
            # collective: call on every rank) and evaluate all heads here
            sd = self.state_dict(ema=use_ema)
            full = FlatParams(self.full_shape, self.device, with_state=False)
            full.load_state_dict(sd)
            return _spectrum_of(self.full_shape, full.pack(full.flat, True), self.problem, self.device, self.path, lim,
                                val_eps, chunk)
        return _spectrum_of(self.shape, self._ema_params if use_ema else self._params, self.problem, self.device,
                            self.path, lim, val_eps, chunk)


@torch.no_grad()
def _spectrum_of(shape, params, problem, device, path, lim, val_eps, chunk):
    """compute_spectrum_evd's accumulation (methods/spectrum.py:56-86) for one parameter set on the uniform grid
    arange(-lim, lim, val_eps)^D (main_pde.py:121-130)."""
    import numpy as np
    D, L = shape.D, shape.L
    with torch.cuda.device(device):
        ax = torch.from_numpy(np.arange(-lim, lim, val_eps))  # numpy's arange values (start + i * step)
        # np.meshgrid's default 'xy' indexing, flattened row-major: the reference's point order
        grid = torch.stack([g.reshape(-1) for g in torch.meshgrid(*(D * [ax]), indexing="xy" if D > 1 else "ij")],
                           dim=1).float().to(device)
        # float64 accumulators (the reference's are float32, methods/spectrum.py:60-61: nsvd.h on why these are not)
        cov = torch.zeros((L, L), dtype=torch.float64, device=device)
        quad = torch.zeros_like(cov)
        ws = H.new_workspace(shape, min(chunk, grid.shape[0]), device)
        for i in range(0, grid.shape[0], chunk):
            xb = grid[i:i + chunk]
            # (a ragged last chunk: hip_ops.operator_forward pads it for the MFMA kernels and drops the padding)
            wsb = ws if xb.shape[0] == chunk else H.new_workspace(shape, xb.shape[0], device)
            f, Tf = H.operator_forward(shape, params, problem, xb, wsb, False, path)
            H.spectrum_accumulate(f, Tf, xb, problem.sigma, bool(problem.use_importance), lim, cov, quad)
    n = grid.shape[0]
    cov, quad = cov.cpu() / n, quad.cpu() / n
    return dict(cov=cov, quad=quad, eigvals=torch.diag(quad) / torch.diag(cov), norms=torch.diag(cov))


class GraphedSteps:
    """`steps` training steps of a FusedTrainer as ONE captured HIP graph: the loop body of
    examples/operator/__init__.py:55-74 (sample, forward, loss, backward, RMSprop, cosine schedule, EMA) replayed
    with one host call per `steps` steps and no launch arguments that depend on the step (hip_ops.StepState)."""

    def __init__(self, tr: FusedTrainer, steps: int = 2):
        if tr.state is None:
            raise H.NsvdError("capture_graph needs FusedTrainer(device_schedule=True)")
        if tr.multi:
            raise H.NsvdError("capture_graph: single-process steps only (no collective inside the graph)")
        if not tr.device_sampler:
            raise H.NsvdError("capture_graph needs the device sampler (the batch must be drawn inside the graph)")
        if steps < 1 or (tr.guest_features and steps % 2):
            raise ValueError("steps must be positive, and even when the workspace sets alternate (guest features)")
        self.tr, self.steps = tr, int(steps)
        with torch.cuda.device(tr.device):
            if tr.guest_features and not tr._next_ready:
                tr.step()  # the first batch is drawn by a launch of its own; from then on by the previous backward
            # without guest features every step draws its batch with nsvd_operator_sample_features_dev (below)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            self.graph = torch.cuda.CUDAGraph()
            before = (tr.t, tr.num_updates, tr.batches_drawn)
            with torch.cuda.stream(side):
                with torch.cuda.graph(self.graph, stream=side):
                    for _ in range(self.steps):
                        tr.step()
            torch.cuda.current_stream().wait_stream(side)
            # capture only recorded the launches: the host counters go back, replay() advances them
            tr.t, tr.num_updates, tr.batches_drawn = before
        # host-side decisions frozen into the graph (whether the forward may read the bf16 planes of the weights, the
        # device-resident schedule position): replay() re-validates them
        self._version = tr.P.version
        self._expect = (before[1] - before[0], before[2] - before[0])

    def replay(self, n: int = 1) -> None:
        """n replays = n * steps optimiser steps"""
        tr = self.tr
        if tr.P.version != self._version:
            # P.load / load_state_dict / apply changed the weights behind the graph's back: a captured bf16x3 forward
            # would go on reading the planes of the OLD weights
            raise H.NsvdError("GraphedSteps.replay: the parameters were changed from the host since the capture "
                              "(P.version moved): capture the steps again")
        if (tr.num_updates - tr.t, tr.batches_drawn - tr.t) != self._expect:
            # the step counter itself lives on the device (nsvd_step_state, advanced by eager and replayed steps alike);
            # what the graph froze are the OFFSETS of the EMA and sampler counters from it
            raise H.NsvdError("GraphedSteps.replay: the trainer's counters (t / num_updates / batches_drawn) were "
                              "changed from the host since the capture: capture the steps again")
        for _ in range(n):
            self.graph.replay()
        k = n * self.steps
        tr.t += k
        tr.num_updates += k
        tr.batches_drawn += k
        tr._loss_stale = not tr._direct_loss
        tr._moments_stale = True
