"""Kernel operators for ``NestedLoRA.compute_loss_kernel`` (methods/nestedlora.py:230-252).

The reference defines only the consumer contract - ``get_approx_kernel_op(x)(model, x, importance) -> (Kf, f)`` - and
ships no kernel operator. This module provides the dense one of the kernel-operator configuration (SURVEY.md 8,
cfg4): a fixed symmetric PSD matrix K on N points z_j, minibatches are point INDICES drawn with replacement,

    f  = model(x) = net(z[x]),        Kf = K[x][:, x_ref] @ model(x_ref) / len(x_ref)

(the model handed to NestedLoRA takes indices: ``op.index_model(net)`` wraps a coordinate network, so that the
contract's own ``self.model(x2)`` call works on an index batch),

with the (B x B) . (B x L) contraction done by ``nsvd_kernel_apply`` on the fp32 MFMA (gathered rows of K, the batch
scattered into the index space). No gradient flows through Kf (the EVD loss function returns none for it).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip_ops as H


class IndexedModel(nn.Module):
    """model(idx) = net(points[idx]): lets a coordinate network consume minibatches of point indices."""

    def __init__(self, net: nn.Module, points: torch.Tensor):
        super().__init__()
        self.net = net
        self.register_buffer("points", points, persistent=False)

    def forward(self, idx):
        return self.net(self.points[idx.to(torch.int64)])


class DenseKernelOperator:
    """K: (N, N) float32 symmetric PSD; points: (N, D) coordinates the model is evaluated at."""

    def __init__(self, K: torch.Tensor, points: torch.Tensor):
        if K.dim() != 2 or K.shape[0] != K.shape[1] or points.shape[0] != K.shape[0]:
            raise ValueError("K must be (N, N) and points (N, D)")
        if not K.is_cuda:
            raise H.NsvdError("DenseKernelOperator: K must live on the GPU (no CPU path)")
        N = K.shape[0]
        ld = (N + 63) // 64 * 64
        self.N = N
        # rows are read in whole 64-float chunks: keep a zero-padded copy with a leading dimension of ceil64(N)
        self.K = torch.zeros((N, ld), dtype=torch.float32, device=K.device)
        self.K[:, :N] = K.float()
        self.points = points.to(K.device).float().contiguous()

    def get_approx_kernel_op(self, x_ref: torch.Tensor):
        """x_ref: (B2,) int64 indices of the reference batch -> op(model, x, importance=None) -> (Kf, f)."""
        x_ref = x_ref.to(torch.int64).contiguous()

        def op(model, x, importance=None):
            if importance is not None:
                raise NotImplementedError("DenseKernelOperator: importance weights are not defined for an index batch")
            x = x.to(torch.int64).contiguous()
            f = model(x)  # an index model (index_model below)
            same = x.data_ptr() == x_ref.data_ptr() and x.numel() == x_ref.numel()
            with torch.no_grad():
                f_ref = f.detach() if same else model(x_ref).detach()
                Kf = H.kernel_apply(self.K, self.N, x, x_ref, f_ref.contiguous(), 1.0 / x_ref.numel())
            return Kf, f
        return op

    def index_model(self, net: nn.Module) -> IndexedModel:
        return IndexedModel(net, self.points)

    def sample_indices(self, batch_size: int, generator=None) -> torch.Tensor:
        return torch.randint(self.N, (batch_size,), device=self.K.device, generator=generator)


def synthetic_psd_kernel(N: int = 10000, rank: int = 256, dim: int = 16, seed: int = 0, device="cuda:0"):
    """cfg4's operator: z_j ~ N(0, I_dim), K = A A^T / rank + 1e-3 I with A ~ randn(N, rank); seeded on the host."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(N, dim, generator=g)
    A = torch.randn(N, rank, generator=g).to(device)
    K = A @ A.T / rank
    K.diagonal().add_(1e-3)
    return DenseKernelOperator(K, z.to(device))


class FusedKernelTrainer:
    """The kernel-operator training step (NestedLoRA.compute_loss_kernel with split_batch = False on a
    DenseKernelOperator, then loss.backward(); RMSprop (+ cosine schedule); EMA - reference methods/nestedlora.py:230-252
    with the optimiser of examples/utils.py:50-57) as a fixed sequence of C-ABI calls on flat parameter buffers: index
    batch -> gather of the coordinates -> model evaluation (nsvd_model_forward) -> Kf = K[x][:, x] f / B
    (nsvd_kernel_apply) -> moments (nsvd_evd_moments) -> d loss / d f, backward and the optimiser step inside
    the backward kernels (nsvd_model_backward_evd_step). No torch autograd, no torch.optim; what torch still does is
    draw the indices and gather the coordinates. The counterpart of trainer.FusedTrainer for BASELINE configs[3].

    Multi-GPU (comm: parallel.Communicator with world > 1): HEADS sharded, as trainer.FusedTrainer's "hp". Rank r owns
    the heads [r L / W, (r + 1) L / W) - weights, gradients, optimiser state: nothing replicated, no gradient traffic.
    Every rank draws the same index batch (equal generator seeds), evaluates its heads on it, and applies K to ITS
    columns of f only (Kf[:, l] = K[x][:, x] f[:, l] / B needs no other head): the MFMA work of the step is split W
    ways. One all-gather of the packed (2, B, L / W) block [f | Kf] per step (any L >= W: the first L % W ranks own one head more) (2 B L floats in total: 4 MB at cfg4) is
    the only exchange; moments, loss gradient and backward of the local heads are then local."""

    def __init__(self, op: DenseKernelOperator, L: int, m: int, hidden=(128, 128), batch_size: int = 8192,
                 sequential: bool = False, step: int = 1, lr: float = 1e-4, rmsprop_decay: float = 0.99,
                 rmsprop_eps: float = 1e-8, ema_decay: float = 0.0, num_iters: int = 0, fourier_scale: float = 0.05,
                 hard_mul_const: float = 1.0, seed: int = 0, index_seed: int = 1, comm=None):
        from .nested_lowrank import nesting_masks
        from .trainer import FlatParams, reference_init
        self.op = op
        dev = op.K.device
        self.device = dev
        D = op.points.shape[1]
        self.comm = comm if comm is not None and comm.multi else None
        world = self.comm.world if self.comm is not None else 1
        rank = self.comm.rank if self.comm is not None else 0
        # any L >= world: L // world heads each, the first L % world ranks one more (parallel.head_range)
        from .parallel import head_block, head_range
        self.world, self.Lg = world, L
        self.l_off, Ll = head_range(L, rank, world)
        Lb = head_block(L, world)
        self.full_shape = H.ModelShape(L=L, D=D, m=m, hidden=tuple(hidden), has_exp_mask=False)
        self.shape = H.ModelShape(L=Ll, D=D, m=m, hidden=tuple(hidden), has_exp_mask=False)
        self.B = int(batch_size)
        if any(h != 128 for h in hidden) or self.B % 32 != 0 or not 1 <= D <= 64 or (2 * m) % 128 != 0:
            raise H.NsvdError("FusedKernelTrainer needs the MFMA model kernels: 128-wide hidden layers, batch % 32 == 0, "
                              "2 m % 128 == 0, input dimension <= 64 (nsvd_model_backward_evd_step)")
        self.P = FlatParams(self.shape, dev)
        fB0, ws0, bs0, _ = reference_init(self.full_shape, fourier_scale, None, seed)
        sl = slice(self.l_off, self.l_off + Ll)  # this rank's heads of the (identically seeded) full model
        self.P.load(fB0, [w[sl] for w in ws0], [b[sl] for b in bs0], None)
        if self.comm is not None:
            self.comm.broadcast(self.P.fourier_B, 0)  # the frozen Fourier matrix is shared whatever the ranks drew
        self._params = self.P.pack(self.P.flat, True)
        self._sq = self.P.pack(self.P.sq, False)
        self._ema = self.P.pack(self.P.ema, True) if ema_decay > 0 else None
        self.vector_mask, self.matrix_mask, self.mask_kind = nesting_masks(L, sequential, step)
        cust = self.mask_kind == H.MASK_CUSTOM
        self.v = self.vector_mask.to(dev) if cust else None
        self.M = self.matrix_mask.to(dev).contiguous() if cust else None
        self.lr, self.alpha, self.eps, self.ema_decay, self.num_iters = lr, rmsprop_decay, rmsprop_eps, ema_decay, num_iters
        self.c = float(hard_mul_const)
        self.ws = H.model_workspace(self.shape, self.B, dev)
        self.ka_ws = torch.empty(H._lib.load().nsvd_kernel_apply_workspace_bytes(int(op.N), self.B, Ll),
                                 dtype=torch.uint8, device=dev)
        # this rank's outputs packed [f | Kf] so that one all-gather moves both
        # (the all-gather block, as long as the largest rank's, begins with it: nsvd_evd_gather_head_blocks)
        self._blk = torch.zeros(2 * self.B * Lb, dtype=torch.float32, device=dev)
        self.fKf_loc = self._blk[:2 * self.B * Ll].view(2, self.B, Ll)
        self.f_loc, self.Kf_loc = self.fKf_loc[0], self.fKf_loc[1]
        if self.comm is not None:
            self.gath = torch.empty((world, 2 * self.B * Lb), dtype=torch.float32, device=dev)
            self.fKf = torch.empty((2, self.B, L), dtype=torch.float32, device=dev)
            self.f, self.Kf = self.fKf[0], self.fKf[1]
        else:
            self.f, self.Kf = self.f_loc, self.Kf_loc
        self.moments = torch.empty(2 * L * L + 1, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(3, dtype=torch.float32, device=dev)
        self.scratch = H.evd_scratch(self.B, L, dev)
        self.gen = torch.Generator(device=dev).manual_seed(index_seed)  # the same stream on every rank
        self.probe = None  # parallel.CommProbe while bench.py measures the exposed wait of the all-gather
        self.t = 0

    def step(self, idx: torch.Tensor = None) -> torch.Tensor:
        """one optimiser step on the index batch idx (or a fresh draw; sharded runs: the SAME batch on every rank);
        returns the device loss triple (no sync)"""
        from .trainer import cosine_lr
        if idx is None:
            idx = self.op.sample_indices(self.B, self.gen)
        idx = idx.to(torch.int64).contiguous()
        x = self.op.points.index_select(0, idx)
        H.model_forward(self.shape, self._params, x, self.c, self.ws, save_for_backward=True, out=self.f_loc)
        H.kernel_apply(self.op.K, self.op.N, idx, idx, self.f_loc, 1.0 / self.B, ws=self.ka_ws, out=self.Kf_loc)
        if self.comm is not None:
            if self.probe is not None:
                with self.probe.span("f_Kf_all_gather_wait"):
                    self.comm.all_gather(self.gath, self._blk)  # blocking: on the compute stream (parallel.dp_step)
            else:
                self.comm.all_gather(self.gath, self._blk)
            H.evd_gather_head_blocks(self.gath, self.Lg, self.f, self.Kf, self.mask_kind, self.v)
        # the reduced moment vector (partials + one reduction launch): at B = 8192 every workgroup of the backward
        # summing the 128 per-chunk partials of its 2 L moments itself would cost 3 x the reduction
        H.evd_moments(self.f, self.Kf, self.mask_kind, self.v, self.moments, self.scratch)
        lr = cosine_lr(self.lr, self.t, self.num_iters) if self.num_iters > 0 else self.lr
        decay = min(self.ema_decay, (2 + self.t) / (11 + self.t)) if self._ema is not None else 0.0
        opt = H.rmsprop_state(self._sq, self._ema, lr, self.alpha, self.eps, decay)
        H.model_backward_evd_step(self.shape, self._params, x, self.f, self.Kf, self.mask_kind, self.v, self.M,
                                  self.moments, True, None, self.loss, None, opt, self.ws, l_offset=self.l_off)
        if self.probe is not None:
            self.probe.step_done()
        self.t += 1
        return self.loss
