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
