"""Data parallelism for the NestedLoRA step: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) across the xGMI links; "gloo" on CPU for the tests.

What crosses ranks per step:
  1. the moment vector [lam_f1 | lam_f2 | mean f.Tf] = 2 L^2 + 1 floats  (all-reduce, mean) - the only
     cross-sample coupling of the loss (methods/nestedlora.py:89);
  2. the flat gradient buffer, P floats (all-reduce, sum; the optimiser kernel folds the 1/world).
Every rank then applies the identical RMSprop/EMA update, so parameters stay bit-identical.
The reference itself has no live distributed code (tools/generic.py:65-180 is never imported).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


class Communicator:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    @classmethod
    def from_env(cls, device: Optional[torch.device] = None, backend: Optional[str] = None) -> "Communicator":
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run)."""
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, init_method="env://", **kw)
        return cls()

    def all_reduce_mean(self, t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.div_(self.world)

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)

    def max_float(self, v: float) -> float:
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self) -> None:
        if dist.is_initialized():
            dist.destroy_process_group()


def dp_step(backend, comm: Optional[Communicator], x_local, state) -> dict:
    """One data-parallel NestedLoRA step written against an abstract compute ``backend`` so the
    exchange logic is testable on CPU (tests inject the oracle; the product injects nothing: the
    FusedTrainer runs this same sequence on HIP kernels).

    backend.forward(x) -> (f, Tf, ctx); backend.moments(f, Tf) -> (2L^2+1,) tensor;
    backend.loss_grad(f, Tf, moments) -> (loss, df); backend.backward(ctx, df) -> flat grad;
    backend.apply(flat_grad, grad_scale) -> None
    """
    f, Tf, ctx = backend.forward(x_local)
    mom = backend.moments(f, Tf)
    if comm is not None and comm.world > 1:
        comm.all_reduce_mean(mom)
    loss, df = backend.loss_grad(f, Tf, mom)
    g = backend.backward(ctx, df)
    scale = 1.0
    if comm is not None and comm.world > 1:
        comm.all_reduce_sum(g)
        scale = 1.0 / comm.world
    backend.apply(g, scale)
    return dict(loss=loss, moments=mom, grad=g, grad_scale=scale)
