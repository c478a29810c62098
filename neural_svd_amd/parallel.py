"""Data parallelism for the NestedLoRA step: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) across the xGMI links; "gloo" on CPU for the tests.

Two shardings are implemented (FusedTrainer(parallelism=...)):

  "dp"  samples sharded (what BASELINE.json's north_star describes). What crosses ranks per step:
  1. the moment vector [lam_f1 | lam_f2 | mean f.Tf] = 2 L^2 + 1 floats  (all-reduce, mean) - the only
     cross-sample coupling of the loss (methods/nestedlora.py:89);
  2. the flat gradient buffer, P floats (all-reduce, sum; the optimiser kernel folds the 1/world).
     Every rank then applies the identical RMSprop/EMA update, so parameters stay bit-identical.
  "hp"  heads sharded (SURVEY 8(e) "alternative worth measuring"): the L heads of ParallelMLP share nothing but
     the input, so rank r owns heads [r L/W, (r+1) L/W) - weights, gradients and optimiser state are not
     replicated and there is NO gradient traffic; every rank evaluates its heads on the whole global batch and
     the only exchange is one all-gather of 2 B L floats (f and Tf). On xGMI the 18.9 MB gradient ring
     all-reduce of "dp" costs about as much as the whole compute step; the all-gather is ~0.5 MB.
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

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out: (world, *inp.shape) contiguous; out[r] = rank r's inp. async_op: returns the work handle (wait()
        orders the current stream after the collective) so that independent kernels can be enqueued meanwhile."""
        # concatenated-along-dim-0 view: the one output shape both RCCL and gloo accept
        return dist.all_gather_into_tensor(out.view(-1, *inp.shape[1:]), inp.contiguous(), group=self.group,
                                           async_op=async_op)

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


def hp_step(backend, comm: Optional[Communicator], x_global, state) -> dict:
    """One HEAD-PARALLEL NestedLoRA step: every rank owns L/world heads (weights, gradients, optimiser state:
    nothing is replicated, no gradient traffic) and evaluates them on the WHOLE global batch. The only
    exchange is an all-gather of the rank's (B, L/world) blocks of f and Tf; the L x L moments, the loss and
    d loss / d f of the local heads are then computed locally from the gathered (B, L) arrays.

    backend.forward(x) -> (f_loc, Tf_loc, ctx); backend.moments(f, Tf) -> (2L^2+1,);
    backend.backward_from_moments(ctx, f, Tf, moments) -> (loss, flat local grad); backend.apply(grad, 1.0)
    """
    f_loc, Tf_loc, ctx = backend.forward(x_global)
    if comm is not None and comm.world > 1:
        W, (B, Ll) = comm.world, f_loc.shape
        buf = torch.empty((W, 2, B, Ll), dtype=f_loc.dtype, device=f_loc.device)
        comm.all_gather(buf, torch.stack([f_loc, Tf_loc]).contiguous())
        both = buf.permute(1, 2, 0, 3).reshape(2, B, W * Ll).contiguous()  # head index = rank * Ll + local head
        f, Tf = both[0], both[1]
    else:
        f, Tf = f_loc, Tf_loc
    mom = backend.moments(f, Tf)
    loss, g = backend.backward_from_moments(ctx, f, Tf, mom)
    backend.apply(g, 1.0)
    return dict(loss=loss, moments=mom, grad=g, f=f, Tf=Tf)


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
