"""Multi-GPU NestedLoRA steps: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
across the xGMI links; "gloo" on CPU for the tests.

Two shardings, each written ONCE here against a small compute-backend protocol. `trainer.FusedTrainer` is the
HIP backend (every protocol method is a kernel launch on the current stream); `tests/test_dp_gloo.py` plugs the
CPU oracle into the same functions, and `tests/test_multirank_gpu.py` runs the HIP backend with two ranks on one
device - so the exchange sequence the product runs is the one the tests execute.

  "dp"  samples sharded (what BASELINE.json's north_star describes). What crosses ranks per step:
  1. the moment vector [lam_f1 | lam_f2 | mean f.Tf] = 2 L^2 + 1 floats  (all-reduce, mean) - the only
     cross-sample coupling of the loss (methods/nestedlora.py:89);
  2. the flat gradient buffer, P floats (all-reduce, sum), cut into buckets on head boundaries of W_0 (89 % of the
     bytes): every bucket's collective is issued asynchronously, the next batch and its features (they depend on
     no weight) are produced while the first one is in flight, and the optimiser pass over bucket k runs while
     bucket k+1 is on the wire. Every rank applies the identical RMSprop/EMA update: parameters stay bit-identical.
  "hp"  heads sharded (SURVEY 8(e) "alternative worth measuring"): the L heads of ParallelMLP share nothing but
     the input, so rank r owns heads [r L/W, (r+1) L/W) - weights, gradients and optimiser state are not
     replicated and there is NO gradient traffic; every rank evaluates its heads on the whole global batch and
     the only exchange is one all-gather of 2 B L floats (f and Tf), under which the next batch's features are
     produced. On xGMI the 18.9 MB gradient all-reduce of "dp" costs about as much as the whole compute step; the
     all-gather is ~0.5 MB.
The reference itself has no live distributed code (tools/generic.py:65-180 is never imported).

Backend protocol (all methods enqueue work and return immediately on the HIP backend):
  forward(x)                       evaluate the operator on this rank's rows / heads (keeps f, Tf)
  backward(moments, take_step)     gradients of the local rows given the GLOBAL moments (None: the backend's own
                                   batch is the global batch); take_step=True: the optimiser step may be fused in
  local_moments() -> tensor        dp: (2 L^2 + 1) moments of this rank's rows, to be averaged in place
  grad_buffer() -> tensor          dp: the flat local gradient, to be summed in place
  grad_buckets() -> [(lo, hi)]     dp: contiguous element ranges of grad_buffer(), exchange order
  begin_apply()                    dp: advance the lr / EMA schedules once per step
  apply(lo, hi, grad_scale)        dp: optimiser step on the elements [lo, hi)
  gather_buffers() -> (out, inp)   hp: out (world, *inp.shape) receives every rank's packed [f | Tf] block
  after_gather()                   hp: gathered blocks -> the (B, L) arrays the backward reads
  prefetch()                       work for the NEXT step that depends on no weight (issued under a collective)
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
        self.backend = dist.get_backend(group)

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
        if self.backend == "nccl":  # RCCL averages inside the collective: no extra kernel
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
            return
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.div_(self.world)

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = False):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out: (world, *inp.shape) contiguous; out[r] = rank r's inp. async_op: returns the work handle (wait()
        orders the current stream after the collective) so that independent kernels can be enqueued meanwhile."""
        # concatenated-along-dim-0 view: the one output shape both RCCL and gloo accept
        return dist.all_gather_into_tensor(out.view(-1, *inp.shape[1:]), inp.contiguous(), group=self.group,
                                           async_op=async_op)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(t, src=src, group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)

    def max_float(self, v: float) -> float:
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self) -> None:
        if dist.is_initialized():
            dist.destroy_process_group()


def _world(comm: Optional[Communicator]) -> int:
    return comm.world if comm is not None else 1


def dp_step(backend, comm: Optional[Communicator], x_local, take_step: bool = True) -> None:
    """One SAMPLE-SHARDED NestedLoRA step (see the module docstring for the protocol). With one rank this is the
    plain step (the backend is free to fuse the optimiser into its backward)."""
    backend.forward(x_local)
    world = _world(comm)
    if world == 1:
        backend.backward(None, take_step)
        return
    mom = backend.local_moments()
    comm.all_reduce_mean(mom)                      # exchange 1: 2 L^2 + 1 floats
    backend.backward(mom, False)
    if not take_step:
        return
    grad = backend.grad_buffer()
    works = [(lo, hi, comm.all_reduce_sum(grad[lo:hi], async_op=True))   # exchange 2, bucket by bucket
             for lo, hi in backend.grad_buckets()]
    backend.prefetch()                             # next batch + features under the first bucket
    backend.begin_apply()
    for lo, hi, work in works:
        work.wait()
        backend.apply(lo, hi, 1.0 / world)         # optimiser on bucket k while bucket k+1 is on the wire


def hp_step(backend, comm: Optional[Communicator], x_global, take_step: bool = True) -> None:
    """One HEAD-SHARDED NestedLoRA step: every rank owns L/world heads (weights, gradients, optimiser state:
    nothing is replicated, no gradient traffic) and evaluates them on the WHOLE global batch. The only exchange
    is an all-gather of the rank's (B, L/world) blocks of f and Tf; the moments, the loss and d loss / d f of the
    local heads are then computed locally from the gathered (B, L) arrays."""
    backend.forward(x_global)
    if _world(comm) > 1:
        out, inp = backend.gather_buffers()
        work = comm.all_gather(out, inp, async_op=True)
        if take_step:
            backend.prefetch()
        work.wait()
        backend.after_gather()
    backend.backward(None, take_step)
