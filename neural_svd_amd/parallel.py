"""Multi-GPU NestedLoRA steps: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
across the xGMI links; "gloo" on CPU for the tests.

Two shardings, each written ONCE here against a small compute-backend protocol. `trainer.FusedTrainer` is the
HIP backend (every protocol method is a kernel launch on the current stream); `tests/test_dp_gloo.py` plugs the
CPU oracle into the same functions, and `tests/test_multirank_gpu.py` runs the HIP backend with two ranks on one
device - so the exchange sequence the product runs is the one the tests execute.

  "dp"  samples sharded (what BASELINE.json's north_star describes). What crosses ranks per step:
  1. the moment vector [lam_f1 | lam_f2 | mean f.Tf] = 2 L^2 + 1 floats  (all-reduce, mean) - the only
     cross-sample coupling of the loss (methods/nestedlora.py:89);
  2. the flat gradient buffer, P floats, in buckets. The backend produces the gradient in STAGES (head windows of
     the backward: the heads of ParallelMLP share nothing but the input), and a bucket's collective is issued the
     moment the launches that complete it are enqueued, so it runs under the remaining backward launches; the next
     batch and its features (they depend on no weight) are produced while the first one is in flight. Two exchange
     algorithms, both leaving every rank with bit-identical parameters:
       "allreduce"  all-reduce(sum) per bucket, then the identical RMSprop/EMA pass over the whole bucket on every
                    rank (optimiser on bucket k while bucket k+1 is on the wire);
       "rs_ag"      reduce-scatter(sum) per bucket, RMSprop/EMA on this rank's 1/world of the bucket only (the
                    square averages and the EMA shadow stay sharded; `gather` them for evaluation / checkpoints),
                    then all-gather of the updated parameters: the same bytes on the wire, 1/world of the optimiser
                    traffic per rank, and the two-phase form SURVEY 5 asks for on fully connected xGMI.
       "a2a"        the same two phases spelled as all-to-alls: every rank sends slice j of the bucket straight to
                    rank j (point-to-point over the link the two share: all seven links of a rank at once, no ring),
                    adds the `world` slices it received in rank order, steps its slice, and sends the updated slice
                    straight to every rank. Whatever RCCL's reduce-scatter / all-gather do internally, this one IS
                    the direct algorithm.
  "hp"  heads sharded (SURVEY 8(e) "alternative worth measuring"): the L heads of ParallelMLP share nothing but
     the input, so rank r owns the consecutive heads `head_range(L, r, W)` (L // W each, the first L % W ranks one
     more: ANY L >= W, e.g. the reference scripts' --neigs 36 / 55 on 8 GPUs) - weights, gradients and optimiser
     state are not replicated and there is NO gradient traffic; every rank evaluates its heads on the whole global
     batch and the only exchange is one all-gather of the ranks' packed [f | Tf] blocks (2 B ceil(L / W) floats per
     rank: equally long blocks, a short rank's tail unused), under which the next batch's features are produced. On
     xGMI the 18.9 MB gradient exchange of "dp" costs about as much as the whole compute step; the all-gather is
     ~0.5 MB.
The reference itself has no live distributed code (tools/generic.py:65-180 is never imported).

Backend protocol (all methods enqueue work and return immediately on the HIP backend):
  forward(x)                       evaluate the operator on this rank's rows / heads (keeps f, Tf)
  backward(moments, take_step)     gradients of the local rows given the GLOBAL moments (None: the backend's own
                                   batch is the global batch); take_step=True: the optimiser step may be fused in
  local_moments() -> tensor        dp: (2 L^2 + 1) moments of this rank's rows, to be averaged in place
  grad_buffer() -> tensor          dp: the flat local gradient, to be summed in place
  backward_staged(moments)         dp: generator; each item (lo, hi) says "the launches that complete the gradient
                                   elements [lo, hi) have just been enqueued"; the items partition the buffer
  begin_apply()                    dp: advance the lr / EMA schedules once per step
  apply(lo, hi, grad_scale)        dp: optimiser step on the elements [lo, hi) from grad_buffer()[lo:hi]
  grad_shard(lo, hi) -> tensor     dp rs_ag: (hi - lo) / world floats receiving this rank's reduced slice of a bucket
  apply_shard(lo, hi, g, scale)    dp rs_ag: optimiser step on the elements [lo, hi) from the reduced slice g
  param_buffer() -> tensor         dp rs_ag / a2a: the flat parameter buffer (gathered in place, bucket by bucket)
  a2a_buffers(lo, hi) -> (r, s)    dp a2a: two (world, (hi - lo) / world) staging buffers of a bucket (received gradient
                                   slices; the updated slice replicated once per destination)
  sum_slices(recv, out)            dp a2a: out = recv[0] + recv[1] + ... (rank order)
  gather_buffers() -> (out, inp)   hp: inp = this rank's block of 2 B head_block(L, world) floats beginning with its
                                   packed f (B, n_r) | Tf (B, n_r); out (world, inp.numel()) receives every rank's
  after_gather()                   hp: gathered blocks -> the (B, L) arrays the backward reads
  prefetch()                       work for the NEXT step that depends on no weight (issued under a collective)

Collectives are asynchronous with late waits by default (a bucket on the wire under the remaining backward launches)
or blocking on the compute stream (`sync=True`: no second stream and no cross-stream event - each of which idles the
compute stream for ~10 us on this runtime - and no overlap either); what an RCCL world of one measures for both, and
why bench.py lets the first hardware contact choose: DESIGN.md 6. `Communicator.force_exchange` runs the whole exchange
sequence in a world of ONE (every collective a one-rank library call): how a one-GPU box exercises the RCCL calls.

Diagnostics (bench.py's `comm` block): `CommProbe` brackets every wait on a collective with two events on the
compute stream - what they measure is the time the compute stream sat idle for that collective, i.e. its EXPOSED
cost - and `Communicator.stub = True` turns every collective into a no-op so that the same step can be timed
compute-only.
"""
from __future__ import annotations

import contextlib
import os
import time
from collections import OrderedDict
from typing import Optional

import torch
import torch.distributed as dist

DP_EXCHANGES = ("allreduce", "rs_ag", "a2a")


class _Done:
    """work handle of a collective that did not run (stub mode)"""

    def wait(self):
        return True


class Communicator:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.stub = False  # True: collectives are skipped (compute-only timing of the multi-rank code path)
        # True: a world of ONE still runs the whole exchange sequence (every collective a one-rank call into the
        # library) instead of the plain single-GPU step: how a box with one GPU exercises the RCCL calls, their
        # argument shapes and their stream ordering (tests/test_multirank_gpu.py)
        self.force_exchange = False

    @property
    def multi(self) -> bool:
        """does a step go through the exchange sequence?"""
        return self.world > 1 or self.force_exchange

    @classmethod
    def from_env(cls, device: Optional[torch.device] = None, backend: Optional[str] = None,
                 timeout_s: Optional[float] = None) -> "Communicator":
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run).
        timeout_s: the process group's collective timeout (a rank stuck in a collective exits instead of waiting for
        the library's default of ten minutes or more)."""
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
            kw = {}
            if timeout_s is not None:
                import datetime
                kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, init_method="env://", **kw)
        return cls()

    def all_reduce_mean(self, t: torch.Tensor) -> None:
        if self.stub:
            return
        if self.backend == "nccl":  # RCCL averages inside the collective: no extra kernel
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
            return
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.div_(self.world)

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = False):
        if self.stub:
            return _Done() if async_op else None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def reduce_scatter_sum(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out (n / world) = this rank's slice of the element-wise sum of every rank's inp (n)."""
        if self.stub:
            return _Done() if async_op else None
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_gather_flat(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out (n) = concatenation over ranks of inp (n / world). `inp` may be out's own slice of this rank (in
        place, what RCCL expects of a sharded-parameter gather); gloo gets a private copy of it."""
        if self.stub:
            return _Done() if async_op else None
        if self.backend != "nccl":
            inp = inp.clone()
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out[j] = rank j's inp[self.rank]; inp, out: (world, q) contiguous."""
        if self.stub:
            return _Done() if async_op else None
        return dist.all_to_all_single(out.view(-1), inp.reshape(-1), group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out: (world, *inp.shape) contiguous; out[r] = rank r's inp. async_op: returns the work handle (wait()
        orders the current stream after the collective) so that independent kernels can be enqueued meanwhile."""
        if self.stub:
            return _Done() if async_op else None
        # concatenated-along-dim-0 view: the one output shape both RCCL and gloo accept
        return dist.all_gather_into_tensor(out.view(-1, *inp.shape[1:]), inp.contiguous(), group=self.group,
                                           async_op=async_op)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(t, src=src, group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)

    def _host_device(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def max_float(self, v: float) -> float:
        t = torch.tensor([v], dtype=torch.float64, device=self._host_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def count_ranks(self) -> int:
        """the world size as the collective library itself sees it: a sum of ones over the ranks"""
        t = torch.ones(1, dtype=torch.float32, device=self._host_device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(round(float(t.item())))

    def close(self) -> None:
        if dist.is_initialized():
            dist.destroy_process_group()


class CommProbe:
    """Stopwatch for the waits on collectives: two events on the compute stream around each wait. The first is
    reached when everything enqueued before the wait has finished, the second when the collective has, so their
    distance is the time the compute stream was idle for that collective - its exposed cost (a collective hidden
    under other kernels reads ~0). On CPU tensors (tests) the host clock is used."""

    def __init__(self, device=None):
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.spans = OrderedDict()
        self.steps = 0

    @contextlib.contextmanager
    def span(self, name: str):
        if self.cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            yield
            e1.record()
            self.spans.setdefault(name, []).append((e0, e1))
        else:
            t0 = time.perf_counter()
            yield
            self.spans.setdefault(name, []).append(time.perf_counter() - t0)

    def step_done(self) -> None:
        self.steps += 1

    def summary(self) -> "OrderedDict[str, float]":
        """name -> mean exposed microseconds per step"""
        if self.cuda:
            torch.cuda.synchronize()
        out = OrderedDict()
        n = max(self.steps, 1)
        for name, v in self.spans.items():
            tot = sum(a.elapsed_time(b) for a, b in v) * 1e3 if self.cuda else sum(v) * 1e6
            out[name] = tot / n
        return out


@contextlib.contextmanager
def _nospan(_name):
    yield


def _world(comm: Optional[Communicator]) -> int:
    return comm.world if comm is not None else 1


def _multi(comm: Optional[Communicator]) -> bool:
    return comm is not None and comm.multi


def head_range(L: int, rank: int, world: int):
    """heads sharded: (first head, number of heads) of `rank` - L // world each, the first L % world ranks one more
    (the layout nsvd_evd_gather_head_blocks unpacks, include/nsvd.h)"""
    if L < world:
        raise ValueError(f"head-parallel needs at least one head per rank: L = {L} < world size {world}")
    base, rem = divmod(L, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def head_block(L: int, world: int) -> int:
    """heads per all-gather block = the largest rank's head count (blocks are equally long)"""
    return -(-L // world)


def shard_range(lo: int, hi: int, rank: int, world: int):
    """rank's slice of the bucket [lo, hi) (equal slices: reduce-scatter / all-gather need them)"""
    n = hi - lo
    if n % world != 0:
        raise ValueError(f"bucket of {n} elements does not split over {world} ranks")
    q = n // world
    return lo + rank * q, lo + (rank + 1) * q


def dp_step(backend, comm: Optional[Communicator], x_local, take_step: bool = True, exchange: str = "allreduce",
            probe: Optional[CommProbe] = None, sync: bool = False) -> None:
    """One SAMPLE-SHARDED NestedLoRA step (see the module docstring for the protocol). With one rank this is the
    plain step (the backend is free to fuse the optimiser into its backward).
    sync: every collective is a blocking call (torch runs those on the CURRENT stream: no second stream, no
    cross-stream event per bucket - each of which idles the compute stream for ~10 us on this runtime - and no overlap
    either); the default issues them asynchronously and waits as late as possible."""
    span = probe.span if probe is not None else _nospan
    backend.forward(x_local)
    world = _world(comm)
    if not _multi(comm):
        backend.backward(None, take_step)
        return
    mom = backend.local_moments()
    with span("moments_allreduce"):
        comm.all_reduce_mean(mom)                  # exchange 1: 2 L^2 + 1 floats
    if not take_step:
        backend.backward(mom, False)
        return
    if exchange not in DP_EXCHANGES:
        raise ValueError(f"dp exchange must be one of {DP_EXCHANGES}")

    def issue(name, fn, *a):
        """the collective now: blocking inside its own span (sync), or asynchronous -> (span name, work handle)"""
        if sync:
            with span(name):
                fn(*a)
            return None
        return name, fn(*a, async_op=True)

    def wait(h):
        if h is not None:
            with span(h[0]):
                h[1].wait()

    if exchange == "allreduce":
        # exchange 2: a bucket's all-reduce goes out as soon as the launches completing it are enqueued
        works = []
        for k, (lo, hi) in enumerate(backend.backward_staged(mom)):
            works.append((lo, hi, issue(f"grad_bucket{k}_allreduce_wait", comm.all_reduce_sum,
                                        backend.grad_buffer()[lo:hi])))
        backend.prefetch()                         # next batch + features under the first bucket
        backend.begin_apply()
        for lo, hi, h in works:
            wait(h)
            backend.apply(lo, hi, 1.0 / world)     # optimiser on bucket k while bucket k+1 is on the wire
    elif exchange == "rs_ag":
        rs = []
        for k, (lo, hi) in enumerate(backend.backward_staged(mom)):
            shard = backend.grad_shard(lo, hi)
            rs.append((lo, hi, shard, issue(f"grad_bucket{k}_reduce_scatter_wait", comm.reduce_scatter_sum, shard,
                                            backend.grad_buffer()[lo:hi])))
        backend.prefetch()
        backend.begin_apply()
        params = backend.param_buffer()
        ag = []
        for k, (lo, hi, shard, h) in enumerate(rs):
            wait(h)
            slo, shi = shard_range(lo, hi, comm.rank, world)
            backend.apply_shard(slo, shi, shard, 1.0 / world)   # 1/world of the optimiser traffic per rank
            ag.append(issue(f"param_bucket{k}_all_gather_wait", comm.all_gather_flat, params[lo:hi], params[slo:shi]))
        for h in ag:
            wait(h)
    else:  # "a2a"
        stages = []
        for k, (lo, hi) in enumerate(backend.backward_staged(mom)):
            recv, send = backend.a2a_buffers(lo, hi)
            q = (hi - lo) // world
            stages.append((lo, hi, recv, send, issue(f"grad_bucket{k}_all_to_all_wait", comm.all_to_all, recv,
                                                     backend.grad_buffer()[lo:hi].view(world, q))))
        backend.prefetch()
        backend.begin_apply()
        params = backend.param_buffer()
        gathers = []
        for k, (lo, hi, recv, send, h) in enumerate(stages):
            wait(h)
            slo, shi = shard_range(lo, hi, comm.rank, world)
            shard = backend.grad_shard(lo, hi)
            backend.sum_slices(recv, shard)                      # rank order: the same sum wherever it is formed
            backend.apply_shard(slo, shi, shard, 1.0 / world)
            send.copy_(params[slo:shi].unsqueeze(0).expand_as(send))
            q = (hi - lo) // world
            gathers.append(issue(f"param_bucket{k}_all_to_all_wait", comm.all_to_all, params[lo:hi].view(world, q),
                                 send))
        for h in gathers:
            wait(h)
    if probe is not None:
        probe.step_done()


def hp_step(backend, comm: Optional[Communicator], x_global, take_step: bool = True,
            probe: Optional[CommProbe] = None, sync: bool = False) -> None:
    """One HEAD-SHARDED NestedLoRA step: every rank owns L/world heads (weights, gradients, optimiser state:
    nothing is replicated, no gradient traffic) and evaluates them on the WHOLE global batch. The only exchange
    is an all-gather of the rank's (B, L/world) blocks of f and Tf; the moments, the loss and d loss / d f of the
    local heads are then computed locally from the gathered (B, L) arrays."""
    span = probe.span if probe is not None else _nospan
    backend.forward(x_global)
    if _multi(comm):
        out, inp = backend.gather_buffers()
        if sync:  # blocking call on the current stream (see dp_step); the next batch then rides in the backward
            with span("f_Tf_all_gather_wait"):
                comm.all_gather(out, inp)
        else:
            work = comm.all_gather(out, inp, async_op=True)
            if take_step:
                backend.prefetch()
            with span("f_Tf_all_gather_wait"):
                work.wait()
        backend.after_gather()
    backend.backward(None, take_step)
    if probe is not None:
        probe.step_done()
