"""Drop-in surface of the reference's ``methods/`` package for the NestedLoRA (NeuralSVD) path,
backed by the HIP C ABI.  Same names, argument meaning and return values as the reference:

    get_evd_method(args, 'neuralsvd', model)                     methods/general.py:7-39
    NestedLoRA(model, neigs, step, sort, sequential)             methods/nestedlora.py:167-267
    NestedLoRALossFunctionEVD.apply(f, Tf, f1, f2, vmask, mmask) methods/nestedlora.py:67-111
    get_sequential_nesting_masks / get_joint_nesting_masks       methods/nestedlora.py:40-54

``compute_loss_operator(operator, x, importance)`` takes the fused HIP path when ``operator`` is
this package's OperatorWrapper(NegativeHamiltonian) and ``self.model`` its WaveFunctions. Any other
callable with the reference's contract ``operator(model, x, importance) -> (Tf, f)`` is CALLED (its
model evaluations and the loss still run on the HIP kernels, ``apply_operator``). The loss Function
takes ``f1, f2`` that are chunks of ``f`` (one fused call) or independent tensors of any row counts
(the reference's lower seam, :84 "f1 and f2 must be independent"). There is no CPU / eager fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import hip_ops as H
from ._lib import NsvdError


# ------------------------------------------------------------------------------------- masks
def get_sequential_nesting_masks(L: int, set_first_mode_const: bool = False):
    if set_first_mode_const:
        L += 1
    return torch.ones(L), torch.triu(torch.ones(L, L))


def get_joint_nesting_masks(weights: np.ndarray, set_first_mode_const: bool = False):
    """weights: per-index step weights (sum 1); v = reverse cumulative sum, M = min(v_i, v_j)."""
    tail = np.cumsum(np.asarray(weights, dtype=np.float64)[::-1])[::-1]
    v = list(tail)
    if set_first_mode_const:
        v = [v[0]] + v
    v = torch.tensor(np.array(v)).float()
    return v, torch.minimum(v.view(-1, 1), v.view(1, -1)).float()


def joint_step_weights(neigs: int, step: int) -> np.ndarray:
    """weight 1/#ends on every step-th index and on the last one (methods/nestedlora.py:185-190)."""
    ends = list(range(step, neigs + 1, step))
    if neigs not in ends:
        ends.append(neigs)
    w = np.zeros(neigs)
    w[np.array(ends) - 1] = 1.0
    return w / w.sum()


def nesting_masks(neigs: int, sequential: bool, step: int = 1):
    """-> (vector_mask, matrix_mask, mask_kind for the kernels)."""
    if sequential:
        v, M = get_sequential_nesting_masks(neigs)
        return v, M, H.MASK_SEQUENTIAL
    v, M = get_joint_nesting_masks(joint_step_weights(neigs, step))
    # step-1 joint masks are regenerated in registers; they must agree bit for bit with the table
    gen = (torch.arange(neigs, 0, -1, dtype=torch.float32) / float(neigs))
    kind = H.MASK_JOINT if (step == 1 and torch.equal(gen, v)) else H.MASK_CUSTOM
    return v, M, kind


def _is_chunk_of(f: torch.Tensor, f1: torch.Tensor, f2: torch.Tensor) -> bool:
    if f.dim() != 2 or f1.dim() != 2 or f2.dim() != 2 or not f.is_contiguous():
        return False
    B, L = f.shape
    B1 = (B + 1) // 2
    return (f1.shape == (B1, L) and f2.shape == (B - B1, L) and f1.data_ptr() == f.data_ptr()
            and (f2.numel() == 0 or f2.data_ptr() == f.data_ptr() + B1 * L * f.element_size())
            and f1.is_contiguous() and f2.is_contiguous())


_MASK_KIND_CACHE: dict = {}


def _mask_kind(vector_mask: torch.Tensor, matrix_mask: torch.Tensor) -> int:
    """Which of the library's built-in nesting masks these are (the kernels then build them in registers). Decided once
    per (tensor objects, in-place version): the comparison below costs milliseconds on a many-core host."""
    key = (id(vector_mask), id(matrix_mask), vector_mask._version, matrix_mask._version, vector_mask.numel())
    hit = _MASK_KIND_CACHE.get(key)
    if hit is not None and hit[0]() is vector_mask and hit[1]() is matrix_mask:
        return hit[2]
    kind = _mask_kind_uncached(vector_mask, matrix_mask)
    if len(_MASK_KIND_CACHE) > 64:
        _MASK_KIND_CACHE.clear()
    import weakref
    _MASK_KIND_CACHE[key] = (weakref.ref(vector_mask), weakref.ref(matrix_mask), kind)
    return kind


def _mask_kind_uncached(vector_mask: torch.Tensor, matrix_mask: torch.Tensor) -> int:
    L = vector_mask.numel()
    v, M = vector_mask.detach().float().cpu(), matrix_mask.detach().float().cpu()
    if torch.equal(v, torch.ones(L)) and torch.equal(M, torch.triu(torch.ones(L, L))):
        return H.MASK_SEQUENTIAL
    gen = torch.arange(L, 0, -1, dtype=torch.float32) / float(L)
    if torch.equal(v, gen) and torch.equal(M, torch.minimum(gen.view(-1, 1), gen.view(1, -1))):
        return H.MASK_JOINT
    return H.MASK_CUSTOM


class NestedLoRALossFunctionEVD(torch.autograd.Function):
    """loss = -2 mean_b sum_l v_l f_bl Tf_bl + sum(M * lam_f1 * lam_f2); gradients to f, f1, f2 only
    (Tf gets none: the operator is assumed self-adjoint, reference :108-111).

    f1, f2 = torch.chunk(f, 2) (what compute_loss_operator passes, reference :263): one moment call + one loss call on
    f, the three gradients summed in d loss / d f. INDEPENDENT f1, f2 (reference :84; what
    compute_loss_kernel(split_batch=True) passes, :239-244 - any row counts B, B1, B2): the operator term and its
    gradient -(4 / B) v Tf come from the kernels on (f, Tf) with a zero matrix mask; the metric term from the kernels
    on X = [f1; f2] with TX = 0, the shorter of the two padded with zero rows to the longer's B' rows - that scales
    its lam by B_short / B', so the metric term and BOTH its gradients (2 / B1) f1 (M lam_f2), (2 / B2) f2 (M lam_f1)
    come out of the kernels times B_short / B' and are multiplied back. Gradients return separately to f, f1, f2
    (autograd adds them when the caller passed one tensor twice)."""

    @staticmethod
    def forward(ctx, f, Tf, f1, f2, vector_mask, matrix_mask):
        kind = _mask_kind(vector_mask, matrix_mask)
        dev = f.device
        v = vector_mask.to(dev).float().contiguous() if kind == H.MASK_CUSTOM else None
        M = matrix_mask.to(dev).float().contiguous() if kind == H.MASK_CUSTOM else None
        ctx.kind, ctx.v, ctx.M = kind, v, M
        ctx.chunked = _is_chunk_of(f, f1, f2)
        if ctx.chunked:
            fd, Tfd = f.detach(), Tf.detach().contiguous()
            moments = H.evd_moments(fd, Tfd, kind, v)
            loss, _ = H.evd_loss_grad(fd, Tfd, kind, v, M, moments, want_grad=False)
            ctx.save_for_backward(fd, Tfd, moments)
            return loss[0].clone()
        if f.dim() != 2 or Tf.shape != f.shape or f1.dim() != 2 or f2.dim() != 2 or \
                f1.shape[1] != f.shape[1] or f2.shape[1] != f.shape[1] or min(f1.shape[0], f2.shape[0]) < 1:
            raise NsvdError(f"NestedLoRALossFunctionEVD (HIP): f, Tf (B, L) and f1 (B1, L), f2 (B2, L) expected, got "
                            f"{tuple(f.shape)}, {tuple(Tf.shape)}, {tuple(f1.shape)}, {tuple(f2.shape)} (the (B, L, O) "
                            f"form of the reference is not on this path)")
        L = f.shape[1]
        B1, B2 = f1.shape[0], f2.shape[0]
        fd, Tfd = f.detach().float().contiguous(), Tf.detach().float().contiguous()
        # operator term: the caller's vector mask, a zero matrix mask
        vv = vector_mask.to(dev).float().contiguous()
        Z = torch.zeros(L, L, dtype=torch.float32, device=dev)
        mom_op = H.evd_moments(fd, Tfd, H.MASK_CUSTOM, vv)
        loss_op, _ = H.evd_loss_grad(fd, Tfd, H.MASK_CUSTOM, vv, Z, mom_op, want_grad=False)
        # metric term on [f1; f2] (zero rows up to equal halves), no operator term
        Bm = max(B1, B2)
        X = torch.zeros(2 * Bm, L, dtype=torch.float32, device=dev)
        X[:B1] = f1.detach()
        X[Bm:Bm + B2] = f2.detach()
        TX = torch.zeros_like(X)
        mom = H.evd_moments(X, TX, kind, v)
        loss_m, _ = H.evd_loss_grad(X, TX, kind, v, M, mom, want_grad=False)
        ctx.rows = (B1, B2, Bm)
        ctx.metric_scale = float(Bm) / float(min(B1, B2))
        ctx.vv, ctx.Z = vv, Z
        ctx.save_for_backward(fd, Tfd, mom_op, X, TX, mom)
        return (loss_op[1] + ctx.metric_scale * loss_m[2]).to(f.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.chunked:
            fd, Tfd, moments = ctx.saved_tensors
            _, df = H.evd_loss_grad(fd, Tfd, ctx.kind, ctx.v, ctx.M, moments, 1.0, True)
            df = df * grad_output
            # the f1 / f2 contributions are already summed into df (they are views of f)
            return df, None, None, None, None, None
        fd, Tfd, mom_op, X, TX, mom = ctx.saved_tensors
        B1, B2, Bm = ctx.rows
        _, df = H.evd_loss_grad(fd, Tfd, H.MASK_CUSTOM, ctx.vv, ctx.Z, mom_op, 1.0, True)
        _, dX = H.evd_loss_grad(X, TX, ctx.kind, ctx.v, ctx.M, mom, 1.0, True)
        dX = dX * (grad_output * ctx.metric_scale)
        return df * grad_output, None, dX[:B1], dX[Bm:Bm + B2], None, None


class NestedLoRALossFunctionSVD(torch.autograd.Function):
    """reference methods/nestedlora.py:114-164 (it has no caller there; the Function itself is complete):
        loss = -2 mean_b sum_l v_l f_bl Tg_bl + sum(M * lam_f * lam_g),   lam_f = f^T f / B1, lam_g = g^T g / B2
        d loss / d f = -(2 / B1) v * Tg    + (2 / B1) f @ (M * lam_g)
        d loss / d g = -(2 / B2) v * Tadjf + (2 / B2) g @ (M * lam_f)
    On the EVD kernels: with X = [f; g] and TX = [Tg; Tadjf] (B1 = B2 rows each) the two chunks of X are f and g, so
    the moment kernel yields lam_f, lam_g and the metric term, and nsvd_evd_loss_grad's d loss / d X is exactly
    [d loss / d f; d loss / d g] (-(4 / 2B1) v TX = -(2 / B1) v TX). Only the VALUE of the operator term differs
    (the reference leaves g . Tadjf out of it): it is taken from a second moment call on [Tg; 0]."""

    @staticmethod
    def forward(ctx, f, Tg, g, Tadjf, vector_mask, matrix_mask):
        if f.dim() != 2 or f.shape != g.shape or Tg.shape != f.shape or Tadjf.shape != g.shape:
            raise NsvdError("NestedLoRALossFunctionSVD (HIP): f, Tg, g, Tadjf must be (B, L) with B1 == B2")
        kind = _mask_kind(vector_mask, matrix_mask)
        dev = f.device
        v = vector_mask.to(dev).float().contiguous() if kind == H.MASK_CUSTOM else None
        M = matrix_mask.to(dev).float().contiguous() if kind == H.MASK_CUSTOM else None
        B1 = f.shape[0]
        X = torch.cat([f.detach(), g.detach()]).float().contiguous()
        TX = torch.cat([Tg.detach(), Tadjf.detach()]).float().contiguous()
        moments = H.evd_moments(X, TX, kind, v)
        loss3, dX = H.evd_loss_grad(X, TX, kind, v, M, moments, 1.0, True)
        TX0 = TX.clone()
        TX0[B1:].zero_()
        op = H.evd_moments(X, TX0, kind, v)[-1]  # mean over the 2 B1 rows of sum_l v_l f Tg
        ctx.B1 = B1
        ctx.save_for_backward(dX)
        return (-4.0 * op + loss3[2]).to(f.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        (dX,) = ctx.saved_tensors
        dX = dX * grad_output
        return dX[:ctx.B1], None, dX[ctx.B1:], None, None, None


class _OperatorFn(torch.autograd.Function):
    """(Tf, f) = operator(model, x, importance) through nsvd_operator_forward; the backward is
    nsvd_operator_backward (gradient through f only)."""

    @staticmethod
    def forward(ctx, x, method, operator, prob, *params):
        model = method.model
        shape = model.shape
        packed = model.packed_params()
        ws = H.new_workspace(shape, x.shape[0], x.device)
        need_grad = any(ctx.needs_input_grad)  # grad mode is off inside Function.forward; ask the ctx
        f, Tf = H.operator_forward(shape, packed, prob, x, ws, save_for_backward=need_grad, path=method.path)
        ctx.model, ctx.prob, ctx.path = model, prob, method.path
        ctx.ws = ws if need_grad else None
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(Tf)
        return Tf, f

    @staticmethod
    def backward(ctx, dTf, df):
        (x,) = ctx.saved_tensors
        model = ctx.model
        grads = model.grad_buffers()
        H.operator_backward(model.shape, model.packed_params(), ctx.prob, x, df.contiguous(), grads.packed, ctx.ws,
                            ctx.path)
        return (None, None, None, None) + tuple(grads.tensors)


_FOREIGN_LOGGED = False


class NestedLoRA(nn.Module):
    def __init__(self, model, neigs, step=1, sort=False, sequential=False, path: int = H.PATH_AUTO):
        self.name = "nestedlora"
        super().__init__()
        self.neigs, self.sort, self.sequential = neigs, sort, sequential
        self.eigvals = None
        self.sort_indices = None
        self.vector_mask, self.matrix_mask, _ = nesting_masks(neigs, bool(sequential), step)
        self.model = model
        self.path = path

    def forward(self, *args):
        out = self.model(*args)
        if self.sort_indices is not None and self.training:
            return out[:, self.sort_indices, ...]
        return out

    def register_eigvals(self, eigvals):
        self.eigvals = torch.Tensor(eigvals)
        self.sort_indices = torch.sort(self.eigvals)[1].flip(0)

    def reset_eigvals(self):
        self.eigvals = None
        self.sort_indices = None

    def _compute_loss(self, *args, evd=True) -> torch.Tensor:
        if not evd:  # (f, Tg, g, Tadjf); like the reference, neither compute_loss_* reaches this with evd=False
            return NestedLoRALossFunctionSVD.apply(*args, self.vector_mask, self.matrix_mask)
        return NestedLoRALossFunctionEVD.apply(*args, self.vector_mask, self.matrix_mask)

    def compute_loss_operator(self, operator, x, importance=None, evd: bool = True):
        if not evd:
            raise NotImplementedError
        Tf, f = self.apply_operator(operator, x, importance)
        f1, f2 = torch.chunk(f, 2)
        loss = self._compute_loss(f, Tf, f1, f2, evd=True)
        return loss, dict(f=f, Tf=Tf, eigvals=None)

    def compute_loss_kernel(self, get_approx_kernel_op, x, importance, split_batch: bool, evd: bool = True):
        """reference methods/nestedlora.py:230-252 (no caller in the reference; kept for the interface).
        ``get_approx_kernel_op(x)(self, x, importance)`` returns (Kf, f) built from ``self(x)`` (which is
        differentiable here); the loss itself runs on the HIP EVD kernels.
        split_batch=True (operator term on the first half only, f2 = model(x2) an independent batch) is the same
        kernel call on f = [f1; f2], Tf = [(B/B1) Kf1; 0]: the operator term and its gradient then carry the
        reference's 1/B1, the metric term is unchanged."""
        if not evd:
            raise NotImplementedError
        if split_batch:
            x1, x2 = torch.chunk(x, 2)
            Kf1, f1 = get_approx_kernel_op(x2)(self, x1, importance=importance)
            f2 = self.model(x2)
            B1, B2 = f1.shape[0], f2.shape[0]
            f = torch.cat([f1, f2]).contiguous()
            Tf = torch.cat([Kf1.detach() * (float(B1 + B2) / B1), torch.zeros_like(f2)]).contiguous()
            loss = self._compute_loss(f, Tf, *torch.chunk(f, 2), evd=True)
            return loss, dict(f=f1, Tf=Kf1, eigvals=None)
        Kf, f = get_approx_kernel_op(x)(self, x, importance=importance)
        f = f.contiguous()
        f1, f2 = torch.chunk(f, 2)
        loss = self._compute_loss(f, Kf.contiguous(), f1, f2, evd=True)
        return loss, dict(f=f, Tf=Kf, eigvals=None)

    def apply_operator(self, operator, x, importance=None):
        """Tf, f = operator(self, x, importance).
        This package's OperatorWrapper (finite-difference / exact Hamiltonian with Gaussian or no importance): ONE fused
        forward on the HIP kernels (nsvd_operator_forward) and its backward.
        Any other callable with the reference's contract `operator(model, x, importance=None) -> (Tf, f)`
        (examples/__init__.py:7-9, methods/nestedlora.py:254-267; SURVEY 8(b): "otherwise fall back to python operator
        + fused loss kernel"): it is CALLED, with this module as `model` - `self(x)` runs on nsvd_model_forward and is
        differentiable through nsvd_model_backward - and its (Tf, f) go to the HIP loss kernels. Still no CPU path: the
        model evaluations inside the foreign operator and the loss are the HIP kernels, only the operator's own
        elementwise algebra is torch's."""
        from .operators import OperatorWrapper, fused_problem_of
        if not isinstance(operator, OperatorWrapper):
            if not callable(operator):
                raise NsvdError("compute_loss_operator: operator must be callable as operator(model, x, importance)")
            global _FOREIGN_LOGGED
            if not _FOREIGN_LOGGED:
                _FOREIGN_LOGGED = True
                print(f"neural_svd_amd: operator {type(operator).__name__} is not this package's OperatorWrapper: the "
                      f"fused operator kernel is bypassed (it is called with the HIP model; the loss runs on the HIP "
                      f"EVD kernels)")
            Tf, f = operator(self, x, importance=importance) if importance is not None else operator(self, x)
            if f.dim() != 2 or Tf.shape != f.shape or f.shape[1] != self.neigs:
                raise NsvdError(f"operator returned Tf {tuple(Tf.shape)}, f {tuple(f.shape)}: expected two "
                                f"(B, {self.neigs}) tensors")
            return Tf.float().contiguous(), f.float().contiguous()
        if not operator.fused(importance):
            # Laplace / uniform / any other density: the wrapper applies the stencil around HIP model evaluations
            Tf, f = operator.apply_stencil(self, x, importance)
            return Tf.contiguous(), f.contiguous()
        prob = fused_problem_of(operator, importance, self.model)
        x = x.reshape(x.shape[0], -1).float().contiguous()
        Tf, f = _OperatorFn.apply(x, self, operator, prob, *self.model.trainable_tensors())
        if self.sort_indices is not None and self.training:
            # register_eigvals (reference methods/nestedlora.py:195-210): forward() hands the operator the model's
            # columns in eigenvalue order, so Tf and f both come out permuted. The operator acts head by head: the
            # same result is the permutation applied to its two outputs (autograd routes d loss / d f back through it)
            idx = self.sort_indices.to(f.device)
            Tf, f = Tf[:, idx].contiguous(), f[:, idx].contiguous()
        return Tf, f


def get_evd_method(args, method_name, model):
    if method_name != "neuralsvd":
        raise NotImplementedError(f"{method_name}: only 'neuralsvd' (NestedLoRA) is built on this path")
    return NestedLoRA(model=model, neigs=args.neigs, step=args.loss.neuralsvd.step, sort=args.sort,
                      sequential=args.loss.neuralsvd.sequential)
