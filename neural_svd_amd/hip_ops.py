"""Tensor-level wrappers over the C ABI (``include/nsvd.h``): validation, pointers, current stream.

PyTorch is used here only as plumbing (device memory + the current HIP stream). Every function
requires contiguous float32 tensors on a GPU and raises otherwise - there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (MASK_CUSTOM, MASK_JOINT, MASK_SEQUENTIAL, PATH_AUTO, PATH_FUSED, PATH_FUSED_BF16X3,  # noqa: F401
                   PATH_GENERIC,
                   POT_HARMONIC, POT_HYDROGEN, ModelDesc, NsvdError, Params, Problem, check)


# ---- which binding carries the hot-path calls: ctypes (default) or the tensor-level torch extension -----------------
_TB = None
_TB_SHAPES: dict = {}


def torch_binding():
    """The tensor-level binding neural_svd_amd/_nsvd_torch.so (csrc/torch_binding.cpp: tensors checked in C++, current
    HIP stream taken in C++, the same C ABI underneath) when NSVD_BINDING=torch, else None. Like the ctypes binding it
    fails loudly when its shared object is missing."""
    global _TB
    import os
    if os.environ.get("NSVD_BINDING", "ctypes") != "torch":
        return None
    if _TB is None:
        import importlib.util
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nsvd_torch.so")
        if not os.path.exists(path):
            raise NsvdError(f"{path} not found: NSVD_BINDING=torch needs the torch binding built "
                            f"(`make -C neural_svd_amd/csrc torch_binding`, or __graft_entry__.build())")
        _lib.load()  # libnsvd_hip.so first: the extension links against it
        spec = importlib.util.spec_from_file_location("_nsvd_torch", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.abi_version() != _lib.ABI_VERSION:
            raise NsvdError("torch binding built against another ABI version; rebuild the extension")
        _TB = mod
    return _TB


def _tb_shape(shape: "ModelShape"):
    sh = _TB_SHAPES.get(shape)
    if sh is None:
        sh = _TB_SHAPES[shape] = _TB.Shape(shape.L, shape.D, shape.m, list(shape.dims), bool(shape.has_exp_mask))
    return sh


def _tb_params(shape: "ModelShape", params: Params):
    """the extension's ParamSet twin of a packed parameter set (built once, from the tensors pack_params pinned)"""
    ps = getattr(params, "_tb", None)
    if ps is None:
        ws, bs, fB, sc = params._keepalive
        ps = params._tb = _TB.ParamSet(_tb_shape(shape), list(ws), list(bs), fB, sc)
    return ps


def _tb_problem(prob: Problem):
    q = getattr(prob, "_tb", None)
    if q is None:
        q = prob._tb = _TB.Problem(prob.potential, prob.charge_or_k, prob.eps, prob.op_scale, prob.op_shift, prob.sigma,
                                   prob.scale_kinetic, prob.hard_mul_const, bool(prob.use_importance))
    return q


def _ptr(t: Optional[torch.Tensor], name: str = "tensor", dtype: torch.dtype = torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise NsvdError(f"{name} must live on the GPU (got {t.device}); neural_svd_amd has no CPU path")
    if t.dtype != dtype:
        raise NsvdError(f"{name} must be {str(dtype).replace('torch.', '')} (got {t.dtype})")
    if not t.is_contiguous():
        raise NsvdError(f"{name} must be contiguous")
    return t.data_ptr()


def _stream() -> int:
    """The current HIP stream of the CURRENT device; every launching wrapper below runs under `_on_tensor_device`,
    which makes the tensors' device the current one first (the C library never calls hipSetDevice)."""
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """Run ``fn`` with the device of its GPU tensor arguments as the current device (so that `_stream()` is a stream
    of THAT device and the kernels launch where the pointers live); tensors of one call must share a device."""
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (tuple, list)):
                cand = [t for t in a if isinstance(t, torch.Tensor)]
            else:
                cand = [a] if isinstance(a, torch.Tensor) else []
            for t in cand:
                if not t.is_cuda:
                    continue
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise NsvdError(f"{fn.__name__}: tensors on different devices ({dev} and {t.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return guarded


@dataclass(frozen=True)
class ModelShape:
    """Shape of WaveFunctions(ParallelMLP(FourierFeatures)); ``hidden`` excludes the final width 1."""
    L: int
    D: int
    m: int
    hidden: Tuple[int, ...]
    has_exp_mask: bool = False

    @property
    def dims(self) -> Tuple[int, ...]:
        return tuple(self.hidden) + (1,)

    def desc(self) -> ModelDesc:
        d = ModelDesc()
        d.L, d.D, d.m = self.L, self.D, self.m
        dims = self.dims
        if len(dims) > _lib.NSVD_MAX_LAYERS:
            raise NsvdError(f"at most {_lib.NSVD_MAX_LAYERS} layers are supported")
        d.nlayers = len(dims)
        for i, h in enumerate(dims):
            d.dims[i] = h
        d.has_exp_mask = int(self.has_exp_mask)
        return d

    def param_shapes(self) -> List[Tuple[int, ...]]:
        """Trainable tensors in the reference's parameter order: ws..., bs..., [scales]."""
        shapes, prev = [], 2 * self.m
        for h in self.dims:
            shapes.append((self.L, h, prev))
            prev = h
        for h in self.dims:
            shapes.append((self.L, h, 1))
        if self.has_exp_mask:
            shapes.append((self.L,))
        return shapes


def make_problem(potential: int, charge_or_k: float, eps: float, op_scale: float, op_shift: float, sigma: float,
                 scale_kinetic: float = 1.0, hard_mul_const: float = 1.0, use_importance: bool = True) -> Problem:
    p = Problem()
    p.potential = int(potential)
    p.charge_or_k = float(charge_or_k)
    p.scale_kinetic = float(scale_kinetic)
    p.eps = float(eps)
    p.op_scale = float(op_scale)
    p.op_shift = float(op_shift)
    p.sigma = float(sigma)
    p.hard_mul_const = float(hard_mul_const)
    p.use_importance = int(bool(use_importance))
    return p


def pack_params(shape: ModelShape, ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor],
                fourier_B: Optional[torch.Tensor], scales: Optional[torch.Tensor]) -> Params:
    """Fill an ``nsvd_params`` struct after checking every tensor against ``shape``."""
    dims = shape.dims
    if len(ws) != len(dims) or len(bs) != len(dims):
        raise NsvdError("number of weight/bias tensors does not match the model shape")
    p = Params()
    if fourier_B is not None:
        if tuple(fourier_B.shape) != (shape.D, shape.m):
            raise NsvdError(f"fourier_B must be {(shape.D, shape.m)}, got {tuple(fourier_B.shape)}")
        p.fourier_B = _ptr(fourier_B, "fourier_B")
    prev = 2 * shape.m
    for i, h in enumerate(dims):
        if tuple(ws[i].shape) != (shape.L, h, prev):
            raise NsvdError(f"W[{i}] must be {(shape.L, h, prev)}, got {tuple(ws[i].shape)}")
        if ws[i].numel() != 0 and bs[i].numel() != shape.L * h:
            raise NsvdError(f"b[{i}] must have {shape.L * h} elements")
        p.W[i] = _ptr(ws[i], f"W[{i}]")
        p.b[i] = _ptr(bs[i], f"b[{i}]")
        prev = h
    if shape.has_exp_mask:
        if scales is None or scales.numel() != shape.L:
            raise NsvdError("scales (L,) required when has_exp_mask")
        p.scales = _ptr(scales, "scales")
    # the struct only holds raw addresses: pin the tensors to it so a temporary (e.g. `fB.to(dev)`)
    # cannot be freed and recycled by the caching allocator while the struct is still in use
    p._keepalive = (list(ws), list(bs), fourier_B, scales)
    return p


def workspace_bytes(shape: ModelShape, B: int) -> int:
    d = shape.desc()
    n = _lib.load().nsvd_workspace_bytes(C.byref(d), int(B))
    if n == 0:
        raise NsvdError("nsvd_workspace_bytes: invalid model description")
    return int(n)


def path_name(shape: ModelShape, B: int, path: int = PATH_AUTO, prob: Optional[Problem] = None) -> str:
    """"fused_mfma" / "generic" (/ "unsupported": exact-Laplacian mode on a shape the MFMA path does not take)."""
    d = shape.desc()
    if prob is not None:
        return _lib.load().nsvd_path_name_for(C.byref(d), C.byref(prob), int(B), int(path)).decode()
    return _lib.load().nsvd_path_name(C.byref(d), int(B), int(path)).decode()


def step_emits_planes(shape: ModelShape, B: int, path: int) -> bool:
    """Does a fused training step on this shape and path leave the bf16 planes of the updated weights for the next
    forward (operator_forward(planes_ready=True))? PATH_FUSED_BF16X3, MFMA shapes, no batch slices."""
    d = shape.desc()
    return bool(_lib.load().nsvd_step_emits_planes(C.byref(d), int(B), int(path)))


def new_workspace(shape: ModelShape, B: int, device) -> torch.Tensor:
    return torch.empty(workspace_bytes(shape, B), dtype=torch.uint8, device=device)


def fourier_features(x: torch.Tensor, fourier_B: torch.Tensor, eps: float, nstencil: int) -> torch.Tensor:
    B, D = x.shape
    m = fourier_B.shape[1]
    R = nstencil * B
    out = torch.empty((2 * m, R), dtype=torch.float32, device=x.device)
    rc = _lib.load().nsvd_fourier_features(_ptr(x, "x"), _ptr(fourier_B, "fourier_B"), _ptr(out), B, D, m, float(eps),
                                           int(nstencil), R, _stream())
    check(rc, "nsvd_fourier_features")
    return out


def operator_forward(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor, ws: torch.Tensor,
                     save_for_backward: bool = True, path: int = PATH_AUTO,
                     out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                     features_ready: bool = False, planes_ready: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Tf, f = operator(model, x, importance). Returns (f, Tf), each (B, L).
    planes_ready (PATH_FUSED_BF16X3): the bf16 planes of the current weights are in `ws` already, left there by the
    fused training step that produced these weights (step_emits_planes; include/nsvd.h: NSVD_W_PLANES_READY)."""
    B = x.shape[0]
    if x.dim() != 2 or x.shape[1] != shape.D:
        raise NsvdError(f"x must be (B, {shape.D})")
    if ((B % 32 != 0 or B > 8192) and not save_for_backward and not features_ready and B > 0 and path != PATH_GENERIC
            and path_name(shape, 32, path, prob) == "fused_mfma"):
        # EVALUATION batches of any size on a model the MFMA kernels take (they want a multiple of 32 rows, at most
        # 8192 of them without a backward layout): in pieces of <= 8192 rows, the last one padded with copies of its
        # last row, and the padding dropped again - a row never sees its neighbours (bit-exact row equivariance is a
        # test). The exact-Laplacian mode exists on that path only; in stencil mode it keeps the reference's validation
        # batches (not multiples of 32 rows) on the fast kernels (the generic ones compute the same even / odd stencil
        # with FMA GEMMs, several times slower).
        fs, Tfs = [], []
        wsc = None
        for i in range(0, B, 8192):
            xc = x[i:i + 8192]
            n = xc.shape[0]
            npad = (n + 31) // 32 * 32
            if npad != n:
                xc = torch.cat([xc, xc[-1:].expand(npad - n, -1)])
            if wsc is None or wsc.numel() < workspace_bytes(shape, npad):
                wsc = new_workspace(shape, npad, x.device)
            fp, Tfp = operator_forward(shape, params, prob, xc.contiguous(), wsc, False, path)
            fs.append(fp[:n])
            Tfs.append(Tfp[:n])
        fa, Tfa = torch.cat(fs), torch.cat(Tfs)
        if out is not None:
            out[0].copy_(fa)
            out[1].copy_(Tfa)
            return out
        return fa, Tfa
    if out is None:
        f = torch.empty((B, shape.L), dtype=torch.float32, device=x.device)
        Tf = torch.empty_like(f)
    else:
        f, Tf = out
    flags = int(save_for_backward) | (_lib.FEATURES_READY if features_ready else 0) | \
        (_lib.W_PLANES_READY if planes_ready else 0)
    if torch_binding() is not None:
        _TB.operator_forward(_tb_shape(shape), _tb_params(shape, params), _tb_problem(prob), x, f, Tf, ws, flags,
                             int(path))
        return f, Tf
    d = shape.desc()
    rc = _lib.load().nsvd_operator_forward(C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B, _ptr(f, "f"),
                                           _ptr(Tf, "Tf"), ws.data_ptr(), ws.numel(), flags, int(path), _stream())
    check(rc, "nsvd_operator_forward")
    return f, Tf


def operator_features(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor, ws: torch.Tensor,
                      save_for_backward: bool = True, path: int = PATH_AUTO) -> None:
    """Weight-independent prologue of operator_forward (Fourier features of x into ws)."""
    d = shape.desc()
    rc = _lib.load().nsvd_operator_features(C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), x.shape[0],
                                            ws.data_ptr(), ws.numel(), int(save_for_backward), int(path), _stream())
    check(rc, "nsvd_operator_features")


def operator_sample_features(shape: ModelShape, params: Params, prob: Problem, seed: int, offset: int,
                             x: torch.Tensor, ws: torch.Tensor, save_for_backward: bool = True,
                             path: int = PATH_AUTO) -> None:
    """x <- sigma * N(0, 1) (counter-based, keyed by seed / offset) and its features into ws, one launch."""
    B = x.shape[0]
    if torch_binding() is not None:
        _TB.operator_sample_features(_tb_shape(shape), _tb_params(shape, params), _tb_problem(prob),
                                     int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), None, x, ws,
                                     bool(save_for_backward), int(path))
        return
    d = shape.desc()
    rc = _lib.load().nsvd_operator_sample_features(C.byref(d), C.byref(params), C.byref(prob),
                                                   int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
                                                   _ptr(x, "x"), B, ws.data_ptr(), ws.numel(),
                                                   int(bool(save_for_backward)), int(path), _stream())
    check(rc, "nsvd_operator_sample_features")


def operator_sample_features_dev(shape: ModelShape, params: Params, prob: Problem, seed: int, offset_base: int,
                                 state: "StepState", x: torch.Tensor, ws: torch.Tensor,
                                 save_for_backward: bool = True, path: int = PATH_AUTO) -> None:
    """operator_sample_features whose batch counter is offset_base + state.step, read on the device."""
    B = x.shape[0]
    if torch_binding() is not None:
        _TB.operator_sample_features(_tb_shape(shape), _tb_params(shape, params), _tb_problem(prob),
                                     int(seed) & (2 ** 64 - 1), int(offset_base) & (2 ** 64 - 1), state.buf, x, ws,
                                     bool(save_for_backward), int(path))
        return
    d = shape.desc()
    rc = _lib.load().nsvd_operator_sample_features_dev(C.byref(d), C.byref(params), C.byref(prob),
                                                       int(seed) & (2 ** 64 - 1), int(offset_base) & (2 ** 64 - 1),
                                                       state.ptr, _ptr(x, "x"), B, ws.data_ptr(), ws.numel(),
                                                       int(bool(save_for_backward)), int(path), _stream())
    check(rc, "nsvd_operator_sample_features_dev")


def operator_backward(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor, df: torch.Tensor,
                      grads: Params, ws: torch.Tensor, path: int = PATH_AUTO) -> None:
    B = x.shape[0]
    if tuple(df.shape) != (B, shape.L):
        raise NsvdError(f"df must be {(B, shape.L)}")
    if torch_binding() is not None:
        _TB.operator_backward(_tb_shape(shape), _tb_params(shape, params), _tb_problem(prob), x, df,
                              _tb_params(shape, grads), ws, int(path))
        return
    d = shape.desc()
    rc = _lib.load().nsvd_operator_backward(C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B,
                                            _ptr(df, "df"), C.byref(grads), ws.data_ptr(), ws.numel(), int(path),
                                            _stream())
    check(rc, "nsvd_operator_backward")


def model_forward(shape: ModelShape, params: Params, x: torch.Tensor, hard_mul_const: float,
                  ws: torch.Tensor, save_for_backward: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B = x.shape[0]
    if out is None:
        out = torch.empty((B, shape.L), dtype=torch.float32, device=x.device)
    d = shape.desc()
    rc = _lib.load().nsvd_model_forward(C.byref(d), C.byref(params), _ptr(x, "x"), B, float(hard_mul_const),
                                        _ptr(out), ws.data_ptr(), ws.numel(), int(save_for_backward), _stream())
    check(rc, "nsvd_model_forward")
    return out


def model_backward(shape: ModelShape, params: Params, x: torch.Tensor, dout: torch.Tensor, grads: Params,
                   ws: torch.Tensor) -> None:
    d = shape.desc()
    rc = _lib.load().nsvd_model_backward(C.byref(d), C.byref(params), _ptr(x, "x"), x.shape[0], _ptr(dout, "dout"),
                                         C.byref(grads), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_model_backward")


def evd_scratch(B: int, L: int, device) -> torch.Tensor:
    n = int(_lib.load().nsvd_evd_scratch_bytes(int(B), int(L)))
    return torch.empty(max(n, 256), dtype=torch.uint8, device=device)


def evd_moments(f: torch.Tensor, Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                moments: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None) -> torch.Tensor:
    """moments = [lam_f1 (L*L) | lam_f2 (L*L) | mean_b sum_l v_l f Tf]  with f1, f2 = chunk(f, 2)."""
    B, L = f.shape
    if moments is None:
        moments = torch.empty(2 * L * L + 1, dtype=torch.float32, device=f.device)
    if scratch is None:
        scratch = evd_scratch(B, L, f.device)
    if torch_binding() is not None:
        _TB.evd_moments(f, Tf, int(mask_kind), v, moments, scratch)
        return moments
    rc = _lib.load().nsvd_evd_moments(_ptr(f, "f"), _ptr(Tf, "Tf"), B, L, int(mask_kind), _ptr(v, "v"),
                                      _ptr(moments, "moments"), scratch.data_ptr(), _stream())
    check(rc, "nsvd_evd_moments")
    return moments


def evd_loss_grad(f: torch.Tensor, Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                  M: Optional[torch.Tensor], moments: torch.Tensor, grad_scale: float = 1.0, want_grad: bool = True,
                  loss: Optional[torch.Tensor] = None,
                  df: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    B, L = f.shape
    if loss is None:
        loss = torch.empty(3, dtype=torch.float32, device=f.device)
    if want_grad and df is None:
        df = torch.empty_like(f)
    if torch_binding() is not None:
        _TB.evd_loss_grad(f, Tf, int(mask_kind), v, M, moments, float(grad_scale), loss, df if want_grad else None)
        return loss, (df if want_grad else None)
    rc = _lib.load().nsvd_evd_loss_grad(_ptr(f, "f"), _ptr(Tf, "Tf"), B, L, int(mask_kind), _ptr(v, "v"),
                                        _ptr(M, "M"), _ptr(moments, "moments"), float(grad_scale),
                                        _ptr(loss, "loss"), _ptr(df, "df") if want_grad else None, _stream())
    check(rc, "nsvd_evd_loss_grad")
    return loss, (df if want_grad else None)


def evd_loss_fused(f: torch.Tensor, Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                   M: Optional[torch.Tensor], moments: torch.Tensor, loss: torch.Tensor, df: Optional[torch.Tensor],
                   scratch: torch.Tensor, grad_scale: float = 1.0) -> None:
    """moments + loss + d loss / d f in one call (single GPU: nothing is exchanged in between)."""
    B, L = f.shape
    rc = _lib.load().nsvd_evd_loss_fused(_ptr(f, "f"), _ptr(Tf, "Tf"), B, L, int(mask_kind), _ptr(v, "v"),
                                         _ptr(M, "M"), float(grad_scale), _ptr(moments, "moments"),
                                         _ptr(loss, "loss"), _ptr(df, "df"), scratch.data_ptr(), _stream())
    check(rc, "nsvd_evd_loss_fused")


def evd_partial(f: torch.Tensor, Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                scratch: torch.Tensor) -> None:
    B, L = f.shape
    rc = _lib.load().nsvd_evd_partial(_ptr(f, "f"), _ptr(Tf, "Tf"), B, L, int(mask_kind), _ptr(v, "v"),
                                      scratch.data_ptr(), _stream())
    check(rc, "nsvd_evd_partial")


def evd_gather_heads(gathered: torch.Tensor, f: torch.Tensor, Tf: torch.Tensor, mask_kind: int,
                     v: Optional[torch.Tensor], scratch: Optional[torch.Tensor] = None) -> None:
    """gathered (world, 2, B, L_local) packed [f | Tf] blocks of the ranks -> f, Tf (B, world * L_local); with
    scratch also the partial moments of evd_partial(f, Tf) (same bits), in one launch."""
    W, two, B, Ll = gathered.shape
    if two != 2 or tuple(f.shape) != (B, W * Ll) or tuple(Tf.shape) != (B, W * Ll):
        raise NsvdError("evd_gather_heads: gathered (world, 2, B, L_local), f / Tf (B, world * L_local)")
    rc = _lib.load().nsvd_evd_gather_heads(_ptr(gathered, "gathered"), int(W), int(B), int(Ll), int(mask_kind),
                                           _ptr(v, "v"), _ptr(f, "f"), _ptr(Tf, "Tf"),
                                           scratch.data_ptr() if scratch is not None else None, _stream())
    check(rc, "nsvd_evd_gather_heads")


def evd_gather_head_blocks(gathered: torch.Tensor, L: int, f: torch.Tensor, Tf: torch.Tensor, mask_kind: int,
                           v: Optional[torch.Tensor], scratch: Optional[torch.Tensor] = None) -> None:
    """gathered (world, 2 * B * ceil(L / world)): rank w's block begins with its packed f (B, n_w) | Tf (B, n_w),
    n_w = L // world + (w < L % world) -> f, Tf (B, L) (+ the partial moments with scratch): any L >= world."""
    W = gathered.shape[0]
    B = f.shape[0]
    Lb = -(-int(L) // W)
    if gathered.dim() != 2 or gathered.shape[1] != 2 * B * Lb or tuple(f.shape) != (B, L) or tuple(Tf.shape) != (B, L):
        raise NsvdError("evd_gather_head_blocks: gathered (world, 2 B ceil(L / world)), f / Tf (B, L)")
    rc = _lib.load().nsvd_evd_gather_head_blocks(_ptr(gathered, "gathered"), int(W), int(B), int(L), int(mask_kind),
                                                 _ptr(v, "v"), _ptr(f, "f"), _ptr(Tf, "Tf"),
                                                 scratch.data_ptr() if scratch is not None else None, _stream())
    check(rc, "nsvd_evd_gather_head_blocks")


def operator_backward_evd(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor, f: torch.Tensor,
                          Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor], M: Optional[torch.Tensor],
                          moments: torch.Tensor, moments_reduced: bool, evd_scratch: Optional[torch.Tensor],
                          loss: torch.Tensor, grads: Params, ws: torch.Tensor, grad_scale: float = 1.0,
                          path: int = PATH_AUTO, l_offset: int = 0) -> None:
    """loss gradient + operator backward in one call (d loss / d f is never materialised).
    f, Tf: (B, L_total) with L_total >= shape.L; this model owns heads [l_offset, l_offset + shape.L)."""
    B = x.shape[0]
    L_total = f.shape[1]
    if tuple(Tf.shape) != (B, L_total) or L_total < shape.L or \
            (moments is not None and moments.numel() != 2 * L_total * L_total + 1):
        raise NsvdError("operator_backward_evd: f/Tf must be (B, L_total), moments 2*L_total^2+1")
    d = shape.desc()
    rc = _lib.load().nsvd_operator_backward_evd(
        C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B, _ptr(f, "f"), _ptr(Tf, "Tf"), int(mask_kind),
        _ptr(v, "v"), _ptr(M, "M"), _ptr(moments, "moments"), int(bool(moments_reduced)),
        evd_scratch.data_ptr() if evd_scratch is not None else None, int(L_total), int(l_offset),
        float(grad_scale), _ptr(loss, "loss"),
        C.byref(grads), ws.data_ptr(), ws.numel(), int(path), _stream())
    check(rc, "nsvd_operator_backward_evd")


def backward_head_window_ok(shape: ModelShape, prob: Problem, B: int, path: int, l_count: int) -> bool:
    d = shape.desc()
    return bool(_lib.load().nsvd_backward_head_window_ok(C.byref(d), C.byref(prob), int(B), int(path), int(l_count)))


def operator_backward_evd_heads(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor, f: torch.Tensor,
                                Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                                M: Optional[torch.Tensor], moments: torch.Tensor, moments_reduced: bool,
                                evd_scratch: Optional[torch.Tensor], loss: torch.Tensor, grads: Params,
                                ws: torch.Tensor, l_begin: int, l_count: int, grad_scale: float = 1.0,
                                path: int = PATH_AUTO, l_offset: int = 0) -> None:
    """operator_backward_evd for the heads [l_begin, l_begin + l_count) only (the other gradients are untouched)."""
    B = x.shape[0]
    L_total = f.shape[1]
    if tuple(Tf.shape) != (B, L_total) or L_total < shape.L or \
            (moments is not None and moments.numel() != 2 * L_total * L_total + 1):
        raise NsvdError("operator_backward_evd_heads: f/Tf must be (B, L_total), moments 2*L_total^2+1")
    d = shape.desc()
    rc = _lib.load().nsvd_operator_backward_evd_heads(
        C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B, _ptr(f, "f"), _ptr(Tf, "Tf"), int(mask_kind),
        _ptr(v, "v"), _ptr(M, "M"), _ptr(moments, "moments"), int(bool(moments_reduced)),
        evd_scratch.data_ptr() if evd_scratch is not None else None, int(L_total), int(l_offset),
        float(grad_scale), _ptr(loss, "loss"), C.byref(grads), ws.data_ptr(), ws.numel(), int(path), int(l_begin),
        int(l_count), _stream())
    check(rc, "nsvd_operator_backward_evd_heads")


class StepState:
    """The device-resident schedule state (include/nsvd.h: nsvd_step_state): step counter, CosineAnnealingLR /
    RMSprop / torch_ema constants and the scheduled values of the step being taken, in DEVICE memory - what lets a
    captured training step replay along the schedule (examples/operator/__init__.py:69-73)."""

    def __init__(self, device, lr0: float, T_max: int, alpha: float, eps: float, ema_decay: float, eta_min: float = 0.0,
                 step: int = 0):
        self.buf = torch.zeros(C.sizeof(_lib.StepState) // 8, dtype=torch.int64, device=device)
        self.args = (float(lr0), float(eta_min), int(T_max), float(alpha), float(eps), float(ema_decay))
        self.reset(step)

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    def reset(self, step: int) -> None:
        lr0, eta_min, T_max, alpha, eps, ema_decay = self.args
        with torch.cuda.device(self.buf.device):
            check(_lib.load().nsvd_step_state_init(self.ptr, lr0, eta_min, T_max, alpha, eps, ema_decay, int(step),
                                                   _stream()), "nsvd_step_state_init")

    def begin(self) -> None:
        """cur <- the scheduled values of step `step` (loop bodies whose first backward kernel does not do it)"""
        with torch.cuda.device(self.buf.device):
            check(_lib.load().nsvd_step_state_begin(self.ptr, _stream()), "nsvd_step_state_begin")

    def read(self) -> "_lib.StepState":
        """host copy (synchronises)"""
        raw = self.buf.cpu().numpy().tobytes()
        return _lib.StepState.from_buffer_copy(raw)


def rmsprop_state(sq: Params, ema: Optional[Params], lr: float, alpha: float, eps: float,
                  ema_decay: float = 0.0, state: Optional[StepState] = None) -> _lib.Rmsprop:
    """nsvd_rmsprop for operator_backward_evd_step; sq / ema are pack_params() sets in the parameters' layouts.
    state: the device-resident schedule - lr / ema_decay are then read (and advanced) on the device."""
    o = _lib.Rmsprop()
    o.sq = sq
    if ema is not None:
        o.ema = ema
    o.lr, o.alpha, o.eps, o.ema_decay = float(lr), float(alpha), float(eps), float(ema_decay)
    o.has_ema = int(ema is not None)
    o.state = state.ptr if state is not None else None
    o._keepalive = (sq, ema, state)
    return o


def _tb_rmsprop(shape: ModelShape, opt: "_lib.Rmsprop"):
    sq, ema, state = opt._keepalive
    return _TB.Rmsprop(_tb_params(shape, sq), _tb_params(shape, ema) if ema is not None else None, opt.lr, opt.alpha,
                       opt.eps, opt.ema_decay, state.buf if state is not None else None)


def operator_backward_evd_step(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor, f: torch.Tensor,
                               Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                               M: Optional[torch.Tensor], moments: torch.Tensor, moments_reduced: bool,
                               evd_scratch: Optional[torch.Tensor], loss: torch.Tensor, grads: Optional[Params],
                               opt: "_lib.Rmsprop", ws: torch.Tensor, grad_scale: float = 1.0,
                               path: int = PATH_AUTO, l_offset: int = 0) -> None:
    """operator_backward_evd + RMSprop/EMA step inside the weight-gradient kernel; params are updated in place."""
    B = x.shape[0]
    L_total = f.shape[1]
    if tuple(Tf.shape) != (B, L_total) or L_total < shape.L or \
            (moments is not None and moments.numel() != 2 * L_total * L_total + 1):
        raise NsvdError("operator_backward_evd_step: f/Tf must be (B, L_total), moments 2*L_total^2+1")
    if torch_binding() is not None:
        _TB.operator_backward_evd_step(_tb_shape(shape), _tb_params(shape, params), _tb_problem(prob), x, f, Tf,
                                       int(mask_kind), v, M, moments, bool(moments_reduced), evd_scratch, int(l_offset),
                                       float(grad_scale), loss, _tb_params(shape, grads) if grads is not None else None,
                                       _tb_rmsprop(shape, opt), ws, int(path), None, None, 0, 0)
        return
    d = shape.desc()
    rc = _lib.load().nsvd_operator_backward_evd_step(
        C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B, _ptr(f, "f"), _ptr(Tf, "Tf"), int(mask_kind),
        _ptr(v, "v"), _ptr(M, "M"), _ptr(moments, "moments"), int(bool(moments_reduced)),
        evd_scratch.data_ptr() if evd_scratch is not None else None, int(L_total), int(l_offset),
        float(grad_scale), _ptr(loss, "loss"), C.byref(grads) if grads is not None else None, C.byref(opt),
        ws.data_ptr(), ws.numel(), int(path), _stream())
    check(rc, "nsvd_operator_backward_evd_step")


def operator_backward_evd_step_window(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor,
                                      f: torch.Tensor, Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                                      M: Optional[torch.Tensor], moments: Optional[torch.Tensor], moments_reduced: bool,
                                      evd_scratch: Optional[torch.Tensor], loss: torch.Tensor, opt: "_lib.Rmsprop",
                                      ws: torch.Tensor, l_begin: int, l_count: int, last_window: bool,
                                      ev_after_chain: Optional["torch.cuda.Event"] = None, next_seed: int = 0,
                                      next_offset: int = 0, x_next: Optional[torch.Tensor] = None,
                                      ws_next: Optional[torch.Tensor] = None, grad_scale: float = 1.0,
                                      path: int = PATH_AUTO, l_offset: int = 0) -> None:
    """operator_backward_evd_step for the heads [l_begin, l_begin + l_count) of ONE fused step taken as several head
    windows (nsvd_operator_backward_evd_step_window: include/nsvd.h). Launches on the CURRENT stream; ev_after_chain is
    recorded between the window's two launches."""
    B = x.shape[0]
    L_total = f.shape[1]
    ev = None
    if ev_after_chain is not None:
        if ev_after_chain.cuda_event == 0:  # torch creates the hipEvent lazily, at the first record
            ev_after_chain.record()
        ev = ev_after_chain.cuda_event
    d = shape.desc()
    rc = _lib.load().nsvd_operator_backward_evd_step_window(
        C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B, _ptr(f, "f"), _ptr(Tf, "Tf"), int(mask_kind),
        _ptr(v, "v"), _ptr(M, "M"), _ptr(moments, "moments"), int(bool(moments_reduced)),
        evd_scratch.data_ptr() if evd_scratch is not None else None, int(L_total), int(l_offset),
        float(grad_scale), _ptr(loss, "loss"), None, C.byref(opt), ws.data_ptr(), ws.numel(), int(path),
        int(l_begin), int(l_count), int(bool(last_window)), ev, int(next_seed) & (2 ** 64 - 1),
        int(next_offset) & (2 ** 64 - 1), _ptr(x_next, "x_next"), ws_next.data_ptr() if ws_next is not None else None,
        ws_next.numel() if ws_next is not None else 0, _stream())
    check(rc, "nsvd_operator_backward_evd_step_window")


def operator_backward_evd_step_next(shape: ModelShape, params: Params, prob: Problem, x: torch.Tensor,
                                    f: torch.Tensor, Tf: torch.Tensor, mask_kind: int, v: Optional[torch.Tensor],
                                    M: Optional[torch.Tensor], moments: torch.Tensor, moments_reduced: bool,
                                    evd_scratch: Optional[torch.Tensor], loss: torch.Tensor, grads: Optional[Params],
                                    opt: "_lib.Rmsprop", ws: torch.Tensor, next_seed: int, next_offset: int,
                                    x_next: torch.Tensor, ws_next: torch.Tensor, grad_scale: float = 1.0,
                                    path: int = PATH_AUTO, l_offset: int = 0) -> None:
    """operator_backward_evd_step + operator_sample_features(next batch) in the same launches (MFMA path only)."""
    B = x.shape[0]
    L_total = f.shape[1]
    if tuple(Tf.shape) != (B, L_total) or L_total < shape.L or tuple(x_next.shape) != tuple(x.shape) or \
            (moments is not None and moments.numel() != 2 * L_total * L_total + 1):
        raise NsvdError("operator_backward_evd_step_next: f/Tf (B, L_total), moments 2*L_total^2+1, x_next like x")
    if torch_binding() is not None:
        _TB.operator_backward_evd_step(_tb_shape(shape), _tb_params(shape, params), _tb_problem(prob), x, f, Tf,
                                       int(mask_kind), v, M, moments, bool(moments_reduced), evd_scratch, int(l_offset),
                                       float(grad_scale), loss, _tb_params(shape, grads) if grads is not None else None,
                                       _tb_rmsprop(shape, opt), ws, int(path), x_next, ws_next,
                                       int(next_seed) & (2 ** 64 - 1), int(next_offset) & (2 ** 64 - 1))
        return
    d = shape.desc()
    rc = _lib.load().nsvd_operator_backward_evd_step_next(
        C.byref(d), C.byref(params), C.byref(prob), _ptr(x, "x"), B, _ptr(f, "f"), _ptr(Tf, "Tf"), int(mask_kind),
        _ptr(v, "v"), _ptr(M, "M"), _ptr(moments, "moments"), int(bool(moments_reduced)),
        evd_scratch.data_ptr() if evd_scratch is not None else None, int(L_total), int(l_offset),
        float(grad_scale), _ptr(loss, "loss"), C.byref(grads) if grads is not None else None, C.byref(opt),
        ws.data_ptr(), ws.numel(), int(path), int(next_seed) & (2 ** 64 - 1), int(next_offset) & (2 ** 64 - 1),
        _ptr(x_next, "x_next"), ws_next.data_ptr(), ws_next.numel(), _stream())
    check(rc, "nsvd_operator_backward_evd_step_next")


def model_backward_evd_step(shape: ModelShape, params: Params, x: torch.Tensor, f: torch.Tensor, Tf: torch.Tensor,
                            mask_kind: int, v: Optional[torch.Tensor], M: Optional[torch.Tensor],
                            moments: torch.Tensor, moments_reduced: bool, evd_scratch: Optional[torch.Tensor],
                            loss: torch.Tensor, grads: Optional[Params], opt: Optional["_lib.Rmsprop"],
                            ws: torch.Tensor, grad_scale: float = 1.0, l_offset: int = 0) -> None:
    """EVD loss gradient + backward of the plain model evaluation (+ the optimiser step when opt is given) after
    model_forward(save_for_backward=True); Tf is the operator output computed from f (no gradient through it).
    f, Tf: (B, L_total) over ALL heads; shape describes the heads [l_offset, l_offset + shape.L) (head sharding)."""
    B, L = f.shape
    if tuple(Tf.shape) != (B, L) or L < l_offset + shape.L or x.shape[0] != B or moments.numel() != 2 * L * L + 1:
        raise NsvdError("model_backward_evd_step: f, Tf (B, L_total); moments 2 L_total^2 + 1")
    d = shape.desc()
    rc = _lib.load().nsvd_model_backward_evd_step(
        C.byref(d), C.byref(params), _ptr(x, "x"), B, _ptr(f, "f"), _ptr(Tf, "Tf"), int(mask_kind), _ptr(v, "v"),
        _ptr(M, "M"), _ptr(moments, "moments"), int(bool(moments_reduced)),
        evd_scratch.data_ptr() if evd_scratch is not None else None, int(L), int(l_offset), float(grad_scale),
        _ptr(loss, "loss"), C.byref(grads) if grads is not None else None, C.byref(opt) if opt is not None else None,
        ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_model_backward_evd_step")


def model_workspace(shape: ModelShape, B: int, device) -> torch.Tensor:
    """workspace of model_forward / model_backward alone (no stencil rows; input dimension up to 64)."""
    d = shape.desc()
    n = _lib.load().nsvd_model_workspace_bytes(C.byref(d), int(B))
    if n == 0:
        raise NsvdError("nsvd_model_workspace_bytes: invalid model description")
    return torch.empty(n, dtype=torch.uint8, device=device)


def kernel_apply(K: torch.Tensor, N: int, rows: torch.Tensor, cols: torch.Tensor, f: torch.Tensor, scale: float,
                 ws: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Kf = scale * K[rows][:, cols] @ f on the MFMA. K: (N_rows >= N, ldk) float32 with ldk >= N rounded up to 64;
    rows (B1), cols (B2): int64; f: (B2, L). Returns (B1, L)."""
    if K.dim() != 2 or not K.is_cuda or K.dtype != torch.float32 or K.stride(1) != 1:
        raise NsvdError("kernel_apply: K must be a 2-D float32 GPU tensor with unit column stride")
    for t, n in ((rows, "rows"), (cols, "cols")):
        if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous() or t.dim() != 1:
            raise NsvdError(f"kernel_apply: {n} must be a contiguous 1-D int64 GPU tensor")
    B1, B2 = rows.numel(), cols.numel()
    if f.dim() != 2 or f.shape[0] != B2:
        raise NsvdError("kernel_apply: f must be (len(cols), L)")
    L = f.shape[1]
    lib = _lib.load()
    if ws is None:
        ws = torch.empty(lib.nsvd_kernel_apply_workspace_bytes(int(N), B1, L), dtype=torch.uint8, device=f.device)
    if out is None:
        out = torch.empty((B1, L), dtype=torch.float32, device=f.device)
    rc = lib.nsvd_kernel_apply(K.data_ptr(), K.stride(0), int(N), rows.data_ptr(), B1, cols.data_ptr(), B2,
                               _ptr(f, "f"), L, float(scale), _ptr(out, "out"), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_kernel_apply")
    return out


def cdk_workspace(B: int, L: int, set_first_mode_const: bool, device) -> torch.Tensor:
    n = _lib.load().nsvd_cdk_workspace_bytes(int(B), int(L), int(bool(set_first_mode_const)))
    return torch.empty(max(n, 256), dtype=torch.uint8, device=device)


def cdk_loss_forward(f: torch.Tensor, g: torch.Tensor, batch_weights: Optional[torch.Tensor], v: torch.Tensor,
                     M: torch.Tensor, set_first_mode_const: bool, loss: torch.Tensor,
                     rs_joint: Optional[torch.Tensor], rs_indep: Optional[torch.Tensor], ws: torch.Tensor) -> None:
    """NestedLoRALossFunctionForCDK.forward: loss (3), rs_joint (B), rs_indep (B(B-1)); state for the backward in ws."""
    B, L = f.shape
    Lp = L + int(bool(set_first_mode_const))
    if tuple(g.shape) != (B, L) or v.numel() != Lp or tuple(M.shape) != (Lp, Lp) or loss.numel() < 3:
        raise NsvdError(f"cdk_loss_forward: f, g (B, L); v ({Lp}); M ({Lp}, {Lp}); loss (3)")
    if batch_weights is not None and batch_weights.numel() != B:
        raise NsvdError("cdk_loss_forward: batch_weights must hold one weight per row")
    if (rs_joint is not None and rs_joint.numel() != B) or (rs_indep is not None and rs_indep.numel() != B * (B - 1)):
        raise NsvdError("cdk_loss_forward: rs_joint (B), rs_indep (B (B-1))")
    rc = _lib.load().nsvd_cdk_loss_forward(_ptr(f, "f"), _ptr(g, "g"), _ptr(batch_weights, "batch_weights"),
                                           _ptr(v, "v"), _ptr(M, "M"), B, L, int(bool(set_first_mode_const)),
                                           _ptr(loss, "loss"), _ptr(rs_joint, "rs_joint"),
                                           _ptr(rs_indep, "rs_indep"), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_cdk_loss_forward")


def cdk_loss_backward(v: torch.Tensor, B: int, L: int, set_first_mode_const: bool, grad_out: Optional[torch.Tensor],
                      grad_f: Optional[torch.Tensor], grad_g: Optional[torch.Tensor], ws: torch.Tensor) -> None:
    for t, n in ((grad_f, "grad_f"), (grad_g, "grad_g")):
        if t is not None and tuple(t.shape) != (B, L):
            raise NsvdError(f"cdk_loss_backward: {n} must be (B, L)")
    rc = _lib.load().nsvd_cdk_loss_backward(_ptr(v, "v"), int(B), int(L), int(bool(set_first_mode_const)),
                                            _ptr(grad_out, "grad_out"), _ptr(grad_f, "grad_f"),
                                            _ptr(grad_g, "grad_g"), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_cdk_loss_backward")


def rmsprop_ema_step(p: torch.Tensor, grad: torch.Tensor, sq: torch.Tensor, ema: Optional[torch.Tensor], lr: float,
                     alpha: float, eps: float, ema_decay: float, grad_scale: float = 1.0) -> None:
    n = p.numel()
    if grad.numel() != n or sq.numel() != n or (ema is not None and ema.numel() != n):
        raise NsvdError("rmsprop_ema_step: size mismatch")
    if torch_binding() is not None:
        _TB.rmsprop_ema_step(p, grad, sq, ema, float(lr), float(alpha), float(eps), float(ema_decay), float(grad_scale))
        return
    rc = _lib.load().nsvd_rmsprop_ema_step(_ptr(p, "p"), _ptr(grad, "grad"), _ptr(sq, "sq"), _ptr(ema, "ema"), n,
                                           float(lr), float(alpha), float(eps), float(ema_decay), float(grad_scale),
                                           _stream())
    check(rc, "nsvd_rmsprop_ema_step")


def rmsprop_ema_step_dev(p: torch.Tensor, grad: torch.Tensor, sq: torch.Tensor, ema: Optional[torch.Tensor],
                         state: "StepState", grad_scale: float = 1.0, advance: bool = True) -> None:
    """rmsprop_ema_step with lr / EMA decay read from the device-resident schedule; advance: last optimiser launch of
    the step (state.step += 1 on the device)."""
    n = p.numel()
    if grad.numel() != n or sq.numel() != n or (ema is not None and ema.numel() != n):
        raise NsvdError("rmsprop_ema_step_dev: size mismatch")
    rc = _lib.load().nsvd_rmsprop_ema_step_dev(_ptr(p, "p"), _ptr(grad, "grad"), _ptr(sq, "sq"), _ptr(ema, "ema"), n,
                                               state.ptr, float(grad_scale), int(bool(advance)), _stream())
    check(rc, "nsvd_rmsprop_ema_step_dev")


def spectrum_accumulate(f: torch.Tensor, Tf: torch.Tensor, x: torch.Tensor, sigma: float, use_importance: bool,
                        lim: float, cov: torch.Tensor, quad: torch.Tensor, first_mode_const: bool = False) -> None:
    """cov += phi^T phi, quad += phi^T Tphi. float32 accumulators: the reference's (methods/spectrum.py:60-75);
    float64 accumulators (cov.dtype == torch.float64): products and sums in float64 (nsvd_spectrum_accumulate_f64).
    first_mode_const (float64 accumulators of shape (L + 1, L + 1)): a constant-one column in front of phi and Tphi
    (methods/spectrum.py:68-70)."""
    B, L = f.shape
    D = x.shape[1]
    if cov.dtype != quad.dtype or cov.dtype not in (torch.float32, torch.float64):
        raise NsvdError("spectrum_accumulate: cov / quad must both be float32 or both float64")
    Lp = L + int(bool(first_mode_const))
    if tuple(cov.shape) != (Lp, Lp) or tuple(quad.shape) != (Lp, Lp) or Tf.shape != f.shape:
        raise NsvdError(f"spectrum_accumulate: cov / quad must be ({Lp}, {Lp}), Tf like f")
    if first_mode_const and cov.dtype != torch.float64:
        raise NsvdError("spectrum_accumulate(first_mode_const=True) takes float64 accumulators")
    if cov.dtype == torch.float64:
        fn = "nsvd_spectrum_accumulate_const_f64" if first_mode_const else "nsvd_spectrum_accumulate_f64"
        rc = getattr(_lib.load(), fn)(_ptr(f, "f"), _ptr(Tf, "Tf"), _ptr(x, "x"), B, L, D, float(sigma),
                                      int(bool(use_importance)), float(lim),
                                      _ptr(cov, "cov", torch.float64), _ptr(quad, "quad", torch.float64), _stream())
        check(rc, fn)
        return
    if torch_binding() is not None:
        _TB.spectrum_accumulate(f, Tf, x, float(sigma), bool(use_importance), float(lim), cov, quad)
        return
    rc = _lib.load().nsvd_spectrum_accumulate(_ptr(f, "f"), _ptr(Tf, "Tf"), _ptr(x, "x"), B, L, D, float(sigma),
                                              int(bool(use_importance)), float(lim), _ptr(cov, "cov"),
                                              _ptr(quad, "quad"), _stream())
    check(rc, "nsvd_spectrum_accumulate")


def profile_next_forward(ev_start: "torch.cuda.Event", ev_stop: "torch.cuda.Event") -> None:
    """Bracket the dominant kernel of the next operator_forward with two timing events (bench.py)."""
    for e in (ev_start, ev_stop):
        if e.cuda_event == 0:  # torch creates the hipEvent lazily, at the first record
            e.record()
    check(_lib.load().nsvd_profile_next_forward(ev_start.cuda_event, ev_stop.cuda_event), "nsvd_profile_next_forward")


def dominant_kernel_name(shape: ModelShape, B: int, path: int = PATH_AUTO) -> str:
    if path_name(shape, B, path).startswith("fused"):
        return "pmlp_fused_fwd_kernel"
    # generic path: the bracket spans the whole forward (features, every layer's contraction + activation pass, epilogue)
    return "generic forward [fourier_evenodd_kernel, gemm_generic3_kernel + softplus_inplace_kernel per layer, fd_epilogue_kernel]"


# ------------------------------------------------------------------------------ row normalisation (CDK towers)
def row_normalize(z: torch.Tensor, r_up: float, mode: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """normalize(z, r_up, 'l2_ball' | 'l2_sphere') of examples/models/siam.py:170-183 on (B, L) float32 rows."""
    if z.dim() != 2 or z.dtype != torch.float32 or not z.is_contiguous():
        raise NsvdError("row_normalize: z must be a contiguous (B, L) float32 tensor")
    if out is None:
        out = torch.empty_like(z)
    check(_lib.load().nsvd_row_normalize_forward(_ptr(z, "z"), z.shape[0], z.shape[1], float(r_up), int(mode),
                                                 _ptr(out, "out"), _stream()), "nsvd_row_normalize_forward")
    return out


def row_normalize_backward(z: torch.Tensor, dout: torch.Tensor, r_up: float, mode: int) -> torch.Tensor:
    dz = torch.empty_like(z)
    check(_lib.load().nsvd_row_normalize_backward(_ptr(z, "z"), _ptr(dout, "dout"), z.shape[0], z.shape[1], float(r_up),
                                                  int(mode), _ptr(dz, "dz"), _stream()), "nsvd_row_normalize_backward")
    return dz


# ------------------------------------------------------------------------------ CDK towers (Linear-BN-lrelu-Linear-BN)
TOWER_KEYS = ("W1", "b1", "g1", "be1", "rm1", "rv1", "W2", "b2", "g2", "be2", "rm2", "rv2")


def tower_supported(B: int, d0: int, d1: int, d2: int) -> bool:
    return int(_lib.load().nsvd_tower_workspace_bytes(int(B), int(d0), int(d1), int(d2))) > 0


def tower_mixed_supported(B: int, d0: int, d1: int, d2: int) -> bool:
    """does the mixed-precision mode (gemm_bf16) take this shape? (B, d1, d2 multiples of 256, d0 of 128, B <= 1024)"""
    return bool(_lib.load().nsvd_tower_mixed_supported(int(B), int(d0), int(d1), int(d2)))


def tower_mixed_fused(B: int, d0: int, d1: int, d2: int, slope: float) -> bool:
    """does the mixed-precision tower of this shape run the wide layer with BatchNorm inside the contraction
    (csrc/tower_col.h; include/nsvd.h: nsvd_tower_mixed_fused)? The oracle mode that restates its roundings:
    tower_forward_backward(gemm_bf16="fused") - otherwise gemm_bf16=True."""
    return bool(_lib.load().nsvd_tower_mixed_fused(int(B), int(d0), int(d1), int(d2), float(slope)))


def tower_workspace(B: int, d0: int, d1: int, d2: int, device) -> torch.Tensor:
    n = int(_lib.load().nsvd_tower_workspace_bytes(int(B), int(d0), int(d1), int(d2)))
    if n == 0:
        raise NsvdError(f"tower: unsupported shape B={B}, sizes {(d0, d1, d2)} (multiples of 128, B <= 1024)")
    return torch.empty(n, dtype=torch.uint8, device=device)


def _tower_struct(t: dict, need_running: bool) -> "_lib.TowerParams":
    p = _lib.TowerParams()
    for k in TOWER_KEYS:
        v = t.get(k)
        if v is None:
            if k.startswith("r") and not need_running:
                continue
            raise NsvdError(f"tower: tensor {k} missing")
        setattr(p, k, _ptr(v, k))
    p._keepalive = dict(t)
    return p


def tower_y2(ws: torch.Tensor, B: int, d0: int, d1: int, d2: int) -> torch.Tensor:
    """the (B, d2) float32 view of a tower workspace where phase 1 leaves this rank's partial A1 W2^T (hidden width
    sharded over processes): all-reduce it in place between tower_forward(phase=1) and tower_forward(phase=2)"""
    off = int(_lib.load().nsvd_tower_y2_offset(int(B), int(d0), int(d1), int(d2)))
    return ws[off:off + 4 * B * d2].view(torch.float32).view(B, d2)


def tower_forward(x: torch.Tensor, params: dict, slope: float, eps: float, momentum: float, update_running: bool,
                  ws: torch.Tensor, gemm_bf16: bool = False, phase: int = 0) -> Optional[torch.Tensor]:
    """z = BN2(Linear2(lrelu(BN1(Linear1(x))))) in training mode; params: dict over TOWER_KEYS (torch layouts).
    gemm_bf16: mixed precision - bfloat16 operands and wide activations, float32 accumulation and statistics (nsvd.h;
    shapes: tower_mixed_supported). The value 3 (bit 1) says the bfloat16 weight copies inside ws are current.
    phase (hidden width sharded over processes, params = this rank's slices): 1 = up to the partial product in
    tower_y2(ws, ...) (returns None), 2 = bias + second BatchNorm from the all-reduced tower_y2; 0 = the whole tower."""
    B, d0 = x.shape
    d1, d2 = params["W1"].shape[0], params["W2"].shape[0]
    if tuple(params["W1"].shape) != (d1, d0) or tuple(params["W2"].shape) != (d2, d1):
        raise NsvdError("tower_forward: W1 must be (d1, d0), W2 (d2, d1)")
    for k, n in (("b1", d1), ("g1", d1), ("be1", d1), ("b2", d2), ("g2", d2), ("be2", d2)):
        if params[k].numel() != n:
            raise NsvdError(f"tower_forward: {k} must have {n} elements")
    z = torch.empty((B, d2), dtype=torch.float32, device=x.device) if phase != 1 else None
    p = _tower_struct(params, bool(update_running))
    rc = _lib.load().nsvd_tower_forward_phase(_ptr(x, "x"), C.byref(p), B, d0, d1, d2, float(slope), float(eps),
                                              float(momentum), int(bool(update_running)), int(gemm_bf16),
                                              int(phase), _ptr(z, "z"), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_tower_forward")
    return z


def tower_backward(x: torch.Tensor, params: dict, dz: torch.Tensor, slope: float, ws: torch.Tensor,
                   gemm_bf16: bool = False) -> dict:
    """gradients of sum(dz * z) w.r.t. W1, b1, g1, be1, W2, b2, g2, be2 (same workspace as the forward call)."""
    B, d0 = x.shape
    d1, d2 = params["W1"].shape[0], params["W2"].shape[0]
    if tuple(dz.shape) != (B, d2):
        raise NsvdError("tower_backward: dz must be (B, d2)")
    grads = {k: torch.empty_like(params[k]) for k in TOWER_KEYS if not k.startswith("r")}
    p, g = _tower_struct(params, False), _tower_struct(grads, False)
    rc = _lib.load().nsvd_tower_backward(_ptr(x, "x"), C.byref(p), _ptr(dz, "dz"), B, d0, d1, d2, float(slope),
                                         int(gemm_bf16), C.byref(g), ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_tower_backward")
    return grads


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """bfloat16(x), round to nearest even (nsvd_to_bf16) - the cast the mixed-precision towers' operands go through"""
    if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() % 8:
        raise NsvdError("to_bf16: contiguous float32 tensor with a multiple of 8 elements")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.load().nsvd_to_bf16(_ptr(x, "x"), out.data_ptr(), x.numel(), _stream()), "nsvd_to_bf16")
    return out


def gemm_bf16(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, a_kstrided: bool = False,
              b_kstrided: bool = False, out_bf16: bool = False, slices: int = 1, want_sumsq: bool = False):
    """C = A B^T on the bf16 MFMA with float32 accumulation (nsvd_gemm_bf16). A, B: bfloat16, 2-D, contiguous.
    a_kstrided: A is (K, M) - the contraction index first (A^T as stored), else (M, K); likewise B with N.
    slices > 1: C has a leading slice dimension (split-K partial products). Returns C, or (C, sumsq per tile)."""
    for t, n in ((A, "A"), (B, "B")):
        if t.dtype != torch.bfloat16 or t.dim() != 2 or not t.is_contiguous() or not t.is_cuda:
            raise NsvdError(f"gemm_bf16: {n} must be a contiguous 2-D bfloat16 GPU tensor")
    K, M = (A.shape if a_kstrided else A.shape[::-1])
    Kb, N = (B.shape if b_kstrided else B.shape[::-1])
    if K != Kb:
        raise NsvdError("gemm_bf16: contraction lengths differ")
    shape = (M, N) if slices == 1 else (slices, M, N)
    Cm = torch.empty(shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=A.device)
    ss = torch.zeros((M // 256) * (N // 128) * slices, dtype=torch.float32, device=A.device) if want_sumsq else None
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_cuda):
        raise NsvdError("gemm_bf16: bias must be N float32 values on the GPU")
    rc = _lib.load().nsvd_gemm_bf16(A.data_ptr(), B.data_ptr(), Cm.data_ptr(),
                                    bias.data_ptr() if bias is not None else None, M, N, K, A.shape[1], B.shape[1], N,
                                    int(bool(a_kstrided)), int(bool(b_kstrided)), int(bool(out_bf16)), int(slices),
                                    M * N, ss.data_ptr() if ss is not None else None, _stream())
    check(rc, "nsvd_gemm_bf16")
    return (Cm, ss) if want_sumsq else Cm


TOWER16_F16 = 16  # gemm_bf16 bit 4: the half type of the mixed-precision towers is IEEE float16 (include/nsvd.h)


class GradScaler:
    """torch.cuda.amp.GradScaler's state on the DEVICE (include/nsvd.h: nsvd_grad_scaler; the reference's AMP branch,
    examples/cdk/sketchy/main_sketchy.py:161,194-208): nsvd_cdk_step scales the loss gradient, skips the optimiser step
    on inf / NaN gradients and grows / backs off the scale without any host round trip. ``state()`` copies it back."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        self.buf = torch.zeros(C.sizeof(_lib.GradScalerState) // 4, dtype=torch.int32, device=device)
        with torch.cuda.device(self.buf.device):
            check(_lib.load().nsvd_grad_scaler_init(self.buf.data_ptr(), float(init_scale), float(growth_factor),
                                                    float(backoff_factor), int(growth_interval), _stream()),
                  "nsvd_grad_scaler_init")

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    def state(self) -> dict:
        """(synchronises) scale, growth_tracker, steps_ok (optimiser steps taken), steps_skipped, ..."""
        raw = self.buf.cpu().numpy().tobytes()
        st = _lib.GradScalerState.from_buffer_copy(raw)
        return {n: getattr(st, n) for n, _ in _lib.GradScalerState._fields_}


def cdk_step_desc(B: int, d0: int, d1: int, d2: int, slope: float, bn_eps: float, bn_momentum: float, mu: float,
                  normalize_mode: int, set_first_mode_const: bool, lr: float, momentum: float, max_grad_norm: float,
                  first_step: bool, gemm_bf16: int = 0, grad_scaler: Optional["GradScaler"] = None) -> "_lib.CdkStepDesc":
    d = _lib.CdkStepDesc()
    d.B, d.d0, d.d1, d.d2 = int(B), int(d0), int(d1), int(d2)
    d.slope, d.bn_eps, d.bn_momentum, d.mu = float(slope), float(bn_eps), float(bn_momentum), float(mu)
    d.normalize_mode, d.set_first_mode_const = int(normalize_mode), int(bool(set_first_mode_const))
    d.lr, d.momentum, d.max_grad_norm = float(lr), float(momentum), float(max_grad_norm or 0.0)
    d.first_step = int(bool(first_step))
    d.gemm_bf16 = int(gemm_bf16)
    d.grad_scaler = grad_scaler.ptr if grad_scaler is not None else None
    return d


def cdk_step_workspace(desc: "_lib.CdkStepDesc", device) -> torch.Tensor:
    n = int(_lib.load().nsvd_cdk_step_workspace_bytes(C.byref(desc)))
    if n == 0:
        raise NsvdError("cdk_step: unsupported description (tower shapes must be multiples of 128, batch <= 1024; "
                        "l2_ball / l2_sphere normalisation; mu > 0)")
    return torch.empty(n, dtype=torch.uint8, device=device)


def cdk_step(desc: "_lib.CdkStepDesc", x: torch.Tensor, y: torch.Tensor, towers: Sequence[dict],
             momentum_bufs: Sequence[dict], v: torch.Tensor, M: torch.Tensor, loss: torch.Tensor, ws: torch.Tensor,
             rs_joint: Optional[torch.Tensor] = None, rs_indep: Optional[torch.Tensor] = None) -> None:
    """one CDK training step (nsvd_cdk_step): towers / momentum_bufs are two dicts each over TOWER_KEYS (the momentum
    dicts without the running statistics), updated in place; loss: (4) = loss, operator term, metric term, grad norm"""
    if tuple(x.shape) != (desc.B, desc.d0) or tuple(y.shape) != (desc.B, desc.d0) or loss.numel() < 4:
        raise NsvdError("cdk_step: x, y must be (B, d0) and loss hold 4 floats")
    Lp = desc.d2 + desc.set_first_mode_const
    if v.numel() != Lp or tuple(M.shape) != (Lp, Lp):
        raise NsvdError(f"cdk_step: v ({Lp}), M ({Lp}, {Lp})")
    tw = (_lib.TowerParams * 2)(_tower_struct(towers[0], True), _tower_struct(towers[1], True))
    mb = (_lib.TowerParams * 2)(_tower_struct(momentum_bufs[0], False), _tower_struct(momentum_bufs[1], False))
    rc = _lib.load().nsvd_cdk_step(C.byref(desc), _ptr(x, "x"), _ptr(y, "y"), tw, mb, _ptr(v, "v"), _ptr(M, "M"),
                                   _ptr(loss, "loss"), _ptr(rs_joint, "rs_joint"), _ptr(rs_indep, "rs_indep"),
                                   ws.data_ptr(), ws.numel(), _stream())
    check(rc, "nsvd_cdk_step")


# every wrapper that launches kernels runs on the device of its tensors (see _on_tensor_device)
for _name in ("fourier_features", "operator_forward", "operator_features", "operator_sample_features",
              "operator_sample_features_dev", "rmsprop_ema_step_dev",
              "operator_backward", "model_forward", "model_backward", "evd_moments", "evd_loss_grad", "evd_loss_fused",
              "evd_partial", "operator_backward_evd", "operator_backward_evd_heads", "operator_backward_evd_step", "operator_backward_evd_step_next", "operator_backward_evd_step_window", "model_backward_evd_step", "kernel_apply", "cdk_loss_forward",
              "cdk_loss_backward", "rmsprop_ema_step", "spectrum_accumulate", "row_normalize",
              "row_normalize_backward", "tower_forward", "tower_backward", "cdk_step", "to_bf16", "gemm_bf16"):
    globals()[_name] = _on_tensor_device(globals()[_name])
del _name
