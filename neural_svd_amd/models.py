"""Model containers of the PDE path with the reference's constructor signatures and state_dict keys
(``base.ws.{i}``, ``base.bs.{i}``, ``base.feature_map._B``, ``boundary_mask.scales``) - parameters
are ordinary leaf ``nn.Parameter``s so torch optimisers / EMA / checkpoints consume them - but with
every evaluation done by the HIP library.

    GaussianFourierFeatureTransform   examples/utils.py:90-143
    ParallelMLP                       examples/models/mlp.py:167-221
    ExponentialMask                   examples/operator/pde/boundary.py:39-53
    WaveFunctions / get_wavefunctions examples/operator/pde/__init__.py:8-55
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import hip_ops as H
from ._lib import NsvdError


class GaussianFourierFeatureTransform(nn.Module):
    def __init__(self, input_dim, mapping_size=256, scale=10, deterministic=False, append_raw=False):
        super().__init__()
        if append_raw:
            raise NotImplementedError("fourier_append_raw is not used by the PDE scripts and not on the HIP path")
        self.input_dim = input_dim
        self.deterministic = deterministic
        if deterministic:
            # integer harmonics: scale * [1*I, 2*I, ...]^T  -> (input_dim, input_dim * mapping_size)
            blocks = [k * torch.eye(input_dim) for k in range(1, mapping_size + 1)]
            self._B = nn.Parameter(scale * torch.cat(blocks, dim=0).T.contiguous(), requires_grad=False)
            self._mapping_size = input_dim * mapping_size
        else:
            self._B = nn.Parameter(2 * torch.pi * scale * torch.randn((input_dim, mapping_size)).float(),
                                   requires_grad=False)
            self._mapping_size = mapping_size
        self.feature_dim = 2 * self._mapping_size
        self.append_raw = False

    @torch.no_grad()
    def forward(self, x):
        x = x.reshape(x.shape[0], -1).float().contiguous()
        return H.fourier_features(x, self._B.contiguous(), 0.0, 1).T  # (B, 2m) view of the feature-major result


class ParallelMLP(nn.Module):
    """L independent MLPs evaluated together; weights (L, h_i, h_{i-1}), biases (L, h_i, 1)."""

    def __init__(self, input_dim, mlp_hidden_dims, output_dim, num_copies, nonlinearity, bias=False,
                 weight_normalization=False, feature_map=None, debug=False):
        super().__init__()
        if nonlinearity != "softplus":
            raise NotImplementedError("HIP path: softplus only (what the PDE scripts use)")
        if not bias or weight_normalization or feature_map is None or output_dim != 1:
            raise NotImplementedError("HIP path: bias=True, weight_normalization=False, a Fourier feature_map and "
                                      "output_dim=1 are required (the reference's PDE configuration)")
        self.feature_map = feature_map
        self.hidden = tuple(int(h) for h in mlp_hidden_dims)
        self.num_copies = num_copies
        ws, bs = nn.ParameterList(), nn.ParameterList()
        prev = feature_map.feature_dim
        for h in list(self.hidden) + [output_dim]:
            if debug:
                ws.append(nn.Parameter(0.1 * torch.ones(num_copies, h, prev)))
                bs.append(nn.Parameter(0.1 * torch.ones(num_copies, h, 1)))
            else:
                ws.append(nn.Parameter(math.sqrt(2.0 / prev) * torch.randn(num_copies, h, prev)))
                bs.append(nn.Parameter(torch.zeros(num_copies, h, 1)))
            prev = h
        self.ws, self.bs = ws, bs
        self.bias, self.weight_normalization = True, False

    def model_shape(self, has_exp_mask: bool) -> H.ModelShape:
        fm = self.feature_map
        return H.ModelShape(L=self.num_copies, D=fm._B.shape[0], m=fm._B.shape[1], hidden=self.hidden,
                            has_exp_mask=has_exp_mask)

    @torch.no_grad()
    def forward(self, x):
        shape = self.model_shape(False)
        x = x.reshape(x.shape[0], -1).float().contiguous()
        p = H.pack_params(shape, [w.data for w in self.ws], [b.data for b in self.bs], self.feature_map._B.data, None)
        return H.model_forward(shape, p, x, 1.0, H.model_workspace(shape, x.shape[0], x.device))


class ExponentialMask(nn.Module):
    def __init__(self, output_dim, init_scale=1000, boundary_mask=None):
        super().__init__()
        if boundary_mask is not None and not _is_unit_mask(boundary_mask):
            raise NotImplementedError("Dirichlet box masks are off in both PDE scripts and not on the HIP path")
        self.output_dim = output_dim
        self.scales = nn.Parameter(init_scale * torch.ones(output_dim))
        self.boundary_mask = None

    def forward(self, x):
        r = torch.norm(x, p=2, dim=-1).view(-1, 1)
        return torch.exp(-r / self.scales.view(1, -1))


def _is_unit_mask(fn) -> bool:
    try:
        return fn(None) == 1.0
    except Exception:  # noqa: BLE001
        return False


class _GradBuffers:
    def __init__(self, shape, tensors):
        nl = len(shape.dims)
        self.tensors = [torch.empty_like(t) for t in tensors]
        self.packed = H.pack_params(shape, self.tensors[:nl], self.tensors[nl:2 * nl], None,
                                    self.tensors[2 * nl] if shape.has_exp_mask else None)


class WaveFunctions(nn.Module):
    """hard_mul_const * base(x) * boundary_mask(x)."""

    def __init__(self, base, boundary_mask, hard_mul_const=1.0):
        super().__init__()
        if not isinstance(base, ParallelMLP):
            raise NotImplementedError("HIP path: base must be this package's ParallelMLP (--parallel 1)")
        self.base = base
        if isinstance(boundary_mask, ExponentialMask):
            self.boundary_mask = boundary_mask
        elif _is_unit_mask(boundary_mask):
            self.boundary_mask = boundary_mask  # plain callable: not a submodule, like the reference
        else:
            raise NotImplementedError("HIP path: boundary_mask must be ExponentialMask or the constant 1")
        self.hard_mul_const = hard_mul_const

    @property
    def has_exp_mask(self) -> bool:
        return isinstance(self.boundary_mask, ExponentialMask)

    @property
    def shape(self) -> H.ModelShape:
        return self.base.model_shape(self.has_exp_mask)

    def trainable_tensors(self) -> List[torch.Tensor]:
        t = list(self.base.ws) + list(self.base.bs)
        if self.has_exp_mask:
            t.append(self.boundary_mask.scales)
        return t

    def packed_params(self) -> H.Params:
        sc = self.boundary_mask.scales.data if self.has_exp_mask else None
        return H.pack_params(self.shape, [w.data for w in self.base.ws], [b.data for b in self.base.bs],
                             self.base.feature_map._B.data, sc)

    def grad_buffers(self) -> _GradBuffers:
        return _GradBuffers(self.shape, [t.data for t in self.trainable_tensors()])

    def forward(self, x):
        """model(x) -> (B, L). Differentiable w.r.t. the parameters (nsvd_model_forward / _backward)."""
        x = x.reshape(x.shape[0], -1).float().contiguous()
        return _ModelFn.apply(x, self, *self.trainable_tensors())


class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        shape = model.shape
        need_grad = any(ctx.needs_input_grad)
        ws = H.model_workspace(shape, x.shape[0], x.device)
        out = H.model_forward(shape, model.packed_params(), x, float(model.hard_mul_const), ws,
                              save_for_backward=need_grad)
        ctx.model, ctx.ws = model, (ws if need_grad else None)
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        model = ctx.model
        grads = model.grad_buffers()
        H.model_backward(model.shape, model.packed_params(), x, dout.contiguous(), grads.packed, ctx.ws)
        return (None, None) + tuple(grads.tensors)


def parse_str(dims_str: str):
    return [int(s) for s in dims_str.split(",")] if dims_str != "" else []


def get_wavefunctions(args):
    """Same argument object as the reference (examples/operator/pde/__init__.py:19-55)."""
    if not args.use_fourier_feature:
        raise NotImplementedError("HIP path: --use_fourier_feature is required (both PDE scripts set it)")
    if not args.parallel:
        raise NotImplementedError("HIP path: --parallel 1 (ParallelMLP) is required")
    if getattr(args, "apply_boundary", 0):
        raise NotImplementedError("HIP path: --apply_boundary 0 (both PDE scripts)")
    fm = GaussianFourierFeatureTransform(input_dim=args.ndim * args.n_particles,
                                         mapping_size=args.fourier_mapping_size, scale=args.fourier_scale,
                                         deterministic=args.fourier_deterministic,
                                         append_raw=args.fourier_append_raw)
    base = ParallelMLP(input_dim=args.ndim * args.n_particles, mlp_hidden_dims=parse_str(args.mlp_hidden_dims),
                       output_dim=1, num_copies=args.neigs, bias=True, nonlinearity=args.nonlinearity,
                       weight_normalization=bool(getattr(args, "weight_normalization", False)), feature_map=fm)
    mask = lambda x: 1.0  # noqa: E731
    if args.apply_exp_mask:
        mask = ExponentialMask(output_dim=args.neigs, init_scale=args.exp_mask_init_scale, boundary_mask=mask)
    return WaveFunctions(base, boundary_mask=mask, hard_mul_const=args.hard_mul_const)
