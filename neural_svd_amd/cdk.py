"""Drop-in surface of the reference's CDK (two-tower / cross-domain) NestedLoRA loss, backed by the HIP C ABI
(``nsvd_cdk_loss_forward`` / ``nsvd_cdk_loss_backward``). Same names, arguments and return values:

    NestedLoRALossFunctionForCDK.apply(f, g, vmask, mmask, set_first_mode_const, batch_weights)
                                                                  methods/nestedlora.py:270-332
    NestedLoRAForCDK(model, neigs, step, sequential, set_first_mode_const).compute_loss(f, g, batch_weights)
                                                                  methods/nestedlora.py:335-378
    get_cdk_method(args, model)                                   methods/cdk.py:4-16
    normalize(z, r_up, regularize_mode), HeteroNetwork(...)       examples/models/siam.py:132-183
    get_mlp(sizes, ...)                                           examples/models/mlp.py:129-164 (torch modules)

Differences, all deliberate: the arithmetic is float32 on the MFMA whatever the autocast state (the reference's
un-decorated Function runs its matmuls in half precision under ``torch.cuda.amp.autocast``; half inputs are
upcast here and the gradients returned in the input dtype); ``batch_weights`` must be one weight per row; the
inputs are never modified in place (the reference's ``f *= batch_weights`` writes into the caller's tensor when
``set_first_mode_const`` is off).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import hip_ops as H
from ._lib import NsvdError
from .nested_lowrank import get_joint_nesting_masks, get_sequential_nesting_masks, joint_step_weights


class NestedLoRALossFunctionForCDK(torch.autograd.Function):
    """-> (loss, loss_operator, loss_metric, rs_joint, rs_indep); gradients flow to f and g only."""

    @staticmethod
    def forward(ctx, f, g, vector_mask, matrix_mask, set_first_mode_const=False, batch_weights=None):
        if f.dim() != 2 or f.shape != g.shape:
            raise NsvdError("NestedLoRALossFunctionForCDK: f and g must both be (B, L)")
        if not f.is_cuda:
            raise NsvdError("NestedLoRALossFunctionForCDK: tensors must live on the GPU (no CPU path)")
        B, L = f.shape
        first = bool(set_first_mode_const)
        dev = f.device
        f32, g32 = f.detach().float().contiguous(), g.detach().float().contiguous()
        v = vector_mask.detach().to(device=dev, dtype=torch.float32).contiguous()
        M = matrix_mask.detach().to(device=dev, dtype=torch.float32).contiguous()
        bw = None
        if batch_weights is not None:
            if batch_weights.numel() != B:
                raise NotImplementedError("HIP path: batch_weights must be one weight per row, (B,) or (B, 1)")
            bw = batch_weights.detach().to(device=dev, dtype=torch.float32).reshape(B).contiguous()
        ws = H.cdk_workspace(B, L, first, dev)
        loss = torch.empty(3, device=dev, dtype=torch.float32)
        rs_joint = torch.empty(B, device=dev, dtype=torch.float32)
        rs_indep = torch.empty(B * (B - 1), device=dev, dtype=torch.float32)
        H.cdk_loss_forward(f32, g32, bw, v, M, first, loss, rs_joint, rs_indep, ws)
        ctx.ws, ctx.v, ctx.meta = ws, v, (B, L, first, f.dtype, g.dtype)
        ctx.mark_non_differentiable(rs_joint, rs_indep)
        return loss[0], loss[1], loss[2], rs_joint, rs_indep

    @staticmethod
    def backward(ctx, grad_output, *unused):
        B, L, first, fdt, gdt = ctx.meta
        need_f, need_g = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dev = ctx.v.device
        gf = torch.empty(B, L, device=dev, dtype=torch.float32) if need_f else None
        gg = torch.empty(B, L, device=dev, dtype=torch.float32) if need_g else None
        if need_f or need_g:
            go = grad_output.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
            H.cdk_loss_backward(ctx.v, B, L, first, go, gf, gg, ctx.ws)
        return (None if gf is None else gf.to(fdt), None if gg is None else gg.to(gdt), None, None, None, None)


class NestedLoRAForCDK(nn.Module):
    def __init__(self, model, neigs, step=1, sequential=False, set_first_mode_const=True):
        self.name = "nestedlora"
        super().__init__()
        self.neigs = neigs
        self.sequential = sequential
        if sequential:
            self.vector_mask, self.matrix_mask = get_sequential_nesting_masks(neigs, set_first_mode_const)
        else:
            self.vector_mask, self.matrix_mask = get_joint_nesting_masks(joint_step_weights(neigs, step),
                                                                         set_first_mode_const)
        self.set_first_mode_const = set_first_mode_const
        self.model = model
        self._dev_masks = None

    def forward(self, *args):
        return self.model(*args)

    def _masks_on(self, device):
        if self._dev_masks is None or self._dev_masks[0].device != device:
            self._dev_masks = (self.vector_mask.to(device), self.matrix_mask.to(device))
        return self._dev_masks

    def compute_loss(self, f, g, batch_weights=None):
        v, M = self._masks_on(f.device)
        return NestedLoRALossFunctionForCDK.apply(f, g, v, M, self.set_first_mode_const, batch_weights)


def get_cdk_method(args, model):
    """Same nested ``args.loss.neuralsvd.*`` attribute layout as the reference's config object."""
    if args.loss.name == "neuralsvd":
        return NestedLoRAForCDK(model, neigs=args.neigs, step=args.loss.neuralsvd.step,
                                sequential=args.loss.neuralsvd.sequential,
                                set_first_mode_const=args.loss.neuralsvd.set_first_mode_const)
    raise NotImplementedError


# ------------------------------------------------------------------------------ towers (examples/models/siam.py)
class _RowNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, r_up, mode):
        zd = z.detach().float().contiguous()
        ctx.r_up, ctx.mode, ctx.dtype = float(r_up), int(mode), z.dtype
        ctx.save_for_backward(zd)
        return H.row_normalize(zd, ctx.r_up, ctx.mode).to(z.dtype)

    @staticmethod
    def backward(ctx, dout):
        (zd,) = ctx.saved_tensors
        dz = H.row_normalize_backward(zd, dout.detach().float().contiguous(), ctx.r_up, ctx.mode)
        return dz.to(ctx.dtype), None, None


def normalize(z, r_up, regularize_mode):
    """examples/models/siam.py:170-191 on the HIP path: 'l2_ball' (what the Sketchy script uses, sketchy.sh) and
    'l2_sphere' are one row-wise kernel each way (nsvd_row_normalize_forward / _backward); 'clip' and 'tanh' are
    single elementwise torch ops in the reference and stay that."""
    if not r_up > 0:
        return z
    if regularize_mode == "l2_ball":
        return _RowNormalize.apply(z, r_up, H._lib.NORMALIZE_L2_BALL)
    if regularize_mode == "l2_sphere":
        return _RowNormalize.apply(z, r_up, H._lib.NORMALIZE_L2_SPHERE)
    if regularize_mode == "clip":
        return torch.clip(z, min=-r_up, max=r_up)
    if regularize_mode == "tanh":
        return r_up * torch.tanh(z)
    raise NotImplementedError(regularize_mode)


class HeteroNetwork(nn.Module):
    """The two-tower network of the CDK script behind the reference's interface (examples/models/siam.py:132-166:
    constructor arguments, ``forward(x, y) -> [x_rep, x_emb, y_rep, y_emb]``, ``forward_single(x, side, classify)``,
    ``output_dims``, and the ``backbones.{x,y}.* / projectors.{x,y}.* / online_heads.{x,y}.*`` state_dict keys).
    One method, ``embed``, does the work of a side: representation = backbone(input), embedding = the projector's
    output scaled into the radius-sqrt(mu) ball / sphere by the HIP row kernel (``normalize`` above)."""

    SIDES = ("x", "y")
    MODES = ("l2_ball", "l2_sphere", "clip", "tanh")

    def __init__(self, backbones, projectors, online_heads=None, mu=1.0, regularize_mode=None):
        super().__init__()
        if regularize_mode not in self.MODES:
            raise AssertionError(f"regularize_mode must be one of {self.MODES}, got {regularize_mode!r}")
        per_side = lambda mods: nn.ModuleDict(dict(zip(self.SIDES, mods)))  # noqa: E731
        self.backbones, self.projectors = per_side(backbones), per_side(projectors)
        self.online_heads = per_side(online_heads) if online_heads else None
        self.mu, self.regularize_mode = mu, regularize_mode

    @property
    def output_dims(self):
        """embedding width per side: the projector's, or the backbone's behind an Identity projector"""
        widths = {}
        for side in self.SIDES:
            proj = self.projectors[side]
            widths[side] = (self.backbones[side] if isinstance(proj, nn.Identity) else proj).output_dim
        return widths

    def embed(self, inp, side):
        if side not in self.SIDES:
            raise AssertionError(f"side must be one of {self.SIDES}")
        rep = self.backbones[side](inp)
        return rep, normalize(self.projectors[side](rep), math.sqrt(self.mu), self.regularize_mode)

    def forward_single(self, x, x_or_y, classify=False):
        rep, emb = self.embed(x, x_or_y)
        if not classify:
            return rep, emb
        return rep, emb, self.online_heads[x_or_y](emb.detach())  # the probe never trains the tower

    def forward(self, x, y):
        out = []
        for side, inp in zip(self.SIDES, (x, y)):
            out.extend(self.embed(inp, side))
        return out


def _activation_factory(name: str):
    """The activation names the tower builder understands (reference examples/models/mlp.py:65-88; its custom erf /
    sine activations are not used by the CDK script and are not provided)."""
    if name == "relu":
        return lambda: nn.ReLU(inplace=True)
    if name.startswith("lrelu"):
        slope = float(name[len("lrelu"):])
        return lambda: nn.LeakyReLU(negative_slope=slope)
    if name.startswith("elu"):
        alpha = float(name[len("elu"):])
        return lambda: nn.ELU(alpha=alpha)
    table = {"tanh": nn.Tanh, "linear": nn.Identity, "softplus": nn.Softplus}
    if name in table:
        return table[name]
    raise NotImplementedError(f"activation {name!r}")


class _TowerFn(torch.autograd.Function):
    """Linear -> BatchNorm1d -> LeakyReLU -> Linear -> BatchNorm1d in training mode on the HIP tower kernels
    (csrc/tower.hip: five fp32-MFMA contractions and four BatchNorm strip kernels for forward + backward).
    Under ``torch.autocast`` (the reference's Sketchy loop wraps ``method(x, y)`` in it unless --disable_amp,
    main_sketchy.py:182) the tower runs in this library's mixed-precision mode - 16-bit operands and wide
    activations, float32 accumulation, statistics and parameter gradients (nsvd.h: gemm_bf16; shapes:
    nsvd_tower_mixed_supported, float32 kernels otherwise) - with the AUTOCAST DTYPE as the half type: float16 under
    the reference's ``torch.cuda.amp.autocast()`` (its default dtype), bfloat16 under autocast(dtype=torch.bfloat16).
    The output stays float32. With float16 the script's GradScaler does what it is there for: the scaled gradients it
    sends into this backward are stored as float16, an overflow comes back as inf / NaN parameter gradients, and
    ``scaler.step`` skips the update (main_sketchy.py:194-208: torch's own GradScaler, unchanged). With bfloat16 a
    GradScaler is harmless (float32's exponent range: its scale neither rescues nor overflows anything)."""

    @staticmethod
    def forward(ctx, x, W1, b1, g1, be1, W2, b2, g2, be2, seq):
        lin1, bn1, act, lin2, bn2 = seq[0], seq[1], seq[2], seq[3], seq[4]
        xd = x.detach().float().contiguous()
        t = dict(W1=W1.detach(), b1=b1.detach(), g1=g1.detach(), be1=be1.detach(), rm1=bn1.running_mean,
                 rv1=bn1.running_var, W2=W2.detach(), b2=b2.detach(), g2=g2.detach(), be2=be2.detach(),
                 rm2=bn2.running_mean, rv2=bn2.running_var)
        ws = H.tower_workspace(xd.shape[0], W1.shape[1], W1.shape[0], W2.shape[0], xd.device)
        track = bn1.track_running_stats and bn1.running_mean is not None
        # (autocast on a shape the mixed-precision kernels do not take: the float32 kernels - more precise, never less)
        mixed = bool(torch.is_autocast_enabled()) and H.tower_mixed_supported(xd.shape[0], W1.shape[1], W1.shape[0],
                                                                              W2.shape[0])
        if mixed:
            try:
                f16 = torch.get_autocast_dtype("cuda") == torch.float16
            except (AttributeError, TypeError):  # (older torch)
                f16 = torch.get_autocast_gpu_dtype() == torch.float16
            mixed = 1 | (H.TOWER16_F16 if f16 else 0)
        else:
            mixed = 0
        z = H.tower_forward(xd, t, seq.slope, bn1.eps, bn1.momentum, track, ws, gemm_bf16=mixed)
        if track:
            bn1.num_batches_tracked += 1
            bn2.num_batches_tracked += 1
        ctx.t, ctx.ws, ctx.xd, ctx.slope, ctx.mixed = t, ws, xd, seq.slope, mixed
        return z

    @staticmethod
    def backward(ctx, dz):
        g = H.tower_backward(ctx.xd, ctx.t, dz.detach().float().contiguous(), ctx.slope, ctx.ws, gemm_bf16=ctx.mixed)
        return (None, g["W1"], g["b1"], g["g1"], g["be1"], g["W2"], g["b2"], g["g2"], g["be2"], None)


class TowerSequential(nn.Sequential):
    """What get_mlp returns for the CDK script's towers: the same five modules under the same names as the reference's
    nn.Sequential (state_dict keys 0.weight, 1.running_mean, ... interchange), whose TRAINING forward on a GPU batch
    of a supported shape (batch and widths multiples of 128, batch <= 1024: BASELINE configs[4] is 1024 x 512 -> 8192
    -> 512) runs on the HIP tower kernels. Evaluation mode (running statistics) and other shapes use the modules
    themselves (torch's library kernels)."""

    def __init__(self, *mods, slope: float = 0.0):
        super().__init__(*mods)
        self.slope = float(slope)

    def __getitem__(self, idx):
        # a slice of the tower is no longer "the tower": plain nn.Sequential over the same modules
        if isinstance(idx, slice):
            return nn.Sequential(*list(self._modules.values())[idx])
        return super().__getitem__(idx)

    def hip_ready(self, x) -> bool:
        lin1, bn1, lin2, bn2 = self[0], self[1], self[3], self[4]
        # the tower kernels produce no gradient for their input (nsvd_tower_backward): behind a trainable module
        # (a backbone in front of a projector, siam.py:156-166) the torch modules run instead, so that the gradient
        # reaches everything upstream
        if torch.is_grad_enabled() and x.requires_grad:
            return False
        return (self.training and x.is_cuda and x.dim() == 2 and bn1.momentum is not None and bn1.eps == bn2.eps
                and bn1.momentum == bn2.momentum and bn1.affine and bn2.affine
                and lin1.weight.dtype == torch.float32
                and H.tower_supported(x.shape[0], lin1.in_features, lin1.out_features, lin2.out_features))

    def forward(self, x):
        if not self.hip_ready(x):
            return super().forward(x)
        lin1, bn1, lin2, bn2 = self[0], self[1], self[3], self[4]
        return _TowerFn.apply(x, lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias, bn2.weight,
                              bn2.bias, self)


def _tower_slope(nonlinearity: str):
    """negative-side slope of the activations the tower kernels implement (None: another activation)"""
    if nonlinearity == "relu":
        return 0.0
    if nonlinearity.startswith("lrelu"):
        return float(nonlinearity[len("lrelu"):])
    return None


def get_mlp(sizes, bias=True, nonlinearity="relu", use_bn=True, weight_normalization=False, last_layer_bn=True,
            feature_map=None):
    """Tower builder of the Sketchy script (reference examples/models/mlp.py:129-164, used at
    examples/cdk/sketchy/main_sketchy.py:109-112): Linear (+ BatchNorm1d) (+ activation) per layer, no activation after
    the last layer, BatchNorm after it only with ``last_layer_bn``; ``output_dim`` attribute. The two-layer form the
    script builds (Linear, BatchNorm, relu / lrelu, Linear, BatchNorm) comes back as a TowerSequential, whose training
    step runs on the HIP tower kernels; any other stack is plain torch modules."""
    make_act = _activation_factory(nonlinearity)
    n = len(sizes) - 1
    if n == 0:
        model = nn.BatchNorm1d(sizes[0]) if (use_bn and last_layer_bn) else nn.Identity()
    else:
        mods = [] if feature_map is None else [feature_map]
        for i, (d_in, d_out) in enumerate(zip(sizes[:-1], sizes[1:])):
            lin = nn.Linear(d_in, d_out, bias=bias)
            mods.append(nn.utils.weight_norm(lin) if weight_normalization else lin)
            last = i == n - 1
            if use_bn and (not last or last_layer_bn):
                mods.append(nn.BatchNorm1d(d_out))
            if not last:
                mods.append(make_act())
        slope = _tower_slope(nonlinearity)
        if (n == 2 and bias and use_bn and last_layer_bn and not weight_normalization and feature_map is None
                and slope is not None):
            model = TowerSequential(*mods, slope=slope)  # Linear, BN, act, Linear, BN: the CDK script's tower
        else:
            model = nn.Sequential(*mods)
    model.output_dim = sizes[-1]
    return model


# ------------------------------------------------------------------------------ the fused training step
class FusedCdkStep:
    """The loop body of the Sketchy script (examples/cdk/sketchy/main_sketchy.py:180-212 with scripts/exps/sketchy.sh's
    switches: sgd with momentum, --clip_grad_norm, --use_lr_scheduler; ``use_amp`` below) as
    ONE C call per step, ``nsvd_cdk_step``: both towers forward and backward, normalisation, the NestedLoRAForCDK loss,
    the global gradient-norm clip and the SGD momentum update, on the modules' OWN parameter tensors (updated in place:
    ``method.model`` is always current, its BatchNorm running statistics included). Takes the place of

        optimizer.zero_grad(); _, fx, _, fy = method(x, y); loss, *_ = method.compute_loss(fx, fy); loss.backward()
        nn.utils.clip_grad_norm_(model.parameters(), max_norm); optimizer.step(); lr_scheduler.step()

    for a model that is HeteroNetwork(two TowerSequential towers, Identity projectors, 'l2_ball' / 'l2_sphere') on a
    batch the tower kernels take; ``supported(method, batch_size)`` says whether it is.

    use_amp: the script runs its step under ``torch.cuda.amp.autocast`` + ``GradScaler`` unless ``--disable_amp``
    (main_sketchy.py:161,182). True selects this library's MIXED-PRECISION mode, the counterpart of that branch with
    bfloat16 as the half type: operands, the wide activations and their gradients are bfloat16, the ten contractions of
    a step run on the bf16 MFMA with float32 accumulation (both towers through every launch together, no transposed or
    re-cast copies: the optimiser kernel leaves the bfloat16 weights of the next step), BatchNorm statistics, the loss,
    parameter gradients, clipping and the update stay float32 - no loss scaling, no skipped step. NOT bit-comparable with
    float16 autocast; pinned to the float64 oracle with the same roundings. Batch and the towers' two output widths must
    be multiples of 256. False (default): float32
    throughout - the script's --disable_amp.

    amp_dtype (with use_amp): "bfloat16" (default: the mode above) or "float16" - the reference's own half type. With
    float16 the step runs the reference's GradScaler as well (grad_scaler=None -> on; a hip_ops.GradScaler to share or
    pre-set one; False -> off): the loss gradient is multiplied by the scale where the backward starts, a step whose
    (scaled) gradient norm is inf / NaN is SKIPPED as a whole - no parameter or momentum buffer
    changes (the schedule still advances), the scale halves - otherwise the gradients are unscaled before the clip and the scale doubles every
    growth_interval clean steps (main_sketchy.py:194-208; torch defaults 65536 / 2 / 0.5 / 2000). All of it on the
    device: ``scaler_state()`` reads it back. The learning-rate schedule advances on every iteration, skipped or not,
    as the script's does (main_sketchy.py:205-206: no scheduler gate there, unlike the PDE loop). Same kernels as the bfloat16 mode with the float16 MFMA; pinned to the float64 oracle
    with the same roundings and the same scaler arithmetic (oracle.cdk_train_step(half="f16", scaler=...))."""

    def __init__(self, method: "NestedLoRAForCDK", lr: float, momentum: float = 0.9, max_grad_norm: float = 1.0,
                 t_max: int = 0, batch_size: int = 1024, use_amp: bool = False, amp_dtype: str = "bfloat16",
                 grad_scaler=None, init_scale: float = 65536.0, growth_interval: int = 2000):
        ok, why = self.supported(method, batch_size, use_amp)
        if not ok:
            raise H.NsvdError(f"FusedCdkStep: {why}")
        if amp_dtype not in ("bfloat16", "float16"):
            raise H.NsvdError("FusedCdkStep: amp_dtype must be 'bfloat16' or 'float16'")
        self.amp_f16 = bool(use_amp) and amp_dtype == "float16"
        if grad_scaler is None:
            grad_scaler = self.amp_f16
        if grad_scaler and not use_amp:
            raise H.NsvdError("FusedCdkStep: a GradScaler needs use_amp=True")
        model = method.model
        self.method, self.model = method, model
        self.lr0, self.momentum, self.max_grad_norm, self.t_max = float(lr), float(momentum), float(max_grad_norm or 0.0), int(t_max)
        self.t = 0
        self._pending = 0
        self._weight_versions = None
        self.use_amp = bool(use_amp)
        tx = model.backbones["x"]
        self.B, self.d0, self.d1, self.d2 = int(batch_size), tx[0].in_features, tx[0].out_features, tx[3].out_features
        dev = tx[0].weight.device
        self.towers, self.bufs = [], []
        for side in HeteroNetwork.SIDES:
            t = model.backbones[side]
            lin1, bn1, lin2, bn2 = t[0], t[1], t[3], t[4]
            d = dict(W1=lin1.weight.data, b1=lin1.bias.data, g1=bn1.weight.data, be1=bn1.bias.data,
                     rm1=bn1.running_mean, rv1=bn1.running_var, W2=lin2.weight.data, b2=lin2.bias.data,
                     g2=bn2.weight.data, be2=bn2.bias.data, rm2=bn2.running_mean, rv2=bn2.running_var)
            self.towers.append(d)
            self.bufs.append({k: torch.zeros_like(v) for k, v in d.items() if not k.startswith("r")})
        self.slope, self.bn_eps, self.bn_momentum = float(tx.slope), float(tx[1].eps), float(tx[1].momentum)
        self.mode = H._lib.NORMALIZE_L2_BALL if model.regularize_mode == "l2_ball" else H._lib.NORMALIZE_L2_SPHERE
        self.v = method.vector_mask.detach().float().to(dev).contiguous()
        self.M = method.matrix_mask.detach().float().to(dev).contiguous()
        self.first_const = bool(method.set_first_mode_const)
        self.loss = torch.zeros(4, dtype=torch.float32, device=dev)  # loss, operator term, metric term, grad norm
        self.scaler = None
        if grad_scaler:
            self.scaler = grad_scaler if isinstance(grad_scaler, H.GradScaler) else \
                H.GradScaler(dev, init_scale=init_scale, growth_interval=growth_interval)
        self.ws = H.cdk_step_workspace(self._desc(self.lr0, True), dev)

    @staticmethod
    def supported(method, batch_size: int, use_amp: bool = False):
        model = getattr(method, "model", None)
        if not isinstance(method, NestedLoRAForCDK) or not isinstance(model, HeteroNetwork):
            return False, "needs NestedLoRAForCDK over a HeteroNetwork"
        if model.regularize_mode not in ("l2_ball", "l2_sphere") or not model.mu > 0:
            return False, "needs l2_ball / l2_sphere normalisation with mu > 0"
        shapes = set()
        for side in HeteroNetwork.SIDES:
            t, pr = model.backbones[side], model.projectors[side]
            if not isinstance(t, TowerSequential) or not isinstance(pr, nn.Identity):
                return False, "needs TowerSequential backbones (get_mlp([d0, d1, d2], use_bn=True)) and Identity projectors"
            if not t[0].weight.is_cuda or t[1].momentum is None or not t[1].track_running_stats:
                return False, "needs GPU towers with BatchNorm running statistics"
            shapes.add((t[0].in_features, t[0].out_features, t[3].out_features, t.slope, t[1].eps, t[1].momentum))
        if len(shapes) != 1:
            return False, "both towers must have the same shape"
        d0, d1, d2 = list(shapes)[0][:3]
        if method.neigs != d2:
            return False, "neigs must equal the towers' output width"
        if not H.tower_supported(batch_size, d0, d1, d2):
            return False, f"tower shape {(batch_size, d0, d1, d2)} outside the tower kernels (multiples of 128, B <= 1024)"
        if use_amp and not H.tower_mixed_supported(batch_size, d0, d1, d2):
            return False, (f"tower shape {(batch_size, d0, d1, d2)} outside the mixed-precision kernels (batch and the "
                           f"two output widths multiples of 256)")
        return True, ""

    def _weights(self):
        """the four weight Parameters whose bfloat16 copies the mixed-precision step keeps in its workspace"""
        return [self.model.backbones[side][i].weight for side in HeteroNetwork.SIDES for i in (0, 3)]

    def weights_changed(self) -> None:
        """Mixed precision: tell the step that W1 / W2 were modified from outside (the next step casts them again).
        In-place operations on the Parameters themselves (optimizer steps, load_state_dict, copy_) are noticed without
        this call - they move the Parameters' version counters; writes through ``.data`` or raw pointers are not."""
        self._weight_versions = None

    def _desc(self, lr, first, weights_ready=False):
        # gemm_bf16 bit 1: the bfloat16 copies of W1 / W2 inside the workspace are the ones the previous step's
        # optimiser kernel wrote, and nothing has touched the float32 masters since (their torch version counters:
        # the C call updates them through raw pointers, which bumps nothing)
        flags = ((3 if weights_ready else 1) | (H.TOWER16_F16 if self.amp_f16 else 0)) if self.use_amp else 0
        return H.cdk_step_desc(self.B, self.d0, self.d1, self.d2, self.slope, self.bn_eps, self.bn_momentum,
                               self.model.mu, self.mode, self.first_const, lr, self.momentum, self.max_grad_norm, first,
                               gemm_bf16=flags, grad_scaler=self.scaler)

    def scaler_state(self):
        """the GradScaler's device state (synchronises), or None without one"""
        return self.scaler.state() if self.scaler is not None else None

    def current_lr(self) -> float:
        """CosineAnnealingLR(optimizer, t_max) after self.t scheduler steps (t_max = 0: constant). The Sketchy script
        steps its scheduler on every iteration, skipped by the GradScaler or not (main_sketchy.py:205-206)."""
        if self.t_max <= 0:
            return self.lr0
        return self.lr0 * (1.0 + math.cos(math.pi * self.t / self.t_max)) / 2.0

    @torch.no_grad()
    def step(self, x: torch.Tensor, y: torch.Tensor, rs_joint=None, rs_indep=None) -> torch.Tensor:
        """one training step on the batch (x, y); returns the device tensor (loss, operator term, metric term, total
        gradient norm before clipping) - no synchronisation"""
        if not self.model.training:
            raise H.NsvdError("FusedCdkStep.step: the model must be in training mode")
        vers = [(w.data_ptr(), w._version) for w in self._weights()]
        ready = self.use_amp and self.t > 0 and vers == self._weight_versions
        H.cdk_step(self._desc(self.current_lr(), self.t == 0, ready), x.float().contiguous(), y.float().contiguous(),
                   self.towers, self.bufs, self.v, self.M, self.loss, self.ws, rs_joint, rs_indep)
        self._weight_versions = vers
        self.t += 1
        self._pending += 1
        if self._pending >= 256:
            self.flush_counters()
        return self.loss

    @torch.no_grad()
    def flush_counters(self) -> None:
        """BatchNorm1d.num_batches_tracked of the four BatchNorm modules (a bookkeeping counter: the running
        statistics themselves are updated inside the step) is advanced in batches of steps rather than by four one-
        element kernels per step; call before a checkpoint."""
        if self._pending:
            for side in HeteroNetwork.SIDES:
                t = self.model.backbones[side]
                t[1].num_batches_tracked += self._pending
                t[4].num_batches_tracked += self._pending
            self._pending = 0


class ShardedCdkStep:
    """The Sketchy training step on several GPUs with the towers' HIDDEN width sharded (one process per GPU,
    ``comm`` = parallel.Communicator): rank r holds rows [r d1/W, (r+1) d1/W) of Linear1 / BatchNorm1 and the same
    columns of Linear2 of BOTH towers; Linear2's bias and BatchNorm2 are replicated. BatchNorm statistics are per column,
    so the arithmetic is the single-process step's on the same batch (the reference is single-process:
    examples/cdk/sketchy/main_sketchy.py:180-212) with TWO collectives per step:
      1. all-reduce(sum) of the two towers' partial products A1 W2^T, packed (2, B, d2) floats (4 MB at configs[4]);
      2. all-reduce(sum) of one float: the squared gradient norm of the sharded tensors (the replicated ones counted once)
         for clip_grad_norm_.
    No gradient traffic: every weight gradient is local to the rank that owns the slice; the replicated parameters get
    identical gradients on every rank. The batch is the same on every rank (strong scaling: the work of one step is
    split W ways). Stage calls of the C ABI (nsvd_tower_forward_phase, nsvd_row_normalize_*, nsvd_cdk_loss_*,
    nsvd_tower_backward) + torch for the clip coefficient and the momentum update of the local tensors.
    ``gather_into_model()`` writes the whole parameters and running statistics back into ``method.model``."""

    SHARDED = ("W1", "b1", "g1", "be1", "rm1", "rv1")   # rows of the hidden width
    REPLICATED = ("b2", "g2", "be2", "rm2", "rv2")

    def __init__(self, method: "NestedLoRAForCDK", comm, lr: float, momentum: float = 0.9, max_grad_norm: float = 1.0,
                 t_max: int = 0, batch_size: int = 1024, use_amp: bool = False):
        ok, why = FusedCdkStep.supported(method, batch_size, use_amp)
        if not ok:
            raise H.NsvdError(f"ShardedCdkStep: {why}")
        self.method, self.model, self.comm = method, method.model, comm
        W, r = comm.world, comm.rank
        tx = self.model.backbones["x"]
        self.B, self.d0, d1, self.d2 = int(batch_size), tx[0].in_features, tx[0].out_features, tx[3].out_features
        q = 256 if use_amp else 128
        if d1 % (q * W) != 0:
            raise H.NsvdError(f"ShardedCdkStep: hidden width {d1} must split into multiples of {q} over {W} ranks")
        self.d1_full, self.d1 = d1, d1 // W
        self.lo, self.hi = r * self.d1, (r + 1) * self.d1
        self.lr0, self.momentum, self.max_grad_norm, self.t_max = float(lr), float(momentum), float(max_grad_norm or 0.0), int(t_max)
        self.use_amp, self.t = bool(use_amp), 0
        dev = tx[0].weight.device
        self.towers, self.bufs = [], []
        for side in HeteroNetwork.SIDES:
            t = self.model.backbones[side]
            lin1, bn1, lin2, bn2 = t[0], t[1], t[3], t[4]
            sl = slice(self.lo, self.hi)
            d = dict(W1=lin1.weight.data[sl].clone(), b1=lin1.bias.data[sl].clone(), g1=bn1.weight.data[sl].clone(),
                     be1=bn1.bias.data[sl].clone(), rm1=bn1.running_mean[sl].clone(), rv1=bn1.running_var[sl].clone(),
                     W2=lin2.weight.data[:, sl].contiguous(), b2=lin2.bias.data.clone(), g2=bn2.weight.data.clone(),
                     be2=bn2.bias.data.clone(), rm2=bn2.running_mean.clone(), rv2=bn2.running_var.clone())
            self.towers.append(d)
            self.bufs.append({k: torch.zeros_like(v) for k, v in d.items() if not k.startswith("r")})
        self.slope, self.bn_eps, self.bn_momentum = float(tx.slope), float(tx[1].eps), float(tx[1].momentum)
        self.mode = H._lib.NORMALIZE_L2_BALL if self.model.regularize_mode == "l2_ball" else H._lib.NORMALIZE_L2_SPHERE
        self.r_up = float(self.model.mu) ** 0.5
        self.v = method.vector_mask.detach().float().to(dev).contiguous()
        self.M = method.matrix_mask.detach().float().to(dev).contiguous()
        self.first_const = bool(method.set_first_mode_const)
        self.loss = torch.zeros(4, dtype=torch.float32, device=dev)
        self.ws = [H.tower_workspace(self.B, self.d0, self.d1, self.d2, dev) for _ in HeteroNetwork.SIDES]
        self.y2 = [H.tower_y2(w, self.B, self.d0, self.d1, self.d2) for w in self.ws]
        self.y2_packed = torch.empty((2, self.B, self.d2), dtype=torch.float32, device=dev)
        self.cdk_ws = H.cdk_workspace(self.B, self.d2, self.first_const, dev)
        self.ge = [torch.empty((self.B, self.d2), dtype=torch.float32, device=dev) for _ in range(2)]
        self.norm2 = torch.zeros(1, dtype=torch.float64, device=dev)
        self.probe = None  # parallel.CommProbe while bench.py measures the exposed waits

    def current_lr(self) -> float:
        if self.t_max <= 0:
            return self.lr0
        return self.lr0 * (1.0 + math.cos(math.pi * self.t / self.t_max)) / 2.0

    def _span(self, name):
        import contextlib
        return self.probe.span(name) if self.probe is not None else contextlib.nullcontext()

    @torch.no_grad()
    def step(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """one training step on the batch (x, y) - the SAME batch on every rank; returns the device tensor (loss,
        operator term, metric term, total gradient norm before clipping)"""
        B, d0, d1, d2 = self.B, self.d0, self.d1, self.d2
        xs = (x.float().contiguous(), y.float().contiguous())
        for t in range(2):  # Linear1, BatchNorm1, activation, this rank's partial A1 W2^T
            H.tower_forward(xs[t], self.towers[t], self.slope, self.bn_eps, self.bn_momentum, True, self.ws[t],
                            gemm_bf16=self.use_amp, phase=1)
            self.y2_packed[t].copy_(self.y2[t])
        with self._span("y2_partials_allreduce"):
            self.comm.all_reduce_sum(self.y2_packed)          # collective 1: (2, B, d2) floats
        zs, es = [], []
        for t in range(2):  # + b2, BatchNorm2, normalisation
            self.y2[t].copy_(self.y2_packed[t])
            z = H.tower_forward(xs[t], self.towers[t], self.slope, self.bn_eps, self.bn_momentum, True, self.ws[t],
                                gemm_bf16=self.use_amp, phase=2)
            zs.append(z)
            es.append(H.row_normalize(z, self.r_up, self.mode))
        H.cdk_loss_forward(es[0], es[1], None, self.v, self.M, self.first_const, self.loss, None, None, self.cdk_ws)
        H.cdk_loss_backward(self.v, B, d2, self.first_const, None, self.ge[0], self.ge[1], self.cdk_ws)
        grads = []
        for t in range(2):
            dz = H.row_normalize_backward(zs[t], self.ge[t], self.r_up, self.mode)
            grads.append(H.tower_backward(xs[t], self.towers[t], dz, self.slope, self.ws[t], gemm_bf16=self.use_amp))
        # total gradient norm: the sharded tensors' squares summed over the ranks, the replicated ones counted once
        sharded = [g[k] for g in grads for k in ("W1", "b1", "g1", "be1", "W2")]
        repl = [g[k] for g in grads for k in ("b2", "g2", "be2")]
        n2 = torch.stack([t.double().pow(2).sum() for t in sharded]).sum()
        if self.comm.rank == 0:
            n2 = n2 + torch.stack([t.double().pow(2).sum() for t in repl]).sum()
        self.norm2.copy_(n2.reshape(1))
        with self._span("grad_norm_allreduce"):
            self.comm.all_reduce_sum(self.norm2)                # collective 2: one float64
        total = self.norm2.sqrt().float()
        coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0) if self.max_grad_norm > 0 else torch.ones_like(total)
        lr = self.current_lr()
        for P, Bf, g in zip(self.towers, self.bufs, grads):
            keys = list(g.keys())
            gs = [g[k] * coef for k in keys]
            bs = [Bf[k] for k in keys]
            if self.t == 0:
                torch._foreach_copy_(bs, gs)
            else:
                torch._foreach_mul_(bs, self.momentum)
                torch._foreach_add_(bs, gs)
            torch._foreach_add_([P[k] for k in keys], bs, alpha=-lr)
        self.loss[3:4].copy_(total)
        self.t += 1
        if self.probe is not None:
            self.probe.step_done()
        return self.loss

    @torch.no_grad()
    def gather_into_model(self) -> None:
        """the whole parameters, momentum-free, and running statistics -> method.model on every rank"""
        W = self.comm.world
        for side, P in zip(HeteroNetwork.SIDES, self.towers):
            t = self.model.backbones[side]
            lin1, bn1, lin2, bn2 = t[0], t[1], t[3], t[4]

            def rows(v):
                out = torch.empty((W,) + tuple(v.shape), dtype=v.dtype, device=v.device)
                self.comm.all_gather(out, v.contiguous())
                return out.view((W * v.shape[0],) + tuple(v.shape[1:]))
            lin1.weight.data.copy_(rows(P["W1"]))
            lin1.bias.data.copy_(rows(P["b1"]))
            bn1.weight.data.copy_(rows(P["g1"]))
            bn1.bias.data.copy_(rows(P["be1"]))
            bn1.running_mean.copy_(rows(P["rm1"]))
            bn1.running_var.copy_(rows(P["rv1"]))
            lin2.weight.data.copy_(rows(P["W2"].t().contiguous()).t())  # columns of W2 = rows of W2^T
            lin2.bias.data.copy_(P["b2"])
            bn2.weight.data.copy_(P["g2"])
            bn2.bias.data.copy_(P["be2"])
            bn2.running_mean.copy_(P["rm2"])
            bn2.running_var.copy_(P["rv2"])
            bn1.num_batches_tracked.fill_(self.t)
            bn2.num_batches_tracked.fill_(self.t)
