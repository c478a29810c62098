"""CPU ORACLE for the NestedLoRA / PDE hot path.  *** TEST INFRASTRUCTURE, NOT PRODUCT ***

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product path (``neural_svd_amd``) never does; it fails loudly when the HIP
library is missing.

It is a from-scratch restatement, in plain torch-CPU tensor algebra with a HAND-DERIVED backward
(no autograd), of what the reference computes on the path below.  Every function cites the
reference ``file:line`` it follows (paths relative to the upstream repo root).  dtype is a
parameter: float64 is the truth the HIP kernels are compared against, float32 reproduces the
reference's own arithmetic.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against the golden vectors in
``tests/golden/*.npz`` that were produced by importing the reference itself
(``tests/golden/make_golden.py``).  Unpinned piece: EMA (``torch_ema`` is an un-vendored,
un-versioned dependency of the reference that is not installed in the build image) - the formula
below is torch_ema's documented update; DESIGN.md says so too.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

SOFTPLUS_THRESHOLD = 20.0  # torch.nn.Softplus default (reference examples/models/mlp.py:84-85)
SQRT_P_CLAMP = 1e-5        # reference examples/operator/pde/diff_ops.py:15

POT_HYDROGEN = 0
POT_HARMONIC = 1


# ----------------------------------------------------------------------------- masks
def sequential_nesting_masks(L: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference methods/nestedlora.py:49-54: v = 1_L, M = triu(1_{LxL})."""
    return torch.ones(L), torch.triu(torch.ones(L, L))


def joint_nesting_masks(L: int, step: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference methods/nestedlora.py:183-192 (step weights) + :40-46 (reverse cumsum, min)."""
    ends = list(range(step, L + 1, step))
    if L not in ends:
        ends.append(L)
    w = np.zeros(L)
    w[np.array(ends) - 1] = 1.0
    w = w / w.sum()
    v = np.cumsum(w[::-1])[::-1].copy()
    v = torch.tensor(v).float()
    M = torch.minimum(v.unsqueeze(1), v.unsqueeze(0)).float()
    return v, M


# ----------------------------------------------------------------------------- loss
def evd_loss_forward(f, Tf, v, M):
    """reference methods/nestedlora.py:70-94 with f1,f2 = torch.chunk(f, 2) (:263).

    returns loss, lam1, lam2, loss_op, loss_metric
    """
    B = f.shape[0]
    B1 = (B + 1) // 2  # torch.chunk: first chunk gets the ceil
    f1, f2 = f[:B1], f[B1:]
    lam1 = f1.T @ f1 / f1.shape[0]            # :10-11
    lam2 = f2.T @ f2 / f2.shape[0]
    loss_metric = (M * lam1 * lam2).sum()     # :64
    loss_op = -2.0 * ((f * Tf) @ v).mean()    # :92
    return loss_op + loss_metric, lam1, lam2, loss_op, loss_metric


def svd_loss(f, Tg, g, Tadjf, v, M):
    """NestedLoRALossFunctionSVD forward + backward (reference methods/nestedlora.py:114-164): f, Tg (B1, L);
    g, Tadjf (B2, L). Returns loss, grad_f, grad_g. Pinned by tests/golden/svd_loss.npz."""
    lam_f = f.T @ f / f.shape[0]              # compute_loss_metric, :57-64
    lam_g = g.T @ g / g.shape[0]
    loss = -2.0 * ((f * Tg) @ v).mean() + (M * lam_f * lam_g).sum()                      # :139-141
    grad_f = -(2.0 / f.shape[0]) * Tg * v.unsqueeze(0) + (2.0 / f.shape[0]) * (f @ (M * lam_g))     # :157-159
    grad_g = -(2.0 / g.shape[0]) * Tadjf * v.unsqueeze(0) + (2.0 / g.shape[0]) * (g @ (M * lam_f))  # :161-163
    return loss, grad_f, grad_g


def evd_loss_backward(f, Tf, v, M, lam1, lam2, grad_output=1.0):
    """reference methods/nestedlora.py:98-111; returns d loss / d f with the f1/f2 terms summed in.

    NOTE (deliberate, reference behaviour): no gradient flows through Tf, and the metric term uses
    f1 @ (M*lam2) (not the symmetrised autograd gradient).
    """
    B = f.shape[0]
    B1 = (B + 1) // 2
    f1, f2 = f[:B1], f[B1:]
    g = -(4.0 / B) * Tf * v.unsqueeze(0)
    g1 = (2.0 / f1.shape[0]) * (f1 @ (M * lam2))
    g2 = (2.0 / f2.shape[0]) * (f2 @ (M * lam1))
    g = g.clone()
    g[:B1] += g1
    g[B1:] += g2
    return grad_output * g


# ----------------------------------------------------------------------------- dense kernel operator
def evd_loss_independent(f, Tf, f1, f2, v, M, grad_output=1.0):
    """NestedLoRALossFunctionEVD with f1, f2 that are NOT chunks of f (reference methods/nestedlora.py:70-111; :84 "f1
    and f2 must be independent"; the call of compute_loss_kernel(split_batch=True), :239-244). f, Tf (B, L), f1 (B1,
    L), f2 (B2, L). Returns loss, grad_f, grad_f1, grad_f2 - the three gradients SEPARATE (when the caller passes one
    tensor as f and f1, autograd adds them). Pinned by tests/golden/evd_loss_indep.npz."""
    lam1 = f1.T @ f1 / f1.shape[0]                                     # compute_lambda, :10-11
    lam2 = f2.T @ f2 / f2.shape[0]
    loss = -2.0 * ((f * Tf) @ v).mean() + (M * lam1 * lam2).sum()      # :88-93
    g = -(4.0 / f.shape[0]) * Tf * v.unsqueeze(0)                      # :108
    g1 = (2.0 / f1.shape[0]) * (f1 @ (M * lam2))                       # :109  'lm,lm,bl->bm'
    g2 = (2.0 / f2.shape[0]) * (f2 @ (M * lam1))                       # :110
    return loss, grad_output * g, grad_output * g1, grad_output * g2


def kernel_apply(K, rows, cols, f, scale=None):
    """Kf = scale * K[rows][:, cols] @ f (default scale 1 / len(cols)). PARITY UNPINNED: the reference has no kernel
    operator (only the consumer contract methods/nestedlora.py:230-252); this restates the build's own definition
     in the dtype of its inputs."""
    if scale is None:
        scale = 1.0 / len(cols)
    return scale * (K[rows][:, cols] @ f)


# ----------------------------------------------------------------------------- CDK loss
def off_diagonal(x):
    """reference methods/utils.py:16-22."""
    n = x.shape[0]
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


def joint_nesting_masks_from_weights(weights, set_first_mode_const=False):
    """reference methods/nestedlora.py:40-46 incl. the duplicated first entry for the constant mode."""
    v = list(np.cumsum(list(weights)[::-1])[::-1])
    if set_first_mode_const:
        v = [v[0]] + v
    v = torch.tensor(np.array(v)).float()
    return v, torch.minimum(v.unsqueeze(1), v.unsqueeze(0)).float()


def cdk_masks(L, sequential, step=1, set_first_mode_const=True):
    """NestedLoRAForCDK.__init__ (reference methods/nestedlora.py:345-359)."""
    if sequential:
        Lp = L + 1 if set_first_mode_const else L
        return torch.ones(Lp), torch.triu(torch.ones(Lp, Lp))
    ends = list(range(step, L + 1, step))
    if L not in ends:
        ends.append(L)
    w = np.zeros(L)
    w[np.array(ends) - 1] = 1.0
    return joint_nesting_masks_from_weights(w / w.sum(), set_first_mode_const)


def cdk_loss(f, g, v, M, set_first_mode_const=True, batch_weights=None):
    """NestedLoRALossFunctionForCDK forward + backward (reference methods/nestedlora.py:273-332).
    Returns loss, loss_operator, loss_metric, rs_joint, rs_indep, grad_f, grad_g. NOTE (reference
    behaviour): the returned gradients are w.r.t. the batch-weighted, padded features; the weights are
    not chain-ruled back, and the constant column's gradient is dropped."""
    if set_first_mode_const:
        one = torch.ones(f.shape[0], 1, dtype=f.dtype)
        f, g = torch.cat([one, f], 1), torch.cat([one, g], 1)
    if batch_weights is not None:
        f, g = f * batch_weights, g * batch_weights
    B = f.shape[0]
    lam_f, lam_g = f.T @ f / B, g.T @ g / B
    loss_metric = (M * lam_f * lam_g).sum()
    loss_op = -2.0 * ((f * g) @ v).mean()
    gram = f @ g.T
    gf = -(2.0 / B) * g * v.unsqueeze(0) + (2.0 / B) * (f @ (M * lam_g))
    gg = -(2.0 / B) * f * v.unsqueeze(0) + (2.0 / B) * (g @ (M * lam_f))
    if set_first_mode_const:
        gf, gg = gf[:, 1:], gg[:, 1:]
    return loss_op + loss_metric, loss_op, loss_metric, gram.diag(), off_diagonal(gram), gf, gg


def row_normalize(z, r_up, mode):
    """normalize(z, r_up, mode) of the CDK towers (reference examples/models/siam.py:170-183), modes 'l2_ball' and
    'l2_sphere'; F.normalize(z, p=2, dim=1) = z / max(||z||, 1e-12). Differentiable (plain torch ops): the tests take
    its autograd gradient. Pinned by tests/golden/normalize.npz."""
    nrm = torch.linalg.vector_norm(z, dim=1, keepdim=True)
    unit = z / torch.clamp(nrm, min=1e-12)
    if mode == "l2_sphere":
        return r_up * unit
    # a constant of the graph, and float32 as in the reference (`.float()`): `(1 - mask) * r_up` is therefore formed in
    # float32 - in a float64 run r_up reaches the scaled rows rounded to float32
    mask = (nrm < r_up).float()
    return mask * z + (1 - mask) * r_up * unit


# ----------------------------------------------------------------------------- model
@dataclass
class Params:
    """Trainable + frozen tensors of WaveFunctions(ParallelMLP(FourierFeatures)) in the reference's
    state_dict layout: ws[i]: (L, h_i, h_{i-1}), bs[i]: (L, h_i, 1), fourier_B: (D, m),
    scales: (L,) or None (ExponentialMask)."""
    ws: List[torch.Tensor]
    bs: List[torch.Tensor]
    fourier_B: torch.Tensor
    scales: Optional[torch.Tensor] = None

    def to(self, dtype):
        return Params([w.to(dtype) for w in self.ws], [b.to(dtype) for b in self.bs],
                      self.fourier_B.to(dtype), None if self.scales is None else self.scales.to(dtype))

    def clone(self):
        return Params([w.clone() for w in self.ws], [b.clone() for b in self.bs],
                      self.fourier_B.clone(), None if self.scales is None else self.scales.clone())

    def trainable(self) -> List[torch.Tensor]:
        """Order of ``method.parameters()`` in the reference with requires_grad=True:
        model.base.ws.*, model.base.bs.*, then model.boundary_mask.scales."""
        out = list(self.ws) + list(self.bs)
        if self.scales is not None:
            out.append(self.scales)
        return out


@dataclass
class Problem:
    """Scalar configuration of OperatorWrapper(NegativeHamiltonian(...)) + Gaussian importance."""
    potential: int = POT_HYDROGEN     # reference examples/operator/pde/schrodinger/potentials.py:5-8 / :24-27
    charge_or_k: float = 1.0
    scale_kinetic: float = 1.0        # reference examples/operator/pde/problems.py:26
    eps: float = 0.01                 # --laplacian_eps
    op_scale: float = 1.0             # reference examples/__init__.py:9
    op_shift: float = 0.0
    sigma: float = 16.0               # --sampling_scale (Gaussian importance, main_pde.py:94-100)
    hard_mul_const: float = 1.0       # reference examples/operator/pde/__init__.py:16
    use_importance: bool = True


def init_params(L, D, mapping_size, hidden, fourier_scale, exp_mask_init=None, seed=None) -> Params:
    """Same draw ORDER as the reference so that torch.manual_seed(seed) reproduces its weights:
    Fourier _B first (examples/operator/pde/__init__.py:21-28 -> examples/utils.py:116-119), then
    per layer W ~ sqrt(2/fan_in) randn(L, h, h_prev), b = 0 (examples/models/mlp.py:185-189)."""
    if seed is not None:
        torch.manual_seed(seed)
    fB = 2 * torch.pi * fourier_scale * torch.randn((D, mapping_size)).float()
    ws, bs = [], []
    prev = 2 * mapping_size
    for h in list(hidden) + [1]:
        ws.append(math.sqrt(2.0 / prev) * torch.randn(L, h, prev))
        bs.append(torch.zeros(L, h, 1))
        prev = h
    scales = None if exp_mask_init is None else exp_mask_init * torch.ones(L)
    return Params(ws, bs, fB, scales)


def softplus(z):
    """torch.nn.Softplus(beta=1, threshold=20): z if z > 20 else log1p(exp(z))."""
    return torch.where(z > SOFTPLUS_THRESHOLD, z, torch.log1p(torch.exp(torch.clamp(z, max=SOFTPLUS_THRESHOLD))))


def softplus_grad(z):
    """d softplus / dz = sigmoid(z) (1 above the threshold)."""
    return torch.where(z > SOFTPLUS_THRESHOLD, torch.ones_like(z), torch.sigmoid(z))


def fourier_features(x, fB):
    """reference examples/utils.py:139-140: [sin(x B), cos(x B)]."""
    proj = x @ fB
    return torch.cat([torch.sin(proj), torch.cos(proj)], dim=1)


def mlp_forward(phi, p: Params, keep=False):
    """reference examples/models/mlp.py:204-221 (norm()==1, bias=True). phi: (R, F).
    returns out (R, L) and, if keep, the list of pre-activations z_i (L, h_i, R)."""
    zs = []
    h = torch.einsum("lhd,bd->lhb", p.ws[0], phi) + p.bs[0]
    zs.append(h)
    h = softplus(h)
    n = len(p.ws)
    for i in range(1, n):
        h = torch.einsum("lhp,lpb->lhb", p.ws[i], h) + p.bs[i]
        if i < n - 1:
            zs.append(h)
            h = softplus(h)
    out = h.permute(2, 0, 1).reshape(phi.shape[0], -1)  # (R, L); output_dim == 1
    return (out, zs) if keep else out


def boundary_mask(x, p: Params):
    """ExponentialMask: exp(-|x| / s_l) (reference examples/operator/pde/boundary.py:46-53);
    1 when absent (examples/operator/pde/__init__.py:46)."""
    if p.scales is None:
        return None
    r = torch.linalg.norm(x, dim=-1).view(-1, 1)
    return torch.exp(-r / p.scales.view(1, -1))


def sqrt_importance(x, sigma):
    """sqrt of the isotropic Gaussian pdf (reference main_pde.py:94-100 via MultivariateNormal)."""
    D = x.shape[1]
    logp = -0.5 * (x * x).sum(-1) / sigma ** 2 - D * math.log(sigma) - 0.5 * D * math.log(2 * math.pi)
    return torch.exp(logp).sqrt().view(-1, 1)


def potential(x, prob: Problem):
    r = torch.linalg.norm(x, dim=1)
    if prob.potential == POT_HYDROGEN:
        return (-prob.charge_or_k / r).view(-1, 1)
    return (prob.charge_or_k * r ** 2).view(-1, 1)


def stencil_points(x, eps):
    """Order used everywhere in this repo: [x, x+e*e_0, x-e*e_0, x+e*e_1, x-e*e_1, ...]
    (reference examples/operator/pde/diff_ops.py:36-45). eps is added as a float32 value, exactly
    like the reference's float32 ``epsi`` tensor."""
    D = x.shape[1]
    e32 = float(np.float32(eps))
    pts = [x]
    for i in range(D):
        d = torch.zeros(1, D, dtype=x.dtype)
        d[0, i] = e32
        pts.append(x + d)
        pts.append(x - d)
    return pts


@dataclass
class OperatorCache:
    x: torch.Tensor
    phi0: torch.Tensor
    zs: List[torch.Tensor]
    base0: torch.Tensor
    mask0: Optional[torch.Tensor]
    sp0: torch.Tensor
    spc0: torch.Tensor
    f: torch.Tensor
    Tf: torch.Tensor


def exact_jets(x, p: Params, prob: Problem):
    """Value, gradient and Laplacian of g(x) = sqrt p(x) * c * model(x) * mask(x) in closed form: what the reference's
    exact mode (laplacian_eps <= 0: VectorizedLaplacian.exact_laplacian, diff_ops.py:54-61, via double autograd
    :64-111) differentiates. Forward-mode propagation of (value, D first derivatives, Laplacian):
      features   phi = [sin t, cos t], t = x B:   d_d phi = [B_d cos t, -B_d sin t],  Lap phi = -|B_j|^2 phi
      linear     all streams through W; the bias joins the value stream only
      softplus   a = sp(z):  d a = s z',  Lap a = s Lap z + s (1 - s) sum_d (d_d z)^2,   s = sigmoid(z)
      product    Lap(u v) = u Lap v + 2 grad u . grad v + v Lap u   with u = sqrt p * mask (radial, closed form).
    returns g (B, L), lap_g (B, L) and the pieces the backward needs (phi, zs, base, mask, sp)."""
    D = x.shape[1]
    fB = p.fourier_B
    t = x @ fB
    sn, cs = torch.sin(t), torch.cos(t)
    phi = torch.cat([sn, cs], 1)
    dphi = [torch.cat([cs * fB[d], -sn * fB[d]], 1) for d in range(D)]
    lphi = -torch.cat([(fB * fB).sum(0), (fB * fB).sum(0)]) * phi
    n = len(p.ws)
    zs = []
    z = torch.einsum("lhd,bd->lhb", p.ws[0], phi) + p.bs[0]
    dz = [torch.einsum("lhd,bd->lhb", p.ws[0], dphi[d]) for d in range(D)]
    lz = torch.einsum("lhd,bd->lhb", p.ws[0], lphi)
    for i in range(1, n):
        zs.append(z)
        s1 = softplus_grad(z)
        s2 = torch.where(z > SOFTPLUS_THRESHOLD, torch.zeros_like(z), s1 * (1 - s1))
        a = softplus(z)
        la = s1 * lz + s2 * sum(d_ * d_ for d_ in dz)
        da = [s1 * d_ for d_ in dz]
        z = torch.einsum("lhp,lpb->lhb", p.ws[i], a) + p.bs[i]
        dz = [torch.einsum("lhp,lpb->lhb", p.ws[i], d_) for d_ in da]
        lz = torch.einsum("lhp,lpb->lhb", p.ws[i], la)
    B = x.shape[0]
    to_bl = lambda h: h.permute(2, 0, 1).reshape(B, -1)  # noqa: E731
    base, dbase, lbase = to_bl(z), [to_bl(d_) for d_ in dz], to_bl(lz)
    # u = c * sqrt p * mask and its derivatives
    r = torch.linalg.norm(x, dim=-1).view(-1, 1)
    if prob.use_importance:
        sp = sqrt_importance(x, prob.sigma)
        dsp = [-(x[:, d:d + 1] / (2 * prob.sigma ** 2)) * sp for d in range(D)]
        lsp = (-D / (2 * prob.sigma ** 2) + (r * r) / (4 * prob.sigma ** 4)) * sp
    else:
        sp = torch.ones(B, 1, dtype=x.dtype)
        dsp = [torch.zeros(B, 1, dtype=x.dtype) for _ in range(D)]
        lsp = torch.zeros(B, 1, dtype=x.dtype)
    mask = boundary_mask(x, p)
    if mask is None:
        mk = torch.ones(B, 1, dtype=x.dtype)
        dmk = [torch.zeros(B, 1, dtype=x.dtype) for _ in range(D)]
        lmk = torch.zeros(B, 1, dtype=x.dtype)
    else:
        sc = p.scales.view(1, -1)
        mk = mask
        dmk = [-(x[:, d:d + 1] / r) / sc * mk for d in range(D)]
        lmk = (1.0 / sc ** 2 - (D - 1) / (r * sc)) * mk
    c = prob.hard_mul_const
    u = c * sp * mk
    du = [c * (dsp[d] * mk + sp * dmk[d]) for d in range(D)]
    lu = c * (lsp * mk + 2 * sum(dsp[d] * dmk[d] for d in range(D)) + sp * lmk)
    g = u * base
    lap_g = u * lbase + 2 * sum(du[d] * dbase[d] for d in range(D)) + base * lu
    return g, lap_g, (phi, zs, base, mask, sp)


def operator_forward(x, p: Params, prob: Problem) -> OperatorCache:
    """Tf, f = OperatorWrapper(NegativeHamiltonian)(method, x, importance)
    reference: examples/__init__.py:7-9 -> schrodinger/__init__.py:16-22 -> diff_ops.py:9-52.
    prob.eps <= 0 selects the exact Laplacian like the reference does (diff_ops.py:7)."""
    D = x.shape[1]
    if prob.eps <= 0:
        g, lap, (phi0, zs, base0, mask0, sp0) = exact_jets(x, p, prob)
        spc0 = torch.clamp(sp0, min=SQRT_P_CLAMP) if prob.use_importance else sp0
        lap = lap / spc0
        fs = g / spc0
        Tf = -(-prob.scale_kinetic * lap + potential(x, prob) * fs)
        Tf = prob.op_scale * Tf + prob.op_shift * fs
        return OperatorCache(x, phi0, zs, base0, mask0, sp0, spc0, fs, Tf)
    pts = stencil_points(x, prob.eps)
    gs = []
    cache0 = None
    for j, xe in enumerate(pts):
        phi = fourier_features(xe, p.fourier_B)
        if j == 0:
            base, zs = mlp_forward(phi, p, keep=True)
        else:
            base = mlp_forward(phi, p)
        m = boundary_mask(xe, p)
        model = prob.hard_mul_const * base
        if m is not None:
            model = model * m
        sp = sqrt_importance(xe, prob.sigma) if prob.use_importance else torch.ones(x.shape[0], 1, dtype=x.dtype)
        gs.append(sp * model)
        if j == 0:
            cache0 = (phi, zs, base, m, sp)
    lap = -2 * D * gs[0]
    for i in range(D):
        lap = lap + (gs[1 + 2 * i] + gs[2 + 2 * i])
    lap = lap / (prob.eps ** 2)
    phi0, zs, base0, mask0, sp0 = cache0
    spc0 = torch.clamp(sp0, min=SQRT_P_CLAMP) if prob.use_importance else sp0
    lap = lap / spc0
    fs = gs[0] / spc0
    kinetic = -prob.scale_kinetic * lap
    pot = potential(x, prob) * fs
    Tf = -(kinetic + pot)
    Tf = prob.op_scale * Tf + prob.op_shift * fs
    return OperatorCache(x, phi0, zs, base0, mask0, sp0, spc0, fs, Tf)


def operator_backward(c: OperatorCache, p: Params, prob: Problem, df):
    """Hand-derived gradient of sum(df * f) w.r.t. the trainable parameters; f = g(x)/clamp(sqrt p)
    is the ONLY differentiable output (Tf carries no gradient: reference nestedlora.py:108-111).
    Returns grads in ``Params.trainable()`` order."""
    n = len(p.ws)
    dmodel = df * (c.sp0 / c.spc0)
    dscales = None
    if c.mask0 is not None:
        r = torch.linalg.norm(c.x, dim=-1).view(-1, 1)
        # d/ds exp(-r/s) = exp(-r/s) * r / s^2
        dscales = (dmodel * prob.hard_mul_const * c.base0 * c.mask0 * r / p.scales.view(1, -1) ** 2).sum(0)
        dbase = dmodel * prob.hard_mul_const * c.mask0
    else:
        dbase = dmodel * prob.hard_mul_const
    # dbase: (B, L) -> (L, 1, B)
    dz = dbase.T.unsqueeze(1)
    dws = [None] * n
    dbs = [None] * n
    hs = [softplus(z) for z in c.zs]  # (L, h, B)
    for i in range(n - 1, -1, -1):
        a_prev = hs[i - 1] if i > 0 else None
        if i > 0:
            dws[i] = torch.einsum("lhb,lpb->lhp", dz, a_prev)
        else:
            dws[i] = torch.einsum("lhb,bd->lhd", dz, c.phi0)
        dbs[i] = dz.sum(-1, keepdim=True)
        if i > 0:
            dh = torch.einsum("lhp,lhb->lpb", p.ws[i], dz)
            dz = dh * softplus_grad(c.zs[i - 1])
    out = dws + dbs
    if dscales is not None:
        out.append(dscales)
    return out


def loss_and_grads(x, p: Params, prob: Problem, v, M):
    """One NestedLoRA.compute_loss_operator + backward (reference nestedlora.py:254-267)."""
    c = operator_forward(x, p, prob)
    v = v.to(x.dtype)
    M = M.to(x.dtype)
    loss, lam1, lam2, loss_op, loss_metric = evd_loss_forward(c.f, c.Tf, v, M)
    df = evd_loss_backward(c.f, c.Tf, v, M, lam1, lam2)
    grads = operator_backward(c, p, prob, df)
    return dict(loss=loss, f=c.f, Tf=c.Tf, lam1=lam1, lam2=lam2, df=df, grads=grads)


def gaussian_kernel_apply(x, x_ref, f_ref, ell):
    """The toy kernel operator of tests/golden/kernel_loss.npz (tests/golden/make_golden.py:gaussian_kernel_op_factory):
    Kf(x) = (1 / B_ref) sum_j exp(-|x - x_ref_j|^2 / (2 ell^2)) f(x_ref_j)."""
    d2 = ((x[:, None, :] - x_ref[None, :, :]) ** 2).sum(-1)
    return torch.exp(-d2 / (2.0 * ell ** 2)) @ f_ref / x_ref.shape[0]


def kernel_loss_and_grads(x, p: Params, kernel_fn, v, M, split_batch: bool, hard_mul_const: float = 1.0):
    """NestedLoRA.compute_loss_kernel + backward (reference methods/nestedlora.py:230-252) for an operator of the form
    Kf = kernel_fn(x_eval, x_ref, model(x_ref)) that carries no gradient (the loss Function returns none for Tf,
    :108-111). Pinned by tests/golden/kernel_loss.npz. Returns dict(loss, f, Kf, grads in Params.trainable() order).
      split_batch=False (:246-249): Kf, f on the whole batch; f1, f2 = chunk(f, 2).
      split_batch=True  (:239-245): x1, x2 = chunk(x, 2); Kf1 = K applied to x1 against the reference batch x2,
        f2 = model(x2); loss(f1, Kf1, f1, f2): operator term and its gradient carry 1 / B1."""
    plain = Problem(potential=POT_HARMONIC, eps=0.01, use_importance=False, hard_mul_const=hard_mul_const)
    v, M = v.to(x.dtype), M.to(x.dtype)
    if not split_batch:
        c = operator_forward(x, p, plain)             # c.f = model(x): no importance, the stencil part is unused
        Kf = kernel_fn(x, x, c.f)
        loss, lam1, lam2, _, _ = evd_loss_forward(c.f, Kf, v, M)
        grads = operator_backward(c, p, plain, evd_loss_backward(c.f, Kf, v, M, lam1, lam2))
        return dict(loss=loss, f=c.f, Kf=Kf, grads=grads)
    B = x.shape[0]
    B1 = (B + 1) // 2
    c1, c2 = operator_forward(x[:B1], p, plain), operator_forward(x[B1:], p, plain)
    f1, f2 = c1.f, c2.f
    Kf1 = kernel_fn(x[:B1], x[B1:], f2)
    lam1, lam2 = f1.T @ f1 / B1, f2.T @ f2 / (B - B1)                                   # :10-11
    loss = -2.0 * ((f1 * Kf1) @ v).mean() + (M * lam1 * lam2).sum()                     # :92, :64
    df1 = -(4.0 / B1) * Kf1 * v.unsqueeze(0) + (2.0 / B1) * (f1 @ (M * lam2))           # :108-109
    df2 = (2.0 / (B - B1)) * (f2 @ (M * lam1))                                          # :110
    grads = [a + b for a, b in zip(operator_backward(c1, p, plain, df1), operator_backward(c2, p, plain, df2))]
    return dict(loss=loss, f=f1, Kf=Kf1, grads=grads)


# ----------------------------------------------------------------------------- CDK tower
def _bf16_round(t):
    """round to nearest even to bfloat16 and back to t's dtype (the operand rounding of the mixed-precision mode)"""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def _f16_round(t):
    """round to nearest even to IEEE float16 (overflow -> inf) and back: the half type of the reference's autocast
    (examples/cdk/sketchy/main_sketchy.py:182)"""
    return t.to(torch.float32).to(torch.float16).to(t.dtype)


def tower_forward_backward(x, P, dz, slope, eps=1e-5, gemm_bf16=False, half="bf16"):
    """Linear -> BatchNorm1d (training) -> LeakyReLU(slope) -> Linear -> BatchNorm1d (training) and the hand-derived
    gradient of sum(dz * z): what get_mlp(sizes=[d0, d1, d2], bias=True, nonlinearity='lrelu<slope>', use_bn=True)
    (reference examples/models/mlp.py:129-164) computes with autograd. P: dict W1 (d1, d0), b1, g1, be1 (d1), W2
    (d2, d1), b2, g2, be2 (d2) in torch's layouts. Pinned by tests/golden/tower.npz.
    Returns z, dict of gradients, and the batch statistics (mean, biased var, unbiased var) of both BatchNorms.
    gemm_bf16: the build's mixed-precision mode (no reference counterpart to pin it to: the reference's autocast is
    float16, has a GradScaler and covers more ops): the values the HIP path STORES as bfloat16 are rounded to bfloat16
    here - X, W1, W2; Y1 = X W1^T + b1; A1 = lrelu(BN1(Y1)); dY2, dA1, dY1 - products, sums, BatchNorm statistics, the
    narrow end (Y2, Z) and every parameter gradient exact in this dtype; the bias gradients are column sums of the
    unrounded dY (include/nsvd.h, nsvd_tower_forward).
    gemm_bf16 == "fused": the mixed-precision form with BatchNorm-1 inside the wide contractions (csrc/tower_col.h;
    include/nsvd.h, nsvd_tower_mixed_fused): Y1 and dA1 are NOT rounded (they never leave the accumulators), and the
    backward of BatchNorm-1 recovers the normalised value from the ROUNDED activation it stored,
    h = A1 > 0 ? A1 : A1 / slope, yhat = (h - beta1) / gamma1, lrelu' from the sign of A1 (slope > 0).
    half: "bf16" or "f16" - the 16-bit type those roundings go to (f16: gemm_bf16 flag bit 4 of include/nsvd.h)."""
    B = x.shape[0]
    fused = isinstance(gemm_bf16, str)
    if fused and gemm_bf16 != "fused":
        raise ValueError("gemm_bf16: False, True or 'fused'")
    rh = {"bf16": _bf16_round, "f16": _f16_round}[half]
    r = rh if gemm_bf16 else (lambda t: t)
    r1 = (lambda t: t) if fused else r  # the rounding of Y1 and dA1

    def bn(y, g, be):
        mu = y.mean(0)
        var = ((y - mu) ** 2).mean(0)             # biased: what the normalisation uses
        inv = 1.0 / torch.sqrt(var + eps)
        yh = (y - mu) * inv
        return yh * g + be, yh, inv, (mu, var, var * B / (B - 1))

    def bn_back(dh, yh, inv, g):
        return g * inv * (dh - dh.mean(0) - yh * (dh * yh).mean(0)), (dh * yh).sum(0), dh.sum(0)

    xh, W1h, W2h = r(x), r(P["W1"]), r(P["W2"])
    y1 = r1(xh @ W1h.T + P["b1"])
    h1, yh1, inv1, st1 = bn(y1, P["g1"], P["be1"])
    a1 = r(torch.where(h1 > 0, h1, slope * h1))
    y2 = a1 @ W2h.T + P["b2"]
    z, yh2, inv2, st2 = bn(y2, P["g2"], P["be2"])
    dy2, dg2, dbe2 = bn_back(dz, yh2, inv2, P["g2"])
    da1 = r1(r(dy2) @ W2h)
    if fused:  # what the backward kernel sees is the stored activation, not Y1
        h1 = torch.where(a1 > 0, a1, a1 / slope)
        yh1 = (h1 - P["be1"]) / P["g1"]
    dh1 = da1 * torch.where(h1 > 0, torch.ones_like(h1), torch.full_like(h1, slope))
    dy1, dg1, dbe1 = bn_back(dh1, yh1, inv1, P["g1"])
    grads = dict(W1=r(dy1).T @ xh, b1=dy1.sum(0), g1=dg1, be1=dbe1, W2=r(dy2).T @ a1, b2=dy2.sum(0), g2=dg2,
                 be2=dbe2)
    return z, grads, (st1, st2)



def cdk_train_step(x, y, towers, bufs, running, v, M, mu, lr, momentum, max_norm, slope, first_step, eps=1e-5,
                   bn_momentum=0.1, gemm_bf16=False, half="bf16", scaler=None):
    """One Sketchy-style CDK training step (reference examples/cdk/sketchy/main_sketchy.py:180-212 with
    scripts/exps/sketchy.sh's switches, AMP off): two towers (get_mlp, examples/models/mlp.py:129-164) behind Identity
    projectors and normalize('l2_ball', sqrt(mu)) (examples/models/siam.py:156-183), NestedLoRAForCDK loss
    (methods/nestedlora.py:273-332, set_first_mode_const), clip_grad_norm_(max_norm) over all 16 parameter tensors
    (torch: coefficient min(1, max_norm / (total_norm + 1e-6))), then torch.optim.SGD with momentum (no dampening, no
    nesterov, no weight decay: buf = g on the first step, else momentum * buf + g; p -= lr * buf). In place on
    `towers` (two dicts W1 b1 g1 be1 W2 b2 g2 be2), `bufs` (same keys) and `running` (two dicts rm1 rv1 rm2 rv2).
    Returns (loss, operator term, metric term), total gradient norm. Pinned by tests/golden/cdk_step.npz.
    scaler: None, or a dict - torch.cuda.amp.GradScaler as the script's AMP branch drives it (main_sketchy.py:161,194-208:
    scaler.scale(loss).backward(); scaler.unscale_(optimizer); clip_grad_norm_; scaler.step(optimizer); scaler.update();
    lr_scheduler.step() on every iteration, :205-206 - `lr` stays the caller's scheduled value) with keys scale, growth_factor, backoff_factor, growth_interval,
    growth_tracker, steps_ok, steps_skipped (updated in place): the loss gradient is multiplied by scale before the
    towers' backward; found_inf = the norm of the SCALED gradients is not finite -> nothing is updated, scale *= backoff;
    else gradients * (1 / scale), clip, SGD, steps_ok += 1, and every growth_interval clean steps scale *= growth.
    first_step is then steps_ok == 0 (torch.optim.SGD creates its momentum buffers in the first step it executes)."""
    r_up = float(mu) ** 0.5
    if scaler is not None:
        first_step = scaler["steps_ok"] == 0
    zs, embs = [], []
    for inp, P in ((x, towers[0]), (y, towers[1])):
        z, _, _ = tower_forward_backward(inp, P, torch.zeros(inp.shape[0], P["W2"].shape[0], dtype=inp.dtype), slope, eps,
                                         gemm_bf16, half)
        zr = z.detach().clone().requires_grad_(True)
        zs.append(zr)
        embs.append(row_normalize(zr, r_up, "l2_ball"))
    loss, lop, lmet, _, _, gf, gg = cdk_loss(embs[0].detach(), embs[1].detach(), v, M, True)
    grads = []
    for inp, P, zr, e, ge, run in ((x, towers[0], zs[0], embs[0], gf, running[0]),
                                   (y, towers[1], zs[1], embs[1], gg, running[1])):
        (dz,) = torch.autograd.grad(e, zr, ge * (scaler["scale"] if scaler is not None else 1.0))
        _, g, (st1, st2) = tower_forward_backward(inp, P, dz, slope, eps, gemm_bf16, half)
        grads.append(g)
        for tag, st in (("1", st1), ("2", st2)):
            run["rm" + tag].mul_(1 - bn_momentum).add_(bn_momentum * st[0])
            run["rv" + tag].mul_(1 - bn_momentum).add_(bn_momentum * st[2])
    total = torch.sqrt(sum((g[k].double() ** 2).sum() for g in grads for k in g)).to(x.dtype)
    if scaler is not None:
        if not bool(torch.isfinite(total)):  # GradScaler.step() skips optimizer.step(); update() backs the scale off
            scaler["scale"] *= scaler["backoff_factor"]
            scaler["growth_tracker"] = 0
            scaler["steps_skipped"] += 1
            return (loss, lop, lmet), total
        inv = 1.0 / scaler["scale"]
        grads = [{k: gk * inv for k, gk in g.items()} for g in grads]  # scaler.unscale_()
        total = total * inv
        scaler["steps_ok"] += 1
        scaler["growth_tracker"] += 1
        if scaler["growth_tracker"] == scaler["growth_interval"]:
            scaler["scale"] *= scaler["growth_factor"]
            scaler["growth_tracker"] = 0
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0) if max_norm and max_norm > 0 else torch.ones_like(total)
    for P, B, g in zip(towers, bufs, grads):
        for k in g:
            gk = g[k] * coef
            if first_step:
                B[k].copy_(gk)
            else:
                B[k].mul_(momentum).add_(gk)
            P[k].sub_(lr * B[k])
    return (loss, lop, lmet), total

# ----------------------------------------------------------------------------- optimiser
def cosine_lr(base_lr, t, T, eta_min=0.0):
    """closed form of torch.optim.lr_scheduler.CosineAnnealingLR after t scheduler steps
    (reference examples/operator/__init__.py:35,71-72)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / T)) / 2


def rmsprop_step(params, grads, sq, lr, alpha=0.999, eps=1e-10):
    """torch.optim.RMSprop(momentum=0, centered=False, weight_decay=0) as configured by the
    reference (examples/utils.py:50-57): v = a v + (1-a) g^2 ; p -= lr g / (sqrt(v) + eps)."""
    for p_, g, s in zip(params, grads, sq):
        s.mul_(alpha).addcmul_(g, g, value=1 - alpha)
        p_.addcdiv_(g, s.sqrt().add_(eps), value=-lr)


def ema_update(shadow, params, decay, num_updates):
    """torch_ema.ExponentialMovingAverage.update (documented behaviour; PARITY UNPINNED):
    num_updates += 1; d = min(decay, (1+n)/(10+n)); s -= (1-d)(s-p). Returns new num_updates."""
    num_updates += 1
    d = min(decay, (1 + num_updates) / (10 + num_updates))
    for s, p_ in zip(shadow, params):
        s.sub_((1.0 - d) * (s - p_))
    return num_updates


# ----------------------------------------------------------------------------- spectrum
def validation_grid(lim, val_eps, D=2):
    """reference main_pde.py:121-125."""
    ax = np.arange(-lim, lim, val_eps)
    xxs = np.meshgrid(*(D * [ax]))
    pts = np.array(list(zip(*[xx.flatten() for xx in xxs])))
    return torch.tensor(pts).float()


def spectrum_evd(grid, p: Params, prob: Problem, lim, chunk=4096):
    """reference methods/spectrum.py:29-102 with importance_val = uniform on [-lim,lim]^D
    (main_pde.py:129-130): cov = sum phi^T phi / n, quad = sum phi^T Tphi / n, RQ eigvals."""
    L = p.ws[0].shape[0]
    D = grid.shape[1]
    cov = torch.zeros(L, L, dtype=grid.dtype)
    quad = torch.zeros(L, L, dtype=grid.dtype)
    n = 0
    # the reference builds this density as a float32 tensor (main_pde.py:130 `.float()`)
    sqrt_val = math.sqrt(float(np.float32(1.0 / (2 * lim) ** D)))
    for i in range(0, grid.shape[0], chunk):
        x = grid[i:i + chunk]
        c = operator_forward(x, p, prob)
        sp_train = sqrt_importance(x, prob.sigma) if prob.use_importance else 1.0
        w = sp_train / sqrt_val                                  # spectrum.py:56-61
        phi = torch.nan_to_num(w * c.f)
        Tphi = torch.nan_to_num(w * c.Tf)
        zero = torch.all(torch.isclose(x, torch.zeros_like(x[0])), dim=1)  # :73
        Tphi[zero] = 0.0
        cov += phi.T @ phi
        quad += phi.T @ Tphi
        n += x.shape[0]
    cov /= n
    quad /= n
    eig = torch.diag(quad) / torch.diag(cov)
    norms = torch.diag(cov)
    return dict(cov=cov, quad=quad, eigvals=eig, norms=norms)


# ----------------------------------------------------------------------------- ground truth
def hydrogen2d_eigvals(neigs, charge=1.0):
    """reference examples/operator/pde/schrodinger/ground_truths.py:120-132."""
    q = []
    n = 0
    while len(q) < neigs:
        q += [n] * (2 * n + 1)
        n += 1
    q = np.array(q[:neigs], dtype=np.float64)
    return -charge ** 2 / (4 * (q + 0.5) ** 2)


def oscillator2d_eigvals(neigs, k=1.0):
    """reference ground_truths.py:78-90 INCLUDING its quirk: it emits every shell up to and
    including one shell past the one that reaches ``neigs`` (never truncated)."""
    nend, states = 0, 0
    while True:
        states += nend + 1  # binom(2+n-1, n) = n+1
        nend += 1
        if states >= neigs:
            break
    return math.sqrt(k) * np.concatenate([(n + 1) * [2 * n + 2] for n in range(nend + 1)]).astype(np.float64)
