"""Operator side of the PDE path with the reference's names: potentials, NegativeHamiltonian,
OperatorWrapper, get_problem, the Gaussian sampler / importance and the analytic spectra.

    hydrogen_potential / harmonic_oscillator_potential  examples/operator/pde/schrodinger/potentials.py:5-8,24-27
    NegativeHamiltonian                                 examples/operator/pde/schrodinger/__init__.py:4-22
    OperatorWrapper                                     examples/__init__.py:1-9
    get_problem                                         examples/operator/pde/problems.py:23-130 (sch: hydrogen, oscillator)
    get_dataloader                                      examples/operator/pde/main_pde.py:89-130 (gaussian sampler)
    Hydrogen2D / HarmonicOscillator .get_eigvals        examples/operator/pde/schrodinger/ground_truths.py:78-90,120-132

These objects are DESCRIPTORS on the scripts' configuration (Gaussian sampler / importance, or none): calling
``operator(method, x, importance)`` forwards to the fused HIP kernel (nsvd_operator_forward). With the two other
samplers of main_pde.py:101-118 (`--sampling_mode laplacian / uniform`: their importance densities are not in the fused
kernel's epilogue) the wrapper applies the reference's finite-difference stencil itself (diff_ops.py:9-52,
schrodinger/__init__.py:16-22, examples/__init__.py:7-9) around 1 + 2D evaluations of the HIP model
(nsvd_model_forward / _backward): "python operator + fused loss kernel"  - still all on the GPU.
"""
from __future__ import annotations

import math
from functools import partial

import numpy as np
import torch

from . import hip_ops as H
from ._lib import NsvdError


# ----------------------------------------------------------------------------------- potentials
def hydrogen_potential(x, charge=1.0):
    """-Z / ||x|| (potentials.py:5-8). The fused kernels evaluate it in their epilogue (fd_math.h); this torch form
    serves the stencil applied outside them (non-Gaussian importance) and foreign operators."""
    x = x.reshape(x.shape[0], -1)
    return -(charge / x.norm(dim=1, p=2)).reshape(-1, 1)


def harmonic_oscillator_potential(x, k=1.0):
    """k ||x||^2 (potentials.py:24-27)."""
    x = x.reshape(x.shape[0], -1)
    return (k * x.norm(dim=1, p=2) ** 2).reshape(-1, 1)


def _potential_kind(ftn):
    base, kw = ftn, {}
    if isinstance(ftn, partial):
        base, kw = ftn.func, ftn.keywords
    if base is hydrogen_potential:
        return H.POT_HYDROGEN, float(kw.get("charge", 1.0))
    if base is harmonic_oscillator_potential:
        return H.POT_HARMONIC, float(kw.get("k", 1.0))
    raise NsvdError("HIP path supports hydrogen_potential and harmonic_oscillator_potential only")


class NegativeHamiltonian:
    def __init__(self, local_potential_ftn, scale_kinetic=1.0, laplacian_eps=1e-5, n_particles=1):
        self.potential_kind, self.potential_param = _potential_kind(local_potential_ftn)
        self.local_potential_ftn = local_potential_ftn
        self.scale_kinetic = scale_kinetic
        self.laplacian_eps = laplacian_eps
        self.n_particles = n_particles
        # laplacian_eps <= 0: exact Laplacian (reference diff_ops.py:7,54-61) - forward-mode jets on the MFMA path

    def __call__(self, f, xs, importance=None, threshold=1e5):
        return OperatorWrapper(self)(f, xs, importance)


class OperatorWrapper:
    def __init__(self, operator, scale=1.0, shift=0.0):
        if not isinstance(operator, NegativeHamiltonian):
            raise NotImplementedError("HIP path: OperatorWrapper wraps this package's NegativeHamiltonian")
        self.operator, self.scale, self.shift = operator, scale, shift

    def __call__(self, model, x, importance=None):
        """returns (scale * Tf + shift * f, f) like the reference; ``model`` is the NestedLoRA method."""
        return model.apply_operator(self, x, importance)

    def fused(self, importance) -> bool:
        """does the fused kernel (nsvd_operator_forward) implement this importance density?"""
        return importance is None or isinstance(importance, GaussianImportance)

    def apply_stencil(self, model, x, importance):
        """The reference's own op sequence for densities the fused kernel does not carry (Laplace, uniform, any
        callable): g = sqrt(p) f at the 1 + 2D stencil points, lap_g = (sum g(x +- eps e_i) - 2D g(x)) / eps^2, divided
        by clamp(sqrt(p(x)), 1e-5) (diff_ops.py:9-52), -(-c lap + V fs) (schrodinger/__init__.py:16-22), scale / shift
        (examples/__init__.py:7-9). `model(z)` is the HIP model (nsvd_model_forward). Like the reference's float32 run
        the point-wise stencil carries percent-level rounding noise in Tf - the fused kernel's even / odd
        form does not exist outside it. The shifted evaluations are not recorded for autograd: the EVD loss gives Tf no
        gradient (methods/nestedlora.py:108-111)."""
        ham = self.operator
        eps = float(ham.laplacian_eps)
        if eps <= 0:
            raise NotImplementedError("exact Laplacian with a non-Gaussian importance: not built (use the Gaussian "
                                      "sampler, or laplacian_eps > 0)")
        x = x.reshape(x.shape[0], -1).float()
        D = x.shape[1]

        def g(z):
            return importance(z).sqrt() * model(z)
        gs = g(x)
        lap = -2.0 * D * gs.detach()
        with torch.no_grad():
            for i in range(D):
                e = torch.zeros((1, D), device=x.device)
                e[0, i] = eps
                lap = lap + g(x + e) + g(x - e)
            lap = lap / eps ** 2
            sw = torch.clamp(importance(x).sqrt(), min=1e-5)
            lap = lap / sw
        fs = gs / sw
        with torch.no_grad():
            V = ham.local_potential_ftn(x.reshape(x.shape[0], ham.n_particles, -1)).view(-1, 1)
            Tf = -(-ham.scale_kinetic * lap + V * fs)
            Tf = self.scale * Tf + self.shift * fs
        return Tf, fs


class GaussianImportance:
    """p(x) of the isotropic Gaussian sampler N(0, sigma^2 I) (reference main_pde.py:94-100)."""

    def __init__(self, sigma: float, dim: int):
        self.sigma, self.dim = float(sigma), int(dim)

    def __call__(self, x):
        x = x.reshape(x.shape[0], -1)
        logp = (-0.5 * (x / self.sigma).pow(2).sum(-1) - self.dim * math.log(self.sigma)
                - 0.5 * self.dim * math.log(2 * math.pi))
        return logp.exp().view(-1, 1)


class LaplaceImportance:
    """p(x) = prod_i exp(-|x_i| / b) / (2 b) of the Laplace sampler (main_pde.py:101-112)."""

    def __init__(self, scale: float, dim: int):
        self.scale, self.dim = float(scale), int(dim)

    def __call__(self, x):
        x = x.reshape(x.shape[0], -1)
        logp = (-x.abs() / self.scale - math.log(2 * self.scale)).sum(-1)
        return logp.exp().view(-1, 1)


class UniformImportance:
    """p(x) = 1 / (2 s)^ndim of the uniform sampler on [-s, s]^D (main_pde.py:113-118; the reference's exponent is
    args.ndim, kept)."""

    def __init__(self, scale: float, ndim: int):
        self.scale, self.ndim = float(scale), int(ndim)

    def __call__(self, x):
        return torch.full((x.shape[0], 1), 1.0 / (2 * self.scale) ** self.ndim, device=x.device).float()


class UniformBoxImportance:
    """p(x) = 1 / (2 lim)^D on the validation box (reference main_pde.py:129-130)."""

    def __init__(self, lim: float, dim: int):
        self.lim, self.dim = float(lim), int(dim)

    def __call__(self, x):
        return torch.full((x.shape[0], 1), 1.0 / (2 * self.lim) ** self.dim, device=x.device).float()


def fused_problem_of(operator, importance, model) -> H.Problem:
    """Translate (OperatorWrapper, importance, WaveFunctions) into the nsvd_problem the kernels take."""
    if not isinstance(operator, OperatorWrapper):
        raise NsvdError("fused operator kernel: operator must be neural_svd_amd.operators.OperatorWrapper (other "
                        "callables go through NestedLoRA.apply_operator's call-through, not this translation)")
    if importance is not None and not isinstance(importance, GaussianImportance):
        raise NsvdError("fused operator kernel: importance must be None or GaussianImportance (other densities go "
                        "through OperatorWrapper.apply_stencil)")
    ham = operator.operator
    return H.make_problem(ham.potential_kind, ham.potential_param, ham.laplacian_eps, operator.scale, operator.shift,
                          importance.sigma if importance is not None else 1.0, ham.scale_kinetic,
                          float(model.hard_mul_const), importance is not None)


# ----------------------------------------------------------------------------------- ground truths
class Hydrogen2D:
    def __init__(self, charge=1.0):
        self.charge = charge

    def get_eigvals(self, neigs):
        """E_n = -Z^2 / (4 (n + 1/2)^2), degeneracy 2n + 1, first ``neigs`` states."""
        shells, n = [], 0
        while len(shells) < neigs:
            shells += [n] * (2 * n + 1)
            n += 1
        q = np.array(shells[:neigs], dtype=np.float64)
        return -self.charge ** 2 / (4 * (q + 0.5) ** 2)


class HarmonicOscillator:
    def __init__(self, k=1.0, ndim=2):
        assert ndim == 2, f"dim={ndim} not implemented"
        self.k, self.ndim = k, ndim

    def get_eigvals(self, neigs):
        """sqrt(k) (2n + D) with degeneracy n + 1.  Like the reference, whole shells are emitted up
        to one shell PAST the one that reaches ``neigs`` (never truncated): compare with ``[:neigs]``."""
        nend, states = 0, 0
        while True:
            states += nend + 1
            nend += 1
            if states >= neigs:
                break
        vals = [2 * n + self.ndim for n in range(nend + 1) for _ in range(n + 1)]
        return math.sqrt(self.k) * np.array(vals, dtype=np.float64)


def get_problem(args, device=None):
    if args.problem != "sch":
        raise NotImplementedError("only the Schroedinger problems are on the HIP path")
    args.n_particles = 1
    if args.potential_type == "hydrogen":
        pot = partial(hydrogen_potential, charge=args.charge)
        if args.ndim != 2:
            raise NotImplementedError("hydrogen: ndim 2 only")
        gt = -Hydrogen2D(charge=args.charge).get_eigvals(args.neigs)
    elif args.potential_type == "harmonic_oscillator":
        pot = partial(harmonic_oscillator_potential, k=1.0)
        gt = -HarmonicOscillator(k=1.0, ndim=args.ndim).get_eigvals(args.neigs)
    else:
        raise NotImplementedError(f"potential_type {args.potential_type}: not in scope of the HIP path")
    ham = NegativeHamiltonian(local_potential_ftn=pot, scale_kinetic=1.0, laplacian_eps=args.laplacian_eps,
                              n_particles=1)
    op = OperatorWrapper(ham, scale=args.operator_scale, shift=args.operator_shift)
    return op, args.operator_scale * gt + args.operator_shift


def get_dataloader(args, device):
    """-> make_batch_ftn_train, val_data, batch_ftn_val, importance_train, importance_val.
    The sampler draws on the DEVICE (the reference draws on the host and copies, main_pde.py:92-93)."""
    d = args.n_particles * args.ndim
    shape = (args.batch_size, args.n_particles, args.ndim)
    if args.sampling_mode == "gaussian":
        def make_batch_ftn_train():
            return args.sampling_scale * torch.randn(shape, device=device)

        importance_train = GaussianImportance(args.sampling_scale, d)
    elif args.sampling_mode == "laplacian":  # main_pde.py:101-112 (steps go through OperatorWrapper.apply_stencil)
        lap = torch.distributions.Laplace(torch.zeros(shape, device=device),
                                          args.sampling_scale * torch.ones(shape, device=device))

        def make_batch_ftn_train():
            return lap.sample()

        importance_train = LaplaceImportance(args.sampling_scale, d)
    elif args.sampling_mode == "uniform":  # main_pde.py:113-118
        def make_batch_ftn_train():
            return args.sampling_scale * (2 * torch.rand(shape, device=device) - 1)

        importance_train = UniformImportance(args.sampling_scale, args.ndim)
    else:
        raise NotImplementedError(f"--sampling_mode {args.sampling_mode}")
    if args.ndim in (1, 2) and args.n_particles == 1:
        ax = np.arange(-args.lim, args.lim, args.val_eps)
        xxs = np.meshgrid(*(args.ndim * [ax]))
        val_data = torch.tensor(np.array(list(zip(*[xx.flatten() for xx in xxs])))).to(device).float()

        def batch_ftn_val():
            n = int(np.ceil(len(val_data) / float(args.batch_size)))
            for i in range(n):
                yield val_data[i * args.batch_size:min((i + 1) * args.batch_size, len(val_data))], 0.0

        importance_val = UniformBoxImportance(args.lim, args.ndim)
    else:
        val_data, batch_ftn_val, importance_val = None, None, None
    return make_batch_ftn_train, val_data, batch_ftn_val, importance_train, importance_val
