"""CPU BASELINE PORT of the reference training step.  *** TEST INFRASTRUCTURE, NOT PRODUCT ***

Same rule as nsvd_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline``
leg may import this file.

Where nsvd_oracle.py restates the MATH (hand-derived backward, float64 truth), this file restates
the reference's OP SEQUENCE in eager PyTorch so that timing it on the GPU box's host cores stands in
for "the reference's CPU path" (the reference's Python cannot travel to that box):
  * 1 + 2D separate model evaluations (examples/operator/pde/diff_ops.py:36-45), each doing the
    Fourier map (examples/utils.py:139-140) and the einsum layers INCLUDING the elementwise
    ``W / norm`` division by the constant 1.0 that the reference performs on every weight at every
    evaluation (examples/models/mlp.py:209,216) - it dominates the reference's CPU time;
  * importance re-weighting through torch.distributions.MultivariateNormal (main_pde.py:94-100);
  * the custom autograd.Function loss with its hand-written backward (methods/nestedlora.py:67-111);
  * eager autograd through the centre evaluation, torch.optim.RMSprop, CosineAnnealingLR, EMA
    (examples/operator/__init__.py:55-74).
tests/test_oracle_golden.py pins its outputs to the reference's golden vectors; DESIGN.md records
its step time next to the imported reference's, measured in the build container.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.distributions import MultivariateNormal

from . import nsvd_oracle as O


class PortModel(nn.Module):
    def __init__(self, p: O.Params, hard_mul_const=1.0):
        super().__init__()
        self.fB = nn.Parameter(p.fourier_B.clone(), requires_grad=False)
        self.ws = nn.ParameterList([nn.Parameter(w.clone()) for w in p.ws])
        self.bs = nn.ParameterList([nn.Parameter(b.clone()) for b in p.bs])
        self.scales = None if p.scales is None else nn.Parameter(p.scales.clone())
        self.c = hard_mul_const
        self.act = nn.Softplus()

    @staticmethod
    def norm(w):
        return 1.0  # weight_normalization=False

    def forward(self, x):
        proj = x @ self.fB
        h = torch.cat([torch.sin(proj), torch.cos(proj)], dim=1)
        h = torch.einsum("lhd,bd->lhb", self.ws[0] / self.norm(self.ws[0]), h) + self.bs[0]
        h = self.act(h)
        for i in range(1, len(self.ws)):
            h = torch.einsum("lhp,lpb->lhb", self.ws[i] / self.norm(self.ws[0]), h) + self.bs[i]
            if i < len(self.ws) - 1:
                h = self.act(h)
        out = self.c * h.permute(2, 0, 1).squeeze()
        if self.scales is not None:
            r = torch.norm(x, p=2, dim=-1).view(-1, 1)
            out = out * torch.exp(-r / self.scales.view(1, -1))
        return out


class _EVDLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, Tf, f1, f2, v, M):
        lam1 = torch.einsum("bl,bm->lm", f1, f1) / f1.shape[0]
        lam2 = torch.einsum("bl,bm->lm", f2, f2) / f2.shape[0]
        ctx.v, ctx.M = v, M
        ctx.save_for_backward(f, Tf, f1, f2, lam1, lam2)
        return -2 * torch.einsum("l,bl,bl->b", v, f, Tf).mean() + (M * lam1 * lam2).sum()

    @staticmethod
    def backward(ctx, go):
        f, Tf, f1, f2, lam1, lam2 = ctx.saved_tensors
        g = -(4 / f.shape[0]) * torch.einsum("l,bl->bl", ctx.v, Tf)
        g1 = (2 / f1.shape[0]) * torch.einsum("lm,lm,bl->bm", ctx.M, lam2, f1)
        g2 = (2 / f2.shape[0]) * torch.einsum("lm,lm,bl->bm", ctx.M, lam1, f2)
        return go * g, None, go * g1, go * g2, None, None


class PortStep:
    """One object = the reference's training loop state for the PDE path."""

    def __init__(self, p: O.Params, prob: O.Problem, v, M, lr=1e-4, alpha=0.999, ema_decay=0.995, num_iters=500000,
                 dtype=torch.float32):
        self.model = PortModel(p, prob.hard_mul_const).to(dtype)
        self.prob, self.v, self.M = prob, v.to(dtype), M.to(dtype)
        D = p.fourier_B.shape[0]
        self.mvn = MultivariateNormal(loc=torch.zeros(D, dtype=dtype),
                                      covariance_matrix=prob.sigma ** 2 * torch.eye(D, dtype=dtype))
        self.opt = torch.optim.RMSprop(self.model.parameters(), lr=lr, alpha=alpha, eps=1e-10, weight_decay=0,
                                       momentum=0.0)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, num_iters)
        self.ema_decay, self.num_updates = ema_decay, 0
        self.shadow = [q.detach().clone() for q in self.model.parameters() if q.requires_grad]

    def importance(self, x):
        return self.mvn.log_prob(x).exp().view(-1, 1)

    def operator(self, x):
        prob = self.prob
        g = (lambda y: self.importance(y).sqrt() * self.model(y)) if prob.use_importance else self.model
        D = x.shape[1]
        fs = g(x)
        lap = -2 * D * fs
        for i in range(D):
            e = torch.zeros((1, D))
            e[0, i] = prob.eps
            lap = lap + (g(x + e) + g(x - e))
        lap = lap / (prob.eps ** 2)
        if prob.use_importance:
            sw = torch.clamp(self.importance(x).sqrt(), min=1e-5)
            lap, fs = lap / sw, fs / sw
        r = x.norm(dim=1, p=2)
        V = (-(prob.charge_or_k / r) if prob.potential == O.POT_HYDROGEN else prob.charge_or_k * r ** 2).reshape(-1, 1)
        H = -prob.scale_kinetic * lap + V * fs
        return prob.op_scale * (-H) + prob.op_shift * fs, fs

    def loss(self, x):
        Tf, f = self.operator(x)
        f1, f2 = torch.chunk(f, 2)
        return _EVDLoss.apply(f, Tf, f1, f2, self.v, self.M), f, Tf

    def step(self, x):
        self.opt.zero_grad()
        loss, _, _ = self.loss(x)
        loss.backward()
        self.opt.step()
        self.sched.step()
        self.num_updates += 1
        d = min(self.ema_decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, q in zip(self.shadow, [q for q in self.model.parameters() if q.requires_grad]):
                s.sub_((1.0 - d) * (s - q))
        return loss
