"""dev: 10 000 optimiser steps on a matrix of configurations; reports finiteness, loss trend and speed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
N = int(os.environ.get("STEPS", "10000"))
cases = [
    ("cfg1 hydrogen B=128 seq", dict(L=16, m=1024, hidden=(128,) * 3, B=128, seq=True, pot="h", eps=0.01)),
    ("cfg3/gpu oscillator L=32 B=512 seq expmask", dict(L=32, m=256, hidden=(128,) * 3, B=512, seq=True, pot="o", eps=0.01)),
    ("oscillator exact L=32 B=512", dict(L=32, m=256, hidden=(128,) * 3, B=512, seq=True, pot="o", eps=0.0)),
    ("hydrogen exact bf16x3 L=16 B=512 jnt", dict(L=16, m=1024, hidden=(128,) * 3, B=512, seq=False, pot="h", eps=0.0, path=H.PATH_FUSED_BF16X3)),
    ("oscillator bf16x3 L=32 B=4096", dict(L=32, m=256, hidden=(128,) * 3, B=4096, seq=True, pot="o", eps=0.01, path=H.PATH_FUSED_BF16X3)),
    ("generic path: hidden 64x3, L=6, B=256 oscillator", dict(L=6, m=64, hidden=(64,) * 3, B=256, seq=True, pot="o", eps=0.01, lr=1e-3, fs=0.15)),
    ("generic path: ragged B=100, hidden 128x2 hydrogen", dict(L=4, m=128, hidden=(128,) * 2, B=100, seq=False, pot="h", eps=0.01)),
    ("generic path: hidden 256x3 hydrogen L=16 B=512", dict(L=16, m=1024, hidden=(256,) * 3, B=512, seq=False, pot="h", eps=0.01)),
    ("generic path: hidden (96, 96) oscillator L=32 B=512", dict(L=32, m=256, hidden=(96, 96), B=512, seq=True, pot="o", eps=0.01)),
    ("generic path: hidden (40, 24) B=100 (K tails, clamped edges)", dict(L=5, m=34, hidden=(40, 24), B=100, seq=True, pot="o", eps=0.01)),
    ("generic path: hidden 64x3 B=101 (scalar kernels)", dict(L=3, m=33, hidden=(64,) * 3, B=101, seq=False, pot="h", eps=0.01)),
    ("hydrogen L=16 B=512 jnt step=4", dict(L=16, m=1024, hidden=(128,) * 3, B=512, seq=False, pot="h", eps=0.01, step=4)),
    ("hydrogen L=1 B=512", dict(L=1, m=1024, hidden=(128,) * 3, B=512, seq=True, pot="h", eps=0.01)),
    ("hydrogen L=128 B=64 m=64", dict(L=128, m=64, hidden=(128,) * 3, B=64, seq=True, pot="h", eps=0.01)),
    ("3-D oscillator, split-stencil form, L=8 B=256", dict(L=8, m=256, hidden=(128,) * 3, B=256, seq=True, pot="o", eps=0.01, D=3)),
    ("3-D oscillator exact L=8 B=256", dict(L=8, m=256, hidden=(128,) * 3, B=256, seq=True, pot="o", eps=0.0, D=3)),
]
for name, c in cases:
    shape = H.ModelShape(L=c["L"], D=c.get("D", 2), m=c["m"], hidden=c["hidden"], has_exp_mask=c["pot"] == "o")
    if c["pot"] == "h":
        prob = H.make_problem(H.POT_HYDROGEN, 1.0, c["eps"], 100.0, 0.0, 16.0); kw = dict(sampling_scale=16.0, fourier_scale=c.get("fs", 0.1))
    else:
        prob = H.make_problem(H.POT_HARMONIC, 1.0, c["eps"], 1.0, 16.0, 4.0); kw = dict(sampling_scale=4.0, fourier_scale=c.get("fs", 1.0), exp_mask_init=10.0)
    try:
        tr = FusedTrainer(shape, prob, c["B"], sequential=c["seq"], step=c.get("step", 1), lr=c.get("lr", 1e-4), num_iters=100000, seed=0,
                          device=dev, path=c.get("path", H.PATH_AUTO), **kw)
        losses = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N):
            tr.step()
            if i % (N // 10) == N // 10 - 1:
                losses.append(float(tr.loss[0]))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        fin = all(bool(torch.isfinite(t).all()) for t in (tr.P.flat, tr.P.ema, tr.P.sq, tr.f, tr.Tf))
        print(f"{name:55s} path={H.path_name(shape, c['B'], c.get('path', H.PATH_AUTO), prob):10s} {N / dt:8.0f} steps/s finite={fin} "
              f"loss {losses[0]:.1f} -> {losses[-1]:.1f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name:55s} ERROR {type(e).__name__}: {e}", flush=True)
