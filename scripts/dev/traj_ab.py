"""dev: loss trajectory of small generic-path configurations (the soak matrix's ragged cases) - run once per tree
(this one, a worktree of an older commit) and compare: the first steps agree to float32 rounding, later ones within the
training's own sensitivity.   traj_ab.py  (from the root of the tree to test)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
cases = [("B=100 hidden 128x2 hydrogen", dict(L=4, m=128, hidden=(128, 128), B=100, seq=False)),
         ("B=101 hidden 64x3 hydrogen", dict(L=3, m=33, hidden=(64,) * 3, B=101, seq=False)),
         ("B=512 hidden 64x3 hydrogen", dict(L=16, m=1024, hidden=(64,) * 3, B=512, seq=False))]
marks = [1, 2, 3, 5, 10, 30, 100, 300, 1000, 3000, 10000]
for name, c in cases:
    shape = H.ModelShape(L=c["L"], D=2, m=c["m"], hidden=c["hidden"], has_exp_mask=False)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    tr = FusedTrainer(shape, prob, c["B"], sequential=c["seq"], lr=1e-4, num_iters=100000, seed=0, device=dev,
                      sampling_scale=16.0, fourier_scale=0.1)
    out = []
    for i in range(1, marks[-1] + 1):
        tr.step()
        if i in marks:
            out.append(f"{i}:{float(tr.loss[0]):.4f}")
    print(name, " ".join(out), flush=True)
