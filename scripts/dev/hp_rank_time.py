"""dev: per-rank compute of the head-parallel step at world N, emulated on one GPU (L/N heads on 512 N rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
for N in (1, 2, 4, 8):
    shape = H.ModelShape(L=16 // N, D=2, m=1024, hidden=(128, 128, 128))
    tr = FusedTrainer(shape, prob, 512 * N, sequential=False, device=dev)
    for _ in range(300): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 1500
    for _ in range(n): tr.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
    print(f"N={N}: L_local={16 // N} B_global={512 * N}: {dt:.1f} us/step (compute only, no all-gather)")
    del tr
