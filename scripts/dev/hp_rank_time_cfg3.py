"""dev: per-rank compute of the head-parallel step of configs[2] at world N, emulated on one GPU (32 / N heads on 512 N rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
prob = H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0)
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for N in ((only,) if only else (1, 2, 4, 8)):
    shape = H.ModelShape(L=32 // N, D=2, m=256, hidden=(128, 128, 128), has_exp_mask=True)
    tr = FusedTrainer(shape, prob, 512 * N, sequential=True, device=dev, sampling_scale=4.0, fourier_scale=1.0,
                      exp_mask_init=10.0)
    for _ in range(300): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 1500
    for _ in range(n): tr.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
    print(f"N={N}: L_local={32 // N} B_global={512 * N}: {dt:.1f} us/step (compute only, no all-gather)")
    del tr
