import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
for pipe in (False, True):
    tr = FusedTrainer(shape, prob, 512, sequential=False, device=dev, pipeline=pipe)
    for _ in range(20): tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): tr.step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"pipeline={pipe}: host enqueue {t_enq/200*1e6:.1f} us/step, total {t_all/200*1e6:.1f} us/step")
