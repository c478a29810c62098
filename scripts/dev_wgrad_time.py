import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
tr = FusedTrainer(shape, prob, 512, sequential=False, device=dev)
for _ in range(5): tr.step()
x = tr.sample()
H.operator_forward(tr.shape, tr._params, tr.problem, x, tr.ws, True, tr.path, out=(tr.f, tr.Tf))
H.evd_partial(tr.f, tr.Tf, tr.mask_kind, None, tr.scratch)
def bwd():
    H.operator_backward_evd(tr.shape, tr._params, tr.problem, x, tr.f, tr.Tf, tr.mask_kind, None, None, tr._moments, False, tr.scratch, tr._loss, tr._grads, tr.ws, 1.0, tr.path)
for _ in range(3): bwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    a.record(); bwd(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(f"NCH={os.environ.get('NSVD_WGRAD_NCH','full')} ONLY={os.environ.get('NSVD_WGRAD_ONLY','all')}: chain+wgrad median {ts[10]:.1f} us")
