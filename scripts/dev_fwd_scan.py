import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import reference_init
dev = "cuda:0"
for m, hidden in ((64, (128,128,128)), (512, (128,128,128)), (1024, (128,128,128)), (1024, (128,)), (2048, (128,128,128))):
    shape = H.ModelShape(L=16, D=2, m=m, hidden=hidden)
    fB, ws, bs, sc = reference_init(shape, 0.1, None, 0)
    ws = [w.to(dev) for w in ws]; bs = [b.to(dev) for b in bs]; fB = fB.to(dev)
    p = H.pack_params(shape, ws, bs, fB, None)
    prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
    x = (16 * torch.randn(512, 2)).to(dev)
    wsb = H.new_workspace(shape, 512, dev)
    for _ in range(5):
        H.operator_forward(shape, p, prob, x, wsb, True, H.PATH_FUSED)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in evs:
        H.profile_next_forward(a, b)
        H.operator_forward(shape, p, prob, x, wsb, True, H.PATH_FUSED)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"m={m} hidden={hidden}: fwd kernel median {ts[15]*1e3:.1f} us min {ts[0]*1e3:.1f}")
