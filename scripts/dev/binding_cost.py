"""dev: host cost per C-ABI call of the two bindings (ctypes: neural_svd_amd/_lib.py; torch extension:
csrc/torch_binding.cpp) - a tiny model, so that the GPU is never the bottleneck and the wall time per call is the
host's: operator_forward, evd_moments, and a whole FusedTrainer.step()."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = "cuda:0"
shape = H.ModelShape(L=2, D=2, m=64, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
rec = {}
for binding in ("ctypes", "torch"):
    os.environ["NSVD_BINDING"] = binding
    tr = FusedTrainer(shape, prob, 32, sequential=False, seed=0, device=dev)
    x = 16 * torch.randn(32, 2, device=dev)
    ws = H.new_workspace(shape, 32, dev)
    f, Tf = H.operator_forward(shape, tr._params, prob, x, ws, False)
    mom = torch.empty(9, device=dev)
    scr = H.evd_scratch(32, 2, dev)
    out = {}
    for name, fn in (("operator_forward", lambda: H.operator_forward(shape, tr._params, prob, x, ws, False, out=(f, Tf))),
                     ("evd_moments", lambda: H.evd_moments(f, Tf, H.MASK_JOINT, None, mom, scr)),
                     ("trainer_step", tr.step)):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3000
        for _ in range(n):
            fn()
        t_issue = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        out[name] = round(t_issue, 2)
    rec[binding] = out
    print(binding, out)
print("RECORD " + json.dumps({"host_us_per_call": rec, "note": "wall time per call with the queue never drained (tiny model: L=2, m=64, B=32)"}))
