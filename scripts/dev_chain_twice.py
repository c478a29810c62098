"""dev: does the chain kernel run faster the second time a CU executes it (hot instruction cache / L2)?
NSVD_CHAIN_TWICE diagnostic build: every (head, sample block) is executed by two workgroups, block b and b + 256."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H, _lib
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
tr = FusedTrainer(shape, prob, 512, sequential=False, device=dev)
for _ in range(100): tr.step()
torch.cuda.synchronize()
lib = _lib.load()
G = 512
buf = (ctypes.c_ulonglong * (G * 16))()
lib.nsvd_debug_chain_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nsvd_debug_chain_stamps(buf, G * 16) == 0
st = np.array(buf, dtype=np.uint64).reshape(G, 16).astype(np.int64)
for lo, hi, nm in ((0, 256, "first 256 workgroups"), (256, 512, "second 256 workgroups (same CUs, second pass)")):
    v = st[lo:hi]
    print(f"{nm}: total {(v[:, 7] - v[:, 0]).mean():.0f}, prefetch issue {(v[:, 8] - v[:, 0]).mean():.0f}, "
          f"tile in LDS {(v[:, 9] - v[:, 0]).mean():.0f}, partial sums {(v[:, 10] - v[:, 9]).mean():.0f}, "
          f"d loss/d f at {(v[:, 1] - v[:, 0]).mean():.0f}, layer 1 {(v[:, 3] - v[:, 2]).mean():.0f}")
