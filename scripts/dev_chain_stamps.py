"""dev: phase timeline of pmlp_fused_bwd_chain_kernel from the NSVD_WG_STAMPS diagnostic build
   bash scripts/dev_build_wgst.sh && NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_wgst.so python scripts/dev_chain_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H, _lib
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
tr = FusedTrainer(shape, prob, int(os.environ.get("B", "512")), sequential=False, device=dev)
for _ in range(300): tr.step()
torch.cuda.synchronize()
lib = _lib.load()
G = int(os.environ.get("G", "256"))
buf = (ctypes.c_ulonglong * (G * 16))()
lib.nsvd_debug_chain_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nsvd_debug_chain_stamps(buf, G * 16) == 0
st = np.array(buf, dtype=np.uint64).reshape(G, 16).astype(np.int64)
names = ["start", "d loss / d f ready (moments done)", "dz of the last hidden layer ready", "dz of layer nh-2 ready",
         "dz of layer nh-3 ready", "-", "-", "end of the chain (before the last stores drain)"]
for i in (1, 2, 3, 4, 7):
    v = st[:, i] - st[:, 0]
    print(f"{names[i]:<52} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f} cycles")

for i, nm in ((8, "prefetches issued"), (9, "f tile in LDS (first barrier passed)"), (10, "moment partial sums done"), (11, "moment columns ready")):
    v = st[:, i] - st[:, 0]
    print(f"  {nm:<50} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f} cycles")

print("first chain layer, cycles since its start (stamp 2):")
for i, nm in ((12, "barrier A passed"), (13, "LDS writes issued"), (14, "barrier B passed"), (15, "MFMA loop issued"), (3, "next layer starts (sigmoid etc. done)")):
    v = st[:, i] - st[:, 2]
    print(f"  {nm:<50} mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f}")
