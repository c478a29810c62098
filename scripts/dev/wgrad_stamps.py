"""dev: timeline of the wgrad kernel's blocks from the NSVD_WG_STAMPS diagnostic build."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H, _lib
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
tr = FusedTrainer(shape, prob, 512, sequential=False, device=dev, fused_step=os.environ.get("FUSED", "1") == "1")
for _ in range(200): tr.step()
torch.cuda.synchronize()
lib = _lib.load()
n = 448 * 8
buf = (ctypes.c_ulonglong * n)()
lib.nsvd_debug_wgrad_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nsvd_debug_wgrad_stamps(buf, n) == 0
st = np.array(buf, dtype=np.uint64).reshape(448, 8).astype(np.int64)
t0 = st[:, 1].min()
for kind, name in ((1, "A"), (2, "B"), (3, "C")):
    m = st[:, 0] == kind
    if not m.any():
        continue
    s, e = (st[m, 1] - t0) / 100.0, (st[m, 6] - t0) / 100.0
    print(f"{name}: n={m.sum()} start {s.min():.1f}..{s.max():.1f} us, end {e.min():.1f}..{e.max():.1f} us, dur mean {np.mean(e - s):.1f} max {np.max(e - s):.1f}")
    if kind == 1:
        pro, loop, epi = 0 * st[m, 2], st[m, 4] - st[m, 2], st[m, 5] - st[m, 4]
        print(f"   cycles: prologue {pro.mean():.0f}, loop {loop.mean():.0f} (min {loop.min()}, max {loop.max()}), epilogue {epi.mean():.0f}; ideal loop 65536")
print(f"kernel span {(st[:, 6].max() - t0) / 100.0:.1f} us")
