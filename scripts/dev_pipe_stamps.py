"""dev: phase timeline of pmlp_wgrad_pipe_kernel from the NSVD_WG_STAMPS diagnostic build
   bash scripts/dev_build_wgst.sh && NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_wgst.so python scripts/dev_pipe_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H, _lib
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
tr = FusedTrainer(shape, prob, int(os.environ.get("B", "512")), sequential=False, device=dev,
                  fused_step=os.environ.get("FUSED", "1") == "1")
for _ in range(300): tr.step()
torch.cuda.synchronize()
lib = _lib.load()
G = 256
n = G * 32
buf = (ctypes.c_ulonglong * n)()
lib.nsvd_debug_pipe_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nsvd_debug_pipe_stamps(buf, n) == 0
st = np.array(buf, dtype=np.uint64).reshape(G, 32).astype(np.int64)
t0 = st[:, 0]
def col(i): return (st[:, i] - t0)
def show(name, v): print(f"{name:<44} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}")
print("cycles since the workgroup's start (MFMA side: thread 0; epilogue side: thread 256)")
for k in range(3):
    show(f"item {k}: K loop done", col(1 + 3 * k))
    show(f"item {k}: accumulators in LDS", col(2 + 3 * k))
    show(f"item {k}: hand-off barrier passed", col(3 + 3 * k))
for k in range(1, 3):
    show(f"epilogue side: slots of item {k} done", col(16 + 2 * k))
    show(f"epilogue side: drain before hand-off {k} done", col(17 + 2 * k))
show("epilogue side: left the item loop", col(26))
show("epilogue side: last item's epilogue done", col(27))
show("epilogue side: last layer done (end)", col(28))
w0 = st[:, 31].min()
print(f"wall: workgroup starts spread {(st[:, 31].max() - w0) / 100.0:.2f} us, last end {(st[:, 30].max() - w0) / 100.0:.2f} us "
      f"(100 MHz wall clock)")
for a, b, nm in ((0, 1, "K loop item 0"), (3, 4, "K loop item 1"), (6, 7, "K loop item 2")):
    d = st[:, b] - st[:, a]
    print(f"{nm}: {d.mean():.0f} cycles (min {d.min()}, max {d.max()})")

buf2 = (ctypes.c_ulonglong * (G * 64))()
lib.nsvd_debug_pipe_chunk_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.nsvd_debug_pipe_chunk_stamps(buf2, G * 64) == 0
cs = np.array(buf2, dtype=np.uint64).reshape(G, 4, 16).astype(np.int64)
nchs = (int(os.environ.get("B", "512")) // 64, int(os.environ.get("B", "512")) // 64, int(os.environ.get("B", "512")) // 128)
for k in range(3):
    n = min(nchs[k], 15)
    start = st[:, 0] if k == 0 else st[:, 3 * k]
    print(f"item {k}: start -> barrier 0: {(cs[:, k, 0] - start).mean():.0f}; chunk times (barrier to barrier): "
          + " ".join(f"{(cs[:, k, c + 1] - cs[:, k, c]).mean():.0f}" for c in range(n))
          + f"; last barrier -> K loop done: {(st[:, 1 + 3 * k] - cs[:, k, n]).mean():.0f}")

hs = cs[:, 3, :].reshape(G, 4, 4)
for i, gg in enumerate(range(2, 6)):
    t = hs[:, i, :]
    print(f"staging step {gg}: barrier -> stores issued {(t[:, 1] - t[:, 0]).mean():.0f}, -> loads issued "
          f"{(t[:, 2] - t[:, 1]).mean():.0f}, -> slot done {(t[:, 3] - t[:, 2]).mean():.0f}"
          + (f", -> next barrier passed {(hs[:, i + 1, 0] - t[:, 3]).mean():.0f}" if i < 3 else ""))
