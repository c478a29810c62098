"""dev: cycles per chunk and region of the streaming backward (pmlp_stream_bwd.h) at configs[3]'s size.
   NSVD_STREAM_DBG=4 python scripts/dev/stream_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import _lib
from neural_svd_amd.kernel_ops import FusedKernelTrainer, synthetic_psd_kernel
dev = "cuda:0"
op = synthetic_psd_kernel(10000, 256, 16, 0, dev)
fk = FusedKernelTrainer(op, L=64, m=64, hidden=(128, 128), batch_size=8192, sequential=False, lr=1e-4,
                        rmsprop_decay=0.99, rmsprop_eps=1e-8, fourier_scale=0.05, seed=0)
for _ in range(5):
    fk.step()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
lib.nsvd_debug_stream_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
print("rc", lib.nsvd_debug_stream_stamps(buf))
names = ["region1", "barrierA", "region2", "barrierB", "region3", "chunks"]
print({n: int(buf[i]) for i, n in enumerate(names)})
