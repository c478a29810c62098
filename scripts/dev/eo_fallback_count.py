"""dev: how often the even / odd softplus takes its large-perturbation path (a library built with -DNSVD_EO_COUNT:
NSVD_LIB_PATH=neural_svd_amd/_ab/libnsvd_hip_cnt.so) - per layer, counts of (lane, group of four registers)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H, _lib
from neural_svd_amd.trainer import FusedTrainer
lib = _lib.load()
fn = ctypes.CDLL(os.environ["NSVD_LIB_PATH"]).nsvd_debug_eo_count
buf = (ctypes.c_ulonglong * 8)()
def counts(reset=True):
    torch.cuda.synchronize(); fn(buf, int(reset)); return list(buf)[:4]
dev = torch.device("cuda:0")
for name, shape, prob, kw in (
    ("cfg2", H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128)), H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0), dict(sampling_scale=16.0, fourier_scale=0.1)),
    ("cfg3", H.ModelShape(L=32, D=2, m=256, hidden=(128, 128, 128), has_exp_mask=True), H.make_problem(H.POT_HARMONIC, 1.0, 0.01, 1.0, 16.0, 4.0), dict(sampling_scale=4.0, fourier_scale=1.0, exp_mask_init=10.0))):
    tr = FusedTrainer(shape, prob, 512, sequential=False, step=1, lr=1e-4, num_iters=20000, seed=0, device=dev, **kw)
    counts()
    total = 512 * shape.L * 128 / 4   # (sample, head, group of 4 rows) per layer
    for n in (1, 100, 2000, 20000):
        while tr.t < n: tr.step()
        counts()
        tr.step()
        c = counts()
        print(name, "after", n, "steps: fallback groups per layer", c, "of", int(total), "=", [round(x / total, 5) for x in c])
