"""dev: the wide layer of a mixed-precision tower with BatchNorm inside the contraction (csrc/tower_col.h) against the
contraction + strip pair it replaces (NSVD_TOWER16_FUSED=0): one tower's forward and backward at configs[4]'s size, event
timed, and the stamped phases of the whole-column kernel (cycles of block 0 / wave 0: prologue, K loop, epilogue)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from neural_svd_amd import _lib, hip_ops as H  # noqa: E402

dev = torch.device("cuda:0")
B, d0, d1, d2 = [int(v) for v in (sys.argv[1:5] or (1024, 512, 8192, 512))]
g = torch.Generator().manual_seed(0)
P = dict(W1=torch.randn(d1, d0, generator=g) / d0 ** 0.5, b1=0.1 * torch.randn(d1, generator=g),
         g1=1.0 + 0.3 * torch.randn(d1, generator=g), be1=0.2 * torch.randn(d1, generator=g),
         W2=torch.randn(d2, d1, generator=g) / d1 ** 0.5, b2=0.1 * torch.randn(d2, generator=g),
         g2=1.0 + 0.3 * torch.randn(d2, generator=g), be2=0.2 * torch.randn(d2, generator=g))
P = {k: v.to(dev).contiguous() for k, v in P.items()}
x, dz = torch.randn(B, d0, generator=g).to(dev), torch.randn(B, d2, generator=g).to(dev)
ws = H.tower_workspace(B, d0, d1, d2, dev)


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for fused in ("1", "0"):
    os.environ["NSVD_TOWER16_FUSED"] = fused
    for form in (("a", "b") if fused == "1" else ("-",)):
        if form != "-":
            os.environ["NSVD_TCOL_FORM"] = form  # (read once per process by the library: the first value wins)
        tf = timed(lambda: H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=True))
        tb = timed(lambda: H.tower_backward(x, P, dz, 0.2, ws, gemm_bf16=True))
        print(f"fused={fused} form={form}: one tower forward {tf:.1f} us, backward {tb:.1f} us "
              f"(fused runs: {H.tower_mixed_fused(B, d0, d1, d2, 0.2)})")
        break  # the form switch is read once: run the script again with NSVD_TCOL_FORM=b for the other
os.environ["NSVD_TOWER16_FUSED"] = "1"
lib = _lib.load()
st = (C.c_ulonglong * 4)()
lib.nsvd_debug_tcol_stamps.argtypes = [C.c_void_p]
lib.nsvd_debug_tcol_stamps(st)  # arm
H.tower_forward(x, P, 0.2, 1e-5, 0.1, False, ws, gemm_bf16=True)
torch.cuda.synchronize()
lib.nsvd_debug_tcol_stamps(st)
print("forward  whole-column kernel, cycles (100 MHz counter ticks x 24 at 2.4 GHz): prologue %d, K loop %d (%d stages), epilogue %d"
      % (st[0], st[1], st[3], st[2]))
H.tower_backward(x, P, dz, 0.2, ws, gemm_bf16=True)
torch.cuda.synchronize()
lib.nsvd_debug_tcol_stamps(st)
print("backward whole-column kernel: prologue %d, K loop %d (%d stages), epilogue %d" % (st[0], st[1], st[3], st[2]))
