"""diagnostic (GPU): per-phase cycle shares of the fused forward from s_memtime stamps.
   NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_stamps.so python scripts/dev/stamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import reference_init
dev = "cuda:0"
L, B, m, hidden, D = 16, 512, 1024, (128, 128, 128), 2
shape = H.ModelShape(L=L, D=D, m=m, hidden=hidden)
fB, ws, bs, sc = reference_init(shape, 0.1, None, 0)
ws = [w.to(dev) for w in ws]; bs = [b.to(dev) for b in bs]; fB = fB.to(dev)
p = H.pack_params(shape, ws, bs, fB, None)
prob = H.make_problem(H.POT_HYDROGEN, 1.0, 0.01, 100.0, 0.0, 16.0)
x = (16 * torch.randn(B, D)).to(dev)
wsb = H.new_workspace(shape, B, dev)
al = lambda n: (n * 4 + 255) // 256 * 256
E = 1 + 2 * D; R = E * B; F = 2 * m
off = al(F * B) + al(2 * D * m) + al(F * B) + len(hidden) * al(L * 128 * B) + 2 * al(B * L)  # FusedWs: dz[0]
for _ in range(3):
    H.operator_forward(shape, p, prob, x, wsb, True, int(os.environ.get("NSVD_DEV_PATH", H.PATH_FUSED)))
torch.cuda.synchronize()
nwg = (B // 32) * L
st = wsb[off:off + nwg * 16 * 8].view(torch.int64).view(nwg, 16).cpu().numpy().astype(np.float64)
names = {0: "start", 1: "prologue(chunk0 staged)", 2: "layer0 loop", 3: "softplus0", 4: "wf loads issued", 5: "exchange1 (2 barriers + Hs write)", 6: "layer1 MFMAs",
         7: "softplus1", 8: "wf loads issued", 9: "exchange2", 10: "layer2 MFMAs", 11: "softplus2", 12: "-", 13: "last layer + reduce", 14: "last layer + reduce + FD epilogue"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 14]  # (stamp 13 left the kernel with the point-wise epilogue)
tot = st[:, 14] - st[:, 0]
print(f"per-WG total ticks median {np.median(tot):.0f} (s_memtime ticks; 100 MHz => {np.median(tot)/100:.1f} us)")
prev = 0
for k in order[1:]:
    d = st[:, k] - st[:, prev]
    print(f"  {names[k]:40s} median {np.median(d):9.0f} ticks = {np.median(d)/100:7.2f} us   ({100*np.median(d)/np.median(tot):5.1f} %)")
    prev = k
print("start spread across WGs (us):", (st[:, 0].max() - st[:, 0].min()) / 100, " end spread:", (st[:, 14].max() - st[:, 14].min()) / 100)
