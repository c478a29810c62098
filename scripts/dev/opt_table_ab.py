"""dev: the generic path's fused step with the one-launch table optimiser against one launch per tensor
(NSVD_OPT_TABLE=0), same seed: parameters, square averages and EMA after N steps must be bit-identical.
   opt_table_ab.py <out.pt> [hidden width] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from neural_svd_amd import hip_ops as H
out = sys.argv[1]
hid = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
cfg = dict(bench.ALT["cfg2"], hidden=(hid, hid, hid))
tr, shape, prob = bench.make_trainer(cfg, "dp", None, dev, H.PATH_AUTO)
for _ in range(steps):
    tr.step()
torch.cuda.synchronize()
torch.save(dict(flat=tr.P.flat.cpu(), sq=tr.P.sq.cpu(), ema=tr.P.ema.cpu(), loss=tr.loss.cpu()), out)
print("loss", tr.loss.cpu().tolist(), "finite", bool(torch.isfinite(tr.P.flat).all()))
