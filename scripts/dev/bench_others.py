"""dev: the PDE entries of bench.py's other_configs alone (eager and graph replay)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from neural_svd_amd import hip_ops as H
dev = torch.device("cuda:0")
for name, cfg, st in (("cfg1", bench.ALT["cfg1"], 200), ("cfg3 B=512", bench.ALT["cfg3"], 200), ("cfg3 B=4096", dict(bench.ALT["cfg3"], B=4096), 60),
                      ("L=36", dict(bench.ALT["cfg2"], L=36), 200), ("L=55", dict(bench.ALT["cfg3"], L=55), 200)):
    d = bench.measure_pde_config(cfg, dev, H.PATH_AUTO, st, 20 if st == 200 else 10, 3, 0.3)
    print(name, d["value"], d["timing_mode"], d.get("modes"), d.get("graph_error"), d["roofline"]["step_frac"])
