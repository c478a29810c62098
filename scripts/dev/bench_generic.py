"""dev: the two generic-path entries of bench.py's other_configs alone (hidden widths 256 and 64 at configs[1]'s sizes)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from neural_svd_amd import hip_ops as H
dev = torch.device("cuda:0")
for hidden in ((256, 256, 256), (64, 64, 64)):
    d = bench.measure_pde_config(dict(bench.ALT["cfg2"], hidden=hidden), dev, H.PATH_AUTO, 100, 10, 3, 0.3)
    print(hidden, json.dumps(d))
