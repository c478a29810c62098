import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import _golden as G
from tests import test_hip_parity as T
from neural_svd_amd import hip_ops
T.H = hip_ops
for fn, cases in (("model_small", ["hyd_small", "osc_small", "hyd_ragged"]), ("model_headline", ["hyd_med", "cfg1"])):
    z = G.load(fn)
    for case in cases:
        cfg = G.cfg_of(z, case); prob = G.problem_of(cfg)
        p = G.params_from_golden(z, case) if fn == "model_small" else G.params_from_seed(cfg)
        v, M = G.masks_of(z, case); x = torch.tensor(z[f"{case}_x"][0])
        for path in ("generic", "auto"):
            r = T.run_hip(p, prob, x, v, M, T._path(path))
            k = T.tf_noise_kappa(r["Tf"], z[f"{case}_f64_step0_Tf"], z[f"{case}_f64_step0_f"], cfg)
            kr = T.tf_noise_kappa(z[f"{case}_f32_step0_Tf"], z[f"{case}_f64_step0_Tf"], z[f"{case}_f64_step0_f"], cfg)
            e = T.rel(r["Tf"], z[f"{case}_f64_step0_Tf"]); er = T.rel(z[f"{case}_f32_step0_Tf"], z[f"{case}_f64_step0_Tf"])
            print(f"{case:10s} {path:8s} ({r['path']:10s}): kappa hip {k:6.2f} ref32 {kr:6.2f} | Tf rel-L2 hip {e:.2e} ref32 {er:.2e} | loss hip {float(r['loss'][0]):.4f} ref64 {float(z[f'{case}_f64_step0_loss']):.4f} ref32 {float(z[f'{case}_f32_step0_loss']):.4f}")
