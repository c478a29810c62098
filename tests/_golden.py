"""Helpers shared by the tests: load golden fixtures into oracle ``Params`` / ``Problem``."""
import ast
import os

import numpy as np
import torch

from oracle import nsvd_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def cfg_of(z, case):
    return ast.literal_eval(str(z[f"{case}_cfg"]))


def problem_of(cfg) -> O.Problem:
    pot = O.POT_HYDROGEN if cfg["potential_type"] == "hydrogen" else O.POT_HARMONIC
    return O.Problem(potential=pot, charge_or_k=cfg["charge"] if pot == O.POT_HYDROGEN else 1.0,
                     scale_kinetic=1.0, eps=cfg["laplacian_eps"], op_scale=cfg["operator_scale"],
                     op_shift=cfg["operator_shift"], sigma=cfg["sampling_scale"],
                     hard_mul_const=cfg["hard_mul_const"], use_importance=True)


def hidden_of(cfg):
    return [int(s) for s in cfg["mlp_hidden_dims"].split(",")]


def params_from_golden(z, case, prefix="param0_") -> O.Params:
    names = [str(n) for n in z[f"{case}_param_names"]]
    nl = sum(1 for n in names if ".ws." in n)
    g = lambda n: torch.tensor(z[f"{case}_{prefix}{n}"])  # noqa: E731
    ws = [g(f"model.base.ws.{i}") for i in range(nl)]
    bs = [g(f"model.base.bs.{i}") for i in range(nl)]
    fB = g("model.base.feature_map._B")
    sc = g("model.boundary_mask.scales") if "model.boundary_mask.scales" in names else None
    return O.Params(ws, bs, fB, sc)


def params_from_seed(cfg) -> O.Params:
    return O.init_params(cfg["neigs"], cfg["ndim"], cfg["fourier_mapping_size"], hidden_of(cfg),
                         cfg["fourier_scale"],
                         exp_mask_init=cfg["exp_mask_init_scale"] if cfg["apply_exp_mask"] else None,
                         seed=cfg["seed"])


def masks_of(z, case):
    return torch.tensor(z[f"{case}_v"]), torch.tensor(z[f"{case}_M"])


def trainable_names(z, case):
    names = [str(n) for n in z[f"{case}_param_names"]]
    return [n for n in names if not n.endswith("_B")]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
