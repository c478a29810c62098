"""dev: how close is ONE mixed-precision CDK step at configs[4]'s size to the float64 oracle with the same roundings?
(the numbers behind the tolerances of tests/test_cdk_step_gpu.py::test_cdk_step_at_headline_size_against_the_oracle)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import nsvd_oracle as O  # noqa: E402
from neural_svd_amd import hip_ops as H  # noqa: E402
from neural_svd_amd.cdk import FusedCdkStep  # noqa: E402
from tests.test_cdk_step_gpu import KEYS, _build  # noqa: E402

DEV = "cuda:0"
sizes, B, mu, lr, mom, max_norm, slope = [512, 8192, 512], 1024, 16.0, 5e-3, 0.9, 1.0, 0.2
g = torch.Generator().manual_seed(123)
x, y = torch.randn(B, sizes[0], generator=g), torch.randn(B, sizes[0], generator=g)
for dtype in ("bfloat16", "float16"):
    model, method = _build(sizes, mu, 5)
    sd0 = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
    fs = FusedCdkStep(method, lr=lr, momentum=mom, max_grad_norm=max_norm, t_max=0, batch_size=B, use_amp=True,
                      amp_dtype=dtype, grad_scaler=False)
    out = fs.step(x.to(DEV), y.to(DEV)).cpu().double().clone()
    sd = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    towers = [{k: sd0[f"backbones.{s}.{n}"].clone() for k, n in KEYS.items()} for s in "xy"]
    bufs = [{k: torch.zeros_like(v) for k, v in t.items()} for t in towers]
    running = [dict(rm1=sd0[f"backbones.{s}.1.running_mean"].clone(), rv1=sd0[f"backbones.{s}.1.running_var"].clone(),
                    rm2=sd0[f"backbones.{s}.4.running_mean"].clone(), rv2=sd0[f"backbones.{s}.4.running_var"].clone())
               for s in "xy"]
    v, M = method.vector_mask.double().cpu(), method.matrix_mask.double().cpu()
    omode = "fused" if H.tower_mixed_fused(B, *sizes, slope) else True
    (loss, lop, lmet), total = O.cdk_train_step(x.double(), y.double(), towers, bufs, running, v, M, mu, lr, mom,
                                                max_norm, slope, True, gemm_bf16=omode,
                                                half="f16" if dtype == "float16" else "bf16")
    print(dtype, "loss rel", abs(float(out[0]) - float(loss)) / abs(float(loss)), "norm rel",
          abs(float(out[3]) - float(total)) / float(total))
    worst = {}
    for si, s in enumerate("xy"):
        for k, n in KEYS.items():
            got, want, start = sd[f"backbones.{s}.{n}"], towers[si][k], sd0[f"backbones.{s}.{n}"]
            move = float((want - start).norm())
            worst[k] = max(worst.get(k, 0.0), float((got - want).norm()) / max(move, 1e-30))
    print("   update error / update length per tensor:", {k: round(v, 5) for k, v in worst.items()})
