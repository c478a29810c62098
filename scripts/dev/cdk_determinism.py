"""dev: is the mixed-precision CDK step reproducible run to run? The same three steps from the same weights, repeated N
times in one process and compared bit for bit (a race in a kernel shows up as a run that differs)."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from neural_svd_amd.cdk import FusedCdkStep, HeteroNetwork, NestedLoRAForCDK, get_mlp  # noqa: E402

dev = "cuda:0"
sizes, B = ([int(v) for v in sys.argv[2].split(",")], int(sys.argv[3])) if len(sys.argv) > 3 else ([128, 256, 256], 256)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(77)
xs = torch.randn(3, B, sizes[0], generator=g).to(dev)
ys = torch.randn(3, B, sizes[0], generator=g).to(dev)
torch.manual_seed(11)
model0 = HeteroNetwork([get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True),
                        get_mlp(sizes, bias=True, nonlinearity="lrelu0.2", use_bn=True)],
                       [nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(dev).train()
sd0 = copy.deepcopy(model0.state_dict())
for dtype in ("bfloat16", "float16"):
    ref, bad = None, 0
    for rep in range(N):
        model0.load_state_dict(sd0)
        method = NestedLoRAForCDK(model0, neigs=sizes[-1], step=1, sequential=False, set_first_mode_const=True).to(dev)
        fs = FusedCdkStep(method, lr=5e-3, momentum=0.9, max_grad_norm=1.0, t_max=0, batch_size=B, use_amp=True,
                          amp_dtype=dtype, grad_scaler=False)
        outs = [fs.step(xs[t], ys[t]).clone() for t in range(3)]
        torch.cuda.synchronize()
        cur = [o.cpu() for o in outs] + [v.detach().cpu().clone() for k, v in model0.state_dict().items() if "num_batches" not in k]
        if ref is None:
            ref = cur
        else:
            diff = [i for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
            if diff:
                bad += 1
                i = diff[0]
                print(f"{dtype} rep {rep}: differs in items {diff[:6]}; first: max abs diff {float((cur[i] - ref[i]).abs().max()):.3e}")
    print(f"{dtype}: {bad} of {N - 1} repetitions differ from the first")
