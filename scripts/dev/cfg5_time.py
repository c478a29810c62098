"""dev: one training step of the Sketchy-style CDK configuration (cfg5: two towers 512 -> 8192 -> 512 with BatchNorm and
lrelu0.2, mu = 16 l2_ball, L = 512, batch 1024, SGD 5e-3 momentum 0.9) through this package's mirrors, float32 and
under torch.autocast(float16) (the script's AMP): where the time goes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from neural_svd_amd.cdk import HeteroNetwork, NestedLoRAForCDK, get_mlp
dev = "cuda:0"
torch.manual_seed(0)
sizes = [512, 8192, 512]
model = HeteroNetwork([get_mlp(sizes, nonlinearity="lrelu0.2"), get_mlp(sizes, nonlinearity="lrelu0.2")],
                      [nn.Identity(), nn.Identity()], mu=16.0, regularize_mode="l2_ball").to(dev)
method = NestedLoRAForCDK(model, neigs=512, step=1, sequential=False, set_first_mode_const=True).to(dev)
opt = torch.optim.SGD(method.parameters(), lr=5e-3, momentum=0.9)
x, y = torch.randn(1024, 512, device=dev), torch.randn(1024, 512, device=dev)
scaler = torch.amp.GradScaler("cuda")
def step(amp):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        _, fx, _, fy = method(x, y)
        out = method.compute_loss(fx, fy)
    loss = out[0]
    if amp:
        scaler.scale(loss).backward(); scaler.step(opt); scaler.update()
    else:
        loss.backward(); opt.step()
import torch.nn as _nn
from neural_svd_amd.cdk import TowerSequential
def tower_time(lib):
    """forward + backward of ONE tower alone: HIP tower kernels vs torch's modules on the same weights"""
    t = model.backbones["x"]
    g = torch.randn(1024, 512, device=dev)
    f = (lambda: _nn.Sequential.forward(t, x)) if lib else (lambda: t(x))
    for _ in range(10):
        opt.zero_grad(set_to_none=True); (f() * g).sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        opt.zero_grad(set_to_none=True); (f() * g).sum().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100
print(f"one tower forward + backward: HIP kernels {tower_time(False) * 1e6:.0f} us, torch modules (library GEMMs) {tower_time(True) * 1e6:.0f} us "
      f"(43 GFLOP of contractions: {43.0 / tower_time(False) / 1e3:.1f} TFLOP/s on the HIP path)")
for amp in (False, True):
    for _ in range(10): step(amp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step(amp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    with torch.no_grad():
        _, fx, _, fy = method(x, y)
    fx, fy = fx.float().requires_grad_(True), fy.float().requires_grad_(True)
    for _ in range(10): method.compute_loss(fx, fy)[0].backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): method.compute_loss(fx, fy)[0].backward()
    torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / 100
    print(f"{'autocast fp16' if amp else 'float32':14s}: step {dt * 1e3:.3f} ms ({1 / dt:.0f} steps/s), of which the CDK loss forward + backward {dl * 1e6:.0f} us")
