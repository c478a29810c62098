"""dev: time the pieces of the kernel-operator step (cfg4: N=10000 points, B=8192 indices, L=64)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace as NS
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.kernel_ops import synthetic_psd_kernel
from neural_svd_amd.models import get_wavefunctions
from neural_svd_amd.nested_lowrank import get_evd_method
dev = "cuda:0"
N, D, L, B = 10000, 16, 64, 8192
op = synthetic_psd_kernel(N, 256, D, 0, dev)
args = NS(ndim=D, n_particles=1, use_fourier_feature=True, fourier_mapping_size=64, fourier_scale=0.05,
          fourier_deterministic=False, fourier_append_raw=False, mlp_hidden_dims="128,128", neigs=L, parallel=1,
          nonlinearity="softplus", apply_exp_mask=0, exp_mask_init_scale=1.0, hard_mul_const=1.0, apply_boundary=0,
          sort=0, loss=NS(neuralsvd=NS(step=1, sequential=False)))
torch.manual_seed(0)
method = get_evd_method(args, "neuralsvd", op.index_model(get_wavefunctions(args).to(dev))).to(dev)
opt = torch.optim.RMSprop(method.parameters(), lr=1e-4)
g = torch.Generator(device=dev).manual_seed(1)
idx = op.sample_indices(B, g)
f = torch.randn(B, L, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ka = t(lambda: H.kernel_apply(op.K, op.N, idx, idx, f, 1.0 / B))
print(f"kernel_apply (8192 x 8192 gathered from 10000^2, L=64): {ka:.1f} us = {2*B*N*L/ka*1e-6:.1f} TFLOP/s over the scattered form, {2*B*B*L/ka*1e-6:.1f} TFLOP/s of 2B^2L")
ref = t(lambda: op.K[idx][:, idx] @ f / B)
print(f"torch eager K[idx][:, idx] @ f / B: {ref:.1f} us")
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = method.compute_loss_kernel(op.get_approx_kernel_op, op.sample_indices(B, g), None, split_batch=False)
    loss.backward(); opt.step()
print(f"full step (model forward/backward on the E = 1 MFMA kernels, torch RMSprop): {t(step, 10):.0f} us")
