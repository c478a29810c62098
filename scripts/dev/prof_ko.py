import sys, os, time
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace as NS
import torch
from neural_svd_amd.kernel_ops import DenseKernelOperator
from neural_svd_amd.models import get_wavefunctions
from neural_svd_amd.nested_lowrank import get_evd_method
dev="cuda:0"
g=torch.Generator().manual_seed(0); N=4000
z=torch.randn(N,2,generator=g); d2=(z[:,None,:]-z[None,:,:]).pow(2).sum(-1); K=20*torch.exp(-d2/(2*0.75**2))
op=DenseKernelOperator(K.float().to(dev), z.to(dev))
args=NS(ndim=2,n_particles=1,use_fourier_feature=True,fourier_mapping_size=64,fourier_scale=0.3,fourier_deterministic=False,fourier_append_raw=False,mlp_hidden_dims="128,128,128",neigs=8,parallel=1,nonlinearity="softplus",apply_exp_mask=0,exp_mask_init_scale=1.0,hard_mul_const=1.0,apply_boundary=0,sort=0,loss=NS(neuralsvd=NS(step=1,sequential=True)))
net=get_wavefunctions(args).to(dev); method=get_evd_method(args,"neuralsvd",op.index_model(net)).to(dev)
opt=torch.optim.RMSprop(method.parameters(),lr=1e-3,alpha=0.999,eps=1e-10)
gen=torch.Generator(device=dev).manual_seed(1)
def step():
    idx=op.sample_indices(1024,gen); opt.zero_grad(set_to_none=True)
    loss,_=method.compute_loss_kernel(op.get_approx_kernel_op,idx,None,split_batch=False); loss.backward(); opt.step()
for _ in range(20): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter()-t0)*10)
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
