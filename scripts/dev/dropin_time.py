"""dev: steps/s of the reference-style loop (get_problem / get_wavefunctions / get_evd_method / compute_loss_operator /
torch RMSprop / cosine / EMA, i.e. drop_in.train_operator's body) at configs[1], next to FusedTrainer's."""
import argparse, os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd.drop_in import ExponentialMovingAverage, get_optimizer
from neural_svd_amd.models import get_wavefunctions
from neural_svd_amd.nested_lowrank import get_evd_method
from neural_svd_amd.operators import get_dataloader, get_problem
dev = "cuda:0"
a = argparse.Namespace(problem="sch", potential_type="hydrogen", charge=1.0, ndim=2, n_particles=1, neigs=16, laplacian_eps=0.01,
                       operator_scale=100.0, operator_shift=0.0, sampling_mode="gaussian", sampling_scale=16.0, batch_size=512,
                       lim=50.0, val_eps=0.5, use_fourier_feature=True, fourier_mapping_size=1024, fourier_scale=0.1,
                       fourier_deterministic=False, fourier_append_raw=False, mlp_hidden_dims="128,128,128", parallel=1,
                       nonlinearity="softplus", apply_exp_mask=0, exp_mask_init_scale=1.0, hard_mul_const=1.0, apply_boundary=0,
                       sort=0, optimizer="rmsprop", lr=1e-4, rmsprop_decay=0.999, momentum=0.0, adam_eps=1e-7, num_iters=500000,
                       ema_decay=0.995, use_lr_scheduler=True)
a.loss = argparse.Namespace(name="neuralsvd", neuralsvd=argparse.Namespace(step=1, sequential=False))
torch.manual_seed(0)
operator, gt = get_problem(a, dev)
model = get_wavefunctions(a)
make_batch, val_data, batch_ftn_val, imp_train, imp_val = get_dataloader(a, dev)
method = get_evd_method(a, "neuralsvd", model).to(dev)
opt = get_optimizer(a, method)
sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, a.num_iters)
ema = ExponentialMovingAverage(method.parameters(), decay=a.ema_decay)
def step():
    method.train(); opt.zero_grad()
    x = make_batch().to(dev); x = x.reshape(x.shape[0], -1)
    loss, _ = method.compute_loss_operator(operator, x, importance=imp_train)
    loss.backward(); opt.step(); sched.step(); ema.update()
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 500
for _ in range(n): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"reference-style loop: {n / dt:.0f} steps/s ({1e3 * dt / n:.3f} ms/step)")
# the same through drop_in.train_operator, whose default loop body is the fused trainer on the model's weights
from neural_svd_amd.drop_in import train_operator
import json
rec = {"reference_style_loop_eager_torch_optim_steps_per_s": round(n / dt, 1)}
for fused, graph in ((True, True), (False, True), (False, False)):
    a.num_iters, a.print_freq, a.eval_freq, a.fused_loop, a.graph_loop, a.log_dir = 6000, 10 ** 9, 10 ** 9, fused, graph, None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_operator(a, method, operator, make_batch, val_data, batch_ftn_val, None, None, dev, imp_train, imp_val)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    name = "fused_loop" if fused else ("plain_loop_hip_graph" if graph else "plain_loop_eager")
    rec["train_operator_" + name + "_steps_per_s"] = round(a.num_iters / dt, 1)
    print(f"train_operator(fused_loop={fused}, graph_loop={graph}): {a.num_iters / dt:.0f} steps/s "
          f"({1e3 * dt / a.num_iters:.3f} ms/step, construction / capture included)")
print("RECORD " + json.dumps(rec))
a.num_iters = 500000
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
