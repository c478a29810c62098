"""dev: bf16x3 layer 0 (NSVD_PATH_FUSED_BF16X3) against the native fp32 MFMA path and the float64 oracle at cfg2."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from neural_svd_amd import hip_ops as H
from oracle import nsvd_oracle as O
dev = "cuda:0"
CFG = sys.argv[1] if len(sys.argv) > 1 else "cfg2"   # cfg2 | cfg3 (oscillator: fourier_scale 1, exponential mask)
WSCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0  # scales W_0 (larger perturbations: the Taylor truncation)
if CFG == "cfg2":
    L, D, m, hidden, B = 16, 2, 1024, (128, 128, 128), 512
    p = O.init_params(L, D, m, hidden, 0.1, seed=0)
    SIG, OPS, OPSH = 16.0, 100.0, 0.0
else:
    L, D, m, hidden, B = 32, 2, 256, (128, 128, 128), 512
    p = O.init_params(L, D, m, hidden, 1.0, exp_mask_init=10.0, seed=0)
    SIG, OPS, OPSH = 4.0, 1.0, 16.0
p.ws[0] = p.ws[0] * WSCALE
shape = H.ModelShape(L=L, D=D, m=m, hidden=hidden, has_exp_mask=p.scales is not None)
params = H.pack_params(shape, [w.to(dev).contiguous() for w in p.ws], [b.to(dev).contiguous() for b in p.bs],
                       p.fourier_B.to(dev).contiguous(), None if p.scales is None else p.scales.to(dev).contiguous())
POT_O, POT_H = (O.POT_HYDROGEN, H.POT_HYDROGEN) if CFG == "cfg2" else (O.POT_HARMONIC, H.POT_HARMONIC)
prob_o = O.Problem(potential=POT_O, eps=0.01, op_scale=OPS, op_shift=OPSH, sigma=SIG)
prob = H.make_problem(POT_H, 1.0, 0.01, OPS, OPSH, SIG)
x = (SIG * torch.randn(B, D, generator=torch.Generator().manual_seed(5))).to(dev)
ws = H.new_workspace(shape, B, dev)
out = {}
for name, path in (("fp32", H.PATH_FUSED), ("bf16x3", H.PATH_FUSED_BF16X3)):
    f, Tf = H.operator_forward(shape, params, prob, x, ws, True, path)
    torch.cuda.synchronize()
    out[name] = (f.clone(), Tf.clone())
    for _ in range(20): H.operator_forward(shape, params, prob, x, ws, True, path)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for _ in range(n): H.operator_forward(shape, params, prob, x, ws, True, path)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
    print(f"{name}: forward (features + kernel) {dt:.1f} us")
rows = torch.arange(0, B, 8)
ref = O.operator_forward(x[rows.to(dev)].double().cpu(), p.to(torch.float64), prob_o)
def rel(a, b): return float((a.double().cpu() - b).norm() / b.norm())
u = 2.0 ** -23
rec = {}
for name, (f, Tf) in out.items():
    fr, Tfr = f[rows.to(dev)], Tf[rows.to(dev)]
    s = OPS * u * ref.f.abs().numpy() / 0.01 ** 2
    kappa = float(np.median(np.abs(Tfr.double().cpu().numpy() - ref.Tf.numpy()) / np.maximum(s, 1e-300)))
    rec[name] = dict(f_rel_err_vs_float64=rel(fr, ref.f), Tf_rel_err_vs_float64=rel(Tfr, ref.Tf), fd_noise_kappa_median=kappa)
    print(f"{name}: f rel err vs float64 {rel(fr, ref.f):.2e}; Tf rel err {rel(Tfr, ref.Tf):.2e}; FD-noise kappa (median) {kappa:.2f}")
# the same yardstick for the reference's own arithmetic: the oracle run in float32
c32 = O.operator_forward(x[rows.to(dev)].cpu().float(), p.to(torch.float32), prob_o)
s = OPS * u * ref.f.abs().numpy() / 0.01 ** 2
rec["oracle_float32"] = dict(f_rel_err_vs_float64=rel(c32.f, ref.f), Tf_rel_err_vs_float64=rel(c32.Tf, ref.Tf),
                             fd_noise_kappa_median=float(np.median(np.abs(c32.Tf.double().numpy() - ref.Tf.numpy()) / np.maximum(s, 1e-300))))
import json
print("RECORD " + json.dumps(dict(cfg=CFG, w0_scale=WSCALE, what="forward accuracy at configs[1] (or [2] per cfg) (seed-0 reference initialisation, 64 sampled rows x 16 heads) against the float64 oracle: native fp32 MFMA path, bf16x3 path, and the oracle run in float32 (the reference's own arithmetic); kappa = median |Tf - Tf64| / (op_scale 2^-23 |f| / eps^2)", paths=rec)))
print("bf16x3 vs fp32: f", rel(out["bf16x3"][0], out["fp32"][0].double().cpu()), "Tf", rel(out["bf16x3"][1], out["fp32"][1].double().cpu()))
import hashlib
print("bf16x3 checksum f/Tf:", hashlib.sha1(out["bf16x3"][0].cpu().numpy().tobytes()).hexdigest()[:12],
      hashlib.sha1(out["bf16x3"][1].cpu().numpy().tobytes()).hexdigest()[:12])
