"""dev helper (GPU): fused forward vs generic forward, and a quick timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import reference_init

dev = "cuda:0"
def run(L, B, m, hidden, D=2, exp=False, pot=H.POT_HYDROGEN):
    shape = H.ModelShape(L=L, D=D, m=m, hidden=hidden, has_exp_mask=exp)
    fB, ws, bs, sc = reference_init(shape, 0.1, 10.0 if exp else None, 0)
    ws = [w.to(dev) for w in ws]; bs = [(b + 0.01 * torch.randn_like(b)).to(dev) for b in bs]
    p = H.pack_params(shape, ws, bs, fB.to(dev), sc.to(dev) if exp else None)
    prob = H.make_problem(pot, 1.0, 0.01, 100.0, 0.5, 16.0)
    x = (16 * torch.randn(B, D)).to(dev)
    wsb = H.new_workspace(shape, B, dev)
    fg, Tg = H.operator_forward(shape, p, prob, x, wsb, False, H.PATH_GENERIC)
    fg, Tg = fg.clone(), Tg.clone()
    ff, Tff = H.operator_forward(shape, p, prob, x, wsb, False, H.PATH_FUSED)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(f"L={L} B={B} m={m} hid={hidden} D={D} exp={exp}: f rel {rel(ff, fg):.2e}  Tf rel {rel(Tff, Tg):.2e} "
          f"finite={bool(torch.isfinite(ff).all())}")
    return shape, p, prob, x, wsb

run(4, 64, 64, (128, 128, 128))
run(16, 64, 64, (128, 128, 128))
run(2, 512, 64, (128, 128, 128))
run(2, 64, 1024, (128, 128, 128))
run(16, 512, 64, (128, 128, 128))
run(8, 256, 256, (128, 128, 128))
run(3, 32, 64, (128,))
run(2, 96, 64, (128, 128), D=1)
run(2, 64, 64, (128, 128), D=2, exp=True, pot=H.POT_HARMONIC)
shape, p, prob, x, wsb = run(16, 512, 1024, (128, 128, 128))
for path, name in ((H.PATH_GENERIC, "generic"), (H.PATH_FUSED, "fused")):
    for _ in range(3):
        H.operator_forward(shape, p, prob, x, wsb, False, path)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        H.profile_next_forward(a, b)
        H.operator_forward(shape, p, prob, x, wsb, False, path)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    t0 = time.perf_counter()
    for _ in range(50):
        H.operator_forward(shape, p, prob, x, wsb, False, path)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 50
    fl = 2 * 5 * 512 * 16 * 295040
    print(f"{name}: dominant kernel median {ts[10]*1e3:.1f} us (min {ts[0]*1e3:.1f}); whole forward {tot*1e6:.1f} us; "
          f"fused-kernel TF/s if fused: {fl/ts[10]/1e9:.1f}")
