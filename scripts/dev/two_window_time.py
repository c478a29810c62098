"""dev: the headline step with the backward as ONE pair of launches vs TWO head windows on two streams
(FusedTrainer(backward_windows=2)): steps/s eager and replayed from a HIP graph, interleaved rounds in one process,
and bit-identity of the two after 20 steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
from neural_svd_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
cfgs = {"cfg2": dict(L=16, m=1024, B=512, seq=False, pot=H.POT_HYDROGEN, sc=100.0, sh=0.0, sig=16.0, fs=0.1, mask=None),
        "cfg3": dict(L=32, m=256, B=512, seq=True, pot=H.POT_HARMONIC, sc=1.0, sh=16.0, sig=4.0, fs=1.0, mask=10.0)}
def mk(c, windows, dsched):
    shape = H.ModelShape(L=c["L"], D=2, m=c["m"], hidden=(128, 128, 128), has_exp_mask=c["mask"] is not None)
    prob = H.make_problem(c["pot"], 1.0, 0.01, c["sc"], c["sh"], c["sig"])
    return FusedTrainer(shape, prob, c["B"], sequential=c["seq"], lr=1e-4, num_iters=500000, sampling_scale=c["sig"],
                        fourier_scale=c["fs"], exp_mask_init=c["mask"], seed=0, device=dev, device_schedule=dsched,
                        backward_windows=windows)
for name, c in cfgs.items():
    a, b = mk(c, 1, False), mk(c, 2, False)
    for _ in range(20): a.step(); b.step()
    torch.cuda.synchronize()
    same = torch.equal(a.P.flat, b.P.flat) and torch.equal(a.P.ema, b.P.ema) and torch.equal(a.P.sq, b.P.sq) and torch.equal(a.loss, b.loss)
    print(name, "two windows active:", b.backward_windows == 2 and b._side_stream is not None, "bit-identical after 20 steps:", same, "loss", float(a.loss[0]), float(b.loss[0]))
    res = {1: [], 2: []}
    for r in range(5):
        for w, t in ((1, a), (2, b)):
            for _ in range(100): t.step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(1000): t.step()
            torch.cuda.synchronize(); res[w].append(1000 / (time.perf_counter() - t0))
    print(name, "eager  steps/s one window", [round(v) for v in res[1]], "two windows", [round(v) for v in res[2]])
    ga, gb = mk(c, 1, True), mk(c, 2, True)
    gsa, gsb = ga.capture_graph(2), gb.capture_graph(2)
    res = {1: [], 2: []}
    for r in range(5):
        for w, g in ((1, gsa), (2, gsb)):
            g.replay(50)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g.replay(500)
            torch.cuda.synchronize(); res[w].append(1000 / (time.perf_counter() - t0))
    print(name, "graph  steps/s one window", [round(v) for v in res[1]], "two windows", [round(v) for v in res[2]])
    torch.cuda.synchronize()
    print(name, "graph replays bit-identical:", torch.equal(ga.P.flat, gb.P.flat))
