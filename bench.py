#!/usr/bin/env python3
"""Benchmark of the north-star hot path: NestedLoRA training steps/s on 2D hydrogen, L=16, B=512,
joint nesting (BASELINE.json configs[1]); synthetic Gaussian-sampled coordinate batches, reference
initialisation (random weights of the real architecture).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = sample x ~ N(0, 16^2 I) on the device -> operator forward (5 stencil evaluations of the
16-headed MLP + FD Hamiltonian) -> EVD loss -> backward -> RMSprop(+cosine LR) -> EMA, all HIP kernels.
N > 1: data parallel, every rank draws its own 512 rows (weak scaling, global batch 512 N), one
all-reduce of the 2L^2+1 moment floats and one of the flat gradient per step over RCCL.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak (= f32 vector peak)
PEAK_HBM_GBS = 8000.0

CFG = dict(L=16, D=2, m=1024, hidden=(128, 128, 128), B=512, sequential=False, eps=0.01, op_scale=100.0, op_shift=0.0,
           sigma=16.0, fourier_scale=0.1, lr=1e-4, alpha=0.999, ema_decay=0.995, num_iters=500000,
           potential="hydrogen", exp_mask_init=None)
# the other BASELINE.json configs are parity-test cases, not bench lines; --config lets a developer time them
ALT = {
    "cfg1": dict(CFG, B=128, sequential=True),
    "cfg2": CFG,
    "cfg3": dict(CFG, L=32, m=256, B=512, sequential=True, op_scale=1.0, op_shift=16.0, sigma=4.0, fourier_scale=1.0,
                 num_iters=100000, potential="oscillator", exp_mask_init=10.0),
}


def macs_per_sample_head(m, hidden):
    F, dims = 2 * m, list(hidden) + [1]
    M = F * dims[0] + sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    return M, F * dims[0]


def algorithmic_flops(cfg, B):
    """2 B L (E M + M + (M - M1)) per step; forward kernel: 2 B L E M."""
    M, M1 = macs_per_sample_head(cfg["m"], cfg["hidden"])
    E = 1 + 2 * cfg["D"] if cfg["eps"] > 0 else 2 + cfg["D"]  # stencil points, or the streams of the exact-Laplacian jet
    return 2.0 * B * cfg["L"] * (E * M + M + (M - M1)), 2.0 * B * cfg["L"] * E * M


def cpu_baseline(cfg, seconds_budget=25.0):
    """Time oracle/torch_port.py (eager-PyTorch restatement of the reference's op sequence) on the
    host cores for the SAME workload; bounded sample."""
    from oracle import nsvd_oracle as O
    from oracle import torch_port as TP
    p = O.init_params(cfg["L"], cfg["D"], cfg["m"], cfg["hidden"], cfg["fourier_scale"], seed=0)
    prob = O.Problem(potential=O.POT_HYDROGEN, charge_or_k=1.0, eps=cfg["eps"], op_scale=cfg["op_scale"],
                     op_shift=cfg["op_shift"], sigma=cfg["sigma"])
    v, M = (O.sequential_nesting_masks(cfg["L"]) if cfg["sequential"] else O.joint_nesting_masks(cfg["L"], 1))
    st = TP.PortStep(p, prob, v, M, lr=cfg["lr"], alpha=cfg["alpha"], ema_decay=cfg["ema_decay"],
                     num_iters=cfg["num_iters"])
    g = torch.Generator().manual_seed(0)
    draw = lambda: cfg["sigma"] * torch.randn(cfg["B"], cfg["D"], generator=g)  # noqa: E731

    def one_step():
        x = draw()
        t0 = time.perf_counter()
        st.step(x)
        return time.perf_counter() - t0

    # eager PyTorch on a many-core host is fastest well below the logical CPU count: calibrate the
    # intra-op thread count on one step each (first call per setting also warms the thread pool)
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
    t_start = time.perf_counter()
    best, cores = None, cands[0]
    for c in cands:
        torch.set_num_threads(c)
        one_step()
        t = one_step()
        if best is None or t < best:
            best, cores = t, c
        if time.perf_counter() - t_start > 0.5 * seconds_budget:
            break
    torch.set_num_threads(cores)
    left = seconds_budget - (time.perf_counter() - t_start)
    n = max(3, min(20, int(left / max(best, 1e-3))))
    times = sorted(one_step() for _ in range(n))
    med = times[len(times) // 2]
    return dict(value=1.0 / med, unit="steps/s", cores=cores, kind="port",
                sample=f"{n} timed full optimiser steps of the same workload (B={cfg['B']}), median, after a "
                       f"thread-count calibration over {cands} (best: {cores} of {ncpu} logical CPUs); torch "
                       f"{torch.__version__} CPU eager")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--path", default="auto", choices=["auto", "generic", "fused", "bf16x3"],
                    help="bf16x3: developer option, first layer on the bf16 MFMA with three-way split operands "
                         "(NSVD_PATH_FUSED_BF16X3; not the headline path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel with events")
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--laplacian-eps", type=float, default=None,
                    help="developer option: override the config's finite-difference eps (<= 0: exact Laplacian)")
    ap.add_argument("--config", default="cfg2", choices=sorted(ALT))
    ap.add_argument("--parallelism", default="auto", choices=["auto", "dp", "hp"],
                    help="N > 1: dp = samples sharded (moments + gradient all-reduce); hp = heads sharded (one "
                         "all-gather of f, Tf, no gradient traffic); auto = hp when L %% N == 0")
    args = ap.parse_args()

    from neural_svd_amd import hip_ops as H
    from neural_svd_amd import parallel
    from neural_svd_amd.trainer import FusedTrainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one process per GPU)")
    # NSVD_FORCE_DEVICE / NSVD_DIST_BACKEND: developer aid to exercise the N > 1 code path on a 1-GPU box
    # (all ranks on one device, gloo instead of RCCL); never set by the driver
    if os.environ.get("NSVD_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["NSVD_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = parallel.Communicator.from_env(dev, backend=os.environ.get("NSVD_DIST_BACKEND")) if world > 1 else None

    cfg = dict(ALT[args.config])
    if args.batch_size:
        cfg["B"] = args.batch_size
    if args.laplacian_eps is not None:
        cfg["eps"] = args.laplacian_eps
    osc = cfg["potential"] == "oscillator"
    shape = H.ModelShape(L=cfg["L"], D=cfg["D"], m=cfg["m"], hidden=cfg["hidden"], has_exp_mask=osc)
    prob = H.make_problem(H.POT_HARMONIC if osc else H.POT_HYDROGEN, 1.0, cfg["eps"], cfg["op_scale"], cfg["op_shift"],
                          cfg["sigma"])
    path = {"auto": H.PATH_AUTO, "generic": H.PATH_GENERIC, "fused": H.PATH_FUSED,
            "bf16x3": H.PATH_FUSED_BF16X3}[args.path]
    par = args.parallelism
    if par == "auto":
        par = "hp" if (world > 1 and cfg["L"] % world == 0) else "dp"
    tr = FusedTrainer(shape, prob, cfg["B"], parallelism=par, sequential=cfg["sequential"], lr=cfg["lr"], rmsprop_decay=cfg["alpha"],
                      ema_decay=cfg["ema_decay"], num_iters=cfg["num_iters"], sampling_scale=cfg["sigma"],
                      fourier_scale=cfg["fourier_scale"], exp_mask_init=cfg["exp_mask_init"], seed=0, device=dev,
                      path=path, comm=comm)

    for _ in range(args.warmup):
        tr.step()
    torch.cuda.synchronize()

    use_ev = not args.no_kernel_events
    EV_EVERY = 4  # bracket the dominant kernel on every 4th timed step (two event records cost ~2 us)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range((args.steps + EV_EVERY - 1) // EV_EVERY)] if use_ev else []
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if use_ev and i % EV_EVERY == 0:
            H.profile_next_forward(*ev[i // EV_EVERY])
        tr.step()
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max_float(elapsed)

    loss = float(tr.loss[0])
    finite = bool(torch.isfinite(tr.P.flat).all())
    if rank != 0:
        if comm is not None:
            comm.close()
        return
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed
    flops_step, flops_fwd = algorithmic_flops(cfg, cfg["B"])
    roof = None
    if use_ev:
        kms = sorted(a.elapsed_time(b) for a, b in ev)
        kavg = sum(kms) / len(kms)
        kname = H.dominant_kernel_name(shape, cfg["B"], path)
        if kname.startswith("gemm_generic"):  # generic path: the bracketed launch is the layer-0 GEMM only
            flops_fwd = 2.0 * (1 + 2 * cfg["D"]) * cfg["B"] * cfg["L"] * (2 * cfg["m"]) * cfg["hidden"][0]
        ach = flops_fwd / (kavg * 1e-3) / 1e12
        nh = len(cfg["hidden"])
        alg_mb = 4.0 * (tr.P.n_trainable + cfg["B"] * 2 * cfg["m"] + nh * tr.shape.L * cfg["hidden"][0] * tr.B
                        + 3 * tr.B * tr.shape.L) / 1e6
        traffic, traffic_src = None, "profiles/"
        try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), same workload only
            tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
            if tj.get("workload") == "cfg2" and cfg["B"] == CFG["B"] and kname.startswith("pmlp_fused_fwd"):
                traffic = int(tj["kernels"]["pmlp_fused_fwd"]["hbm_bytes_corrected"])
                traffic_src = tj.get("source", "profiles/")
        except Exception:  # noqa: BLE001
            traffic = None
        roof = dict(bound="mfma", achieved=round(ach, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic,
                    traffic_note=f"HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction) from "
                                 f"{traffic_src}; algorithmic bytes {alg_mb:.1f} MB (parameters + centre features "
                                 f"read once, saved activations + f, Tf, jac written once)",
                    kernel=kname, kernel_avg_us=round(kavg * 1e3, 2),
                    kernel_flops=flops_fwd,
                    step_flops=flops_step, step_tflops=round(flops_step / (ms_per_step * 1e-3) / 1e12, 3),
                    step_frac=round(flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
    out = {
        "metric": "training steps/sec, 2D hydrogen L=16 B=512 (NestedLoRA joint nesting, full optimiser step)",
        "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: 2D hydrogen, L=16, batch_size=512 per GPU, joint nesting, "
                               "MLP 2048(Fourier m=1024)->128->128->128->1 x16 heads, eps=0.01, RMSprop+cosine+EMA",
                   "global_batch": cfg["B"] * world,
                   "parallelism": (f"{par}{world}" if world > 1 else "dp1"),
                   "sharding": ("heads: each GPU owns L/N heads and evaluates them on the whole global batch; one "
                                "all-gather of f,Tf per step, no gradient traffic" if (par == "hp" and world > 1) else
                                "samples: each GPU draws its own 512 rows; all-reduce of 2L^2+1 moments and of the "
                                "flat gradient per step" if world > 1 else "single GPU: no exchange"),
                   "optimiser": ("RMSprop+EMA step fused into the weight-gradient kernel" if tr.fused_step else
                                 "separate RMSprop+EMA kernel after the gradient all-reduce"),
                   "path": H.path_name(tr.shape, tr.B, path, prob), "params": tr.P.n_trainable * (world if tr.hp else 1)},
        "final_loss": loss, "params_finite": finite,
        "roofline": roof,
    }
    try:  # the other half of BASELINE.json's metric: eigenvalue error after the full schedule (committed run records)
        if args.config != "cfg2" or args.laplacian_eps is not None or args.batch_size:
            raise KeyError("headline workload only")
        aj = json.load(open(os.path.join(ROOT, "profiles", "latest_accuracy.json")))
        key = "bf16x3" if args.path == "bf16x3" else "fp32"
        r = aj["runs"][key]
        out["rel_eigenvalue_error"] = {"value": round(r["rel_err_mean"], 5), "max": round(r["rel_err_max"], 5),
                                       "after_steps": r["steps"], "train_seconds": r["train_seconds"],
                                       "source": r["source"], "not_measured_in_this_run": True,
                                       "reference_published": aj["reference_published"]}
    except Exception:  # noqa: BLE001
        pass
    if args.config != "cfg2" or args.laplacian_eps is not None or args.batch_size:
        out["metric"] = f"training steps/sec, developer config {args.config} (not the headline workload)"
        out["config"]["workload"] = f"{args.config}: {cfg}"
    if args.path == "bf16x3":
        out["metric"] += " [developer path: layer 0 as bf16x3 split products, fp32 accumulation]"
        out["config"]["layer0"] = ("bf16 MFMA, operands split into 3 bf16 planes, 6 partial products, fp32 "
                                   "accumulate; roofline.frac stays relative to the fp32 MFMA peak")
    if world == 1 and args.path == "auto" and args.config == "cfg2" and args.laplacian_eps is None \
            and not args.batch_size:
        # side measurement, never the headline: the same K steps with the opt-in forward
        try:
            tr2 = FusedTrainer(shape, prob, cfg["B"], parallelism="dp", sequential=cfg["sequential"], lr=cfg["lr"],
                               rmsprop_decay=cfg["alpha"], ema_decay=cfg["ema_decay"], num_iters=cfg["num_iters"],
                               sampling_scale=cfg["sigma"], fourier_scale=cfg["fourier_scale"],
                               exp_mask_init=cfg["exp_mask_init"], seed=0, device=dev, path=H.PATH_FUSED_BF16X3)
            for _ in range(args.warmup):
                tr2.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tr2.step()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            out["opt_in_path_bf16x3"] = {
                "value": round(args.steps / e2, 3), "unit": "steps/s", "ms_per_step": round(1e3 * e2 / args.steps, 4),
                "final_loss": float(tr2.loss[0]), "params_finite": bool(torch.isfinite(tr2.P.flat).all()),
                "note": "same workload and step count with NSVD_PATH_FUSED_BF16X3 (first layer as 3-way split bf16 "
                        "products, fp32 accumulation, float32-accurate: DESIGN.md 3.7); not the headline value"}
        except Exception as e:  # noqa: BLE001
            out["opt_in_path_bf16x3"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
        cb = cpu_baseline(cfg)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
