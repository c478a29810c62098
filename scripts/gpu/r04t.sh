#!/bin/bash
# round 4: bf16x3 step with the planes emitted by the weight-gradient epilogue against the split launch, one box
out=/root/repo/gpurun_out/r04t
mkdir -p $out
cd /root/repo
python - <<'PY'
import torch
from tests.test_hip_parity import *
import tests.test_hip_parity as T
from neural_svd_amd import hip_ops as H
T.H = H
for path in ("generic", "auto", "bf16x3"):
    for eps, ws in ((0.5, 20.0), (0.5, 4.0), (1.0, 1.0)):
        L, D, m, hidden = 4, 2, 64, (128, 128, 128)
        p = O.init_params(L, D, m, hidden, 1.0, exp_mask_init=10.0, seed=11)
        p.ws[0] = p.ws[0] * ws
        prob_o = O.Problem(potential=O.POT_HARMONIC, eps=eps, op_scale=1.0, op_shift=16.0, sigma=4.0)
        shape, prob = shape_of(p), hip_problem(prob_o)
        ws_t, bs_t, fB, sc = to_dev(p)
        params = H.pack_params(shape, ws_t, bs_t, fB, sc)
        x = (4.0 * torch.randn(96, D, generator=torch.Generator().manual_seed(12))).to(DEV)
        f, Tf = H.operator_forward(shape, params, prob, x, H.new_workspace(shape, 96, DEV), False, T._path(path))
        ref = O.operator_forward(x.double().cpu(), p.to(torch.float64), prob_o)
        c32 = O.operator_forward(x.cpu().float(), p.to(torch.float32), prob_o)
        print(path, eps, ws, "f", rel(f, ref.f), "Tf", rel(Tf, ref.Tf), "| float32 oracle: f", rel(c32.f, ref.f), "Tf", rel(c32.Tf, ref.Tf))
PY
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 /root/repo/bench.py "$@" --no-cpu-baseline --no-extras --accuracy off --graph off > $out/bench_$name.json 2> $out/bench_$name.err
  rm -f $out/${name}_kernel_trace.csv
  python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${name}_kernel_stats.csv")))
print("== $name")
for r in rows[:5]:
    print(f"  {r['Name'][:66]:<68}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
try:
    d = json.load(open("$out/bench_$name.json")); print("  steps/s", d["value"], "ms/step", d["ms_per_step"])
except Exception as e: print("  bench:", e, open("$out/bench_$name.err").read()[-800:])
PY
}
run bf16x3_emit --path bf16x3 --steps 300 --warmup 20 --repeats 3
NSVD_PLANES_FROM_STEP=0 run bf16x3_split --path bf16x3 --steps 300 --warmup 20 --repeats 3
run bf16x3_emit2 --path bf16x3 --steps 300 --warmup 20 --repeats 3
NSVD_PLANES_FROM_STEP=0 run bf16x3_split2 --path bf16x3 --steps 300 --warmup 20 --repeats 3
