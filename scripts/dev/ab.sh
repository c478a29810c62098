#!/bin/bash
# dev (GPU box): kernel-level A/B of the backward pair: per-kernel average durations from rocprofv3 --kernel-trace --stats
# of a short bench run, for the default build and with the environment switches given as arguments, e.g.
#   bash scripts/dev/ab.sh tag NSVD_WGRAD_TILES=1          (BENCH_ARGS="--config cfg3" adds bench.py arguments)
tag=${1:-ab}; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  ( export "$@" _X=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 /root/repo/bench.py --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/$name.json 2> $out/$name.err )
  rm -f $out/${name}_kernel_trace.csv
  echo "== $name"; python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${name}_kernel_stats.csv")))
for r in rows[:6]:
    print(f"{r['Name'][:70]:<72}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
try:
    d = json.load(open("$out/$name.json")); print("steps/s", d["value"], "ms/step", d["ms_per_step"], "loss", d["final_loss"], d["params_finite"])
except Exception as e: print("bench json:", e, open("$out/$name.err").read()[-1500:])
PY
}
run new
for e in "$@"; do run "$(echo $e | tr '=/' '__' | tail -c 40)" "$e"; done
