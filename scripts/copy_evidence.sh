#!/bin/bash
# build-container side: gpurun_out/<tag>* (what scripts/gpu/r06z.sh <tag> left) -> the tracked files under profiles/
#   copy_evidence.sh <tag>
tag=${1:-r06z}
cd "$(dirname "$0")/.."
o=gpurun_out/$tag
for c in cfg1 cfg2 cfg3 cfg4 cfg5 cfg5_amp cfg5_f16; do
  [ -f gpurun_out/${tag}_$c/stats_kernel_stats.csv ] && cp gpurun_out/${tag}_$c/stats_kernel_stats.csv profiles/${tag}_kernel_stats_$c.csv
  [ -s gpurun_out/${tag}_$c/bench.json ] && cp gpurun_out/${tag}_$c/bench.json profiles/${tag}_bench_$c.json
done
[ -s $o/bench_driver_args.json ] && cp $o/bench_driver_args.json profiles/${tag}_bench_cfg2_driver_args.json
[ -s $o/bench_n2_gloo.json ] && cp $o/bench_n2_gloo.json profiles/${tag}_bench_n2_one_device_gloo.json
[ -s $o/bench_rccl_world1.json ] && cp $o/bench_rccl_world1.json profiles/${tag}_bench_rccl_world1.json
for f in generic_h256_by_grid generic_h64_by_grid generic_h256_pmc_sq generic_h256_pmc_l2; do
  [ -s $o/$f.txt ] && grep -v "simple_timer\|output_stream\|^W2026\|^E2026" $o/$f.txt | cut -c1-400 > profiles/${tag}_$f.txt
done
python3 scripts/summarize_pmc.py ${tag}_cfg2 $tag
python3 scripts/summarize_pmc.py ${tag}_cfg5_amp $tag cfg5_amp
git status --short profiles | head -40
