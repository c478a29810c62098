#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 evidence behind bench.py's numbers.
#   collect_profiles.sh <tag> [bench.py arguments, e.g. --config cfg3]
#   1. kernel-trace + stats of the bench command                          -> gpurun_out/<tag>/stats_*
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | the SQ set behind MFMA
#      utilisation: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
#      SQ_ACTIVE_INST_ANY + GRBM_GUI_ACTIVE), kernel-trace only (never combined with sys/hip/hsa tracing),
#      short runs                                                        -> gpurun_out/<tag>/pmc_<counter>_*
#   3. the plain bench line                                               -> gpurun_out/<tag>/bench.json
# The profiler runs are eager steps only (--accuracy off --graph off: no 500 000-step schedule, no graph replays mixed
# into the kernel stats). The program itself follows `--` (python3 ...), as the pool requires. NSVD_PROFILE_PMC=0 skips step 2.
tag=${1:-prof}
shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 /root/repo/bench.py --no-cpu-baseline --no-extras --accuracy off --graph off "$@" > $out/bench_stats_run.log 2>&1
rm -f $out/stats_kernel_trace.csv
if [ "${NSVD_PROFILE_PMC:-1}" != "0" ]; then
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc_$n -- python3 /root/repo/bench.py --steps 12 --warmup 3 --repeats 1 --prewarm-seconds 0.2 --no-cpu-baseline --no-kernel-events --no-extras --accuracy off --graph off "$@" > $out/pmc_$n.log 2>&1
  rm -f $out/pmc_${n}_kernel_trace.csv
done
fi
python3 /root/repo/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
ls -la $out
