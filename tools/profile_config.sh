#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py for one config:  bash tools/profile_config.sh <tag> <config> [steps]
#   -> gpurun_out/<tag>_kernel_stats_<config>.csv (+ the bench line under rocprof)
TAG=$1; CFG=$2; STEPS=${3:-10}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$CFG -o out --output-format csv -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu --config $CFG > $R/gpurun_out/prof_${TAG}_${CFG}_bench.log 2>&1
cp $(find $R/gpurun_out/prof_${TAG}_$CFG -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats_${CFG}_${STEPS}steps.csv
grep "^{\"metric\"" $R/gpurun_out/prof_${TAG}_${CFG}_bench.log > $R/gpurun_out/${TAG}_bench_${CFG}_under_rocprof.json || true
head -12 $R/gpurun_out/${TAG}_kernel_stats_${CFG}_${STEPS}steps.csv | cut -c1-160
