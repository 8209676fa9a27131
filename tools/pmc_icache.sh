#!/bin/bash
# instruction-cache / issue-stall counters of k_iterate (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" ; do
  n=$(echo $set | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$n -o out --output-format csv -- python3 $R/tools/pmc_run.py > $R/gpurun_out/pmc_$n.log 2>&1 || { echo "FAILED $set"; tail -3 $R/gpurun_out/pmc_$n.log; }
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_iterate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "per-launch mean %.4g (n=%d)" % (sum(v) / len(v), len(v)))
PY
