#!/bin/bash
# SQ issue/busy counters of the fused iteration kernel (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v2}
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  n=${TAG}_$(echo $set | tr ' ' '_' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$n -o out --output-format csv -- python3 $R/tools/pmc_run.py > $R/gpurun_out/pmc_$n.log 2>&1 || { echo "FAILED $set"; tail -3 $R/gpurun_out/pmc_$n.log; }
done
python3 - $TAG <<'PY'
import csv, glob, os, collections, sys
R = os.environ["GRAFT_REPO_ROOT"]; tag = sys.argv[1]
out = {}
for f in sorted(glob.glob(R + "/gpurun_out/pmc_" + tag + "_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_iterate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
        print("%-32s per-launch mean %.4g (n=%d)" % (k, out[k], len(v)))
import json
json.dump(out, open(R + "/gpurun_out/pmc_sq_" + tag + ".json", "w"), indent=1)
PY
