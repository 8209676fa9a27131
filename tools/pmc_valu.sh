#!/bin/bash
# VALU / SALU / LDS / MFMA instruction counts of the fused kernel (one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $R/gpurun_out/pmcv -o out --output-format csv -- python3 $R/tools/pmc_run.py > $R/gpurun_out/pmcv.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
acc = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/pmcv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_iterate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()})
PY
