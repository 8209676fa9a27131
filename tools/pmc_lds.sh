#!/bin/bash
# LDS bank-conflict counters of ONE kernel (name substring), one rocprofv3 --pmc pass: tools/pmc_lds.sh <tag> <kernel substring> <script>
TAG=$1; KSUB=$2; SCRIPT=$3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/pmcl_$TAG -o out --output-format csv -- python3 $R/$SCRIPT > $R/gpurun_out/pmcl_$TAG.log 2>&1 || tail -3 $R/gpurun_out/pmcl_$TAG.log
python3 - $TAG "$KSUB" <<'PY'
import csv, glob, os, collections, sys
R = os.environ["GRAFT_REPO_ROOT"]; tag, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(R + "/gpurun_out/pmcl_" + tag + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in acc.items()}
print(tag, {k: "%.4g" % v for k, v in sorted(out.items())}, "conflict share of LDS-active cycles: %.3f" % (out.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, out.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
