#!/bin/bash
# VALU/SALU/LDS instruction counts of the fused kernel with constraint stages switched off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in full nosym nomono none; do
  unset PMC_NOSYM PMC_NOMONO
  case $cfg in nosym) export PMC_NOSYM=1;; nomono) export PMC_NOMONO=1;; none) export PMC_NOSYM=1 PMC_NOMONO=1;; esac
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $R/gpurun_out/pmcph_$cfg -o out --output-format csv -- python3 $R/tools/pmc_run.py > $R/gpurun_out/pmcph_$cfg.log 2>&1 || { echo "FAILED $cfg"; tail -3 $R/gpurun_out/pmcph_$cfg.log; }
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for cfg in ("full", "nosym", "nomono", "none"):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(R + "/gpurun_out/pmcph_%s/**/*counter_collection.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_iterate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(cfg, {k: "%.4g" % (sum(v) / len(v)) for k, v in acc.items()})
PY
