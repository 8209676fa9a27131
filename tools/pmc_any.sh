#!/bin/bash
# SQ + HBM-traffic counters of ONE kernel (name substring) under rocprofv3, separate --pmc passes with
# --kernel-trace only (MI355X_MICROARCH.md: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE in passes of their own).
#   usage: bash tools/pmc_any.sh <tag> <kernel substring> <python script> [args...]
#   -> gpurun_out/pmc_<tag>.json  (per-launch means over the launches of that kernel)
TAG=$1; KSUB=$2; SCRIPT=$3; shift 3
R=$GRAFT_REPO_ROOT
case "$SCRIPT" in /*) ;; *) SCRIPT=$R/$SCRIPT;; esac
set -- "$SCRIPT" "$@"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_${TAG}_$i -o out --output-format csv -- python3 "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "FAILED $set"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.log; }
done
python3 - $TAG "$KSUB" <<'PY'
import csv, glob, os, collections, sys, json
R = os.environ["GRAFT_REPO_ROOT"]; tag, ksub = sys.argv[1], sys.argv[2]
out = {"kernel_substring": ksub}
for f in sorted(glob.glob(R + "/gpurun_out/pmc_" + tag + "_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); out["kernel"] = r["Kernel_Name"][:120]
    for k, v in acc.items():
        out[k] = sum(v) / len(v); out.setdefault("launches", len(v))
if "FETCH_SIZE" in out:
    # KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced read streams (guide, HBM section)
    out["hbm_read_bytes_x2_correction"] = out["FETCH_SIZE"] * 1024 * 2
    out["hbm_read_bytes_raw"] = out["FETCH_SIZE"] * 1024
    out["hbm_write_bytes"] = out.get("WRITE_SIZE", 0) * 1024
json.dump(out, open(R + "/gpurun_out/pmc_" + tag + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
