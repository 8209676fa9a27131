#!/bin/bash
# per-launch durations of the two box kernels of config 3 (kernel trace), with and without the box-size hint
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export SCARLET_NO_BOXHINT=$v
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/boxl_$v -o out --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --config c3 > $R/gpurun_out/boxl_$v.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/boxl_$v/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_source_update_box" in r["Kernel_Name"]]
for tag in ("31>", "63>"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if tag in r["Kernel_Name"]]
    print("NO_BOXHINT=$v", tag, " ".join("%.0f" % x for x in d))
PY
done
