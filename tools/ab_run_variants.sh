#!/bin/bash
# A/B of library variants built by tools/ab_variants.sh: alternate them ROUNDS times on the same box
#   usage: tools/ab_run_variants.sh <out tag> <rounds> name1 name2 ...
out=gpurun_out/$1; rounds=$2; shift 2; mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    SCARLET_LIB_PATH=$PWD/scarlet_amd/csrc/variants/lib_$v.so timeout -k 10 300 python bench.py --no-cpu --no-other --steps 50 --warmup 5 > $out/${v}_$r.json 2> $out/${v}_$r.err || { echo "FAILED $v"; tail -3 $out/${v}_$r.err; }
    python - <<PY
import json
d=json.load(open("$out/${v}_$r.json"))
print("round $r %-8s ms/step %.4f  frac %.4f  loss %s" % ("$v", d["ms_per_step"], d["roofline"]["frac"], d["config"]["mean_loss_first_last"]))
PY
  done
done
