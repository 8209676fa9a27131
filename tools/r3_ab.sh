#!/bin/bash
# round-3 A/B of the headline path: per-iteration launches (NO_PERSIST) against k_fit2x, at several batch sizes
set -e
out=gpurun_out/$1; mkdir -p $out
for S in 10000 1250 2500 5000; do
  for mode in 1 0; do
    SCARLET_NO_PERSIST=$mode timeout -k 10 300 python bench.py --no-cpu --steps 50 --warmup 5 --scenes $S > $out/bench_S${S}_nopersist${mode}.json 2> $out/bench_S${S}_nopersist${mode}.err
    python - <<PY
import json
d=json.load(open("$out/bench_S${S}_nopersist${mode}.json"))
print("S=$S NO_PERSIST=$mode ms/step %.4f  value %.3fM  frac %.4f  launch_ms %.4f" % (d["ms_per_step"], d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
