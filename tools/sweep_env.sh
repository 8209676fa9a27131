#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... ; prints k_iterate ms for each value (10k scenes, 10 steps)
VAR=$1; shift
for v in "$@"; do
  echo -n "$VAR=$v : "
  env $VAR=$v ${EXTRA_ENV} timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],3), 'ms', round(d['value']/1e6,3), 'M/s')"
done
