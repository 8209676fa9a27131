#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 512 1024 2048 4096 10000 20480; do
  python bench.py --steps 50 --warmup 5 --no-cpu --scenes $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S=$s  ms/iter %.4f  ns per scene-iteration %.1f  value %.4g' % (d['ms_per_step'], 1e6*d['ms_per_step']/$s, d['value']))"
done
