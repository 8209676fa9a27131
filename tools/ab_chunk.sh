#!/bin/bash
# config 3's shape at several batch sizes, one pipeline: per-scene class times (does a working set that fits the
# 256 MB Infinity Cache make the streaming passes faster?)
cd $GRAFT_REPO_ROOT
export SCARLET_NO_PIPELINE=1
for n in 128 256 512 1024 4096; do
python bench.py --steps 10 --warmup 3 --config c3 --scenes $n --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); n=$n
print(n, 'us per scene-iteration %.3f' % (1e3*d['ms_per_step']/n), {k: round(1e3*v/n,3) for k,v in d['roofline']['per_class_avg_ms'].items()})"
done
