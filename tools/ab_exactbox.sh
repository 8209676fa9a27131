#!/bin/bash
# configs 5 and 3 with the exact-shape instances of the box kernels and with the generic ones (NO_EXACT)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config $2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'], d['roofline']['per_class_avg_ms'], d['config']['mean_loss_first_last'])"; }
for c in c5 c3; do
run "exact" $c; SCARLET_NO_EXACT=1 run "generic" $c; run "exact" $c; SCARLET_NO_EXACT=1 run "generic" $c
done
