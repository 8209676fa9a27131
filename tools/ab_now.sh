#!/bin/bash
# bench lines of configs 5 and 3 (per-class times), current build
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config $1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'], d['config']['mean_loss_first_last'])"; }
run c5; run c5; run c3; run c3
