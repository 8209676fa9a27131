#!/bin/bash
# config 3 with the two-stream pipeline of scarlet_fit and without it
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config c3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'], d['config']['mean_loss_first_last'])"; }
run "two streams"; SCARLET_NO_PIPELINE=1 run "one stream"; run "two streams"; SCARLET_NO_PIPELINE=1 run "one stream"
