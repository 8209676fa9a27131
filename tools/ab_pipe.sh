#!/bin/bash
# config 3: the two-stream pipeline of scarlet_fit, its convolution stagger, the exact-shape convolution instance
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config c3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'])"; }
run "default (pipeline, stagger, exact)"
SCARLET_NO_STAGGER=1 run "pipeline, exact, no stagger"
SCARLET_NO_CONV_EXACT=1 run "pipeline, stagger, generic conv"
SCARLET_NO_CONV_EXACT=1 SCARLET_NO_STAGGER=1 run "pipeline only"
SCARLET_NO_PIPELINE=1 run "one stream, exact conv"
SCARLET_NO_PIPELINE=1 SCARLET_NO_CONV_EXACT=1 run "one stream, generic conv"
run "default (pipeline, stagger, exact)"
