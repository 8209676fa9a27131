#!/bin/bash
# config 5 with and without the second stream (Gram + Lipschitz beside the morphology step)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config c5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'])"; }
run "side stream"; SCARLET_NO_SIDE_STREAM=1 run "one stream"; run "side stream"; SCARLET_NO_SIDE_STREAM=1 run "one stream"
