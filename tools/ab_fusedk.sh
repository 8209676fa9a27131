#!/bin/bash
# config 5 with the MFMA-fused residual + step pass and with the separate passes
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config c5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'], d['config']['mean_loss_first_last'])"; }
run "fused"; SCARLET_NO_BIGK_FUSED=1 run "separate"; run "fused"; SCARLET_NO_BIGK_FUSED=1 run "separate"
