#!/bin/bash
# config 5 with the MFMA Gram pass and with the chunk-pair one
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config c5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'])"; }
run "mfma gram"; SCARLET_NO_GRAM_MFMA=1 run "chunk pairs"; run "mfma gram"; SCARLET_NO_SIDE_STREAM=1 run "mfma gram, one stream"; SCARLET_NO_SIDE_STREAM=1 SCARLET_NO_GRAM_MFMA=1 run "chunk pairs, one stream"
