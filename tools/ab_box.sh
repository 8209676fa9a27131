#!/bin/bash
# A/B of the box kernel's occupancy target on the GPU box
cd $GRAFT_REPO_ROOT
for w in 4 3; do
  sed -i "s/__launch_bounds__(SC_BLOCK, [0-9]) void k_source_update_box/__launch_bounds__(SC_BLOCK, $w) void k_source_update_box/" scarlet_amd/csrc/boxupdate.h
  make -C scarlet_amd/csrc > gpurun_out/ab_build_box$w.log 2>&1
  python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -1
  for c in c3 c5; do
  python bench.py --steps 10 --warmup 2 --config $c --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves/SIMD=$w $c', d['ms_per_step'], d['roofline']['per_class_avg_ms'])"
  done
done
