#!/bin/bash
# A/B of the small-box kernel's occupancy target.  The target is a macro (boxupdate.h SC_UB_WAVES); the variants are
# built side by side by tools/ab_variants.sh (HERE, in the container, before the gpurun call) and selected on the
# GPU box with SCARLET_LIB_PATH -- no tracked source is edited.
#   container:  tools/ab_variants.sh ub4 "-DSC_UB_WAVES=4" ub3 "-DSC_UB_WAVES=3"
#   GPU box:    bash tools/ab_box.sh ub4 ub3
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export SCARLET_LIB_PATH=$PWD/scarlet_amd/csrc/variants/lib_$v.so
  python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -1
  for c in c3 c5; do
    python bench.py --steps 10 --warmup 2 --config $c --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $c', d['ms_per_step'], d['roofline']['dominant_kernel']['avg_launch_ms'])"
  done
done
