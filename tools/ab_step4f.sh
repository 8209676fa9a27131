#!/bin/bash
# A/B of the occupancy target of k_step_psf4f<8, 6> (BASELINE config 3) on the GPU box
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --config c3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['per_class_avg_ms'])"; }
run "4 waves (36 B scratch)"
SCARLET_NO_PSF3PASS=1 run "four-pass form"
sed -i 's/(KM \* BM <= 48 ? 4 : 2)) void k_step_psf4f/(KM * BM <= 32 ? 4 : 2)) void k_step_psf4f/' scarlet_amd/csrc/psf_path.h
make -C scarlet_amd/csrc > gpurun_out/ab_build_step4f.log 2>&1
run "3 waves (142 VGPRs)"
run "3 waves (142 VGPRs)"
