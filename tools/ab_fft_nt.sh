#!/bin/bash
# A/B of the convolution workgroup size on the GPU box: rebuild with -DSC_FFT_NT=<n>, bench config 3
cd $GRAFT_REPO_ROOT
for nt in 1024 256 512; do
  make -C scarlet_amd/csrc -B EXTRA="-DSC_FFT_NT=$nt" > gpurun_out/ab_build_$nt.log 2>&1
  python bench.py --steps 10 --warmup 2 --config c3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NT=$nt', d['ms_per_step'], d['roofline']['per_class_avg_ms'])"
done
