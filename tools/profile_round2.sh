#!/bin/bash
# Round-2 evidence for profiles/: one call on the GPU box.
#   bash tools/profile_round2.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/${TAG}_rel_err_log.txt
SCARLET_LOG_REL_ERR=$R/gpurun_out/${TAG}_rel_err_log.txt python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1
sort -g -r gpurun_out/${TAG}_rel_err_log.txt > gpurun_out/${TAG}_rel_err_all_gpu_tests.txt
for c in c2 c3 c5; do
  python bench.py --steps 20 --warmup 5 --config $c > gpurun_out/${TAG}_bench_$c.log 2>&1
  grep '^{"metric"' gpurun_out/${TAG}_bench_$c.log > gpurun_out/${TAG}_bench_$c.json
done
SCARLET_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --scenes 5000 --no-cpu 2>/dev/null | grep '^{"metric"' > gpurun_out/${TAG}_bench_c2_2ranks_one_gpu_rehearsal_weak.json
SCARLET_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --scenes 10000 --strong --no-cpu 2>/dev/null | grep '^{"metric"' > gpurun_out/${TAG}_bench_c2_2ranks_one_gpu_rehearsal_strong.json
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}_profile_round.log 2>&1
bash tools/profile_config.sh ${TAG} c3 10 > gpurun_out/${TAG}_prof_c3.log 2>&1
bash tools/profile_config.sh ${TAG} c5 10 > gpurun_out/${TAG}_prof_c5.log 2>&1
bash tools/pmc_any.sh ${TAG}_c3conv k_psf_conv tools/pmc_run_c3.py > gpurun_out/${TAG}_pmc_c3conv.log 2>&1
python tools/stamps_c3.py > gpurun_out/${TAG}_stamps_k_psf_conv.txt 2>&1
python tools/stamps_box.py c3 > gpurun_out/${TAG}_stamps_box_c3.txt 2>&1
python tools/stamps_box.py c5 > gpurun_out/${TAG}_stamps_box_c5.txt 2>&1
tail -2 gpurun_out/${TAG}_gputest.log
