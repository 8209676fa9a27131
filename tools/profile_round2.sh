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
python - ${TAG} <<'PY'
# per-launch HBM traffic of k_psf_conv in the form bench.py looks up (scarlet_fit launches it per half-batch)
import json, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; tag = sys.argv[1]
pm = json.load(open(R + "/gpurun_out/pmc_%s_c3conv.json" % tag))
S = 4096 // 2
out = {"command": "bash tools/pmc_any.sh <tag> k_psf_conv tools/pmc_run_c3.py  (rocprofv3 --kernel-trace --pmc <set>, separate passes; "
                  "4096 scenes, launched by scarlet_fit as two half-batches of 2048)",
       "kernel": pm.get("kernel"), "scenes_per_launch": S, "launches": pm.get("launches"),
       "note": "FETCH_SIZE / WRITE_SIZE in KiB per launch; FETCH_SIZE x 2 on gfx950 for wide coalesced reads (guide, HBM section); "
               "the kernel's own stream is one 64 KiB model plane + one 64 KiB image plane read and one 64 KiB gradient plane "
               "written per (scene, band)",
       "FETCH_SIZE_mean_KiB": pm.get("FETCH_SIZE"), "WRITE_SIZE_mean_KiB": pm.get("WRITE_SIZE"),
       "hbm_read_bytes_per_launch": pm.get("hbm_read_bytes_x2_correction"), "hbm_write_bytes_per_launch": pm.get("hbm_write_bytes"),
       "hbm_bytes_per_launch": pm.get("hbm_read_bytes_x2_correction", 0) + pm.get("hbm_write_bytes", 0),
       "kernel_own_bytes_per_launch (model + image read, G written)": S * 5 * 3 * 128 * 128 * 4,
       "algorithmic_bytes_per_launch (whole iteration, SURVEY 8d)": S * (4 * 128 * 128 * (5 + 16) + 8 * 8 * 5),
       "sq_counters_per_launch": {k: v for k, v in pm.items() if k.startswith("SQ_")}}
json.dump(out, open(R + "/gpurun_out/%s_pmc_traffic_k_psf_conv.json" % tag, "w"), indent=1)
PY
bash tools/pmc_any.sh ${TAG}_c5box "k_source_update_box<16" tools/pmc_run_c5.py > gpurun_out/${TAG}_pmc_c5box.log 2>&1
python - ${TAG} <<'PY'
# per-launch HBM traffic of config 5's dominant kernel (the 63 x 63 box on 256 x 256 planes), in the form bench.py looks up
import json, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; tag = sys.argv[1]
pm = json.load(open(R + "/gpurun_out/pmc_%s_c5box.json" % tag))
S, K, HW = 64, 30, 256 * 256
out = {"command": "bash tools/pmc_any.sh <tag> 'k_source_update_box<16' tools/pmc_run_c5.py  (rocprofv3 --kernel-trace --pmc <set>, separate passes; 64 scenes)",
       "kernel": pm.get("kernel"), "scenes_per_launch": S, "launches": pm.get("launches"),
       "note": "FETCH_SIZE / WRITE_SIZE in KiB per launch; FETCH_SIZE x 2 on gfx950 for wide coalesced reads (guide, HBM section).  "
               "The kernel's own streams per component: the stepped plane (window rows, GEMM 1) and the previous plane (convergence "
               "sums) read, the new plane written: 3 x 256 KiB",
       "FETCH_SIZE_mean_KiB": pm.get("FETCH_SIZE"), "WRITE_SIZE_mean_KiB": pm.get("WRITE_SIZE"),
       "hbm_read_bytes_per_launch": pm.get("hbm_read_bytes_x2_correction"), "hbm_write_bytes_per_launch": pm.get("hbm_write_bytes"),
       "hbm_bytes_per_launch": pm.get("hbm_read_bytes_x2_correction", 0) + pm.get("hbm_write_bytes", 0),
       "kernel_own_bytes_per_launch (two planes read, one written per component)": S * K * 3 * HW * 4,
       "algorithmic_bytes_per_launch (whole iteration, SURVEY 8d)": S * (4 * HW * (6 + 2 * K) + 8 * K * 6),
       "sq_counters_per_launch": {k: v for k, v in pm.items() if k.startswith("SQ_")}}
json.dump(out, open(R + "/gpurun_out/%s_pmc_traffic_k_source_update_box_c5.json" % tag, "w"), indent=1)
PY
python tools/stamps_c3.py > gpurun_out/${TAG}_stamps_k_psf_conv.txt 2>&1
python tools/stamps_box.py c3psf > gpurun_out/${TAG}_stamps_box_c3.txt 2>&1
python tools/stamps_box.py c5 > gpurun_out/${TAG}_stamps_box_c5.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1
tail -1 gpurun_out/${TAG}_smoke.log
tail -2 gpurun_out/${TAG}_gputest.log
