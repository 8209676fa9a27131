#!/bin/bash
# Round-3 evidence for profiles/: one call on the GPU box (about ten minutes).
#   bash tools/profile_round3.sh <tag>          -> gpurun_out/<tag>_*
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
line() { grep '^{"metric"' "$1" > "$2" || true; }

# 1. the GPU suite, with every max-norm error a test computes
rm -f $O/${TAG}_rel_err_log.txt
SCARLET_LOG_REL_ERR=$R/$O/${TAG}_rel_err_log.txt timeout -k 10 600 python -m pytest tests -m gpu -q > $O/${TAG}_gputest.log 2>&1
sort -g -r $O/${TAG}_rel_err_log.txt > $O/${TAG}_rel_err_all_gpu_tests.txt
tail -3 $O/${TAG}_gputest.log

# 2. the driver's invocation (headline + c3 + c5 + CPU legs), then the headline alone at 50 steps and the
#    single-GPU proxies of the strong-scaling shards (10 000 scenes over 2 / 4 / 8 GPUs)
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver.log 2>&1; line $O/${TAG}_bench_driver.log $O/${TAG}_bench_driver_invocation.json
for S in 10000 5000 2500 1250; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu --no-other --scenes $S > $O/${TAG}_b.log 2>&1; line $O/${TAG}_b.log $O/${TAG}_bench_c2_S${S}_50steps.json
done
SCARLET_NO_PERSIST=1 timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu --no-other > $O/${TAG}_b.log 2>&1; line $O/${TAG}_b.log $O/${TAG}_bench_c2_S10000_50steps_one_launch_per_iteration.json
SCARLET_NO_PERSIST=1 timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu --no-other --scenes 1250 > $O/${TAG}_b.log 2>&1; line $O/${TAG}_b.log $O/${TAG}_bench_c2_S1250_50steps_one_launch_per_iteration.json
python - $TAG <<'PY'
import json, sys, os
tag = sys.argv[1]
for f in sorted(os.listdir("gpurun_out")):
    if f.startswith(tag + "_bench_c2_S") and f.endswith(".json"):
        try:
            d = json.load(open("gpurun_out/" + f))
            print("%-62s ms/step %.4f  %.2f M scene-it/s  frac %.4f" % (f, d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"]))
        except Exception as e:
            print(f, "unreadable", e)
PY

# 3. rocprofv3 kernel statistics of the 50-step headline run, of config 3 with and without the two pipelines, of config 5
prof() {   # name, bench arguments...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_${TAG}_$name -o out --output-format csv -- python3 $R/bench.py "$@" > $R/$O/prof_${TAG}_$name.log 2>&1 )
  cp $(find $O/prof_${TAG}_$name -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_$name.csv
  line $O/prof_${TAG}_$name.log $O/${TAG}_bench_under_rocprof_$name.json
}
prof bench_c2_50steps --steps 50 --warmup 5 --no-cpu --no-other
prof bench_c3_10steps --steps 10 --warmup 2 --no-cpu --config c3
SCARLET_NO_PIPELINE=1 prof bench_c3_10steps_nopipeline --steps 10 --warmup 2 --no-cpu --config c3
prof bench_c5_10steps --steps 10 --warmup 2 --no-cpu --config c5
head -4 $O/${TAG}_kernel_stats_bench_c2_50steps.csv | cut -c1-200

# 4. counters of k_fit2x on ONE steady-state launch of 10 iterations (tools/pmc_run.py), separate --pmc passes
bash tools/pmc_any.sh ${TAG}_fit2x k_fit2x tools/pmc_run.py > $O/${TAG}_pmc_fit2x.log 2>&1
python - $TAG <<'PY'
import json, os, sys
tag = sys.argv[1]
pm = json.load(open("gpurun_out/pmc_%s_fit2x.json" % tag))
S, n = 10000, 10
alg = S * (4 * 64 * 64 * (5 + 2 * 4) + 8 * 4 * 5)
rd, wr = pm.get("hbm_read_bytes_x2_correction", 0), pm.get("hbm_write_bytes", 0)
out = {"command": "bash tools/pmc_any.sh <tag> k_fit2x tools/pmc_run.py  (rocprofv3 --kernel-trace --pmc <set>, separate passes: 8 SQ "
                  "counters per pass, FETCH_SIZE and WRITE_SIZE in passes of their own; 10 warm iterations as single-iteration "
                  "launches, then ONE k_fit2x launch of 10 iterations on 10 000 scenes)",
       "kernel": pm.get("kernel"), "scenes_per_launch": S, "iterations_per_launch": n, "launches": pm.get("launches"),
       "note": "FETCH_SIZE / WRITE_SIZE in KiB per launch; FETCH_SIZE x 2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM "
               "section); Infinity-Cache hits are counted.  Per iteration the kernel reads the images (80 KB per scene) and the "
               "rows of the previous morphology that are not zero, and writes those rows of the new one; the morphologies "
               "themselves stay in LDS between iterations.",
       "FETCH_SIZE_KiB": pm.get("FETCH_SIZE"), "WRITE_SIZE_KiB": pm.get("WRITE_SIZE"),
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
       "hbm_bytes_per_launch_and_iteration": (rd + wr) / n,
       "algorithmic_bytes_per_launch_and_iteration": alg, "traffic_over_algorithmic": (rd + wr) / n / alg,
       "sq_counters_per_launch": {k: v for k, v in pm.items() if k.startswith("SQ_")}}
if "SQ_LDS_IDX_ACTIVE" in pm:
    out["lds_bank_conflict_share_of_lds_active_cycles"] = pm["SQ_LDS_BANK_CONFLICT"] / pm["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open("gpurun_out/%s_pmc_traffic_k_fit2x.json" % tag, "w"), indent=1)
print({k: out[k] for k in ("kernel", "hbm_bytes_per_launch_and_iteration", "algorithmic_bytes_per_launch_and_iteration", "traffic_over_algorithmic")})
PY

# 4b. counters of config 3's convolution kernel (k_psf_conv_x128: one launch per half-batch of 2048 scenes), phase stamps
bash tools/pmc_any.sh ${TAG}_c3conv k_psf_conv tools/pmc_run_c3.py > $O/${TAG}_pmc_c3conv.log 2>&1
python - $TAG <<'PY'
import json, os, sys
tag = sys.argv[1]
pm = json.load(open("gpurun_out/pmc_%s_c3conv.json" % tag))
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
if "SQ_INSTS_VALU" in pm and "SQ_WAVES" in pm:
    out["vector_instructions_per_wave"] = pm["SQ_INSTS_VALU"] / pm["SQ_WAVES"]
json.dump(out, open("gpurun_out/%s_pmc_traffic_k_psf_conv.json" % tag, "w"), indent=1)
print({k: out[k] for k in ("kernel", "hbm_bytes_per_launch", "vector_instructions_per_wave") if k in out})
PY
SCARLET_STAMPS=1 python tools/stamps_c3.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_stamps_k_psf_conv.txt
SCARLET_STAMPS=1 SCARLET_NO_EXACT=1 python tools/stamps_c3.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_stamps_k_psf_conv_generic.txt
tail -16 $O/${TAG}_stamps_k_psf_conv.txt

# 5. launch-level diagnostics and phase stamps of k_fit2x
SCARLET_STAMPS=1 python tools/occupancy.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_occupancy_k_fit2x.txt
SCARLET_STAMPS=1 STAMP_PRE=11 STAMP_ITERS=33 python tools/stamps.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_stamps_k_fit2x.txt
SCARLET_NO_PERSIST=1 SCARLET_STAMPS=1 STAMP_PRE=43 STAMP_ITERS=1 python tools/stamps.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_stamps_k_iterate2.txt
cat $O/${TAG}_occupancy_k_fit2x.txt | head -6; tail -3 $O/${TAG}_stamps_k_fit2x.txt

# 6. N > 1 control flow rehearsed on one GPU (two ranks over gloo)
SCARLET_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --scenes 5000 --no-cpu > $O/${TAG}_b.log 2>$O/${TAG}_b.err; line $O/${TAG}_b.log $O/${TAG}_bench_c2_2ranks_one_gpu_rehearsal_weak.json
SCARLET_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --scenes 10000 --strong --no-cpu > $O/${TAG}_b.log 2>$O/${TAG}_b.err; line $O/${TAG}_b.log $O/${TAG}_bench_c2_2ranks_one_gpu_rehearsal_strong.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
