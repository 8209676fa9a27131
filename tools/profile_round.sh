#!/bin/bash
# Round profile of bench.py on one MI355X: (1) rocprofv3 kernel-trace stats, (2) HBM traffic of the
# iteration kernel from FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM
# section: FETCH_SIZE counts half of the bytes of 16 B/lane coalesced reads on gfx950).
#   usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_traffic_k_iterate.json
set -e
TAG=${1:-r01_c}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
cp $(find $R/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats_bench_50steps.csv
grep "^{\"metric\"" $R/gpurun_out/prof_${TAG}_bench.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json || true
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${TAG}_$c -o out --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
python3 - $TAG <<'PY'
import csv, glob, json, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; tag = sys.argv[1]
vals = {}
kname = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(R + "/gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_iterate" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"])); kname = r["Kernel_Name"]
    vals[c] = v
S = 10000
alg = S * (4 * 64 * 64 * (5 + 2 * 4) + 8 * 4 * 5)
fm = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]); wm = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
out = {"command": "rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu",
       "kernel": kname, "scenes_per_launch": S,
       "note": "separate passes per counter; FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced (16 B/lane) read stream (MI355X_MICROARCH.md, HBM section), so fetch bytes = FETCH_SIZE*1024*2; WRITE_SIZE is exact for 16 B/lane stores",
       "FETCH_SIZE_per_launch_KiB": vals["FETCH_SIZE"], "FETCH_SIZE_mean_KiB": fm,
       "WRITE_SIZE_per_launch_KiB": vals["WRITE_SIZE"], "WRITE_SIZE_mean_KiB": wm,
       "hbm_read_bytes_per_launch": fm * 1024 * 2, "hbm_write_bytes_per_launch": wm * 1024,
       "hbm_bytes_per_launch": fm * 1024 * 2 + wm * 1024, "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": (fm * 1024 * 2 + wm * 1024) / alg}
json.dump(out, open(R + "/gpurun_out/%s_pmc_traffic_k_iterate.json" % tag, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
PY
head -5 $R/gpurun_out/${TAG}_kernel_stats_bench_50steps.csv
cat $R/gpurun_out/${TAG}_bench_under_rocprof.json | cut -c1-300
