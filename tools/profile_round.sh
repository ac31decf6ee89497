#!/bin/bash
# The measurements a round commits under profiles/: the default bench line, the rocprofv3 kernel-trace summary of the
# same command, the kernels' HBM-side traffic (two --pmc passes) and the PMC reading of what limits them. Run on the GPU box:
#   tools/profile_round.sh TAG   ->  gpurun_out/bench_c3_TAG.json, bench_c3_kernel_stats_TAG.csv, kernel_traffic.json, pmc_TAG_limiters.json
# (copy kernel_traffic.json -> profiles/rNN/kernel_traffic_c3.json and pmc_TAG_limiters.json -> profiles/rNN/kernel_limiters.json: bench.py reads them)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
timeout -k 10 600 python3 -u bench.py > gpurun_out/bench_c3_$tag.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/bench_c3_$tag.log; exit 1; }
grep '^{"metric"' gpurun_out/bench_c3_$tag.log > gpurun_out/bench_c3_$tag.json
rm -rf gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || { echo "kernel trace failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/bench_c3_kernel_stats_$tag.csv && rm -rf gpurun_out/prof_$tag
grep '^{"metric"' gpurun_out/prof_$tag.log > gpurun_out/bench_c3_under_rocprof_$tag.json
tools/pmc_traffic.sh || exit 1
bash tools/pmc.sh $tag > /dev/null || exit 1
python3 - "$tag" <<'PY'
import json, sys
j = json.load(open(f"gpurun_out/bench_c3_{sys.argv[1]}.json"))
print("frame ms", j["ms_per_step"], "Mrays/s", j["value"], "roofline", j["roofline"]["frac"], "avg extend launch ms", j["roofline"]["avg_launch_ms"],
      "cpu", j.get("cpu_baseline", {}).get("value"))
PY
head -8 gpurun_out/bench_c3_kernel_stats_$tag.csv | cut -c1-60,200-
