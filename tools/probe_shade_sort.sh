#!/bin/bash
# shade_kernel time with and without the class-sorted shade queue (RTX_AMD_SORT_SHADE), kernel-trace summaries of a 32 spp frame
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
for v in 0 1; do
  rm -rf gpurun_out/prof_ss$v
  RTX_AMD_SORT_SHADE=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ss$v --output-format csv -- python3 bench.py --no-cpu-baseline --spp 32 --steps 3 --warmup 1 > gpurun_out/prof_ss$v.log 2>&1 || { echo "trace $v failed"; tail -5 gpurun_out/prof_ss$v.log; exit 1; }
  f=$(find gpurun_out/prof_ss$v -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/shade_sort_stats_$v.csv && rm -rf gpurun_out/prof_ss$v
  echo "== RTX_AMD_SORT_SHADE=$v"; head -9 gpurun_out/shade_sort_stats_$v.csv | cut -c1-70,200-
done
