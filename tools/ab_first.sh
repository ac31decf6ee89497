#!/bin/bash
# Like ab.sh but with max_bounces=1 (every variant shades the same primary hits): isolates per-shade cost.
spp=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$v.so; fi
  timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --spp $spp --max-bounces 1 > gpurun_out/abf_$v.log 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abf_$v.log; exit 1; }
  python - "$v" gpurun_out/abf_$v.log <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line); k=j["kernel_ms_per_frame_rank0"]
        print(f'{sys.argv[1]:>10}: frame {j["ms_per_step"]:8.1f} ms  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f}  mean {j["mean_linear"]:.6f}')
PY
done
