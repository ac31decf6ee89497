#!/bin/bash
# A/B of library variants on the GPU box: tools/ab.sh SPP base w3 w4 ...  ("base" = the in-tree librtx_amd.so)
spp=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$v.so; fi
  timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --spp $spp > gpurun_out/ab_$v.log 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/ab_$v.log; exit 1; }
  python - "$v" gpurun_out/ab_$v.log <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        k=j["kernel_ms_per_frame_rank0"]
        print(f'{sys.argv[1]:>10}: frame {j["ms_per_step"]:8.1f} ms  {j["value"]:7.1f} Mrays/s  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f} (nee {k.get("nee",0):6.1f})  mean {j["mean_linear"]:.6f}')
PY
done
