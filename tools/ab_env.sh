#!/bin/bash
# A/B of a run-time switch on the GPU box: tools/ab_env.sh VAR SPP v0 v1 ...   (values of the environment variable VAR)
var=$1; spp=$2; shift 2
mkdir -p gpurun_out
for v in "$@"; do
  export $var=$v
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp $spp > gpurun_out/abe_$v.txt 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abe_$v.txt; exit 1; }
  python - "$var=$v" gpurun_out/abe_$v.txt <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        k=j["kernel_ms_per_frame_rank0"]
        print(f'{sys.argv[1]:>22} spp {j["config"]["workload"].split(" spp")[0].split()[-1]:>4}: frame {j["ms_per_step"]:8.1f} ms  {j["value"]:7.1f} Mrays/s  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f}  rays {j["rays_per_frame"]:.0f} mean {j["mean_linear"]:.6f}')
PY
done
