#!/bin/bash
# A/B of the tail_kernel threshold on the GPU box: tools/ab_tail.sh SPP n0 n1 ...   (RTX_AMD_TAIL_PATHS values; 0 = off)
spp=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  export RTX_AMD_TAIL_PATHS=$v
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp $spp > gpurun_out/abt_$v.txt 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abt_$v.txt; exit 1; }
  python - "$v" gpurun_out/abt_$v.txt <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        k=j["kernel_ms_per_frame_rank0"]
        print(f'tail<={sys.argv[1]:>7}: frame {j["ms_per_step"]:8.1f} ms  {j["value"]:7.1f} Mrays/s  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f}  rays {j["rays_per_frame"]:.0f} mean {j["mean_linear"]:.6f}')
PY
done
