#!/bin/bash
# Per-bounce cost on the GPU box: tools/ab_bounces.sh SPP b0 b1 ...   (frames limited to that many bounces)
spp=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --spp $spp --max-bounces $v > gpurun_out/abb_$v.txt 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abb_$v.txt; exit 1; }
  python - "$v" gpurun_out/abb_$v.txt <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        k=j["kernel_ms_per_frame_rank0"]
        print(f'bounces<={sys.argv[1]:>3}: frame {j["ms_per_step"]:8.1f} ms  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f}  rays {j["rays_per_frame"]/1e6:9.1f} M  ext rays/launch {j["roofline"]["algorithmic_bytes_per_launch"]/j["roofline"]["bytes_per_ray"]/1e6:8.1f} M x {j["roofline"]["launches"]}')
PY
done
