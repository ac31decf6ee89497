#!/bin/bash
# A/B of library variants on the GPU box: tools/ab_lib.sh SPP name0 name1 ...
# ("base" = raytracer_amd/librtx_amd.so, other names = raytracer_amd/variants/librtx_NAME.so from tools/build_variant.sh)
spp=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$v.so; fi
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp $spp > gpurun_out/abl_$v.txt 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abl_$v.txt; exit 1; }
  python - "$v" gpurun_out/abl_$v.txt <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        k=j["kernel_ms_per_frame_rank0"]
        print(f'{sys.argv[1]:>14}: frame {j["ms_per_step"]:8.1f} ms  {j["value"]:7.1f} Mrays/s  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f}  frac {j["roofline"]["frac"]:.3f} rays {j["rays_per_frame"]:.0f} mean {j["mean_linear"]:.6f}')
PY
done
