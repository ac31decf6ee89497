#!/bin/bash
# A/B of library variants in legacy shading mode on the GPU box: tools/ab_legacy.sh SPP name0 name1 ... ("base" = the built library)
spp=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$v.so; fi
  timeout -k 10 300 python -u bench.py --legacy --steps 2 --warmup 1 --no-cpu-baseline --spp $spp > gpurun_out/abg_$v.txt 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abg_$v.txt; exit 1; }
  python - "$v" gpurun_out/abg_$v.txt <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        print(f'{sys.argv[1]:>14}: frame {j["ms_per_step"]:8.1f} ms  {j["value"]:7.1f} Mrays/s  legacy kernel {j["kernel_ms_per_frame_rank0"]["shade"]:7.1f} ms  rays {j["rays_per_frame"]:.0f} mean {j["mean_linear"]:.6f}')
PY
done
