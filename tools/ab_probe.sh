#!/bin/bash
# extend-only A/B (tools/gpu_probe.py) of library variants: tools/ab_probe.sh base g64 ...
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$v.so; fi
  timeout -k 10 200 python -u tools/gpu_probe.py 1.0 > gpurun_out/abp_$v.log 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abp_$v.log; exit 1; }
  echo "== $v"; grep -E "^(primary|bounce)" gpurun_out/abp_$v.log | cut -c1-75
done
