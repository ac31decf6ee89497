#!/bin/bash
# A/B of environment settings on the GPU box with the in-tree library: tools/ab_env.sh SPP "VAR=a" "VAR=b OTHER=c" ...  ("-" = no setting)
spp=$1; shift
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --spp $spp > gpurun_out/abenv_$i.log 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/abenv_$i.log; exit 1; }
  python - "$v" gpurun_out/abenv_$i.log <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line)
        k=j["kernel_ms_per_frame_rank0"]
        print(f'{sys.argv[1]:>28}: frame {j["ms_per_step"]:8.1f} ms  {j["value"]:7.1f} Mrays/s  extend {k["extend"]:7.1f} shadow {k["shadow"]:7.1f} shade {k["shade"]:7.1f} (nee {k.get("nee",0):6.1f})  mean {j["mean_linear"]:.6f}')
PY
done
