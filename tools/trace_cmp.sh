#!/bin/bash
# per-launch trace of base and a variant, first 10 bounces of the timed frame
for v in base run4; do
  if [ "$v" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$v.so; fi
  RTX_AMD_TRACE=1 timeout -k 10 200 python -u bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 128 > gpurun_out/trace_$v.log 2>&1 || exit 1
done
python3 - <<'PY'
import re
def load(v):
    rows=[]
    for l in open(f"gpurun_out/trace_{v}.log"):
        m=re.match(r"\[rtx trace\] (\S+)\s+(\d+) items\s+([\d.]+) ms", l)
        if m: rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
    return rows
a,b=load("base"),load("run4")
print(len(a),len(b))
for i,(x,y) in enumerate(zip(a,b)):
    if i<48: print(f"{x[0]:10s} {x[1]:10d} {x[2]:8.3f} | {y[0]:10s} {y[1]:10d} {y[2]:8.3f}  {y[2]-x[2]:+7.3f}")
PY
