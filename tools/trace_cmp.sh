#!/bin/bash
# Per-launch times (RTX_AMD_TRACE=1) of the in-tree library against a variant (tools/build_variant.sh NAME ...), launch by
# launch over one frame, on the GPU box:
#   tools/trace_cmp.sh NAME [SPP]     ->  gpurun_out/trace_base.log, trace_NAME.log and the comparison on stdout
# The launches of the full-size frame are matched by kind and item count (both libraries trace the same paths), listed for the
# traversal kernels and summed per kind for launches below and above 3 M items.
v=${1:?variant name}; spp=${2:-128}
mkdir -p gpurun_out
for lib in base "$v"; do
  if [ "$lib" = base ]; then unset RTX_AMD_LIB; else export RTX_AMD_LIB=$PWD/raytracer_amd/variants/librtx_$lib.so; fi
  RTX_AMD_TRACE=1 timeout -k 10 200 python3 -u bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp "$spp" > gpurun_out/trace_$lib.log 2>&1 || { echo "$lib failed"; exit 1; }
done
python3 - "$v" <<'PY'
import collections, re, sys
def load(v):
    rows = []
    for l in open(f"gpurun_out/trace_{v}.log"):
        m = re.match(r"\[rtx trace\] (\S+)\s+(\d+) items\s+([\d.]+) ms", l)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
    return rows
a, b = load("base"), load(sys.argv[1])
# the frame proper starts at the largest extend launch (the counting pass in front of it is a small one)
big = max(r[1] for r in a if r[0] == "extend")
ia = [i for i, r in enumerate(a) if r[0] == "extend" and r[1] == big][0]
ib = [i for i, r in enumerate(b) if r[0] == "extend" and r[1] == big][0]
tot = collections.defaultdict(lambda: [0.0, 0.0])
for x, y in zip(a[ia:], b[ib:]):
    if x[:2] != y[:2]:
        print("launch sequences differ at", x, y)
        break
    cls = x[0] + (" <3M" if x[1] < 3_000_000 else " >=3M")
    tot[cls][0] += x[2]
    tot[cls][1] += y[2]
    if x[0] in ("extend", "shadow"):
        print(f"{x[0]:10s} {x[1]:10d} {x[2]:8.3f} | {y[2]:8.3f}  {y[2] - x[2]:+7.3f}")
for k, t in sorted(tot.items()):
    print(f"{k:20s} {t[0]:8.2f} {t[1]:8.2f} {t[1] - t[0]:+7.2f}")
PY
