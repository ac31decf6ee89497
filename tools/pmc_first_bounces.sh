#!/bin/bash
# Per-dispatch view of the extend / shade / shadow kernels over the first bounces of one 16 spp frame (the coherent camera rays
# against the incoherent bounces): VALU instructions per wave, lanes active per VALU instruction, share of wave time issuing
# VALU / waiting. Two --pmc passes (never mixed with trace domains).   tools/pmc_first_bounces.sh  ->  gpurun_out/pmc_first_bounces.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
args="--steps 1 --warmup 0 --spp 16 --no-cpu-baseline"
rm -rf gpurun_out/pmcfb_1 gpurun_out/pmcfb_2
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/pmcfb_1 --output-format csv -- python3 bench.py $args > gpurun_out/pmcfb_1.log 2>&1 || { echo "pass 1 failed"; tail -5 gpurun_out/pmcfb_1.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES -d gpurun_out/pmcfb_2 --output-format csv -- python3 bench.py $args > gpurun_out/pmcfb_2.log 2>&1 || { echo "pass 2 failed"; tail -5 gpurun_out/pmcfb_2.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/pmc_first_bounces.txt
import csv, glob, re
from collections import defaultdict
rows = defaultdict(dict)
for d in ("gpurun_out/pmcfb_1", "gpurun_out/pmcfb_2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if not k.startswith(("extend_kernel", "shade_kernel", "shadow_kernel", "nee_kernel")):
                continue
            rows[(int(r["Dispatch_Id"]), k)][r["Counter_Name"]] = float(r["Counter_Value"])
big = [(i, k, c) for (i, k), c in sorted(rows.items()) if c.get("SQ_INSTS_VALU", 0) > 3e8 and k.startswith(("extend", "shadow"))]
seen = defaultdict(int)
for i, k, c in big:
    seen[k] += 1
    if seen[k] > 14:
        continue
    w, wc = c["SQ_WAVES"], c.get("SQ_WAVE_CYCLES", 0)
    print(f"{k:44s} launch {seen[k]}  waves {w:9.0f}  VALU/wave {c.get('SQ_INSTS_VALU',0)/w:9.1f}  SALU/wave {c.get('SQ_INSTS_SALU',0)/w:8.1f}  "
          f"lanes/VALU {c.get('SQ_THREAD_CYCLES_VALU',0)/max(1.0,c.get('SQ_ACTIVE_INST_VALU',0)):5.1f}  VALU {c.get('SQ_ACTIVE_INST_VALU',0)/max(1.0,wc):5.1%}  WAIT_ANY {c.get('SQ_WAIT_ANY',0)/max(1.0,wc):5.1%}  "
          f"wave cycles/wave {4*wc/w:9.0f}")
PY
rm -rf gpurun_out/pmcfb_1 gpurun_out/pmcfb_2
