#!/bin/bash
# HBM-side traffic of every kernel of the bench workload: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined
# with trace domains), summed per kernel over the launches of one frame. The file records the SHA-256 of the library it was
# measured on: bench.py reports these figures only for that build.
#   tools/pmc_traffic.sh [bench args]   ->  gpurun_out/kernel_traffic.json
args="--steps 1 --warmup 0 --no-cpu-baseline $*"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_traffic_$c
  timeout -s KILL 400 rocprofv3 --pmc $c -d gpurun_out/pmc_traffic_$c --output-format csv -- python3 bench.py $args > gpurun_out/pmc_traffic_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmc_traffic_$c.log; exit 1; }
done
python3 - "$args" <<'PY'
import csv, glob, hashlib, json, sys
out = {"command": "bench.py " + sys.argv[1], "library_sha256": hashlib.sha256(open("raytracer_amd/librtx_amd.so", "rb").read()).hexdigest(),
       "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; gfx950 tallies 128-B fetches at 64 B, so HBM bytes = 2 x FETCH + WRITE",
       "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_traffic_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            d = out["kernels"].setdefault(k, {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "dispatches": set()})
            d[c + "_KB"] += float(r["Counter_Value"])
            d["dispatches"].add(r["Dispatch_Id"])
for k, d in out["kernels"].items():
    d["dispatches"] = len(d["dispatches"])
json.dump(out, open("gpurun_out/kernel_traffic.json", "w"), indent=1)
for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["FETCH_SIZE_KB"])[:8]:
    print(f'{k:40s} launches {d["dispatches"]:5d}  FETCH {d["FETCH_SIZE_KB"]/1e6:9.3f} GB  WRITE {d["WRITE_SIZE_KB"]/1e6:9.3f} GB')
PY
rm -rf gpurun_out/pmc_traffic_FETCH_SIZE gpurun_out/pmc_traffic_WRITE_SIZE
