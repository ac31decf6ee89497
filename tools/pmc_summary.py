"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and prints derived ratios (tools/pmc.sh)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

# --json FILE: the derived ratios per kernel as JSON as well, with the SHA-256 of the library they were measured on
json_out = None
if "--json" in sys.argv:
    k = sys.argv.index("--json")
    json_out = sys.argv[k + 1]
    del sys.argv[k:k + 2]
summary = {}
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_BUSY_CYCLES", 0)):
    c = tot[k]
    if c.get("SQ_WAVES", 0) < 1000:
        continue
    w = c["SQ_WAVES"]
    print(f"== {k}: dispatches {len(calls[k]['SQ_WAVES'])} waves {w:.3e}")
    g = lambda n: c.get(n, 0.0)
    print(f"   per wave: VALU {g('SQ_INSTS_VALU')/w:8.1f} SALU {g('SQ_INSTS_SALU')/w:8.1f} VMEM_RD {g('SQ_INSTS_VMEM_RD')/w:7.1f} "
          f"VMEM_WR {g('SQ_INSTS_VMEM_WR')/w:6.1f} SMEM {g('SQ_INSTS_SMEM')/w:6.1f} LDS {g('SQ_INSTS_LDS')/w:7.1f} FLAT {g('SQ_INSTS_FLAT')/w:6.1f}")
    wc = g("SQ_WAVE_CYCLES")
    J = summary.setdefault(k, {"waves": w, "valu_insts_per_wave": g("SQ_INSTS_VALU") / w})
    if wc:
        lanes = g("SQ_THREAD_CYCLES_VALU") / max(1.0, g("SQ_ACTIVE_INST_VALU")) 
        J.update({"valu_issue_frac_of_wave_time": g("SQ_ACTIVE_INST_VALU") / wc, "active_lanes_per_valu": lanes, "wait_any_frac": g("SQ_WAIT_ANY") / wc,
                  "wave_cycles_per_wave": 4 * wc / w})
        print(f"   wave cycles/wave {4*wc/w:10.0f}  of wave time: VALU {g('SQ_ACTIVE_INST_VALU')/wc:5.1%} VMEM {g('SQ_ACTIVE_INST_VMEM')/wc:5.1%} "
              f"SCA {g('SQ_ACTIVE_INST_SCA')/wc:5.1%} WAIT_INST_ANY {g('SQ_WAIT_INST_ANY')/wc:5.1%} WAIT_ANY {g('SQ_WAIT_ANY')/wc:5.1%}  "
              f"active lanes/VALU {lanes:4.1f}")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        a = g("TCP_TOTAL_CACHE_ACCESSES_sum")
        print(f"   TCP accesses/wave {a/w:8.1f}  TCP->TCC rd/wave {g('TCP_TCC_READ_REQ_sum')/w:8.1f} wr/wave {g('TCP_TCC_WRITE_REQ_sum')/w:7.1f}  "
              f"L1 miss ratio {g('TCP_TCC_READ_REQ_sum')/a:5.1%}  TA_BUSY {g('TA_BUSY_sum'):.3e} TA_ADDR_STALL {g('TA_ADDR_STALLED_BY_TC_CYCLES_sum'):.3e} "
              f"TCP_PENDING_STALL {g('TCP_PENDING_STALL_CYCLES_sum'):.3e}")
    if g("TCC_HIT_sum") or g("TCC_MISS_sum"):
        h, m = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        rd, rd32 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum")
        J.update({"l2_hit": h / (h + m), "l2_requests_per_wave": (h + m) / w})
        print(f"   L2 hit {h/(h+m):5.1%}  req/wave {(h+m)/w:8.1f}  EA read bytes/wave {((rd-rd32)*64+rd32*32)/w:10.0f}")

if json_out:
    import hashlib
    import json
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "librtx_amd.so")
    json.dump({"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
               "what": "rocprofv3 --pmc passes (tools/pmc.sh): what holds each kernel below its roofline - lanes active per VALU instruction "
                       "(divergence), share of a wave's lifetime spent issuing VALU work / waiting on memory, L2 hit rate",
               "kernels": summary}, open(json_out, "w"), indent=1)
