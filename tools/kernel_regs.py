#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in the assembly files `make -C raytracer_amd/csrc spillcheck` leaves in
/tmp/rtx_amd_asm (reads the amdhsa metadata at the end of each .s file)."""
import re, subprocess, sys, glob
files = sys.argv[1:] or sorted(glob.glob("/tmp/rtx_amd_asm/*.s"))
for f in files:
    txt = open(f).read()
    meta = txt[txt.rfind("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.", meta)[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except Exception:
            pass
        print(f"{name[:60]:60s} vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
