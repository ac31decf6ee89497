#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc --cuda-device-only -S) for a register-spill store that is followed, in the same basic
block, by an instruction that re-enables lanes (s_or_b64 exec, exec, ...): the lanes that come back did not take part
in the store, so a later reload hands them whatever their scratch slot held before. (A mask that was itself saved AFTER
the store - `s_and_saveexec_b64 M` of an if / else nested below the store - only brings back lanes that stored: not
reported.) This placement was seen from the
ROCm 7.2 compiler in legacy_kernel (DESIGN.md "Toolchain notes"): the per-lane `diffuseDepth` of root hits was lost and
renders with photon maps depended on the previous contents of scratch memory.
usage: check_spill_exec.py file.s [...]   (exit status 1 if any site is found)"""
import re, sys

def widening_mov(s, recent):
    # `s_mov_b64 exec, sN` right after `s_and_b64 sN, <saved exec>, cond` is the narrowing half of an if: not a hazard
    m = re.match(r"s_mov_b64\s+exec,\s*(s\[\d+:\d+\])", s)
    if not m: return False
    return not any(r.startswith("s_and_b64 " + m.group(1) + ",") for r in recent[:-1])

def scan(path):
    found, func, pending, recent = [], None, [], []
    for ln, line in enumerate(open(path, errors="replace"), 1):
        s = line.strip()
        recent = (recent + [s])[-8:]
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not s.startswith("."):
            func, pending = m.group(1), []
            continue
        if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm")):
            pending = []
            continue
        if "Folded Spill" in s and s.startswith(("scratch_store", "buffer_store")):
            pending.append((ln, s, set()))
            continue
        m = re.match(r"s_and_saveexec_b64\s+(s\[\d+:\d+\])", s)
        if m:   # exec as of now (a subset of every pending store's) is kept in this mask
            for p in pending: p[2].add(m.group(1))
            continue
        if re.match(r"s_or_b64\s+exec,\s*exec", s) or re.match(r"s_or_saveexec|s_andn2_saveexec", s) or widening_mov(s, recent):
            masks = set(re.findall(r"s\[\d+:\d+\]", s))
            keep = []
            for p in pending:
                if masks and masks <= p[2]: keep.append(p)   # the lanes that come back all took part in the store
                else: found.append((func, p[0], p[1], ln, s))
            pending = keep
    return found

if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for func, l0, st, l1, ex in scan(p):
            bad += 1
            print(f"{p}:{l0}: in {func}: `{st}` precedes `{ex}` (line {l1}) in one block")
    print(f"{bad} suspicious spill site(s)")
    sys.exit(1 if bad else 0)
