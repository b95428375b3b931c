#!/usr/bin/env python
"""Where a kernel's register spills sit: for every scratch load / store of the kernels matching a mangled-name substring,
the number of MFMAs and barriers that precede it (a reload between the MFMAs of the hot loop waits for every load in flight).
usage: python tools/spill_sites.py <file.s> <mangled substring> [...]"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
starts = [(i, l) for i, l in enumerate(s) if re.match(r"^_Z[^\s]*:", l)]
for pat in sys.argv[2:]:
    for idx, (i, l) in enumerate(starts):
        if pat not in l:
            continue
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(s)
        nm = nb = 0
        total = sum(1 for k in range(i, end) if "v_mfma" in s[k])
        print(l.split(":")[0], "mfma", total)
        for k in range(i, end):
            x = s[k]
            if "v_mfma" in x:
                nm += 1
            if "s_barrier" in x:
                nb += 1
            if "scratch_" in x:
                print("   line", k - i, "mfma", nm, "bar", nb, x.strip()[:80])
