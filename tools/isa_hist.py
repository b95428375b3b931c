#!/usr/bin/env python
"""Instruction histogram of one kernel from hipcc -S output: every mnemonic with its count, grouped by unit (MFMA / VALU /
transcendental / LDS / global / scalar / waitcnt).  The WaveFlow layer kernel's tile is straight-line code (slab loop fully
unrolled), so static counts of the working copy are per-tile dynamic counts.
usage: python tools/isa_hist.py <file.s> <mangled substring> [first_label_regex]"""
import collections
import re
import sys

s = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r"^_Z[^\s]*:", l) and pat in l)
end = next((i for i in range(start + 1, len(s)) if s[i].startswith("\t.end_amdhsa_kernel") or re.match(r"^_Z[^\s]*:", s[i])), len(s))
hist = collections.Counter()
for l in s[start:end]:
    m = re.match(r"^\t([a-z_0-9]+)", l)
    if m and not m.group(1).startswith(("amdhsa", "p2align", "section", "globl", "type", "size", "text", "set")):
        hist[m.group(1)] += 1


def unit(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "transcendental (quarter rate)"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return "wait/barrier"
    return "scalar"


groups = collections.defaultdict(list)
for op, n in hist.items():
    groups[unit(op)].append((n, op))
for g in ("mfma", "valu", "transcendental (quarter rate)", "lds", "vmem", "wait/barrier", "scalar"):
    items = sorted(groups.get(g, []), reverse=True)
    print(f"{g}: {sum(n for n, _ in items)}")
    print("   " + ", ".join(f"{op} {n}" for n, op in items[:40]))
