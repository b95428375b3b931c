#!/usr/bin/env python
"""Round 6 (HISTORY 10, DESIGN 4.3 "the exchange rule"): for every kernel of a hipcc -S file, the places where an LDS or vector-memory
instruction reads -- as data or as address -- a VGPR that a VALU instruction wrote FEWER than `min_slots` issue slots earlier.

Why: in the WaveFlow layer kernels hipcc put `ds_bpermute_b32 v4, v28, v2` directly behind `v_pk_fma_f32 v[2:3], ...`.  With
other waves' matrix instructions in the SIMD, lanes 48 - 63 of v2 (the last quarter of the wave's pass through the vector
ALU) were now and then read before the packed FMA had written them: the (logs, b) exchange between half waves returned the
sum without its last term for 16 positions of a tile -- the "cause (ii)" of round 5.  LLVM knows no hazard there (and none
is documented); one wait state cures it (0 wrong of 33 000 replays against 1 in 12).  This lists every such adjacency by
the kind of the writer (packed fp32, transcendental, other VALU) so that they can be padded where they matter.
usage: python tools/valu_to_mem_slack.py <file.s> [min_slots=1] [--all-writers]"""
import collections
import re
import sys

path = sys.argv[1]
MIN = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
ALL = "--all-writers" in sys.argv
lines = open(path).read().split("\n")


def regs(tok):
    tok = tok.strip()
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def parse(l):
    m = re.match(r"^\s*([a-z_0-9]+)\s*(.*?)(\s*;.*)?$", l)
    if not m or not l.startswith(("\t", " ")) or m.group(1).startswith("."):
        return None
    toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", m.group(2)) if t.strip()]
    return m.group(1), [t.split(" ")[0] for t in toks]


def kind(op):
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "packed fp32"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "transcendental"
    if op.startswith(("v_mfma", "v_smfmac")):
        return None                      # tools/mfma_slack.py
    return "other VALU" if op.startswith("v_") else None


MEM = ("ds_", "global_", "buffer_", "flat_", "scratch_")
kernel = None
recent = []          # (slots ago, op, set(written regs)) newest first
report = collections.defaultdict(list)
for n, l in enumerate(lines, 1):
    m = re.match(r"^(_Z[^\s:]*):", l)
    if m:
        kernel, recent = m.group(1), []
        continue
    p = parse(l)
    if not p or kernel is None:
        continue
    op, toks = p
    if op.startswith(MEM):
        store = "store" in op or "write" in op or "atomic" in op
        srcs = set()
        for t in (toks if store else toks[1:]):
            srcs |= regs(t)
        if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
            srcs = set()
            for t in toks[1:]:
                srcs |= regs(t)
        for age, wop, wr in recent:
            hit = wr & srcs
            if hit and age < MIN and (ALL or kind(wop) != "other VALU"):
                report[kernel].append((n, kind(wop), wop, op, sorted(hit)[0], age))
    step = int(toks[0]) + 1 if op == "s_nop" and toks and toks[0].isdigit() else 1
    recent = [(a + step, o, w) for a, o, w in recent if a + step <= 8]
    k = kind(op)
    if k and toks:
        recent.insert(0, (0, op, regs(toks[0])))
total = 0
for kname, ev in report.items():
    by = collections.Counter((e[1], e[2], e[3].split("_b")[0]) for e in ev)
    total += len(ev)
    print(f"{kname[:120]}: {len(ev)} site(s)")
    for (kd, wop, mop), c in by.most_common(6):
        print(f"    {c:4d} x {wop} -> {mop}  [{kd}]")
print(f"{path}: {total} LDS / memory instruction(s) read a register written fewer than {MIN} slot(s) earlier by a"
      f"{'ny' if ALL else ' packed-fp32 or transcendental'} VALU instruction")
