#!/usr/bin/env python
"""For every kernel of a hipcc -S file: the tightest MFMA -> non-MFMA consumer distance, in issue slots (an s_nop N counts N + 1).
Round 5: a VALU read of an accumulator exactly the compiler's minimum (11 slots on gfx950 for v_mfma_f32_32x32x16_f16) after the
last MFMA of two interleaved dependent chains returned a stale value when a second wave of the SIMD had MFMAs in the pipe
(WaveFlow layer kernel, 64 channels, out projection; HISTORY 9.9).  This lists where else the margin is small.
usage: python tools/mfma_slack.py <file.s> [max_slack_to_report=24]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
LIMIT = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def regs(tok):
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def parse(l):
    m = re.match(r"^\t([a-z_0-9]+)\s*(.*?)(\s*;.*)?$", l)
    if not m or m.group(1).startswith("."):
        return None
    op = m.group(1)
    toks = [t.strip() for t in re.split(r",\s*|\s+", m.group(2)) if t.strip()]
    return op, toks


i = 0
while i < len(lines):
    if re.match(r"^_Z[^\s]*:", lines[i]) and i + 1 < len(lines):
        name = lines[i][:-1]
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel") and not re.match(r"^_Z[^\s]*:", lines[j]) and not lines[j].startswith(".Lfunc_end"):
            p = parse(lines[j])
            if p:
                body.append((j - i, p[0], p[1]))
            j += 1
        pending = {}   # vgpr -> (slots since the writing mfma issued, line of the mfma)
        worst = []
        for ln, op, toks in body:
            is_mfma = op.startswith(("v_mfma", "v_smfmac"))
            store = op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store"))
            reads = set()
            for t in (toks if store else toks[1:]):
                reads |= regs(t)
            if not is_mfma and pending:
                hit = [(pending[r][0], pending[r][1]) for r in reads if r in pending]
                if hit:
                    s, src = min(hit)
                    if s <= LIMIT:
                        worst.append((s, src, ln, op))
                    for r in list(pending):
                        if pending[r][1] == src or r in reads:
                            pass
                    for r in reads:
                        pending.pop(r, None)
            step = 1
            if op == "s_nop":
                step = int(toks[0]) + 1
            for r in pending:
                pending[r] = (pending[r][0] + step, pending[r][1])
            if is_mfma:
                for r in regs(toks[0]):
                    pending[r] = (0, ln)
            elif toks and not store:
                for r in regs(toks[0]):      # overwritten by something else: no longer an MFMA result
                    if r not in reads:
                        pending.pop(r, None)
            if op in ("s_barrier",):
                pass
        if worst:
            worst.sort()
            dem = name
            print(f"{dem[:110]}: {len(worst)} consumer(s) within {LIMIT} slots of the MFMA; tightest {worst[0][0]} (mfma line +{worst[0][1]}, consumer +{worst[0][2]} {worst[0][3]})")
        i = j
    else:
        i += 1
