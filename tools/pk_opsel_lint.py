#!/usr/bin/env python
"""THE OP_SEL RULE (round 6, DESIGN 4.3 / HISTORY 10): list every packed fp32 vector instruction (v_pk_fma_f32, v_pk_mul_f32,
v_pk_add_f32) whose LOW half reads a HIGH source register -- `op_sel:[..1..]` -- per kernel of hipcc -S files.

On the MI355X such an instruction now and then drops its product in the low half for lanes 48 - 63 when another wave of
the SIMD executes matrix instructions (WaveFlow layer kernel, round 5's "cause (ii)"; reproduced in isolation by
tools/micro/mfma_chain_hazard.hip `b`: 1 in 3e7 per instruction with two or three waves per SIMD; the same arithmetic by
scalar FMAs, or by packed FMAs with `op_sel_hi` only -- the HIGH half from a LOW register, what a broadcast compiles to --
never failed in 4e9).  hipcc's SLP vectoriser produces the form from scalar source (`pl += w0 z; pb += w1 z; ...`); it is
avoided by keeping such sums scalar (an asm statement with the two sums as separate operands, wf_layer.hip) or by
-fno-slp-vectorize for a whole file (parakeet_amd/build.py FILE_FLAGS).
usage: python tools/pk_opsel_lint.py <file.s> [...]      (exit status 1 if any kernel has one)
       python tools/pk_opsel_lint.py --built             (the .s files parakeet_amd/build.py keeps under csrc/_isa/)"""
import glob
import os
import re
import sys

PAT = re.compile(r"^\s*(v_pk_[a-z0-9]+_f32)\b.*\bop_sel:\[[01,]*1[01,]*\]")


def lint(path):
    """{kernel: [(line, instruction text)]} of one .s file."""
    out, kernel = {}, None
    with open(path) as f:
        for n, line in enumerate(f, 1):
            m = re.match(r"^(_Z[^\s:]*|[A-Za-z_][\w.$]*):\s*(;.*)?$", line)
            if m and not line.startswith((".", " ", "\t")):
                kernel = m.group(1)
                continue
            if PAT.match(line):
                out.setdefault(kernel or "?", []).append((n, line.strip()))
    return out


def main(argv):
    files = argv
    if argv == ["--built"]:
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        files = sorted(glob.glob(os.path.join(here, "parakeet_amd", "csrc", "_isa", "*.s")))
        if not files:
            print("no csrc/_isa/*.s: run parakeet_amd.build first")
            return 2
    total = 0
    for path in files:
        hits = lint(path)
        n = sum(len(v) for v in hits.values())
        total += n
        print(f"{os.path.basename(path)}: {n} packed fp32 instruction(s) with a low half from a high register in {len(hits)} kernel(s)")
        for k, v in list(hits.items())[:8]:
            print(f"    {k[:100]}: {len(v)}, e.g. line {v[0][0]}: {v[0][1]}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
