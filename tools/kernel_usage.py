#!/usr/bin/env python
"""Register / scratch / LDS use of the kernels of one translation unit (cross-compiles here, no GPU needed).
usage: python tools/kernel_usage.py <file.hip> [regex on the demangled kernel name] [--profile] [-DNAME=VALUE ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parakeet_amd.build import CSRC, FILE_FLAGS, INCLUDE, hipcc  # noqa: E402

src = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ".")
cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-x", "hip", "-c",
       os.path.join(CSRC, src), "-o", "/tmp/_usage.o", f"-DPK_PROFILE_BUILD={int('--profile' in sys.argv)}",
       "-Rpass-analysis=kernel-resource-usage"] + FILE_FLAGS.get(src, []) + [a for a in sys.argv[2:] if a.startswith("-D")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
keys = (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if "error:" in line:
        print(line)
    for k, p in keys:
        m = re.search(p, line)
        if m and cur is not None:
            cur[k] = m.group(1)
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = n.replace("(anonymous namespace)::", "")
    if pat.search(n):
        print("%-60s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %s" % (
            n[:60], r.get("vgpr"), r.get("agpr"), r.get("spill"), r.get("scratch"), r.get("occ"), r.get("lds")))
