#!/usr/bin/env python
"""A/B libraries for one-box measurements: the profile build's objects with ONE translation unit recompiled differently
(extra -D flags, or the file as it was at a git revision), linked to parakeet_amd/variants/<name>.so.  A GPU call copies a
variant over libpk_synth_prof.so and runs the same tool under PK_PROFILE_LIB=1 (the library carries the tree's hash, so the
loader accepts it; variants are measurement artefacts, git-ignored, never the product).
usage: python tools/build_variant.py <name> <file.hip> [--rev REV] [-DNAME=VALUE | -f... ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parakeet_amd import build as B  # noqa: E402

name, src = sys.argv[1], sys.argv[2]
rev = sys.argv[sys.argv.index("--rev") + 1] if "--rev" in sys.argv else None
flags = [a for a in sys.argv[3:] if a.startswith("-") and a != "--rev"]     # -DNAME=VALUE, -fno-slp-vectorize, -mllvm <x> ...
flags += [a for i, a in enumerate(sys.argv[3:]) if i > 0 and sys.argv[3:][i - 1] == "-mllvm"]
B.build(profile=True)                      # the other objects (*.prof.o) must be current
out_dir = os.path.join(ROOT, "parakeet_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
path = os.path.join(B.CSRC, src)
if rev:
    text = subprocess.run(["git", "show", f"{rev}:parakeet_amd/csrc/{src}"], cwd=ROOT, check=True, capture_output=True, text=True).stdout
    path = os.path.join(out_dir, f"_{name}_{src}")
    open(path, "wt").write(text)
obj = os.path.join(out_dir, f"{name}.o")
cmd = [B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", B.INCLUDE, "-I", B.CSRC, "-x", "hip", "-c", path,
       "-o", obj, f'-DPK_SOURCE_HASH="{B.source_hash()}"', "-DPK_PROFILE_BUILD=1"] + flags + B.FILE_FLAGS.get(src, [])
subprocess.run(cmd, check=True)
objs = [obj if s == src else os.path.join(B.CSRC, os.path.splitext(s)[0] + ".prof.o") for s in B.SOURCES]
lib = os.path.join(out_dir, name + ".so")
subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
assert B.library_hash(lib) == B.source_hash()
print(lib)
