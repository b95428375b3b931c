#!/usr/bin/env python
"""A variant library from HAND-EDITED device assembly of one translation unit (round 5: experiments that must not move any
instruction -- e.g. widening an existing s_nop or s_waitcnt, same encoding size).  The profile build's device code of <file.hip>
is emitted as assembly, passed through a Python edit function, assembled, linked and bundled, and the host object is compiled
against that device binary; the rest as tools/build_variant.py.
usage: python tools/asm_variant.py <name> <file.hip> <edit.py>     (edit.py defines edit(lines) -> lines; "none" = no edit)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parakeet_amd import build as B  # noqa: E402

name, src, edit_py = sys.argv[1], sys.argv[2], sys.argv[3]
B.build(profile=True)
out_dir = os.path.join(ROOT, "parakeet_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
llvm = os.path.join(os.path.dirname(os.path.realpath(B.hipcc())), "..", "lib", "llvm", "bin")
if not os.path.isdir(llvm):
    llvm = "/opt/rocm/lib/llvm/bin"
common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", B.INCLUDE, "-I", B.CSRC, "-x", "hip",
          os.path.join(B.CSRC, src), f'-DPK_SOURCE_HASH="{B.source_hash()}"', "-DPK_PROFILE_BUILD=1"] + B.FILE_FLAGS.get(src, []) + \
         [a for a in sys.argv[4:] if a.startswith("-D")]
base = os.path.join(out_dir, name)
asm = base + ".dev.s"
if not (os.path.exists(asm + ".orig") and "--reuse" in sys.argv):
    subprocess.run([B.hipcc(), "-S", "--cuda-device-only", "-o", asm + ".orig"] + common, check=True)
lines = open(asm + ".orig").read().split("\n")
if edit_py != "none":
    ns = {}
    exec(open(edit_py).read(), ns)
    lines = ns["edit"](lines)
open(asm, "w").write("\n".join(lines))
run = lambda *c: subprocess.run(list(c), check=True)
run(os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", base + ".dev.o")
run(os.path.join(llvm, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", base + ".dev.out", base + ".dev.o")
run(os.path.join(llvm, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
    "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + base + ".dev.out",
    "-output=" + base + ".hipfb")
obj = base + ".o"
run(B.hipcc(), "--cuda-host-only", "-c", "-o", obj, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", *common)
objs = [obj if s == src else os.path.join(B.CSRC, os.path.splitext(s)[0] + ".prof.o") for s in B.SOURCES]
lib = base + ".so"
run(B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs)
assert B.library_hash(lib) == B.source_hash()
print(lib)
