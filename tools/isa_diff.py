#!/usr/bin/env python
"""Is a kernel's gfx950 ISA the same in the working tree as at a git revision?  Used before claiming that a measured
default path is unchanged by an edit of its translation unit.

    python tools/isa_diff.py parakeet_amd/csrc/rowgemm.hip k_rowgemmILi16E k_rowgemmILi16ELb0E [rev=HEAD]
    python tools/isa_diff.py --all parakeet_amd/csrc/tts.hip c260a29        every kernel of the file, by symbol name

The two patterns select the (mangled) kernel in the old and the new source; labels are compared modulo their numbering."""
import difflib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src_text, name, tmp):
    csrc = os.path.join(ROOT, "parakeet_amd", "csrc")
    path = os.path.join(csrc, name)          # next to its headers
    with open(path, "w") as f:
        f.write(src_text)
    out = os.path.join(tmp, name + ".s")
    try:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                        "-I", csrc, "-x", "hip", "--offload-device-only", "-S", path, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
    finally:
        os.remove(path)
    return open(out).read()


def body(asm, pat):
    m = re.search(r"^(_Z\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M)
    if not m:
        raise SystemExit(f"no kernel matching {pat}")
    lines = [l.split(";")[0].strip() for l in m.group(2).splitlines()]
    return m.group(1), [re.sub(r"\.LBB\d+_", ".LBB_", l) for l in lines if l and not l.startswith(".")]


def all_kernels(asm):
    out = {}
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        lines = [l.split(";")[0].strip() for l in m.group(2).splitlines()]
        out[m.group(1)] = [re.sub(r"\.LBB\d+_", ".LBB_", l) for l in lines if l and not l.startswith(".")]
    return out


def main_all():
    src, rev = sys.argv[2], sys.argv[3]
    old_text = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{src}"], check=True, capture_output=True, text=True).stdout
    new_text = open(os.path.join(ROOT, src)).read()
    # the old source is compiled against the old headers: check the whole csrc + include directories out
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(f"git -C {ROOT} archive {rev} parakeet_amd/csrc include | tar -x -C {tmp}", shell=True, check=True)
        out_old = os.path.join(tmp, "old.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(tmp, "include"),
                        "-I", os.path.join(tmp, "parakeet_amd", "csrc"), "-x", "hip", "--offload-device-only", "-S",
                        os.path.join(tmp, src), "-o", out_old], check=True, stderr=subprocess.DEVNULL)
        old = all_kernels(open(out_old).read())
        new = all_kernels(device_asm(new_text, "_isa_new.hip", tmp))
    bad = 0
    for name, b in sorted(old.items()):
        if name not in new:
            print(f"GONE      {name}")
            bad += 1
        elif new[name] != b:
            print(f"CHANGED   {name}: {len(b)} -> {len(new[name])} instructions")
            bad += 1
        else:
            print(f"identical {name} ({len(b)})")
    for name in sorted(set(new) - set(old)):
        print(f"new       {name} ({len(new[name])})")
    return 1 if bad else 0


def main():
    if sys.argv[1] == "--all":
        return main_all()
    src, old_pat, new_pat = sys.argv[1:4]
    rev = sys.argv[4] if len(sys.argv) > 4 else "HEAD"
    old_text = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{src}"], check=True, capture_output=True, text=True).stdout
    new_text = open(os.path.join(ROOT, src)).read()
    with tempfile.TemporaryDirectory() as tmp:
        n1, b1 = body(device_asm(old_text, "_isa_old.hip", tmp), old_pat)
        n2, b2 = body(device_asm(new_text, "_isa_new.hip", tmp), new_pat)
    print(f"{rev}: {n1}: {len(b1)} instructions\ntree: {n2}: {len(b2)} instructions")
    if b1 == b2:
        print("identical (modulo label numbering)")
        return 0
    for i, l in enumerate(difflib.unified_diff(b1, b2, lineterm="")):
        if i < 60:
            print(l)
    return 1


if __name__ == "__main__":
    sys.exit(main())
