"""Rules the compiled device code has to obey, checked on the assembly parakeet_amd/build.py keeps (csrc/_isa/*.s; CPU only).

THE OP_SEL RULE: no packed fp32 instruction whose low half reads a high source register
(`v_pk_*_f32 ... op_sel:[..1..]`) -- on the MI355X such an instruction drops its product now and then when another wave of the
SIMD runs matrix instructions; it was the WaveFlow layer kernel's sporadic wrong tiles of round 5."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "parakeet_amd", "csrc", "_isa")


def test_no_packed_fp32_instruction_takes_its_low_half_from_a_high_register():
    from parakeet_amd import build as B
    files = sorted(glob.glob(os.path.join(ISA, "*.s")))
    if not files or B.needs_build():
        pytest.skip("no current build with its assembly in this tree (python -m parakeet_amd.build)")
    assert {os.path.basename(f)[:-2] + ".hip" for f in files} == {s for s in B.SOURCES if s.endswith(".hip")}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pk_opsel_lint
    bad = {os.path.basename(f): {k: len(v) for k, v in pk_opsel_lint.lint(f).items()} for f in files}
    bad = {f: v for f, v in bad.items() if v}
    assert not bad, f"the op_sel rule is violated: {bad}"
    # ... and the lint does find the form where it is (the round-5 code of the sums: PK_WF_SCALAR_SUMS=0 compiles to it)
    demo = ("_Zk:\n\tv_pk_fma_f32 v[2:3], v[16:17], v[26:27], v[2:3] op_sel:[0,1,0]\n"
            "\tv_pk_fma_f32 v[2:3], v[14:15], v[26:27], v[2:3] op_sel_hi:[1,0,1]\n\tv_pk_mul_f32 v[2:3], s[10:11], v[8:9] op_sel:[1,0]\n")
    p = os.path.join(ISA, "..", "_lint_demo.s")
    with open(p, "wt") as f:
        f.write(demo)
    try:
        assert {k: len(v) for k, v in pk_opsel_lint.lint(p).items()} == {"_Zk": 2}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pk_opsel_lint.py"), p], capture_output=True, text=True)
        assert r.returncode == 1 and "2 packed fp32" in r.stdout
    finally:
        os.remove(p)
