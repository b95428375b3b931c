"""Build tools/hipemu/_build/libpk_synth_emu.so: the engine's .hip sources compiled as host C++ against the
stand-in <hip/hip_runtime.h> of this directory (DEVELOPMENT ONLY -- see that header).

Two source rewrites happen on the way (copies under _build/src, the originals are untouched):
  * ``extern __shared__ float x[];``  ->  a pointer to the emulator's dynamic-LDS block;
  * the one inline-assembly idiom of the code base (v_fma_mix_f32 d, h[sel], -1.0, x) -> hipemu::fma_mix_sub;
  * a comment line starting with ``// [wave-lds-exchange]`` -> hipemu::wave_sync(): the places where lanes of one wave
    exchange data through LDS without a barrier (legal on the hardware, which runs a wave's LDS instructions in order for
    all lanes; the fibers of the emulation need the rendezvous).
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "parakeet_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libpk_synth_emu.so")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

_DYN_LDS = re.compile(r"extern\s+__shared__\s+(\w+)\s+(\w+)\[\];")
_WAVE_LDS = re.compile(r"^(\s*)// \[wave-lds-exchange\]", re.M)
_STATIC_LDS = re.compile(r"^(\s*)(__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?[\w:<>]+\s+(\w+)(?:\[[^\];]+\])+;)", re.M)
_FMA_MIX = re.compile(r'asm\("v_fma_mix_f32 %0, %1, -1\.0, %2 op_sel:\[(\d),0,0\] op_sel_hi:\[1,0,0\]"\s*:\s*"=v"\((\w+)\)\s*:\s*'
                      r'"v"\((\w+)\),\s*"v"\(([^;]+)\)\);')


_OPAQUE = re.compile(r'asm volatile\(""\s*:\s*"\+[sv]"\(([\w\[\]]+)\)(?:\s*,\s*"\+[sv]"\([\w\[\]]+\))*\);')   # "the compiler may not reason about this value" (one or more operands): nothing to emulate
_NOP = re.compile(r'asm volatile\("s_nop \d+"\);')                               # issue-slot padding (wf_layer.hip, round 5): nothing to emulate


def rewrite(text):
    text = _DYN_LDS.sub(r"\1* \2 = (\1*)hipemu::dynamic_lds();", text)
    text = _FMA_MIX.sub(r"\2 = hipemu::fma_mix_sub(\3, \1, \4);", text)
    text = _OPAQUE.sub(";", text)
    text = _NOP.sub(";", text)
    text = _WAVE_LDS.sub(r"\1hipemu::wave_sync(); //", text)
    # HIPEMU_POISON: LDS is not initialised on the hardware; the first work-item of a workgroup fills every __shared__ array
    # with NaN bytes before anything runs, so that a read-before-write cannot pass on what the previous workgroup left behind
    text = _STATIC_LDS.sub(r"\1\2 hipemu::poison_lds(\3, sizeof(\3));", text)
    if "asm(" in text or "asm volatile" in text:
        raise RuntimeError("inline assembly the emulator has no rewrite for")
    return text


def sources():
    sys.path.insert(0, ROOT)
    from parakeet_amd.build import SOURCES
    return [os.path.join(CSRC, s) for s in SOURCES]


def build(verbose=False, opt="-O1"):
    os.makedirs(os.path.join(OUT, "src"), exist_ok=True)
    flags = [CXX, "-std=c++17", opt, "-g", "-fPIC", "-fno-strict-aliasing", "-Wno-unknown-attributes", "-Wno-unused-value",
             "-Wno-pass-failed", "-I" + HERE, "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]
    jobs = []
    stamp_deps = [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    stamp_deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    stamp_deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    newest_dep = max(os.path.getmtime(p) for p in stamp_deps)
    objs = []
    for src in sources() + [os.path.join(HERE, "hipemu.cpp")]:
        base = os.path.splitext(os.path.basename(src))[0]
        cpp = os.path.join(OUT, "src", base + ".cpp")
        obj = os.path.join(OUT, base + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(newest_dep, os.path.getmtime(src)):
            continue
        with open(src) as f:
            text = f.read()
        with open(cpp, "w") as f:
            f.write(rewrite(text) if src.endswith(".hip") else text)
        jobs.append(flags + ["-c", cpp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("hipemu build failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([CXX, "-shared", "-o", LIB] + objs + ["-lm"])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
