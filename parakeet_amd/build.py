"""Build libpk_synth.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library carries a hash of the sources it was built from (``pk_version()``: ``PK_SOURCE_HASH=<sha256>;``).
``build()`` rebuilds whenever that hash differs from the sources on disk; modification times are not
consulted (the .so is git-ignored but travels to the GPU box, where a stale binary may well be newer than
an edited source).  Without hipcc (never the case in the build image) a mismatching library is an error,
not something to run.

Two libraries come from the same sources:
  libpk_synth.so       the product.  Reads no environment variable: ``pk_prof_env`` (csrc/pk_common.h) is a constant
                       nullptr, the measurement / ablation switches and the kernels instantiated for them are not in it.
  libpk_synth_prof.so  ``build(profile=True)`` / ``python -m parakeet_amd.build --profile``: -DPK_PROFILE_BUILD=1, the
                       switches the tools/ scripts use (PK_WF_ABLATE, PK_PWG_ABLATE, PK_FFNP_ABLATE, PK_GEMM_TILE, ...;
                       several give WRONG results by design).  ``_capi`` loads it only under PK_PROFILE_LIB=1.
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libpk_synth.so")
LIB_PROF = os.path.join(HERE, "libpk_synth_prof.so")
# per-file flags.  wf_layer.hip: the layer kernel's slab loop must unroll completely (26 slabs at 128 channels; a ring slot is a
# register only under a compile-time index) -- beyond LLVM's default budget for `#pragma unroll`, where it silently keeps a
# loop and the operand ring moves to scratch memory
_UNROLL_ALL = ["-mllvm", "-pragma-unroll-threshold=2000000"]
# THE OP_SEL RULE: packed fp32 instructions whose low half reads a high source register drop a product now and
# then beside other waves' matrix instructions.  hipcc's SLP vectoriser made four of them each in k_sinusoid (ops.hip), k_ss_expand
# (speedyspeech.hip) and two in k_ar_dropout (tts.hip / taco2.hip) -- kernels of a few scalar lines that gain nothing from packing and
# may share a SIMD with another stream's GEMMs: these files are compiled without it (tools/pk_opsel_lint.py: none left anywhere)
_NO_SLP = ["-fno-slp-vectorize"]
FILE_FLAGS = {"wf_layer.hip": _UNROLL_ALL, "ffn_planes.hip": _UNROLL_ALL, "ops.hip": _NO_SLP, "speedyspeech.hip": _NO_SLP,
              "tts.hip": _NO_SLP, "taco2.hip": _NO_SLP}
ISA_DIR = os.path.join(CSRC, "_isa")   # the product build's device assembly, one .s per source (kept for tools/pk_opsel_lint.py and the ISA tools)
SOURCES = ["pk_ctx.cpp", "pwg.hip", "gemm.hip", "fs2.hip", "ffn_planes.hip", "waveflow.hip", "wf_layer.hip", "speedyspeech.hip", "tts.hip", "gst.hip", "taco2.hip", "rowgemm.hip", "mel.hip", "ops.hip"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def source_hash():
    """sha256 over every source the library is built from (names and contents, sorted)."""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")))
    for path in [os.path.join(CSRC, f) for f in files] + [os.path.join(INCLUDE, "pk_synth.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(repr(sorted(FILE_FLAGS.items())).encode())   # a flag change is a different library too
    h.update(compiler_id().encode())                       # ... and so is another compiler (the WaveFlow defect of round 5
    return h.hexdigest()                                   # came and went with instruction placement; a new hipcc must rebuild and re-run the gate tests)


_COMPILER_ID = None


def compiler_id():
    """First line of `hipcc --version` that names the HIP / clang version (cached); "unknown" without a compiler."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            out = subprocess.run([hipcc(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode()
            keep = [ln.strip() for ln in out.splitlines() if ln.startswith(("HIP version", "AMD clang version", "clang version"))]
            _COMPILER_ID = " | ".join(keep) or "unknown"
        except Exception:
            _COMPILER_ID = "unknown"
    return _COMPILER_ID


def file_hash(name):
    """sha256 of ONE source file under csrc/ (profiles/*_traffic.json name the kernel source their counters were collected on)."""
    with open(os.path.join(CSRC, name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def library_hash(path=LIB):
    """The PK_SOURCE_HASH string embedded in a built library (read from the file, nothing is loaded)."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(rb"PK_SOURCE_HASH=([0-9a-f]{64});", f.read())
    return m.group(1).decode() if m else None


def needs_build(profile=False):
    return library_hash(LIB_PROF if profile else LIB) != source_hash()


def _keep_isa():
    """-save-temps=obj leaves <name>-hip-amdgcn-amd-amdhsa-gfx950.{s,bc,hipi,o,out,...} and <name>-host-* next to the objects:
    the device .s moves to csrc/_isa/<name>.s, the rest goes."""
    os.makedirs(ISA_DIR, exist_ok=True)
    for f in os.listdir(CSRC):
        m = re.match(r"^(\w+)-hip-amdgcn-amd-amdhsa-gfx950\.s$", f)
        if m:
            os.replace(os.path.join(CSRC, f), os.path.join(ISA_DIR, m.group(1) + ".s"))
    for f in os.listdir(CSRC):
        if re.match(r"^\w+-(hip-amdgcn-amd-amdhsa-gfx950|host-x86_64-unknown-linux-gnu)\.", f) or f.endswith(".hipfb"):
            os.remove(os.path.join(CSRC, f))


def build(force=False, verbose=False, extra_flags=(), profile=False):
    lib = LIB_PROF if profile else LIB
    if not force and not needs_build(profile):
        return lib
    objs = []
    procs = []
    extra_flags = list(extra_flags) + [f'-DPK_SOURCE_HASH="{source_hash()}"', f"-DPK_PROFILE_BUILD={int(profile)}"]
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + (".prof.o" if profile else ".o"))
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
               "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj] + list(extra_flags) + FILE_FLAGS.get(src, [])
        if not profile and src.endswith(".hip"):
            cmd.append("-save-temps=obj")      # the device assembly is kept (below); same code as without
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    if not profile:
        _keep_isa()
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    if library_hash(lib) != source_hash():
        raise RuntimeError(f"{os.path.basename(lib)} does not carry the hash of the sources it was just built from")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, profile="--profile" in sys.argv,
                extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ()))
