"""CPU-side checks of the boundary: the library loads, exports every symbol that
include/pk_synth.h declares, and fails loudly without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pk_synth.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", text)))


def test_build_and_exports():
    import __graft_entry__ as ge
    ge.build()
    from parakeet_amd import _capi
    lib = _capi.lib()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libpk_synth.so does not export {s}"
    assert b"gfx950" in lib.pk_version()


def test_product_library_carries_no_measurement_switches():
    """the ablation / measurement switches (several give wrong results by design) exist in the profile build
    only.  The product library imports no getenv and contains none of the switch names; behaviour a caller may choose goes
    through pk_*_set_math / pk_*_set_option."""
    import subprocess
    from parakeet_amd import build as b
    b.build()
    blob = open(b.LIB, "rb").read()
    for name in (b"ABLATE", b"PK_WF_ACTIVE", b"PK_FS2_ATTN_WAVES", b"PK_GEMM_TILE", b"PK_PWG_PLANES", b"PK_FS2_FFN_PLANES",
                 b"PK_FFNP_VARIANT", b"PK_TTS_KV_PREFIX", b"_MATH="):
        assert name not in blob, name
    assert not re.search(rb"PK_[A-Z0-9]+_MATH\0", blob)
    nm = subprocess.run(["nm", "-D", "--undefined-only", b.LIB], capture_output=True, text=True).stdout
    assert "getenv" not in nm
    assert b"PK_PROFILE_BUILD=0;" in blob


def test_binding_covers_header():
    from parakeet_amd import _capi
    bound = set(_capi._declare(_capi.lib()).keys())
    assert set(_declared_symbols()) == bound


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from parakeet_amd import _capi
    lib = _capi.lib()
    h = C.c_void_p()
    rc = lib.pk_ctx_create(0, C.byref(h))
    assert rc == -4 and b"no HIP device" in lib.pk_last_error()
    from parakeet_amd.parallel_wavegan import PWGGenerator
    with pytest.raises(RuntimeError):
        PWGGenerator()


def test_struct_sizes_match_header():
    from parakeet_amd import _capi
    assert C.sizeof(_capi.PwgCfg) == 4 * (11 + 8 + 1)
    assert C.sizeof(_capi.Fs2Cfg) == 4 * 37
    assert C.sizeof(_capi.WfCfg) == 4 * (1 + 4 + 7)
    assert C.sizeof(_capi.TtsCfg) == 4 * (29 + 7 + 8)   # pk_tts_cfg: 29 int32 fields + 7 + 8 for the style encoder
    assert C.sizeof(_capi.TacoCfg) == 4 * 18 + 4     # pk_taco_cfg: 18 int32 fields + float p_prenet_dropout


def test_smoke_config_subset_is_valid():
    """__graft_entry__.smoke() builds the oracle's config from its own; optional oracle keys must not break it."""
    import __graft_entry__ as g
    import inspect
    src = inspect.getsource(g.smoke)
    assert "if k in fcfg" in src


def test_build_rebuilds_on_source_hash_mismatch_not_on_mtime(tmp_path, monkeypatch):
    """a stale-but-newer .so must not be used.  The library embeds the sha256 of its
    sources; needs_build() compares hashes, file times are irrelevant."""
    import shutil
    from parakeet_amd import build as b
    assert b.library_hash() == b.source_hash(), "in-tree library is stale: run python -m parakeet_amd.build"
    assert not b.needs_build()
    # a copy of the tree's sources with one edited file -> different hash -> rebuild required, although the
    # library file is newer than every source
    csrc = tmp_path / "csrc"
    shutil.copytree(b.CSRC, csrc, ignore=shutil.ignore_patterns("*.o"))
    with open(csrc / "ops.hip", "a") as f:
        f.write("\n// edited\n")
    os.utime(csrc / "ops.hip", (0, 0))
    monkeypatch.setattr(b, "CSRC", str(csrc))
    assert b.source_hash() != b.library_hash()
    assert b.needs_build()
    # and the loader refuses a library whose hash differs from the tree
    from parakeet_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    import pytest
    with pytest.raises(RuntimeError, match="built from other sources"):
        _capi.lib()


def test_library_only_install_and_stale_override(tmp_path, monkeypatch):
    """an install that ships libpk_synth.so without csrc/ (nothing to hash) loads the library instead of
    raising FileNotFoundError, and PK_ALLOW_STALE_LIB is honoured before any hashing."""
    import shutil
    from parakeet_amd import _capi
    from parakeet_amd import build as b
    monkeypatch.setattr(b, "CSRC", str(tmp_path / "no_such_csrc"))
    monkeypatch.setattr(_capi, "_lib", None)
    assert _capi.lib().pk_version()                          # no source tree: loads
    csrc = tmp_path / "csrc"
    shutil.copytree(os.path.join(os.path.dirname(b.LIB), "csrc"), csrc, ignore=shutil.ignore_patterns("*.o"))
    with open(csrc / "ops.hip", "a") as f:
        f.write("\n// edited\n")
    monkeypatch.setattr(b, "CSRC", str(csrc))
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setenv("PK_ALLOW_STALE_LIB", "1")
    assert _capi.lib().pk_version()                          # mismatch, but explicitly allowed


def test_concurrent_library_loads_do_not_race():
    """N ranks of one node start at once (torch.distributed.run), each hashing the sources, checking the
    library's embedded hash and dlopen-ing it.  Four processes doing exactly that concurrently must all succeed, see the same
    hash, and none may rebuild or rewrite the library (its mtime and size are unchanged)."""
    import subprocess
    import sys
    from parakeet_amd import build as b
    b.build()
    st0 = os.stat(b.LIB)
    code = ("import sys; sys.path.insert(0, %r); from parakeet_amd import _capi, build as b; lib = _capi.lib(); "
            "assert not b.needs_build(); print(lib.pk_version().decode())") % ROOT
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(4)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e.decode()[-500:] for _, e in outs]
    versions = {o.decode().strip() for o, _ in outs}
    assert len(versions) == 1 and b.source_hash() in versions.pop()
    st1 = os.stat(b.LIB)
    assert (st0.st_mtime_ns, st0.st_size) == (st1.st_mtime_ns, st1.st_size)
