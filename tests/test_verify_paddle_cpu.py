"""tools/verify_with_paddle.py end to end in the build image, over the stand-in (PARAKEET_REAL_PADDLE=0 -- the code path a
machine with PaddlePaddle runs with the switch on): generators into a scratch directory, archives through the backend's
``paddle.save``, the synthesize_e2e.py loop on checkpoint directories of the released layout, the diff report; then the
consumers of tests/golden_paddle/ (tests/released_cases.py) on what it wrote.  Needs the reference checkout."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import released_cases as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PARAKEET_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "parakeet")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    out = tmp_path_factory.mktemp("golden_paddle")
    env = {k: v for k, v in os.environ.items() if k not in ("PARAKEET_REAL_PADDLE", "PARAKEET_GOLDEN_DIR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_with_paddle.py"), "--out", str(out), "--quick",
                        "--only", "make_golden_speedyspeech.py"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return str(out)


def test_report_and_zero_diff_against_the_committed_goldens(run):
    rep = json.load(open(os.path.join(run, "report.json")))
    assert rep["backend"] == "shim" and not rep["diff"]["infinite"] and rep["diff"]["worst"] == 0.0
    assert len(rep["diff"]["rows"]) >= 20 and set(rep["released"]) == {"standin", "waveflow_standin"}
    # same generator, same stand-in: the scratch copy is the committed file, tensor for tensor
    a, b = np.load(os.path.join(run, "speedyspeech_baker.npz")), np.load(os.path.join(ROOT, "tests", "golden", "speedyspeech_baker.npz"))
    assert sorted(a.files) == sorted(b.files) and all(np.array_equal(a[k], b[k]) for k in a.files)


def test_real_switch_refuses_the_stand_in():
    env = dict(os.environ, PARAKEET_REAL_PADDLE="1")
    code = "import sys; sys.path.insert(0, 'tools'); import ref_import; ref_import.setup()"
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and ("No module named 'paddle'" in r.stderr or "paddle" in r.stderr)   # no Paddle here: must not fall back


def test_consumers_on_the_stand_in_run(run):
    rc.check_paddle_written(run)
    files = rc.released_files(run)
    assert [os.path.basename(f) for f in files] == ["released_standin.npz"]
    worst = rc.check_oracle_released(files[0], os.path.join(run, "released"))
    assert worst["mel"] < 2e-5 and worst["wav"] < 2e-5
    wf = rc.waveflow_files(run)
    assert len(wf) == 1 and rc.check_oracle_waveflow(wf[0], os.path.join(run, "released")) < 1e-5


def test_committed_stand_in_fixture_is_what_the_script_writes(run):
    """tests/golden/released_standin.npz + released_waveflow_standin.npz (consumed on the GPU by
    tests/test_released_ckpt_gpu.py) = this script's --quick stand-in leg."""
    for name in ("released_standin.npz", "released_waveflow_standin.npz"):
        a, b = np.load(os.path.join(run, name)), np.load(os.path.join(ROOT, "tests", "golden", name))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (name, k)


def test_real_paddle_code_paths_over_a_disguised_stand_in(tmp_path):
    """The branches only PARAKEET_REAL_PADDLE=1 takes -- the stand-in kept OFF sys.path, ``F.dropout`` replaced for the prenet's
    always-on dropout, ONE reading of padding="same" named after the oracle reading it equals, archives through ``paddle.save``,
    the typeguard stub -- cannot meet Paddle here; they run against the stand-in under another path, which ref_import cannot tell
    from an installed package.  Everything it writes must equal what the stand-in leg writes."""
    fake = tmp_path / "site"
    fake.mkdir()
    os.symlink(os.path.join(ROOT, "oracle", "paddle_shim", "paddle"), fake / "paddle")
    out = tmp_path / "golden_paddle"
    env = {k: v for k, v in os.environ.items() if k != "PARAKEET_GOLDEN_DIR"}
    env.update(PARAKEET_REAL_PADDLE="1", PYTHONPATH=str(fake) + os.pathsep + env.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_with_paddle.py"), "--out", str(out), "--quick",
                        "--only", "make_golden_speedyspeech.py,make_golden_ar.py"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "backend: paddle" in r.stdout
    rep = json.load(open(out / "report.json"))
    assert rep["backend"] == "paddle" and not rep["diff"]["infinite"] and rep["diff"]["worst"] == 0.0
    gold = os.path.join(ROOT, "tests", "golden")
    # the autoregressive models: the dropout stream injected through the replaced F.dropout gives the stand-in's vectors
    for name in ("transformer_tts.npz", "tacotron2.npz"):
        a, b = np.load(out / name), np.load(os.path.join(gold, name))
        assert sorted(a.files) == sorted(b.files) and all(np.array_equal(a[k], b[k]) for k in a.files), name
    # SpeedySpeech: one reading, recognised as the dilation-resetting one (the disguised stand-in's default), stored under its tag
    a, b = np.load(out / "speedyspeech_baker.npz"), np.load(os.path.join(gold, "speedyspeech_baker.npz"))
    assert str(a["paddle_same_padding_reading"]) == "rd" and not any(k.startswith(("dil_", "real_mel")) for k in a.files)
    assert all(np.array_equal(a[k], b[k]) for k in a.files if k.startswith("rd_"))
    # the released-layout leg and the archives: same numbers as the stand-in leg's committed fixtures
    for name in ("released_standin.npz", "released_waveflow_standin.npz"):
        a, b = np.load(out / name), np.load(os.path.join(gold, name))
        assert all(np.array_equal(a[k], b[k]) for k in a.files), name
    rc.check_paddle_written(str(out))
    assert rc.check_oracle_released(str(out / "released_standin.npz"), str(out / "released"))["mel"] < 2e-5
