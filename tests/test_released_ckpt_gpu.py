"""The HIP engine, built from checkpoint DIRECTORIES of the released layout through parakeet_amd.checkpoint (the way
examples/fastspeech2/ljspeech/synthesize_e2e.py:45-83 builds the reference models), against what the reference's source
computed from the same files: (i) always -- the committed stand-in leg of tools/verify_with_paddle.py
(tests/golden/released_*standin.npz; the checkpoint directories are rebuilt here from seeds, byte for byte), (ii) when
tests/golden_paddle/ exists -- what real Paddle computed on the released LJSpeech checkpoints."""
import os
import sys

import pytest

import released_cases as rc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BASE = os.path.join(HERE, "golden_paddle")


@pytest.fixture(scope="module")
def standin_dirs(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import verify_with_paddle as vp          # make_standin_checkpoints needs neither Paddle nor the reference
    out = tmp_path_factory.mktemp("standin")
    vp.make_standin_checkpoints(str(out), quick=True)
    return os.path.join(str(out), "released")


def test_engine_from_stand_in_checkpoint_directories(standin_dirs):
    got = rc.check_engine_released(os.path.join(HERE, "golden", "released_standin.npz"), standin_dirs)
    print(got)


def test_waveflow_engine_from_stand_in_checkpoint(standin_dirs):
    print(rc.check_engine_waveflow(os.path.join(HERE, "golden", "released_waveflow_standin.npz"), standin_dirs))


@pytest.mark.skipif(not os.path.isdir(BASE), reason="tests/golden_paddle/ not generated (tools/verify_with_paddle.py)")
def test_engine_matches_paddle_on_released_checkpoints():
    for f in rc.released_files(BASE):
        print(os.path.basename(f), rc.check_engine_released(f, os.path.join(BASE, "released")))
    for f in rc.waveflow_files(BASE):
        print(os.path.basename(f), rc.check_engine_waveflow(f, os.path.join(BASE, "released")))
