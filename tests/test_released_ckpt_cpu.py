"""Consumers of tests/golden_paddle/ (written by ``PARAKEET_REAL_PADDLE=1 python tools/verify_with_paddle.py`` on a machine
that has PaddlePaddle): the oracle and the checkpoint reader against what Paddle itself computed and wrote.  Skipped while
the directory does not exist; tests/test_verify_paddle_cpu.py runs the same checks on a stand-in run of the script."""
import os

import pytest

import released_cases as rc

BASE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_paddle")
pytestmark = pytest.mark.skipif(not os.path.isdir(BASE), reason="tests/golden_paddle/ not generated (tools/verify_with_paddle.py)")


def test_archives_written_by_paddle_itself_are_read():
    rc.check_paddle_written(BASE)


def test_oracle_matches_paddle_on_released_checkpoints():
    files = rc.released_files(BASE)
    assert files
    for f in files:
        print(os.path.basename(f), rc.check_oracle_released(f, os.path.join(BASE, "released")))


def test_waveflow_oracle_matches_paddle_on_released_checkpoint():
    files = rc.waveflow_files(BASE)
    assert files
    for f in files:
        print(os.path.basename(f), rc.check_oracle_waveflow(f, os.path.join(BASE, "released")))
