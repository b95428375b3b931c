import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _emulated():
    """PK_EMU=1: run the -m gpu tests against the host emulation of the kernels (tools/hipemu; development aid for
    kernel logic when no GPU box is at hand -- never a parity claim, never set by the driver)."""
    return os.environ.get("PK_EMU") == "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp64 oracles are many small torch CPU ops.  On the GPU boxes (a few hundred logical CPUs) torch's default intra-op
    # pool makes them 10 - 20x slower than on 8 threads (thread wake-ups dominate): the two bench-size autoregressive tests
    # took 122 s + 105 s there against 17 s here.  Oracle results do not depend on the thread count.
    try:
        import torch
        torch.set_num_threads(min(8, os.cpu_count() or 8))
    except Exception:
        pass
    if _emulated():
        sys.path.insert(0, os.path.join(ROOT, "tools", "hipemu"))
        import harness
        harness.install()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# GPU tests of the variants added after the round's GPU minutes were spent: they have run under
# the host emulation only.  They are moved behind the tests that have run on an MI355X, so that under `-x` a failure among
# them cannot hide the results of the validated ones.  Remove an entry once its tests have passed on hardware.
FIRST_GPU_RUN_PENDING = (
    "spk_add", "spk_concat", "unscaled", "linear_in", "-r2-", "r2-f", "r3stop", "gst", "enc_postnorm", "enc_concat", "dec_postnorm",
    "dec_concat", "all_post_concat", "[global-", "test_global_condition", "test_speaker_embeddings", "test_reduction_factor",
    "test_style_tokens", "block_variants", "test_kv_only", "test_conv1d_cell", "test_vocoder_recipe_script", "test_mandarin_multispeaker_recipe")   # block_variants: FastSpeech2 post-norm / concat / r > 1


def pytest_collection_modifyitems(config, items):
    pending = [it for it in items if "gpu" in it.keywords and any(k in it.nodeid for k in FIRST_GPU_RUN_PENDING)]
    if pending:
        ids = {id(it) for it in pending}
        items[:] = [it for it in items if id(it) not in ids] + pending
    if _has_gpu() or _emulated():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
