"""``paddle.amp.auto_cast()`` for the one recipe that synthesises under it (examples/waveflow/synthesize.py:40).

Under Paddle's O1 auto-cast the white-listed ops -- conv2d and matmul, i.e. every contraction of WaveFlow -- take fp16
operands and accumulate in fp32; everything else stays fp32.  The engine's name for that is ``set_math("f16")`` (one fp16
MFMA per product, the layer inputs still stored as 22-bit pairs: DESIGN.md 3).  Inside ``with amp.auto_cast():`` every
``ConditionalWaveFlow`` call runs in that mode and returns to the model's own math afterwards; models whose recipes never
run under auto-cast (FastSpeech2, Parallel WaveGAN, ...) ignore it.
"""
import contextlib
import threading

_state = threading.local()


def enabled():
    return getattr(_state, "depth", 0) > 0


@contextlib.contextmanager
def auto_cast(enable=True, **_unused):
    """Same call shape as ``paddle.amp.auto_cast(enable=True, custom_white_list=None, custom_black_list=None, level='O1')``;
    the lists and the level are accepted and ignored (the engine has one reduced-precision mode)."""
    if not enable:
        yield
        return
    _state.depth = getattr(_state, "depth", 0) + 1
    try:
        yield
    finally:
        _state.depth -= 1
