"""Host-side mirrors added in round 6 that need no GPU: `parakeet_amd.amp.auto_cast` (paddle.amp.auto_cast of
examples/waveflow/synthesize.py:40), `parakeet_amd.utils.layer_tools` (parakeet/utils/layer_tools.py:40-46) and the
`from_pretrained` classmethod's presence and signature (waveflow.py:827-852)."""
import inspect
import threading

import pytest


def test_auto_cast_nests_and_is_per_thread():
    from parakeet_amd import amp
    assert not amp.enabled()
    with amp.auto_cast():
        assert amp.enabled()
        with amp.auto_cast(enable=False):          # paddle's signature: enable=False is a no-op context
            assert amp.enabled()
        with amp.auto_cast(custom_white_list={"conv2d"}, level="O1"):
            assert amp.enabled()
        seen = []
        t = threading.Thread(target=lambda: seen.append(amp.enabled()))
        t.start()
        t.join()
        assert seen == [False]                      # another thread is not inside this thread's context
    assert not amp.enabled()
    with pytest.raises(ZeroDivisionError):
        with amp.auto_cast():
            1 / 0
    assert not amp.enabled()                        # left cleanly on an exception


def test_layer_tools_and_from_pretrained_surface():
    from parakeet_amd.utils import layer_tools
    from parakeet_amd.waveflow import ConditionalWaveFlow

    class Model:
        calls = 0

        def set_state_dict(self, sd):
            pass

        def remove_weight_norm(self):
            Model.calls += 1
    layer_tools.recursively_remove_weight_norm(Model())
    assert Model.calls == 1

    class NoHook:
        def set_state_dict(self, sd):
            pass
    layer_tools.recursively_remove_weight_norm(NoHook())      # folded at packing time: nothing to do, no error
    with pytest.raises(TypeError):
        layer_tools.recursively_remove_weight_norm(object())
    sig = inspect.signature(ConditionalWaveFlow.from_pretrained)
    assert list(sig.parameters) == ["config", "checkpoint_path"]        # the reference's (cls, config, checkpoint_path)
    assert isinstance(inspect.getattr_static(ConditionalWaveFlow, "from_pretrained"), classmethod)
