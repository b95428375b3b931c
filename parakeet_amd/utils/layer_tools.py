"""``parakeet.utils.layer_tools`` for the synthesis recipes (parakeet/utils/layer_tools.py:40-46).

``recursively_remove_weight_norm(model)`` is what examples/waveflow/synthesize.py:32 and the Tacotron2 / WaveFlow
recipes call after ``from_pretrained``.  The engine models have no sublayers and no hooks: ``set_state_dict`` accepts the
``weight_g`` / ``weight_v`` pairs of a weight-normalised checkpoint and the packing step (``pk_*_finalize``) folds
``w = g * v / ||v||`` per output channel, exactly once -- so there is nothing left to remove.  Models that mirror an
explicit ``remove_weight_norm()`` of the reference (``PWGGenerator``) get that call; everything else is a checked no-op.
"""

__all__ = ["recursively_remove_weight_norm"]


def recursively_remove_weight_norm(layer):
    """layer_tools.py:40-46: try ``remove_weight_norm`` on every sublayer, ignore layers without the hook."""
    if not hasattr(layer, "set_state_dict"):
        raise TypeError(f"recursively_remove_weight_norm: {type(layer).__name__} is not an engine model")
    fn = getattr(layer, "remove_weight_norm", None)
    if callable(fn):
        fn()
