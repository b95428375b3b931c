"""paddle.fluid.layers stand-in.  ``sequence_mask`` is imported by parakeet/models/tacotron2.py:18 and used only by the
training-time forward (:741-745), which is out of scope."""


def sequence_mask(*a, **k):
    raise NotImplementedError("paddle.fluid.layers.sequence_mask: training path, not restated")
