"""Host-side helpers with the names of parakeet/modules/nets_utils.py:21-125.

Inside the engine these masks never exist as tensors -- a ragged batch is a row timeline whose per-row
utterance table plays the mask's role -- but recipe code that builds batches calls them, so they are kept
as small numpy functions with the reference's semantics (worked examples: nets_utils.py:36-42,71-75,119-123;
``FastSpeech2._source_mask`` fastspeech2.py:618-641)."""
import numpy as np


def _lengths(lengths):
    return [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]


def pad_list(xs, pad_value):
    """List of (T_i, *) arrays -> (B, Tmax, *) padded with pad_value."""
    xs = [np.asarray(x) for x in xs]
    out = np.full((len(xs), max(x.shape[0] for x in xs)) + xs[0].shape[1:], pad_value, dtype=xs[0].dtype)
    for i, x in enumerate(xs):
        out[i, :x.shape[0]] = x
    return out


def make_pad_mask(lengths, length_dim=-1):
    """(B,) lengths -> bool (B, Tmax), True on the padded part."""
    if length_dim == 0:
        raise ValueError("length_dim cannot be 0: {}".format(length_dim))
    lens = _lengths(lengths)
    return np.arange(max(lens))[None, :] >= np.asarray(lens)[:, None]


def make_non_pad_mask(lengths, length_dim=-1):
    return np.logical_not(make_pad_mask(lengths, length_dim))


def source_mask(ilens):
    """FastSpeech2._source_mask: (B,) -> bool (B, 1, Tmax) for self-attention."""
    return make_non_pad_mask(ilens)[:, None, :]
