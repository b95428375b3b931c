"""Host-side mirrors of the ``parakeet.utils`` helpers the synthesis recipes import (``layer_tools``); checkpoint reading
lives in ``parakeet_amd.checkpoint`` (parakeet/utils/checkpoint.py's role)."""
from . import layer_tools  # noqa: F401
