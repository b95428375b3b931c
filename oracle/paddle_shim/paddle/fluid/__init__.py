"""paddle.fluid stand-in: only what the reference's inference modules import at module level."""
from . import layers  # noqa: F401
