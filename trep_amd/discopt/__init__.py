"""Mirror of ``trep.discopt`` for the part on the MidpointVI hot path."""
from .dsystem import DSystem, BatchDSystem  # noqa: F401
