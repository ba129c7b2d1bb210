"""Namespace mirror of ``trep.potential`` (reference: trep/potential.py)."""
from .dynamics import Potential  # noqa: F401
