"""Namespace mirror of ``trep.force`` (reference: trep/force.py)."""
from .dynamics import Force  # noqa: F401
