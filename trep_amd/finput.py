"""Namespace mirror of ``trep.finput`` (reference: trep/finput.py)."""
from .config import Input  # noqa: F401
