"""Namespace mirror of ``trep.constraint`` (reference: trep/constraint.py)."""
from .dynamics import Constraint  # noqa: F401
