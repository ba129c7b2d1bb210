"""Namespace mirror of ``trep.potentials`` (reference: trep/potentials/__init__.py)."""
from .dynamics import Gravity, ConfigSpring, NonlinearConfigSpring, LinearSpring  # noqa: F401
