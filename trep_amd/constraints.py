"""Namespace mirror of ``trep.constraints`` (reference: trep/constraints/__init__.py)."""
from .dynamics import Distance, PointToPoint1D, PointToPoint2D, PointToPoint3D, PointOnPlane  # noqa: F401
