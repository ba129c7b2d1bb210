"""Namespace mirror of ``trep.forces`` (reference: trep/forces/__init__.py)."""
from .dynamics import Damping, ConfigForce, HybridWrench, SpatialWrench, BodyWrench, LinearDamper  # noqa: F401
