"""Exceptions shared by the host shell (reference: _trep.ConvergenceError, _trep.c:142)."""


class ConvergenceError(Exception):
    """Raised when the DEL Newton solve fails (too many iterations or singular Jacobian)."""
