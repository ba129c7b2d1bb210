"""trep_amd: MI355X-native batched MidpointVI engine behind the trep model API.

Host-side (Python) mirror of the part of ``trep``'s public surface that the
MidpointVI hot path needs: ``System``, ``Frame``, ``Config``,
``Input``, the frame-definition helpers, ``potentials.Gravity``,
``forces.Damping/ConfigForce``, ``constraints.Distance/PointToPoint*``,
``puppets.Puppet``, ``MidpointVI`` and ``discopt.DSystem``.  Numerical work is
done by ``libtrepamd.so`` (hand-written HIP for gfx950) through a ctypes C ABI
(include/trep_amd.h); there is no CPU fallback.
"""
from .frame import (Frame, FrameDef, WORLD, TX, TY, TZ, RX, RY, RZ, CONST_SE3,
                    tx, ty, tz, rx, ry, rz, const_se3, const_txyz)
from .config import Config, Input
from .system import System, save_trajectory, load_trajectory
from .dynamics import Potential, Force, Constraint
from . import potentials, forces, constraints, puppets, systems
from .errors import ConvergenceError
from .spline import Spline
from .tapemeasure import TapeMeasure
from .midpointvi import MidpointVI, BatchMidpointVI
from . import discopt

__all__ = ["Spline", "TapeMeasure", "System", "save_trajectory", "load_trajectory", "Frame", "Config", "Input", "Potential", "Force", "Constraint", "MidpointVI",
           "BatchMidpointVI", "ConvergenceError", "tx", "ty", "tz", "rx", "ry", "rz", "const_se3",
           "const_txyz", "WORLD", "TX", "TY", "TZ", "RX", "RY", "RZ", "CONST_SE3"]
