"""TapeMeasure: the length of the broken line through the origins of a sequence of frames, its rate of change and their
config derivatives (reference: trep/tapemeasure.py:14-70, 125-195; _trep/tapemeasure.c).

Host-side numpy queries for modelling and checks (e.g. tendon lengths); sums of the two-frame segment formulas of
``element_queries`` (the same ones behind LinearSpring / LinearDamper).  Nothing here runs on the device.
"""
from . import element_queries as _eq


class _Segment(object):
    def __init__(self, system, frame1, frame2):
        self.system, self.frame1, self.frame2 = system, frame1, frame2


class TapeMeasure(object):
    def __init__(self, system, frames):
        self._system = system
        self._frames = tuple(system.get_frame(f) for f in frames)
        if any(f is None for f in self._frames) or len(self._frames) < 2:
            raise ValueError("a tape measure needs at least two frames of the system")
        self._segments = [_Segment(system, a, b) for a, b in zip(self._frames[:-1], self._frames[1:])]

    system = property(lambda self: self._system)
    frames = property(lambda self: self._frames)

    def _sum(self, fn, *configs):
        _eq.check_configs(*configs)
        total = 0.0
        for seg in self._segments:
            if all(seg.frame1.uses_config(q) or seg.frame2.uses_config(q) for q in configs):
                total += fn(seg, *configs)
        return total

    def length(self):
        return self._sum(_eq.length)

    def length_dq(self, q1):
        return self._sum(_eq.length_dq, q1)

    def length_dqdq(self, q1, q2):
        return self._sum(_eq.length_dqdq, q1, q2)

    def length_dqdqdq(self, q1, q2, q3):
        return self._sum(_eq.length_dqdqdq, q1, q2, q3)

    def velocity(self):
        return self._sum(_eq.velocity)

    def velocity_dq(self, q1):
        return self._sum(_eq.velocity_dq, q1)

    def velocity_dqdq(self, q1, q2):
        return self._sum(_eq.velocity_dqdq, q1, q2)

    def velocity_ddq(self, dq1):
        return self._sum(_eq.length_dq, dq1)

    def velocity_ddqdq(self, dq1, q2):
        return self._sum(_eq.length_dqdq, dq1, q2)

    # finite-difference checks (tapemeasure.py:125-195)
    def _check(self, kind, lower, upper, outer, delta, tolerance, verbose, name):
        test = self._system.test_derivative_dq if kind == "q" else self._system.test_derivative_ddq
        return all([test(lambda qs=qs: lower(*qs), lambda qn, qs=qs: upper(*(qs + (qn,))) if kind == "q" else upper(qn, *qs), delta, tolerance,
                         verbose=verbose, test_name='TapeMeasure.%s()' % name) for qs in _eq.pairs(self._system, outer)])

    def validate_length_dq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._check("q", self.length, self.length_dq, 0, delta, tolerance, verbose, "length_dq")

    def validate_length_dqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._check("q", self.length_dq, self.length_dqdq, 1, delta, tolerance, verbose, "length_dqdq")

    def validate_length_dqdqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._check("q", self.length_dqdq, self.length_dqdqdq, 2, delta, tolerance, verbose, "length_dqdqdq")

    def validate_velocity_dq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._check("q", self.velocity, self.velocity_dq, 0, delta, tolerance, verbose, "velocity_dq")

    def validate_velocity_dqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._check("q", self.velocity_dq, self.velocity_dqdq, 1, delta, tolerance, verbose, "velocity_dqdq")

    def validate_velocity_ddq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return self._check("dq", self.velocity, self.velocity_ddq, 0, delta, tolerance, verbose, "velocity_ddq")

    def validate_velocity_ddqdq(self, delta=1e-6, tolerance=1e-6, verbose=False):
        return all([self._system.test_derivative_dq(lambda d=d: self.velocity_ddq(d), lambda q2, d=d: self.velocity_ddqdq(d, q2), delta, tolerance,
                                                    verbose=verbose, test_name='TapeMeasure.velocity_ddqdq()') for d in self._system.configs])
