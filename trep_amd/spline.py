"""Piecewise-quintic interpolating curve y = f(x) used by ``NonlinearConfigSpring`` (reference: trep/spline.py:6-259,
_trep/spline.c:8-57).

``Spline(data)`` takes points ``(x, y)``, ``(x, y, dy)`` or ``(x, y, dy, ddy)`` (``None`` = not prescribed), x increasing.
Between two neighbouring points the curve is ``y_i + e t + d t^2 + c t^3 + b t^4 + a t^5`` with ``t = x - x_i``; the pieces
join with continuous first and second derivatives.  That leaves two free coefficients per interior point and two at
each end: a prescribed derivative uses one up, and for every derivative that is NOT prescribed the reference sets the
highest-order coefficient still free in a neighbouring piece to zero -- a first, then b -- choosing the piece by the rule
of ``_give_up`` below.  Outside the data the curve continues as the parabola that matches value, slope and curvature at
the end point (one extra piece on either side, as wide as the nearest data interval).

Host-side set-up code: the device evaluates the resulting tables (``x_points``, ``coefficients``); nothing here runs
per step.
"""
import numpy as np


def _give_up(used, first, second, start):
    """The piece that surrenders its next coefficient for one unprescribed derivative (trep/spline.py:52-101): of the two
    candidates the one that has surrendered fewest (`first` wins ties), as long as that is fewer than two; otherwise the
    nearest piece at or below `start` with fewer than two (piece 0 as the last resort).  Returns (piece, 0 for a / 1 for b)."""
    pick = None
    for level in (0, 1):
        for j in (first, second):
            if pick is None and j >= 0 and used[j] == level:
                pick = j
    if pick is None:
        pick = start
        while pick > 0 and used[pick] >= 2:
            pick -= 1
    used[pick] += 1
    return pick, (0 if used[pick] == 1 else 1)


class Spline(object):
    def __init__(self, data):
        self._data = [tuple(d) for d in data]
        x = [float(d[0]) for d in self._data]
        y = [float(d[1]) for d in self._data]
        n = len(x)
        if n < 2:
            raise ValueError("a spline needs at least two points")
        dy = [float(d[2]) if len(d) > 2 and d[2] is not None else None for d in self._data]
        ddy = [float(d[3]) if len(d) > 3 and d[3] is not None else None for d in self._data]
        prescribed = sum(v is not None for v in dy) + sum(v is not None for v in ddy)
        # with fewer than two prescribed derivatives in total the end curvatures default to zero (first the left end,
        # then the right one), as the reference does (with a printed notice there)
        if ddy[0] is None and prescribed < 2:
            ddy[0] = 0.0
            prescribed += 1
        if ddy[-1] is None and prescribed < 2:
            ddy[-1] = 0.0
            prescribed += 1

        pieces = n - 1                       # unknowns (a, b, c, d, e) of piece i at columns 5 i + 0 .. 4
        A = np.zeros((5 * pieces, 5 * pieces))
        rhs = np.zeros(5 * pieces)
        used = [0] * pieces
        row = 0

        def put(cols_vals, value):
            nonlocal row
            for col, val in cols_vals:
                A[row, col] = val
            rhs[row] = value
            row += 1

        def value_row(i, h):                 # the piece reaches the next data value at t = h
            return [(5 * i + 0, h ** 5), (5 * i + 1, h ** 4), (5 * i + 2, h ** 3), (5 * i + 3, h ** 2), (5 * i + 4, h)]

        def slope_row(i, h):
            return [(5 * i + 0, 5.0 * h ** 4), (5 * i + 1, 4.0 * h ** 3), (5 * i + 2, 3.0 * h ** 2), (5 * i + 3, 2.0 * h), (5 * i + 4, 1.0)]

        def curvature_row(i, h):
            return [(5 * i + 0, 20.0 * h ** 3), (5 * i + 1, 12.0 * h ** 2), (5 * i + 2, 6.0 * h), (5 * i + 3, 2.0)]

        for i in range(pieces):
            h = x[i + 1] - x[i]
            put(value_row(i, h), y[i + 1] - y[i])
            if i + 1 < pieces:               # slope and curvature continue into the next piece
                put(slope_row(i, h) + [(5 * (i + 1) + 4, -1.0)], 0.0)
                put(curvature_row(i, h) + [(5 * (i + 1) + 3, -2.0)], 0.0)
            # the two derivatives at the LEFT point of the piece: prescribed, or a coefficient given up near it
            for prescribed_value, own in ((dy[i], (5 * i + 4, 1.0)), (ddy[i], (5 * i + 3, 2.0))):
                if prescribed_value is None:
                    j, which = _give_up(used, i - 1, i, i)
                    put([(5 * j + which, 1.0)], 0.0)
                else:
                    put([own], prescribed_value)
        last, h = pieces - 1, x[-1] - x[-2]
        for prescribed_value, builder in ((dy[-1], slope_row), (ddy[-1], curvature_row)):     # the right end point
            if prescribed_value is None:
                j, which = _give_up(used, last, last - 1, last)
                put([(5 * j + which, 1.0)], 0.0)
            else:
                put(builder(last, h), prescribed_value)
        sol = np.linalg.solve(A, rhs)
        coeffs = [tuple(sol[5 * i:5 * i + 5]) + (y[i],) for i in range(pieces)]

        # parabolic continuation on the left (an arbitrary knot one last-interval width below the first point) ...
        slope0, curv0 = coeffs[0][4], 2.0 * coeffs[0][3]
        w = x[-1] - x[-2]
        x_left = x[0] - w
        left = (0.0, 0.0, 0.0, 0.5 * curv0, slope0 - curv0 * w, y[0] - slope0 * w + 0.5 * curv0 * w * w)
        # ... and on the right, starting at the last point
        a, b, c, d, e, _ = coeffs[-1]
        slope_n = 5 * a * h ** 4 + 4 * b * h ** 3 + 3 * c * h ** 2 + 2 * d * h + e
        curv_n = 20 * a * h ** 3 + 12 * b * h ** 2 + 6 * c * h + 2 * d
        right = (0.0, 0.0, 0.0, 0.5 * curv_n, slope_n, y[-1])
        xs = [x_left] + x + [x[-1] + h]
        ys = [left[5]] + y + [right[3] * h * h + right[4] * h + right[5]]
        self._x_points = np.array(xs, dtype=np.float64)
        self._y_points = np.array(ys, dtype=np.float64)
        self._coefficients = np.array([left] + coeffs + [right], dtype=np.float64)

    x_points = property(lambda self: self._x_points.copy())
    y_points = property(lambda self: self._y_points.copy())
    coefficients = property(lambda self: self._coefficients.copy())

    def _piece(self, x):
        """Index of the polynomial used at x (spline.c:8-22): the first piece below the second knot, the last one from
        the second-to-last knot on."""
        xp = self._x_points
        if x < xp[0]:
            return 0
        if x >= xp[-1]:
            return len(xp) - 2
        return int(np.searchsorted(xp, x, side="right")) - 1

    def y(self, x):
        i = self._piece(x)
        a, b, c, d, e, f = self._coefficients[i]
        t = x - self._x_points[i]
        return a * t ** 5 + b * t ** 4 + c * t ** 3 + d * t ** 2 + e * t + f

    def dy(self, x):
        i = self._piece(x)
        a, b, c, d, e, _ = self._coefficients[i]
        t = x - self._x_points[i]
        return 5 * a * t ** 4 + 4 * b * t ** 3 + 3 * c * t ** 2 + 2 * d * t + e

    def ddy(self, x):
        i = self._piece(x)
        a, b, c, d, _, _ = self._coefficients[i]
        t = x - self._x_points[i]
        return 20 * a * t ** 3 + 12 * b * t ** 2 + 6 * c * t + 2 * d

    def copy(self):
        return Spline(self._data[:])
