"""trep_amd.Spline against tables recorded from the reference's Spline (trep/spline.py:6-259) for the three force curves of
systems.nonlinear_spring_arm() (tests/golden/nonlinear_spring_arm.npz, tools/gen_golden.py::gen_nonlinear_spring_arm), plus the
evaluation rules of _trep/spline.c:8-57 (piece selection outside the knots, derivatives)."""
import numpy as np

from common import golden


def test_spline_tables_match_reference():
    from trep_amd import systems
    g = golden("nonlinear_spring_arm")
    system = systems.nonlinear_spring_arm()
    curves = [p.spline for p in system.potentials if hasattr(p, "spline")]
    assert len(curves) == 5
    for i, sp in enumerate(curves):
        assert np.abs(sp.x_points - g["spline%d_x" % i]).max() < 1e-14
        assert np.abs(sp.y_points - g["spline%d_y" % i]).max() < 1e-12 * max(1.0, np.abs(g["spline%d_y" % i]).max())
        ref = g["spline%d_c" % i]
        assert sp.coefficients.shape == ref.shape
        assert np.abs(sp.coefficients - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_spline_evaluation_rules():
    from trep_amd import Spline
    sp = Spline([(0.0, 1.0, 0.5), (1.0, 2.0), (2.5, 0.0, None, -1.0)])
    xp, co = sp.x_points, sp.coefficients
    assert len(xp) == 5 and co.shape == (4, 6)                 # one parabolic piece added on either side
    assert np.all(co[0, :3] == 0.0) and np.all(co[-1, :3] == 0.0)
    for x in (0.0, 1.0, 2.5):                                  # interpolation, C1 and C2 joins
        assert abs(sp.y(x) - {0.0: 1.0, 1.0: 2.0, 2.5: 0.0}[x]) < 1e-12
    for x in (1.0,):
        e = 1e-6
        assert abs(sp.dy(x - e) - sp.dy(x + e)) < 1e-4 and abs(sp.ddy(x - e) - sp.ddy(x + e)) < 1e-3
    assert abs(sp.dy(0.0) - 0.5) < 1e-12 and abs(sp.ddy(2.5) + 1.0) < 1e-12
    h = 1e-5                                                   # derivatives are those of y
    for x in (-3.0, -0.2, 0.3, 1.7, 2.9, 9.0):
        assert abs((sp.y(x + h) - sp.y(x - h)) / (2 * h) - sp.dy(x)) < 1e-6 * max(1.0, abs(sp.dy(x)))
        assert abs((sp.dy(x + h) - sp.dy(x - h)) / (2 * h) - sp.ddy(x)) < 1e-6 * max(1.0, abs(sp.ddy(x)))
    # beyond the outer knots the outermost parabolas continue (spline.c:12-15)
    assert abs(sp.ddy(-50.0) - sp.ddy(xp[0] + 1e-9)) < 1e-12 and abs(sp.ddy(50.0) - sp.ddy(xp[-1] - 1e-9)) < 1e-12
    assert isinstance(sp.copy(), Spline) and np.all(sp.copy().coefficients == co)
