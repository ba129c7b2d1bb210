"""Builders for the BASELINE.json benchmark systems (host-side model code).

Each function rebuilds, with this package's model API, the mechanical system a
reference example script defines:

  pendulum(links)      examples/pendulum.py:36-71        (RX joints, unit point masses)
  pend_on_cart()       examples/pend-on-cart-optimization.py:48-64
  scissor_lift(n)      examples/scissor.py:53-105        (closed chain, PointToPoint2D)
  puppet()             trep/puppets/puppets.py:220-253   (Puppet(string_constraints=True))
  puppet_basic()       examples/puppet-basic.py:15-84    (22 dynamic configs, six fixed-length strings)

plus the synthetic initial conditions SURVEY.md §8(d) prescribes for each.
"""
import math

import numpy as np



def _api(api):
    """The model API to build with: this package by default, or any module exposing the same
    names (tools/gen_golden.py passes the reference's ``trep`` to build the identical system there)."""
    if api is None:
        import trep_amd as api
    return api


def pendulum(links=1, q0=math.pi / 4.0, api=None):
    T = _api(api)
    system = T.System()
    T.potentials.Gravity(system, name="Gravity")
    parent = system.world_frame
    for link in range(links):
        joint = T.Frame(parent, T.RX, "link-%d" % link, "link-%d" % link)
        parent = T.Frame(joint, T.TZ, -1)
        parent.set_mass(1.0)
    system.get_config("link-0").q = q0
    return system


def pend_on_cart(torque_force=False, api=None):
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('x', name='Cart', mass=10.0), [
            T.rz('theta', name="PendulumBase"), [
                T.ty(-1.0, name="Pendulum", mass=1.0)]]])
    T.potentials.Gravity(system, (0, -9.8, 0))
    T.forces.Damping(system, 0.01)
    T.forces.ConfigForce(system, 'x', 'x-force')
    if torque_force:
        T.forces.ConfigForce(system, 'theta', 'theta-force')
    return system


def spring_arm(api=None):
    """A 3-D three-joint arm on a sliding base with springs on its configs: exercises the ConfigSpring
    potential (several springs on one config, a spring on a kinematic config) next to gravity, damping and
    an input force.  Synthetic test system (no reference example of this shape)."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.rx('c', name='Wrist'), [
                                T.tz(-0.5, name='Hand', mass=0.5)]]]]]]])
    T.potentials.Gravity(system, (0.3, 0, -9.8))
    T.potentials.ConfigSpring(system, 'a', k=20.0, q0=0.3)
    T.potentials.ConfigSpring(system, 'b', k=5.0)
    T.potentials.ConfigSpring(system, 'c', k=2.0, q0=-0.2)
    T.potentials.ConfigSpring(system, 'c', k=1.0, q0=0.1)
    T.potentials.ConfigSpring(system, 'slide', k=3.0, q0=0.5)
    T.forces.Damping(system, 0.1)
    T.forces.ConfigForce(system, 'a', 'a-torque')
    return system


def nonlinear_spring_arm(api=None):
    """The arm of spring_arm() with NonlinearConfigSpring potentials (force curves given by splines): two curves on one
    config (one of them scaled and shifted, m = -1.5, b = 0.4), one next to a ConfigSpring, one on the kinematic slider,
    curves with prescribed end slopes / curvatures and with none, and states that leave the knot range on both sides.
    Synthetic test system for potentials/nonlinear_config_spring.c and trep/spline.py."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.rx('c', name='Wrist'), [
                                T.tz(-0.5, name='Hand', mass=0.5)]]]]]]])
    T.potentials.Gravity(system, (0.3, 0, -9.8))
    soft = T.Spline([(-1.0, -6.0), (-0.3, -1.0), (0.2, 0.5), (0.9, 4.0), (1.6, 5.0)])
    stiff = T.Spline([(-0.8, 3.0, -2.0), (0.0, 0.0), (0.5, -1.0, None, 0.5), (1.2, -4.0, -6.0, 0.0)])
    bump = T.Spline([(-0.5, 0.0, 0.0, 0.0), (0.0, 1.5), (0.4, 0.2), (0.7, 0.0, 0.0, 0.0)])
    T.potentials.NonlinearConfigSpring(system, 'a', soft)
    T.potentials.NonlinearConfigSpring(system, 'a', bump, m=-1.5, b=0.4)
    T.potentials.NonlinearConfigSpring(system, 'b', stiff, m=0.8, b=-0.1)
    T.potentials.ConfigSpring(system, 'b', k=5.0, q0=0.1)
    T.potentials.NonlinearConfigSpring(system, 'c', bump, m=2.0)
    T.potentials.NonlinearConfigSpring(system, 'slide', soft, m=1.0, b=0.2)
    T.forces.Damping(system, 0.1)
    T.forces.ConfigForce(system, 'a', 'a-torque')
    return system


def spring_link(api=None):
    """Two branches joined by two-point springs: a 3-D arm on a kinematic slider and a telescopic pendulum held at
    unit length by a distance constraint; one LinearSpring from the arm's tip to a fixed anchor (rest length 1), one
    of zero rest length between the fore-arm and the pendulum bob.  Synthetic test system for the LinearSpring
    potential next to constraints, damping and a kinematic config."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.tx(0.2, name='Tip')]]]]],
        T.ty(1.5, name='Pivot'), [
            T.rx('d', name='Swing'), [
                T.tz('e', name='Bob', mass=1.5)]],
        T.tz(2.0, name='Anchor')])
    T.potentials.Gravity(system, (0, 0, -9.8))
    T.potentials.LinearSpring(system, 'Tip', 'Anchor', k=15.0, x0=1.0)
    T.potentials.LinearSpring(system, 'Fore', 'Bob', k=4.0)
    T.constraints.Distance(system, 'Bob', 'Pivot', 1.0)
    T.forces.Damping(system, 0.2)
    return system


def plane_link(api=None):
    """Closed chain held together by PointOnPlane constraints whose plane frames move: a pendulum carries the plane
    (its own y = 0 plane, and a second one with an oblique normal on its bob), a 3-D two-link arm on another pivot
    keeps its tip in them.  At the zero configuration the tip lies in both planes.  Synthetic test system."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.rx('a', name='PlaneBody'), [
            T.tz(-1.0, name='PlaneBob', mass=2.0)],
        T.ty(1.0, name='Pivot2'), [
            T.rx('b', name='Arm1'), [
                T.tz(-1.0, name='Arm1Mass', mass=1.0), [
                    T.ry('c', name='Arm2'), [
                        T.rx('d', name='Arm3'), [
                            T.ty(-1.0, name='Tip', mass=(0.5, 0.05, 0.06, 0.07))]]]]]])
    T.potentials.Gravity(system, (0.5, 0.3, -9.8))
    T.constraints.PointOnPlane(system, 'PlaneBody', (0, 1, 0), 'Tip')
    T.constraints.PointOnPlane(system, 'PlaneBob', (1, 0, 0.5), 'Tip')
    T.forces.Damping(system, 0.05)
    return system


def wrench_arm(api=None):
    """A 3-D arm on a kinematic slider pushed around by HybridWrench forces (world-frame force at a frame's origin):
    one with two input components and a constant one at the hand, a constant one at the fore-arm; plus damping and a
    ConfigForce.  Synthetic test system."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.rx('c', name='Wrist'), [
                                T.tz(-0.5, name='Hand', mass=0.5)]]]]]]])
    T.potentials.Gravity(system, (0, 0, -9.8))
    T.forces.HybridWrench(system, 'Hand', ('hand-fx', 2.0, 'hand-fz', 0, 0, 0), name='hand')
    T.forces.HybridWrench(system, 'Fore', (0, 1.0, -1.0), name='fore')
    T.forces.ConfigForce(system, 'a', 'a-torque')
    T.forces.Damping(system, 0.1)
    return system


def damper_link(api=None):
    """Two pendulums and a 3-D arm joined by LinearDamper forces only (no LinearSpring, so the reference has second
    derivatives for it), config springs for stiffness.  Synthetic test system."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.rx('theta1'), [
            T.tz(2, mass=1, name='pend1')],
        T.ty(1), [
            T.rx('theta2'), [
                T.tz(2, mass=1, name='pend2'), [
                    T.ry('phi'), [T.tx(0.7, mass=0.5, name='tip')]]]]])
    T.forces.LinearDamper(system, 'pend1', 'pend2', c=1.0)
    T.forces.LinearDamper(system, 'pend1', 'tip', c=0.6)
    T.potentials.ConfigSpring(system, 'theta1', k=3.0, q0=0.2)
    T.potentials.ConfigSpring(system, 'phi', k=2.0)
    T.potentials.Gravity(system, name="Gravity")
    system.q = [2.5, -2.0, 0.4]
    return system


def wrench_spatial(api=None):
    """The arm of wrench_arm driven by SpatialWrench forces (components in spatial coordinates): one with inputs and
    constants at the hand, a constant one on the fore-arm.  Synthetic test system."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.rx('c', name='Wrist'), [
                                T.tz(-0.5, name='Hand', mass=0.5)]]]]]]])
    T.potentials.Gravity(system, (0, 0, -9.8))
    T.forces.SpatialWrench(system, 'Hand', ('hand-fx', 0.5, -0.3, 'hand-tx', 0.3, 'hand-tz'), name='hand')
    T.forces.SpatialWrench(system, 'Fore', (0.2, 0, 0.4, 0, 1.0, -0.5), name='fore')
    T.forces.HybridWrench(system, 'Upper', (0, 0.3, 0, 0, 0, 0.2), name='upper')
    T.forces.Damping(system, 0.1)
    return system


def wrench_body(api=None):
    """The arm of wrench_arm driven by BodyWrench forces (components in the coordinates of the frame they act on), one on a
    frame with a constant offset rotation.  Synthetic test system."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.rx('c', name='Wrist'), [
                                T.tz(-0.5, name='Hand', mass=0.5), [
                                    T.ry(0.7), [T.tx(0.1, name='Tool')]]]]]]]]])
    T.potentials.Gravity(system, (0, 0, -9.8))
    T.forces.BodyWrench(system, 'Tool', ('tool-fx', 0.5, -0.3, 'tool-tx', 0.3, 'tool-tz'), name='tool')
    T.forces.BodyWrench(system, 'Fore', (0.2, 0, 0.4, 0, 1.0, -0.5), name='fore')
    T.forces.Damping(system, 0.1)
    return system


def dual_pendulums(api=None):
    """examples/dual_pendulums.py:28-44: two pendulums on neighbouring pivots joined by a linear spring and a linear
    damper, under gravity."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.rx('theta1'), [
            T.tz(2, mass=1, name='pend1')],
        T.ty(1), [
            T.rx('theta2'), [
                T.tz(2, mass=1, name='pend2')]]])
    T.potentials.LinearSpring(system, 'pend1', 'pend2', k=20, x0=1)
    T.forces.LinearDamper(system, 'pend1', 'pend2', c=1)
    T.potentials.Gravity(system, name="Gravity")
    system.q = [3, -3]
    return system


def wrench_torque(api=None):
    """The arm of wrench_arm with HybridWrench torque components: a wrench with force and torque inputs at the hand and a
    constant torque on the fore-arm.  Synthetic test system."""
    T = _api(api)
    system = T.System()
    system.import_frames([
        T.tx('slide', name='Base', kinematic=True), [
            T.rz('a', name='Shoulder'), [
                T.tx(1.0, name='Upper', mass=2.0), [
                    T.ry('b', name='Elbow'), [
                        T.tx(0.8, name='Fore', mass=(1.0, 0.1, 0.2, 0.3)), [
                            T.rx('c', name='Wrist'), [
                                T.tz(-0.5, name='Hand', mass=0.5)]]]]]]])
    T.potentials.Gravity(system, (0, 0, -9.8))
    T.forces.HybridWrench(system, 'Hand', ('hand-fx', 0.5, 0, 'hand-tx', 0.3, 'hand-tz'), name='hand')
    T.forces.HybridWrench(system, 'Fore', (0, 0, 0, 0, 1.0, -0.5), name='fore')
    T.forces.Damping(system, 0.1)
    return system


def puppet_forces(api=None):
    """Puppet(string_forces=True): the marionette driven by a world-frame force with three inputs at each of its six
    string hooks instead of kinematic strings (22 dynamic configs, 18 inputs, no constraints)."""
    T = _api(api)
    return T.puppets.Puppet(joint_forces=False, string_forces=True, string_constraints=False)


EXTENSOR_TENDON_POSE = {
    "theta-1": 0.5, "theta-2": 0.0, "theta-3": -0.5, "theta-4": -1.0, "theta-5": 0.5, "theta-6": -0.5, "theta-8": 0.0,
    "theta-9": 0.5, "x-1": -1.0, "x-2": -1.0, "x-3": -1.0, "x-4": -1.0, "x-5": -1.0, "x-6": -1.0, "x-8": -1.0, "x-9": -1.0,
}


def extensor_tendon(api=None):
    """examples/extensor-tendon-model.py:15-56: a planar network of nine linear springs (the tendon) pulled by three
    constant muscle forces (HybridWrench), heavy damping, no gravity; it settles into a steady state."""
    T = _api(api)
    tz, rx = T.tz, T.rx
    system = T.System()
    system.import_frames([
        rx('theta-1'), [
            tz('x-1', name='A', mass=1), [
                rx('theta-2'), [
                    tz('x-2', name='B', mass=1), [
                        rx('theta-3'), [
                            tz('x-3', name='C', mass=1)]]],
                rx('theta-4'), [
                    tz('x-4', name='D', mass=1), [
                        rx('theta-5'), [
                            tz('x-5', name='E', mass=1)]]]]],
        rx('theta-6'), [
            tz('x-6', name='F', mass=1), [
                rx('theta-8'), [
                    tz('x-8', name='H', mass=1), [
                        rx('theta-9'), [
                            tz('x-9', name='I', mass=1)]]]]]])
    for a, b in (("World", "A"), ("A", "B"), ("B", "C"), ("A", "D"), ("D", "E"), ("World", "F"), ("F", "D"), ("F", "H"),
                 ("H", "I")):
        T.potentials.LinearSpring(system, a, b, 10.0, 1.0)
    T.forces.Damping(system, 4.0)
    T.forces.HybridWrench(system, 'C', (0, 1, -1))
    T.forces.HybridWrench(system, 'E', (0, 0, -1.414))
    T.forces.HybridWrench(system, 'I', (0, -1, -1))
    system.q = EXTENSOR_TENDON_POSE
    return system


def scissor_lift(segments=4, theta_0=0.05 * math.pi, m_link=1.0, I_link=1.0, L_link=5.0, m_slider=1.0,
                 api=None):
    """Scissor lift at its analytic closed configuration (no constraint solver needed)."""
    T = _api(api)
    Frame, RY, TX = T.Frame, T.RY, T.TX
    system = T.System()
    T.potentials.Gravity(system, name="Gravity")
    slider = Frame(system.world_frame, TX, "SLIDER")
    slider.config.q = L_link * math.cos(theta_0)
    slider.set_mass(m_slider)
    left, right = system.world_frame, slider
    for link in range(segments):
        left = Frame(left, RY, "L%02d" % link, "L%02d" % link)
        left.config.q = theta_0 if link == 0 else math.pi + 2.0 * theta_0
        left_mid = Frame(left, TX, L_link / 2.0)
        left_mid.set_mass(m_link, I_link, I_link, I_link)
        left_end = Frame(left, TX, L_link)
        right = Frame(right, RY, "R%02d" % link, "R%02d" % link)
        right.config.q = math.pi - theta_0 if link == 0 else math.pi - 2.0 * theta_0
        right_mid = Frame(right, TX, L_link / 2.0)
        right_mid.set_mass(m_link, I_link, I_link, I_link)
        right_end = Frame(right, TX, L_link)
        T.constraints.PointToPoint2D(system, 'xz', left_mid, right_mid)
        left, right = right_end, left_end  # the two sides swap at every level
    return system


def scissor_q(system, theta_0, L_link=5.0):
    """Analytic closed configuration of scissor_lift for opening angle theta_0."""
    q = {}
    for c in system.configs:
        if c.name == "SLIDER":
            q[c.name] = L_link * math.cos(theta_0)
        elif c.name == "L00":
            q[c.name] = theta_0
        elif c.name == "R00":
            q[c.name] = math.pi - theta_0
        elif c.name.startswith("L"):
            q[c.name] = math.pi + 2.0 * theta_0
        else:
            q[c.name] = math.pi - 2.0 * theta_0
    return np.array([q[c.name] for c in system.configs])


def puppet(api=None):
    T = _api(api)
    return T.puppets.Puppet(joint_forces=False, string_forces=False, string_constraints=True)


def puppet_basic(api=None):
    """The humanoid marionette of examples/puppet-basic.py:15-84: a 6-DOF torso, two 4-DOF arms, two 4-DOF
    legs (22 dynamic configs, no kinematic ones), gravity, uniform damping 0.1 and six strings of constant
    length (Distance constraints) hanging from a fixed frame 14 units up.  Frame and config names are the
    example's, so configurations can be exchanged by name."""
    T = _api(api)
    tx, ty, tz, rx, ry, rz = T.tx, T.ty, T.tz, T.rx, T.ry, T.rz

    def ball(prefix, name, children):
        # three intersecting revolute axes (z, y, x), the last one carries the limb and its frame name
        return [rz(prefix + 'Psi'), [ry(prefix + 'Theta'), [rx(prefix + 'Phi', name=name), children]]]

    def arm(side, letter, sx):
        hand = [tx(0.14 * sx), [ty(-0.173, name=side + ' Finger')]]
        forearm = [tz(-1, name=side + ' Radius', mass=(4, 1, 1, 1)), tz(-2.001), hand]
        upper = [tz(-0.95, name=side + ' Humerus', mass=(5, 1, 1, 1)),
                 tz(-1.9), [rx(letter + 'ElbowTheta', name=side + ' Elbow'), forearm]]
        return [tx(1.3 * sx), [tz(0.4), ball(letter + 'Shoulder', side + ' Shoulder', upper)]]

    def leg(side, letter, sx, knee_name):
        shank = [tz(-1.5, name=side + ' Tibia', mass=(4, 1, 1, 1))]
        thigh = [tz(-1.5, name=side + ' Femur', mass=(5, 1, 1, 1)),
                 tz(-2.59), [ty(-0.322, name=side + ' Knee Hook')],
                 tz(-3.0), [rx(letter + 'KneeTheta', name=knee_name), shank]]
        return [tx(0.5 * sx), [tz(-3.0), ball(letter + 'Hip', side + ' Hip', thigh)]]

    torso = [tz(-1.5, mass=50),
             tx(-1.011), [tz(0.658, name='Right Torso Hook')],
             tx(1.011), [tz(0.658, name='Left Torso Hook')],
             tz(0.9, name='Head'), [tz(0.5, mass=(10, 1, 1, 1))]]
    torso += arm('Left', 'L', 1) + arm('Right', 'R', -1)
    torso += leg('Left', 'L', 1, 'Left Knee') + leg('Right', 'R', -1, 'right Knee')
    body = [tx('TorsoX'), [ty('TorsoY'), [tz('TorsoZ'), [
        rz('TorsoPsi'), [ry('TorsoTheta'), [rx('TorsoPhi', name='Torso'), torso]]]]]]
    spindles = [tx(1, name='Left Torso Spindle'), tx(-1, name='Right Torso Spinde'),
                tx(1), [ty(-1, name='Left Arm Spindle')], tx(-1), [ty(-1, name='Right Arm Spindle')],
                tx(1), [ty(-2, name='Left Leg Spindle')], tx(-1), [ty(-2, name='Right Leg Spindle')]]
    system = T.System()
    system.import_frames(body + [tz(14, name='Frame Plane'), spindles])
    T.potentials.Gravity(system, (0, 0, -9.8))
    T.forces.Damping(system, 0.1)
    for hook, spindle, length in (('Left Torso Hook', 'Left Torso Spindle', 13.4),
                                  ('Right Torso Hook', 'Right Torso Spinde', 13.4),
                                  ('Left Finger', 'Left Arm Spindle', 15.4),
                                  ('Right Finger', 'Right Arm Spindle', 15.5),
                                  ('Left Knee Hook', 'Left Leg Spindle', 18.6),
                                  ('Right Knee Hook', 'Right Leg Spindle', 18.6)):
        T.constraints.Distance(system, hook, spindle, length)
    return system


# Starting guess of examples/puppet-basic.py:87-98 (made consistent by System.satisfy_constraints there; the
# consistent poses used here come from tests/golden/puppet_basic.npz, captured from the reference).
PUPPET_BASIC_POSE = {
    'TorsoX': 1, 'TorsoY': 1, 'LElbowTheta': -1.57, 'RElbowTheta': -1.57, 'LHipTheta': -0.314, 'RHipTheta': 0.314,
    'LHipPhi': -0.785, 'RHipPhi': -0.785, 'LKneeTheta': 0.785, 'RKneeTheta': 0.785,
}


# Base pose of the reference's puppet-optimization example
# (examples/puppet-optimization.py:32-47).
PUPPET_BASE_POSE = {
    'torso_rx': -0.05, 'torso_tz': 0.0,
    'lelbow_rx': 1.57, 'relbow_rx': 1.57,
    'lhip_rx': math.pi / 2 - 0.6, 'rhip_rx': math.pi / 2 - 0.6,
    'lknee_rx': -math.pi / 2 + 0.6, 'rknee_rx': -math.pi / 2 + 0.6,
}
PUPPET_LIMB_JOINTS = [s + j for s in 'lr' for j in ('hip_rz', 'hip_ry', 'hip_rx', 'knee_rx',
                                                    'shoulder_rz', 'shoulder_ry', 'shoulder_rx', 'elbow_rx')]


def puppet_initial_conditions(system, batch, seed=20250 + 3):
    """[batch][nq] constraint-consistent puppet poses: base pose + seeded perturbation, then
    ``project_string_controls`` semantics (strings vertical above their hooks, exact lengths)."""
    rng = np.random.default_rng(seed)
    names = [c.name for c in system.configs]
    Q = np.zeros((batch, system.nQ))
    for b in range(batch):
        system.q = 0.0
        system.q = PUPPET_BASE_POSE
        for j in PUPPET_LIMB_JOINTS:
            c = system.get_config(j)
            c.q = c.q + rng.uniform(-0.05, 0.05)
        for j in ('torso_rx', 'torso_ry'):
            c = system.get_config(j)
            c.q = c.q + rng.uniform(-0.02, 0.02)
        system.project_string_controls()
        Q[b] = system.q
    assert names == [c.name for c in system.configs]
    return Q


def puppet_string_schedule(system, k2_0, n_steps, dt, t0=0.0):
    """Kinematic inputs k2[b, k, :] for n_steps steps: the four limb string lengths move by
    0.1*sin(0.6*pi*t) with the sign pattern of examples/puppet-optimization.py:79-82, t being the
    time at the START of step k (the script evaluates the sine at mvi.t1 after the shift)."""
    k2_0 = np.asarray(k2_0, dtype=float)
    K = np.repeat(k2_0[:, None, :], n_steps, axis=1)
    signs = {'left_leg_string-length': -1.0, 'right_leg_string-length': 1.0,
             'left_arm_string-length': 1.0, 'right_arm_string-length': -1.0}
    t = t0 + dt * np.arange(n_steps)
    wave = 0.1 * np.sin(0.6 * math.pi * t)
    for name, sgn in signs.items():
        K[:, :, system.get_config(name).k_index] += sgn * wave[None, :]
    return K
