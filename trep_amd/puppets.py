"""The marionette ("puppet") model factory: the ~40-DOF benchmark system.

Produces the same mechanical system as the reference's
``trep.puppets.Puppet`` (/root/reference/trep/puppets/puppets.py:9-352): same
frame tree and frame order, same config names and order (strings in dict
insertion order: the Py3 oracle's order, SURVEY.md §8c "ordering hazard"), same
dimensions and inertias.  The tree is assembled from two small limb/torso
builders instead of one literal nested list; the OpenGL visual is out of scope.
``Puppet(string_constraints=True)`` gives nq=40 (22 dynamic + 18 kinematic),
nc=6, 86 frames, 10 masses.
"""
from .system import System
from .frame import rx, ry, rz, tx, ty, tz, const_txyz
from .dynamics import Gravity, Damping, ConfigForce, HybridWrench, Distance

# Lengths from motion-capture data; inertia entries are [M, Ixx, Iyy, Izz].
DEFAULT_DIMENSIONS = {
    'ltibia_length': 0.4000000081268073,
    'upper_torso_length': 0.49134332092915983,
    'lshoulder_width': 0.15811387592769632,
    'final_head_length': 0.079999981957211988,
    'lradius_length': 0.25765900905806788,
    'rshoulder_width': 0.15811388178161004,
    'rfoot_length': 0.20000000361419676,
    'rradius_length': 0.25765900577819512,
    'lfoot_length': 0.20298699715253277,
    'rtibia_length': 0.40000001048670952,
    'rhip_width': 0.10000000072551045,
    'neck_length': 0.04500002226970809,
    'lhip_width': 0.10000000633343277,
    'lhumerus_length': 0.24999999610023391,
    'lower_head_length': 0.044999991144167348,
    'rfemur_length': 0.4000000149386187,
    'upper_head_length': 0.090000011865579349,
    'lhand_length': 0.090000006491074078,
    'rhumerus_length': 0.2500000129383001,
    'rhand_length': 0.089999998759819969,
    'lfemur_length': 0.4000000143639284,
    'pelvis_mass': [10, 1, 1, 1],
    'head_mass': [0.75, 0.08, 0.08, 0.08],
    'femur_mass': [1.5, 0.3, 0.3, 0.1],
    'tibia_mass': [1.5, 0.3, 0.3, 0.1],
    'humerus_mass': [0.5, 0.05, 0.05, 0.01],
    'radius_mass': [0.5, 0.05, 0.05, 0.01],
    'damping': 0.2,
    'string_plane_height': 2,
    # string name -> (frame it hooks to, (x, y, z) offset of the hook in that frame)
    'strings': {
        'upper_torso_string': ('spine_top', (-0.1, 0, 0)),
        'lower_torso_string': ('spine_top', (0.1, 0, 0)),
        'left_arm_string': ('lradius_end', (0, 0.1, 0)),
        'right_arm_string': ('rradius_end', (0, 0.1, 0)),
        'left_leg_string': ('lfemur_end', (0, 0.1, 0)),
        'right_leg_string': ('rfemur_end', (0, 0.1, 0)),
    },
}

_FREE_JOINTS = (['torso_' + a for a in ('tx', 'ty', 'tz', 'rz', 'ry', 'rx')] +
                [s + j for s in 'lr' for j in ('hip_rz', 'hip_ry', 'hip_rx', 'knee_rx',
                                               'shoulder_rz', 'shoulder_ry', 'shoulder_rx', 'elbow_rx')])
_LOCKED_JOINTS = ([s + part + ax for s in 'lr' for part in ('foot_', 'hand_') for ax in ('rx', 'ry', 'rz')] +
                  ['neck_rz', 'neck_ry', 'neck_rx'])

# A string names the config that drives the joint; a number locks the joint there.
DEFAULT_JOINTS = dict([(j, j) for j in _FREE_JOINTS] + [(j, 0.0) for j in _LOCKED_JOINTS])


def fill_dimensions(dimensions={}):
    dim = dict(dimensions)
    for key, default in DEFAULT_DIMENSIONS.items():
        dim.setdefault(key, default)
    return dim


def fill_joints(joints={}):
    out = dict(joints)
    for key, default in DEFAULT_JOINTS.items():
        out.setdefault(key, default)
    return out


def _ball_joint(j, stem, tail):
    """rz -> ry -> rx chain; ``tail`` hangs off the rx frame (given as (rx_def, children))."""
    rx_def, children = tail
    return [rz(j[stem + '_rz']), [ry(j[stem + '_ry']), [rx_def, children]]]


def _limb(side, dim, j, root, offset, upper, hinge, lower, tip, upper_mass, lower_mass, tip_frame):
    """hip/shoulder offset -> 3-axis joint -> upper segment -> hinge -> lower segment -> locked 3-axis tip."""
    n = lambda s: side + s
    upper_len = dim[n(upper) + '_length']
    lower_len = dim[n(lower) + '_length']
    tip_chain = _ball_joint(j, n(tip), (rx(j[n(tip) + '_rx'], name=n(tip)), [tip_frame]))
    lower_seg = [
        tz(-lower_len / 2, name=n(lower) + '_mass', mass=lower_mass),
        tz(-lower_len, name=n(lower) + '_end'), tip_chain]
    upper_seg = [
        tz(-upper_len / 2, name=n(upper) + '_mass', mass=upper_mass),
        tz(-upper_len, name=n(upper) + '_end'), [
            rx(j[n(hinge) + '_rx'], name=n(lower)), lower_seg]]
    return [tx(offset, name=n(root)),
            _ball_joint(j, n(root), (rx(j[n(root) + '_rx'], name=n(upper)), upper_seg))]


def make_skeleton(dimensions={}, joints={}):
    dim = fill_dimensions(dimensions)
    j = fill_joints(joints)

    def leg(side, sign):
        return _limb(side, dim, j, 'hip', sign * dim[side + 'hip_width'], 'femur', 'knee', 'tibia', 'foot',
                     dim['femur_mass'], dim['tibia_mass'],
                     ty(dim[side + 'foot_length'], name=side + 'foot_end'))

    def arm(side, sign):
        return _limb(side, dim, j, 'shoulder', sign * dim[side + 'shoulder_width'], 'humerus', 'elbow',
                     'radius', 'hand', dim['humerus_mass'], dim['radius_mass'],
                     tz(-dim[side + 'hand_length'], name=side + 'hand_end'))

    # The reference drives the neck's rx frame with the 'neck_rz' entry
    # (puppets.py:197); both are locked at 0.0 by default, kept for parity.
    head = [tz(dim['neck_length'], name='neck'),
            _ball_joint(j, 'neck', (rx(j['neck_rz'], name='neck_joint'), [
                tz(dim['lower_head_length'], name='head'), [
                    tz(dim['upper_head_length'], name='head_center', mass=dim['head_mass']), [
                        tz(dim['final_head_length'], name='head_end')]]]))]
    chest = [tz(dim['upper_torso_length'], name='spine_top'), head + arm('l', -1) + arm('r', +1)]
    body = leg('l', -1) + leg('r', +1) + chest
    return [tx(j['torso_tx']), [ty(j['torso_ty']), [tz(j['torso_tz']), [
        rz(j['torso_rz']), [ry(j['torso_ry']), [
            rx(j['torso_rx'], name='pelvis', mass=dim['pelvis_mass']), body]]]]]]


class Puppet(System):
    def __init__(self, dimensions={}, joints={}, joint_forces=False, string_forces=False,
                 string_constraints=False):
        System.__init__(self)
        self.string_plane = None
        self.string_hooks = {}
        self.joint_forces = {}
        self.string_forces = {}
        self.string_constraints = {}
        self.dimensions = fill_dimensions(dimensions)
        self.joints = fill_joints(joints)

        self.import_frames(make_skeleton(self.dimensions, self.joints))
        self.make_string_frames()
        if joint_forces:
            self.make_joint_forces()
        if string_forces:
            self.make_string_forces()
        if string_constraints:
            self.make_string_constraints()
        Gravity(self, (0, 0, -9.8))
        Damping(self, self.dimensions['damping'])

    def make_string_frames(self):
        self.world_frame.import_frames([tz(self.dimensions['string_plane_height'], name='string_plane')])
        self.string_plane = self.get_frame('string_plane')
        self.string_hooks = {}
        for name, (anchor, offset) in self.dimensions['strings'].items():
            self.string_hooks[name] = name + '_hook'
            self.get_frame(anchor).import_frames([const_txyz(offset, name=self.string_hooks[name])])

    def make_joint_forces(self):
        for config in self.dyn_configs:
            self.joint_forces[config.name] = ConfigForce(self, config, config.name, config.name)

    def make_string_forces(self):
        """A world-frame force with three inputs at every string hook (puppets.py:276-288)."""
        for name, hook in self.string_hooks.items():
            force = {'name': name, 'x': name + '-x', 'y': name + '-y', 'z': name + '-z', 'hook': hook}
            HybridWrench(self, hook, (force['x'], force['y'], force['z'], 0, 0, 0), name=name)
            self.string_forces[name] = force

    def make_string_constraints(self):
        for name, hook in self.string_hooks.items():
            info = {'name': name, 'x': name + '-x', 'y': name + '-y', 'length': name + '-length',
                    'control_hook': name + '_control', 'hook': hook}
            self.string_plane.import_frames([
                tx(info['x'], kinematic=True), [ty(info['y'], kinematic=True, name=info['control_hook'])]])
            Distance(self, info['hook'], info['control_hook'], info['length'], name=name)
            self.string_constraints[name] = info

    def _string_infos(self, strings):
        if strings is None:
            return list(self.string_constraints.values())
        return [self.string_constraints[name] for name in strings]

    def project_string_controls(self, strings=None):
        """Move every control point straight above its hook and fix the string lengths."""
        for info in self._string_infos(strings):
            pos = self.get_frame(info['hook']).p()
            self.get_config(info['x']).q = pos[0]
            self.get_config(info['y']).q = pos[1]
        self.correct_string_lengths(strings)

    def correct_string_lengths(self, strings=None):
        for info in self._string_infos(strings):
            self.get_config(info['length']).q = self.get_constraint(info['name']).get_actual_distance()
