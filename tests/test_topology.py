"""Bit-exact frame indexing / tree topology vs tables dumped from the reference (SURVEY.md §8 a-T)."""
import numpy as np
import pytest

from common import BUILDERS, build, golden


@pytest.mark.parametrize("name", sorted(BUILDERS))
def test_topology_tables_match_reference(name):
    g = golden(name)
    system, d = build(name)
    assert [str(f.name) for f in system.frames] == list(g["topo_frame_names"])
    assert [str(c.name) for c in system.configs] == list(g["topo_config_names"])
    sizes = [d.n_configs, d.n_dyn, d.n_kin, d.n_inputs, d.n_constraints, d.n_frames]
    assert sizes == list(g["topo_sizes"])
    for key in ["frame_transform", "frame_parent", "frame_config", "frame_cache_size", "frame_cache_index",
                "config_kinematic", "config_gen", "config_k_index", "masses", "config_masses_off",
                "config_masses"]:
        ours = getattr(d, key)
        ref = g["topo_" + key]
        assert ours.dtype == np.int32
        assert np.array_equal(ours, ref), key


def test_puppet_sizes():
    system, d = build("puppet40")
    assert (d.n_configs, d.n_dyn, d.n_kin, d.n_constraints, d.n_frames, d.n_masses) == (40, 22, 18, 6, 86, 10)
    depth = d.frame_cache_size
    assert depth.max() == 10
    assert int(depth.sum()) == 609  # SURVEY.md Appendix E, "all frames"


def test_uses_config_and_lookup():
    system, _ = build("puppet40")
    knee = system.get_config("lknee_rx")
    assert system.get_frame("ltibia_mass").uses_config(knee)
    assert not system.get_frame("rtibia_mass").uses_config(knee)
    assert len(system.get_config("torso_tx").masses) == 10
    assert len(knee.masses) == 1
    assert system.get_config("left_arm_string-length").config_gen == system.nQ


def test_satisfy_constraints_matches_reference():
    """System.satisfy_constraints (host-side setup, examples/puppet-basic.py:101, scissor.py:104): the consistent pose
    found from the script's starting guess equals the one the reference's own satisfy_constraints produced
    (tests/golden/puppet_basic.npz, ic_set[0]); constraint gradients are checked by finite differences."""
    import numpy as np
    from trep_amd import systems
    from common import golden
    s = systems.puppet_basic()
    s.q = 0.0
    s.q = systems.PUPPET_BASIC_POSE
    assert max(abs(c.h()) for c in s.constraints) > 1.0
    for c in s.constraints[:3]:
        for name in ("TorsoPhi", "LShoulderTheta", "LKneeTheta", "TorsoZ"):
            cfg = s.get_config(name)
            h0, g = c.h(), c.h_dq(cfg)
            cfg.q += 1e-6
            fd = (c.h() - h0) / 1e-6
            cfg.q -= 1e-6
            assert abs(fd - g) < 1e-4 * max(1.0, abs(g))
    q = s.satisfy_constraints()
    assert max(abs(c.h()) for c in s.constraints) < 1e-9
    assert np.abs(q - golden("puppet_basic")["ic_set"][0]).max() < 1e-9
    kept = systems.scissor_lift(4)
    kept.get_config("L01").q += 0.05
    slider = kept.get_config("SLIDER").q
    kept.satisfy_constraints(constant_q_list=["SLIDER"])
    assert max(abs(c.h()) for c in kept.constraints) < 1e-9 and kept.get_config("SLIDER").q == slider


def test_export_frames_round_trip():
    """System.export_frames() text rebuilds the same topology tables through import_frames."""
    import trep_amd
    from trep_amd import systems, descriptor
    for build_system in (systems.puppet, systems.spring_link, systems.scissor_lift):
        a = build_system()
        text = a.export_frames().replace("from trep import", "from trep_amd import")
        scope = {"system": trep_amd.System()}
        exec(text, scope)
        b = scope["system"]
        da, db = descriptor.flatten(a), descriptor.flatten(b)
        for key in ("frame_transform", "frame_parent", "frame_value", "frame_lg", "frame_inertia", "frame_cache_size"):
            assert np.array_equal(da.tables[key], db.tables[key]), (build_system.__name__, key)
        assert [f.name for f in a.frames] == [f.name for f in b.frames]
        # configs of the frames by name (configs that constraints create, e.g. string lengths, are not part of the tree)
        describe = lambda s: [(None if f.config is None else (f.config.name, f.config.kinematic)) for f in s.frames]
        assert describe(a) == describe(b)
