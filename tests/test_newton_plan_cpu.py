"""The bordered-block-diagonal plan of the structured Newton solve (trep_amd/csrc/bbd.hpp, host side: no GPU needed): pattern and plan of
the BASELINE systems through tg_system_newton_plan, and the property the device code relies on -- every structural non-zero of the
Newton matrix lies inside one group's (own | border) rows and columns or inside the border system."""
import ctypes

import numpy as np
import pytest

from common import build


def _plan(name):
    from trep_amd import _lib
    L = _lib.lib()
    system, desc = build(name)
    h = L.tg_system_create(desc.byref())
    assert h
    try:
        out = np.zeros(8, dtype=np.int32)
        _lib.check(L.tg_system_newton_plan(h, out.ctypes.data, None, None))
        nf = int(out[5])
        pat = np.zeros((nf, nf), dtype=np.uint8)
        tab = np.zeros(128, dtype=np.int32)
        _lib.check(L.tg_system_newton_plan(h, out.ctypes.data, pat.ctypes.data, tab.ctypes.data))
    finally:
        L.tg_system_destroy(h)
    return dict(ok=int(out[0]), groups=int(out[1]), ng=int(out[2]), nb=int(out[3]), t=int(out[4]), nf=nf, nd=int(out[6])), pat.astype(bool), tab, system


def _groups(plan, tab):
    own, border = [], []
    for g in range(4):
        rows = [(int(tab[16 * g + r]) & 0xFF) - 1 for r in range(16)]
        own.append([v for v in rows[:plan["ng"]] if v >= 0])
        border.append([v for v in rows[plan["ng"]:plan["ng"] + plan["nb"]] if v >= 0])
    trailing = [((int(tab[i]) >> 16) & 0xFF) - 1 for i in range(plan["t"])]
    return own, border, trailing


def test_puppet_plan_is_four_limbs_around_torso_and_strings():
    plan, pat, tab, system = _plan("puppet40")
    assert plan == dict(ok=1, groups=4, ng=4, nb=7, t=12, nf=28, nd=22)
    assert int(pat[:22, :22].sum()) == 292            # of 484: the mass-matrix pattern of the marionette
    own, border, trailing = _groups(plan, tab)
    names = [c.name for c in system.dyn_configs]
    assert [names[i] for i in trailing[:6]] == ["torso_tx", "torso_ty", "torso_tz", "torso_rz", "torso_ry", "torso_rx"]
    assert trailing[6:] == [22, 23, 24, 25, 26, 27]  # the six string constraints, after the configs
    limbs = sorted(sorted(names[i] for i in g) for g in own)
    assert limbs == sorted([sorted(s + j for j in ("hip_rz", "hip_ry", "hip_rx", "knee_rx")) for s in "lr"] +
                           [sorted(s + j for j in ("shoulder_rz", "shoulder_ry", "shoulder_rx", "elbow_rx")) for s in "lr"])
    for b in border:                                   # every limb touches the torso and exactly one string
        assert b[:6] == trailing[:6] and len(b) == 7 and b[6] >= 22


@pytest.mark.parametrize("name", ["puppet40", "puppet_basic", "scissor4"])
def test_every_structural_nonzero_is_covered_by_the_plan(name):
    plan, pat, tab, _ = _plan(name)
    assert plan["ok"] == 1 and plan["groups"] >= 2 and plan["ng"] + plan["nb"] <= 16 and plan["t"] <= 16
    own, border, trailing = _groups(plan, tab)
    nf, nd = plan["nf"], plan["nd"]
    where = {}
    for g, o in enumerate(own):
        for v in o:
            assert v < nd and v not in where           # own variables are configs, each in one group
            where[v] = g
    assert sorted(list(where) + trailing) == list(range(nf))
    assert all(v < nd for v in trailing[:sum(1 for v in trailing if v < nd)]) and sorted(trailing[sum(1 for v in trailing if v < nd):]) == list(range(nd, nf))
    assert (pat == pat.T).all()
    for i in range(nf):
        for j in range(nf):
            if not pat[i, j]:
                continue
            gi, gj = where.get(i), where.get(j)
            if gi is not None and gj is not None:
                assert gi == gj, (i, j)                # two own variables couple only inside their group
            elif gi is not None:
                assert j in border[gi], (i, j)         # own x border: the border variable is in the group's list
            elif gj is not None:
                assert i in border[gj], (i, j)
    # the column tables repeat the row tables (column j of a group is the variable of its row j)
    for g in range(4):
        for r in range(16):
            assert (int(tab[64 + 16 * g + r]) & 0xFFFF) == (int(tab[16 * g + r]) & 0xFFFF)


@pytest.mark.parametrize("name", ["pendulum1", "pend_on_cart", "spring_arm"])
def test_small_systems_have_no_plan(name):
    plan, pat, tab, _ = _plan(name)
    assert plan["ok"] == 0 and not pat.any() and not tab.any()


def _spec_constants(system):
    """static constexpr ints of the system's specialisation header (tg_system_spec_header: host side, no compiler involved)"""
    import re
    from trep_amd import specialize
    text = specialize.header(system)
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"static constexpr int (\w+) = (-?\d+);", text)}, text


def test_puppet_sweep_plan_and_rollout_item_lists():
    """What the re-cut evaluation phases of the specialised rollout kernel are compiled against (DevProg sw_*, n_dhr, max_cfg_items):
    the quad-lane chain sweep's passes -- a round's 2 x chains instances five to a pass, the longest chain of a pass as its unrolled trip
    count -- and the constraint-derivative items of the dynamic configs only."""
    from trep_amd import systems
    c, text = _spec_constants(systems.puppet())
    assert c["sw_ok"] == 1 and c["n_rounds"] == 2
    # round 0: the torso chain (6 joints) in both pose sets and the six two-joint string-hook chains at q2 only (no body hangs below a
    # hook: its midpoint pose is never read) -> 8 instances in 2 passes; round 1: four limbs of 4 joints in both sets
    assert [c["sw_np_%d" % i] for i in range(4)] == [2, 2, 0, 0]
    assert [c["sw_len_%d" % i] for i in range(8)] == [6, 2, 0, 0, 4, 4, 0, 0]
    assert c["sw_maxlen"] == 6
    inst = [c["sw_inst_%d" % i] for i in range(80)]
    assert inst[:8] == [512, 768] + [768 + s for s in range(1, 7)] and not any(inst[8:20])       # slot | set << 8 | 1 << 9
    assert inst[20:28] == [512, 768, 513, 769, 514, 770, 515, 771] and not any(inst[28:])
    # 2 x 34 (pose set, joint) items, 14 of them never read by the rollout: one trip of the wavefront instead of two
    assert c["n_sj"] == 54 and c["n_sj_rot"] == 36
    # 68 (constraint, config) items, 18 of them of kinematic configs (six string lengths, twelve hook coordinates)
    assert c["n_dh"] == 68 and c["n_dhr"] == 50 and c["nk"] == 18
    assert c["max_cfg_items"] == 10 and c["n_items"] == 88 and c["n_joints"] == 34
    for name in ("j_prm", "at_d", "ae_d", "at_i", "ae_i", "dhr_pack", "sj_list", "sj_full"):
        assert "*%s = " % name in text, name


@pytest.mark.parametrize("name", ["pendulum", "cart", "scissor_lift", "puppet_basic"])
def test_sweep_plans_of_the_other_baseline_systems(name):
    from trep_amd import systems
    system = {"pendulum": lambda: systems.pendulum(1), "cart": systems.pend_on_cart, "scissor_lift": lambda: systems.scissor_lift(4),
              "puppet_basic": systems.puppet_basic}[name]()
    c, _ = _spec_constants(system)
    assert c["n_dhr"] <= c["n_dh"]
    if c["sw_ok"]:
        rounds = c["n_rounds"]
        assert 1 <= rounds <= 4
        for r in range(rounds):
            np_r = c["sw_np_%d" % r]
            assert 1 <= np_r <= 4
            assert sum(c["sw_len_%d" % (4 * r + p)] for p in range(np_r)) <= 16
            assert all(1 <= c["sw_len_%d" % (4 * r + p)] <= c["sw_maxlen"] for p in range(np_r))
            assert all(c["sw_len_%d" % (4 * r + p)] == 0 for p in range(np_r, 4))
