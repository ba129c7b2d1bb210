"""Host-side tables of the round-5 rollout kernels, read back from the generated specialisation header (no GPU needed):

* `wev_lane` -- the lists of the world-frame evaluation (mvi_core.hpp eval_world; program.hpp): lane k < nd sums the twists of the proper
  ancestors of config k, lane (body b, axis r) those of the body's whole path -- checked against the model's own frame tree;
* `bbd_map`, `wev_pairx`, `wev_dhx` -- the Newton image in the structured solve's own row order (bbd.hpp, BbdPacked; opt-in
  -DTG_BBD_PACKED): every structural entry of the matrix has exactly one place, the places are distinct and inside the image, and the
  writers' packed records point at them."""
import re

import numpy as np
import pytest

from common import build


def _header(name):
    from trep_amd import specialize
    system, _ = build(name)
    text = specialize.header(system)
    ints = {m.group(1): int(m.group(2)) for m in re.finditer(r"static constexpr int (\w+) = (-?\d+);", text)}
    pool = np.array([int(x) for x in re.search(r"spec_ipool\[\d+\] = \{([^}]*)\}", text, re.S).group(1).replace("\n", "").split(",")], dtype=np.int64)
    offs = {m.group(1): int(m.group(2)) for m in re.finditer(r"static constexpr const int \*(\w+) = spec_ipool \+ (\d+);", text)}
    return system, ints, pool, offs


def _paths(system):
    """config index -> configs on the path from the root to (and including) it; per massive frame its path configs"""
    idx = {c: i for i, c in enumerate(system.configs)}
    cfg_path, body_paths = {}, []
    for f in system.frames:
        if f.config is not None:
            cfg_path[idx[f.config]] = [idx[x.config] for x in f._path() if x.config is not None]
    for f in system.masses:
        body_paths.append([idx[x.config] for x in f._path() if x.config is not None])
    return cfg_path, body_paths


@pytest.mark.parametrize("name", ["puppet40", "puppet_basic"])
def test_world_evaluation_lists_are_the_frame_tree(name):
    system, ints, pool, offs = _header(name)
    assert ints["wev_ok"] == 1 and ints["cmp_ok"] == 1
    nd, nb, depth = ints["nd"], ints["n_bodies"], ints["wev_depth"]
    lane = pool[offs["wev_lane"]:offs["wev_lane"] + 256].reshape(64, 4)
    cfg_path, body_paths = _paths(system)

    def lst(l):
        b = [(int(lane[l, e >> 2]) >> (8 * (e & 3))) & 0xFF for e in range(12)]
        n = next((i for i, v in enumerate(b) if v == nd), 12)
        assert all(v == nd for v in b[n:])          # padded with the all-zero record
        return b[:n]
    longest = 0
    for k in range(nd):
        assert lst(k) == cfg_path[k][:-1], k         # proper ancestors, root first
        longest = max(longest, len(lst(k)))
        w = int(lane[k, 3])
        assert (w >> 24) & 0x7F < ints["n_cgroups"]
    for b in range(nb):
        for r in range(3):
            l = nd + 3 * b + r
            assert lst(l) == body_paths[b], (b, r)
            assert int(lane[l, 3]) == (b | (r << 8))
            longest = max(longest, len(body_paths[b]))
    for l in range(nd + 3 * nb, 64):
        assert lst(l) == []                           # lanes without a role sum the zero record
    assert depth == longest and nd + 3 * nb < 64


@pytest.mark.parametrize("name", ["puppet40", "puppet_basic"])
def test_packed_newton_image_map_is_a_bijection_onto_its_places(name):
    system, ints, pool, offs = _header(name)
    assert ints["bbd_ok"] == 1 and ints["bbd_pk_ok"] == 1
    nf, nd, size = ints["nf"], ints["nd"], ints["bbd_pk_size"]
    assert size <= nf * ints["df_ld"] and size % 2 == 0 and ints["bbd_pk_nc2"] % 2 == 0 and ints["bbd_pk_tc2"] % 2 == 0
    m = pool[offs["bbd_map"]:offs["bbd_map"] + nf * (nf + 1)].reshape(nf, nf + 1)
    placed = m[m >= 0]
    assert len(set(placed.tolist())) == len(placed) and placed.max() < size          # distinct places inside the image
    assert (m[:, nf] >= 0).all()                                                      # every right-hand-side entry has one
    # the structural pattern of the matrix (tg_system_newton_plan) is covered
    from test_newton_plan_cpu import _plan
    plan, pat, tab, _ = _plan(name)
    assert (m[:, :nf][pat] >= 0).all()
    # the writers' packed records
    pairx = pool[offs["wev_pairx"]:offs["wev_pairx"] + ints["n_cmpairs"]].astype(np.uint32)
    seen = set()
    for x in pairx:
        a, b, ab, ba = int(x) & 63, (int(x) >> 6) & 63, (int(x) >> 12) & 0x3FF, (int(x) >> 22) & 0x3FF
        assert a < nd and b < nd and ab == m[a, b] and ba == m[b, a]
        seen.add((a, b)); seen.add((b, a))
    assert seen == {(i, j) for i in range(nd) for j in range(nd) if pat[i, j]}        # exactly the inertial block's entries
    dhx = pool[offs["wev_dhx"]:offs["wev_dhx"] + ints["n_dhr"]]
    for x in dhx:
        kc, ck = (int(x) >> 8) & 0x3FF, (int(x) >> 18) & 0x3FF
        (k, c), (c2, k2) = np.argwhere(m == kc)[0], np.argwhere(m == ck)[0]
        assert k < nd <= c and (c2, k2) == (c, k)
