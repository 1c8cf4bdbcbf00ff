"""CPU-only: the list schedule that deals the joints of a tree sweep out to a team of wavefronts
(loikb_sweep_schedule, built by build_team_schedule in loik_amd/csrc/loik_host.hip).  Checked as a schedule, i.e.
against the dependency structure of the reference's recursions: BwdPassOptimizedVisitor accumulates into the parent
(loik-loid-optimized.hxx:31-81, leaf -> root), FwdPass2OptimizedVisitor reads the parent's velocity
(loik-loid-optimized.hxx:102-163, root -> leaf)."""
import numpy as np
import pytest

import loik_amd
from loik_amd import capi
from helpers import random_tree

SF_IN_REG, SF_OUT_REG, SF_OUT_LDS, SF_VPAR_REG = 1, 2, 4, 8


def trees():
    out = {"talos32": np.asarray(loik_amd.builtin_model("talos32").parents),
           "panda7": np.asarray(loik_amd.builtin_model("panda7").parents)}
    for nb in (6, 17, 40, 63):
        out["tree%d" % nb] = np.asarray(random_tree(nb, nb).parents)
    out["star"] = np.array([0] + [0] * 9)          # nine children of the universe
    out["broom"] = np.array([0, 0, 1, 2, 3, 3, 3, 3, 3])
    return out


def positions(joint):
    pos = {}
    for w in range(joint.shape[0]):
        for t in range(joint.shape[1]):
            j = int(joint[w, t])
            if j:
                assert j not in pos, "joint %d scheduled twice" % j
                pos[j] = (w, t)
    return pos


@pytest.mark.parametrize("team", [1, 2, 3, 4])
@pytest.mark.parametrize("name", sorted(trees()))
def test_leaf_to_root_schedule(name, team):
    parents = trees()[name]
    nj = len(parents)
    joint, flags, slot, nslots = capi.sweep_schedule(parents, team, 0)
    pos = positions(joint)
    assert sorted(pos) == list(range(1, nj))
    children = {i: [c for c in range(1, nj) if parents[c] == i] for i in range(nj)}
    for j, (w, t) in pos.items():
        for c in children[j]:
            assert pos[c][1] < t, "child %d of %d not handled at an earlier step" % (c, j)
    # hand-over of every contribution: registers (same wavefront, parent is its next joint) or an LDS slot
    nexts = {}
    for w in range(team):
        seq = [int(j) for j in joint[w] if j]
        for a, b in zip(seq, seq[1:]):
            nexts[a] = b
    busy = {}  # slot -> step until which it is occupied (inclusive)
    for t in range(joint.shape[1]):
        for w in range(team):
            j = int(joint[w, t])
            if not j:
                continue
            f = int(flags[w, t])
            p = int(parents[j])
            reg_children = [c for c in children[j] if nexts.get(c) == j and pos[c][0] == w]
            assert bool(f & SF_IN_REG) == (len(reg_children) > 0)
            assert len(reg_children) <= 1
            if p == 0:
                assert not (f & (SF_OUT_REG | SF_OUT_LDS))
            elif nexts.get(j) == p and pos[p][0] == w:
                assert f & SF_OUT_REG and not (f & SF_OUT_LDS)
            else:
                assert f & SF_OUT_LDS and not (f & SF_OUT_REG)
                s = int(slot[w, t])
                assert 0 <= s < nslots
                assert busy.get(s, -1) < t, "LDS slot %d rewritten before its reader ran" % s
                busy[s] = pos[p][1]
    # critical path: no schedule can beat the tree depth, a single wavefront needs one step per joint
    depth = np.zeros(nj, dtype=int)
    for i in range(1, nj):
        depth[i] = depth[parents[i]] + 1
    assert joint.shape[1] >= depth.max()
    if team == 1:
        assert joint.shape[1] == nj - 1 and not (joint == 0).any()
    assert joint.shape[1] <= nj - 1


@pytest.mark.parametrize("team", [1, 2, 4])
@pytest.mark.parametrize("name", sorted(trees()))
def test_root_to_leaf_schedule(name, team):
    parents = trees()[name]
    nj = len(parents)
    joint, flags, slot, nvslots = capi.sweep_schedule(parents, team, 1)
    pos = positions(joint)
    assert sorted(pos) == list(range(1, nj))
    prev = {}
    for w in range(team):
        seq = [int(j) for j in joint[w] if j]
        for a, b in zip(seq, seq[1:]):
            prev[b] = a
    last_reader = {}
    for j, (w, t) in pos.items():
        p = int(parents[j])
        if p:
            assert pos[p][1] < t
            in_reg = bool(int(flags[w, t]) & SF_VPAR_REG)
            assert in_reg == (prev.get(j) == p)
            if not in_reg:
                pw, pt = pos[p]
                assert int(flags[pw, pt]) & SF_OUT_LDS, "parent %d never publishes its velocity" % p
                last_reader[p] = max(last_reader.get(p, -1), t)
    # a velocity slot is not recycled before its last reader
    for p, tl in last_reader.items():
        pw, pt = pos[p]
        s = int(slot[pw, pt])
        assert 0 <= s < nvslots
        for q, tq in last_reader.items():
            if q != p and int(slot[pos[q][0], pos[q][1]]) == s:
                a0, a1 = pt, tl
                b0, b1 = pos[q][1], tq
                assert a1 < b0 or b1 < a0, "velocity slot %d shared by overlapping lifetimes" % s


def test_talos_team_of_four_reaches_the_critical_path():
    """legs, arms and head are independent chains: 4 wavefronts walk Talos in 10 steps instead of 32"""
    parents = np.asarray(loik_amd.builtin_model("talos32").parents)
    up = capi.sweep_schedule(parents, 4, 0)[0]
    down = capi.sweep_schedule(parents, 4, 1)[0]
    assert up.shape[1] == 10 and down.shape[1] == 10
    assert capi.sweep_schedule(parents, 1, 0)[0].shape[1] == 32


def test_argument_errors():
    parents = np.array([0, 0, 1, 1], dtype=np.int32)
    with pytest.raises(capi.LoikError):
        capi.sweep_schedule(parents, 9, 0)
    with pytest.raises(capi.LoikError):
        capi.sweep_schedule(np.array([0, 0, 3, 1], dtype=np.int32), 2, 0)
