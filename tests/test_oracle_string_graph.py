"""CPU: the oracle's restatement of src/main.cpp:305-312 (bad / contained read removal + TransitiveReduction).  The reference holds no
fixture for this stage (the matrix algebra lives in CombBLAS, absent): the oracle is held against hand-derived known answers and
against an independent pure-Python restatement written from the same reference lines (dense 4-slot matrices, the loop followed
literally) on random graphs."""
import numpy as np
import pytest

import string_graph_util as sg
from oracle import pyoracle as po

INF = 2**31 - 1


def test_three_reads_on_a_line_lose_the_long_edge():
    """r0 = [0,1000), r1 = [300,1300), r2 = [600,1600), all forward.  extend_overlap (src/Overlap.cpp:55-61) gives each pair
    direction 1 / directionT 2 and suffix = the target's overhang; the walk r0 -> r1 -> r2 (300 + 300) explains r0 -> r2 (600)."""
    vals = np.array([sg.ov(1, 2, 300, 300), sg.ov(1, 2, 600, 600), sg.ov(1, 2, 300, 300)], dtype=po.OVERLAP_DTYPE)
    S, flags, st = po.string_graph(3, [0, 0, 1], [1, 2, 2], vals)
    assert list(zip(S["rows"], S["cols"])) == [(1, 0), (0, 1), (2, 1), (1, 2)]            # columns ascending, rows within
    assert list(S["vals"]["direction"]) == [2, 1, 2, 1] and list(S["vals"]["suffix"]) == [300] * 4
    assert st == dict(bad_reads=0, edges_passed=3, contained_reads=0, edges_kept=3, products=12, nnzN=2, marked=2, removed=2, nnz=4, iterations=2)
    assert not flags.any()


def test_fuzz_decides_a_near_miss():
    """The direct edge is 1500 longer than the two-edge walk claims: suffix + FUZZ >= path only once FUZZ >= -1500 ... i.e. always here;
    the other way round (direct edge 1500 SHORTER than the walk) it is removed only when FUZZ >= 1500."""
    vals = np.array([sg.ov(1, 2, 1000, 1000), sg.ov(1, 2, 500, 500), sg.ov(1, 2, 1000, 1000)], dtype=po.OVERLAP_DTYPE)     # walk 2000, direct 500
    for fuzz, nnz in ((1000, 6), (1499, 6), (1500, 4), (3000, 4)):
        S, _, st = po.string_graph(3, [0, 0, 1], [1, 2, 2], vals, fuzz=fuzz)
        assert st["nnz"] == nnz, fuzz


def test_walk_must_leave_a_read_by_its_other_end():
    """MinPlusSR::multiply refuses t2 == h1 (include/TransitiveReduction.hpp:98-99).  R(0,1) has head bit 1; make R(1,2) leave read 1 by
    the same end (tail bit 1: direction 2 or 3) and nothing explains R(0,2)."""
    vals = np.array([sg.ov(1, 2, 300, 300), sg.ov(1, 2, 600, 600), sg.ov(3, 3, 300, 300)], dtype=po.OVERLAP_DTYPE)
    S, _, st = po.string_graph(3, [0, 0, 1], [1, 2, 2], vals)
    assert st["marked"] == 0 and st["nnz"] == 6 and st["iterations"] == 1


def test_slot_must_match_the_direct_edges_direction():
    """The walk 0 -> 1 -> 2 lands in slot 2*t1 + h2 = 1; a direct edge with direction 0 reads slot 0 and survives."""
    vals = np.array([sg.ov(1, 2, 300, 300), sg.ov(0, 0, 600, 600), sg.ov(1, 2, 300, 300)], dtype=po.OVERLAP_DTYPE)
    S, _, st = po.string_graph(3, [0, 0, 1], [1, 2, 2], vals)
    assert st["marked"] == 0 and st["nnz"] == 6


def test_one_orientation_marked_removes_both():
    """I += I^T (src/TransitiveReduction.cpp:72-75): make only the (2,0) orientation transitive by giving the transposed side a small suffix."""
    vals = np.array([sg.ov(1, 2, 300, 300), sg.ov(1, 2, 5000, 600), sg.ov(1, 2, 300, 300)], dtype=po.OVERLAP_DTYPE)
    S, _, st = po.string_graph(3, [0, 0, 1], [1, 2, 2], vals)
    # (0,2): 5000 + 1000 >= 600 marked; (2,0): direction 2, suffix 600, walk 2 -> 1 -> 0 = 600: marked as well
    assert st["marked"] == 2
    vals[1] = sg.ov(1, 2, 600, -2000)                    # (2,0) now has suffix -2000: -2000 + 1000 < 600, only (0,2) is marked
    S, _, st = po.string_graph(3, [0, 0, 1], [1, 2, 2], vals)
    assert st["marked"] == 1 and st["removed"] == 2 and st["nnz"] == 4


def test_bad_reads_by_the_reference_ratio():
    """find_bad_reads: (passed + 1) / (aligned + 1) <= cutoff.  Read 0: 1 of 3 pass -> 2/4 = 0.5 <= 0.65 bad.  Read 3: 1 of 1 -> 1.0."""
    vals = np.array([sg.ov(1, 2, 10, 10, passed=0, direction_none=True), sg.ov(1, 2, 10, 10, passed=0, direction_none=True), sg.ov(1, 2, 10, 10),
                     sg.ov(1, 2, 10, 10)], dtype=po.OVERLAP_DTYPE)
    S, flags, st = po.string_graph(5, [0, 0, 0, 3], [1, 2, 3, 4], vals)
    # read 1, 2: 0 of 1 -> 1/2 bad; read 0 bad; read 3: 2 of 2; read 4: 1 of 1
    assert list(flags) == [1, 1, 1, 0, 0] and st["bad_reads"] == 3 and st["edges_passed"] == 1 and st["nnz"] == 2
    # a read nobody aligned to: (0 + 1) / (0 + 1) = 1 > cutoff: not bad; cutoff 1.0 makes everyone bad
    S, flags, st = po.string_graph(6, [0, 0, 0, 3], [1, 2, 3, 4], vals, cutoff=1.0)
    assert st["bad_reads"] == 6 and st["nnz"] == 0


def test_contained_reads_are_taken_from_rows_and_columns():
    """containedQ removes the row's read, containedT the column's (src/main.cpp:575-581); decided on what survived the first prune."""
    vals = np.array([sg.ov(-1, -1, 0, 0, cq=1), sg.ov(1, 2, 10, 10), sg.ov(-1, -1, 0, 0, ct=1), sg.ov(1, 2, 10, 10)], dtype=po.OVERLAP_DTYPE)
    S, flags, st = po.string_graph(5, [0, 1, 1, 2], [1, 2, 3, 4], vals, cutoff=0.0)
    assert list(flags) == [2, 0, 0, 2, 0] and st["contained_reads"] == 2 and st["edges_kept"] == 2 and st["nnz"] == 4


def test_empty_and_degenerate_inputs():
    e = np.zeros(0, dtype=po.OVERLAP_DTYPE)
    S, flags, st = po.string_graph(0, [], [], e)
    assert S["n"] == 0 and st["nnz"] == 0 and st["iterations"] == 1
    S, flags, st = po.string_graph(7, [], [], e)
    assert S["n"] == 0 and len(flags) == 7 and not flags.any()
    with pytest.raises(ValueError):
        po.string_graph(3, [1], [1], np.array([sg.ov(1, 2, 1, 1)], dtype=po.OVERLAP_DTYPE))
    with pytest.raises(ValueError):
        po.string_graph(3, [1], [3], np.array([sg.ov(1, 2, 1, 1)], dtype=po.OVERLAP_DTYPE))


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_the_literal_python_restatement_on_random_graphs(seed):
    rng = np.random.default_rng(seed)
    M = int(rng.integers(3, 40))
    rows, cols, vals = sg.random_overlaps(rng, M, density=float(rng.uniform(0.1, 0.7)), p_fail=0.15, p_contained=0.03 if seed % 3 else 0.0,
                                          p_nodir=0.05, suffix_range=2500)
    cutoff = float(rng.choice([0.0, 0.5, 0.65]))
    fuzz = int(rng.choice([0, 300, 1000]))
    S, flags, st = po.string_graph(M, rows, cols, vals, cutoff=cutoff, fuzz=fuzz)
    want, wflags, wst = sg.python_string_graph(M, rows, cols, vals, cutoff, fuzz)
    assert list(flags) == list(wflags)
    assert [(int(r), int(c)) for r, c in zip(S["rows"], S["cols"])] == [(r, c) for r, c, _ in want]
    for a, (_, _, v) in enumerate(want):
        for f in po.OVERLAP_DTYPE.names:
            assert S["vals"][a][f] == v[f], (a, f)
    for key in ("bad_reads", "edges_passed", "contained_reads", "edges_kept", "nnzN", "marked", "removed", "nnz", "iterations", "products"):
        assert st[key] == wst[key], key
