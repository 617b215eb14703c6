"""-m gpu: bad / contained read removal + transitive reduction on the GPU (elba_transitive_reduction: a masked min-plus product) against
the CPU oracle, which runs the reference's statements literally with the full min-plus SpGEMM (tests/test_oracle_string_graph.py).
Bit-exact: the entries of S in the reference's output order, every field, the read flags and the counts."""
import numpy as np
import pytest

import elba_amd
import gpu_util as gu
import string_graph_util as sg
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

COUNTS = ("bad_reads", "edges_passed", "contained_reads", "edges_kept", "products", "marked", "removed", "nnz", "iterations")


def _same(e, nreads, rows, cols, vals, cutoff=0.65, fuzz=1000):
    st = e.transitive_reduction(cutoff, fuzz)
    g = e.export_string_graph()
    S, flags, ost = po.string_graph(nreads, rows, cols, vals, cutoff=cutoff, fuzz=fuzz)
    for key in COUNTS:
        assert st[key] == ost[key], (key, st, ost)
    assert st["nreads"] == nreads and st["nedges"] == len(rows)
    assert g["n"] == S["n"] and (g["rows"] == S["rows"]).all() and (g["cols"] == S["cols"]).all()
    for f in po.OVERLAP_DTYPE.names:
        if f != "pad":
            assert (g["vals"][f] == S["vals"][f]).all(), f
    assert (e.export_read_flags(nreads) == flags).all()
    return st


@pytest.mark.parametrize("seed", range(8))
def test_random_overlap_graphs(seed):
    """Fields drawn independently (every direction pair, missing directions, contained and failed pairs, negative suffixes)."""
    rng = np.random.default_rng(100 + seed)
    M = int(rng.integers(3, 400))
    rows, cols, vals = sg.random_overlaps(rng, M, density=float(rng.uniform(0.02, 0.5)), p_fail=0.15, p_contained=0.005 if seed % 2 else 0.0, p_nodir=0.05)
    e = elba_amd.Engine(17, 2, 8)
    e.set_overlaps(M, rows, cols, vals)
    st = _same(e, M, rows, cols, vals, cutoff=float(rng.choice([0.0, 0.5, 0.65])), fuzz=int(rng.choice([0, 300, 1000])))
    assert st["products"] > 0
    e.close()


def test_rows_longer_than_the_lds_stage():
    """A hub read with more neighbours than the kernel stages in LDS (2048): its row is searched in global memory instead."""
    rng = np.random.default_rng(7)
    M = 2600
    rows, cols, vals = [], [], []
    for j in range(1, M):                                    # hub 0 overlaps everyone; a sparse chain among the others
        rows.append(0); cols.append(j); vals.append(sg.ov(int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 3000)), int(rng.integers(0, 3000))))
    for i in range(1, M - 1):
        for j in (i + 1, i + 2):
            if j < M and rng.random() < 0.7:
                rows.append(i); cols.append(j); vals.append(sg.ov(int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 3000)), int(rng.integers(0, 3000))))
    order = np.lexsort((cols, rows))
    rows = np.array(rows, dtype=np.int64)[order]; cols = np.array(cols, dtype=np.int64)[order]; vals = np.array(vals, dtype=po.OVERLAP_DTYPE)[order]
    e = elba_amd.Engine(17, 2, 8)
    e.set_overlaps(M, rows, cols, vals)
    st = _same(e, M, rows, cols, vals, cutoff=0.0)
    assert st["marked"] > 0
    e.close()


def test_empty_graphs_and_argument_checks():
    e = elba_amd.Engine(17, 2, 8)
    with pytest.raises(elba_amd.ElbaError):
        e.transitive_reduction()                             # nothing to work on
    z = np.zeros(0, dtype=po.OVERLAP_DTYPE)
    e.set_overlaps(0, [], [], z)
    _same(e, 0, np.zeros(0, np.int64), np.zeros(0, np.int64), z)
    e.set_overlaps(9, [], [], z)
    _same(e, 9, np.zeros(0, np.int64), np.zeros(0, np.int64), z)
    one = np.array([sg.ov(1, 2, 5, 5)], dtype=po.OVERLAP_DTYPE)
    for rows, cols in (([1], [1]), ([2], [1]), ([0], [9]), ([-1], [3])):
        with pytest.raises(elba_amd.ElbaError):
            e.set_overlaps(9, rows, cols, one)
    two = np.array([sg.ov(1, 2, 5, 5)] * 2, dtype=po.OVERLAP_DTYPE)
    with pytest.raises(elba_amd.ElbaError):
        e.set_overlaps(9, [0, 0], [2, 1], two)               # not ascending
    with pytest.raises(elba_amd.ElbaError):
        e.set_overlaps(9, [0, 0], [2, 2], two)               # a pair twice
    e.set_overlaps(9, [0, 0], [1, 2], two)
    with pytest.raises(elba_amd.ElbaError):
        e.transitive_reduction(0.65, -1)
    # everything failed: every read with an alignment is bad
    failed = np.array([sg.ov(0, 0, 0, 0, passed=0, direction_none=True)] * 2, dtype=po.OVERLAP_DTYPE)
    e.set_overlaps(9, [0, 0], [1, 2], failed)
    st = _same(e, 9, np.array([0, 0]), np.array([1, 2]), failed)
    assert st["bad_reads"] == 3 and st["nnz"] == 0
    e.close()


@pytest.mark.parametrize("err,cutoff", [(0.02, 0.65), (0.10, 0.65), (0.10, 0.0)])
def test_string_graph_of_aligned_reads(err, cutoff):
    """End to end on one context: reads -> B -> x-drop alignments -> string graph, against the oracle fed with the oracle's own alignments."""
    packed, off, lens, info = elba_amd.synth_reads(41, 120000, 14, 3000, 600, error_rate=err, min_len=400)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 12)
    o = gu.oracle_run(packed, off, lens, 17, 2, 12)
    a = e.align_seeds()
    rows, cols, ov, _ = o.align_upper(packed, off, lens, nthreads=8)
    s = _same(e, len(lens), rows, cols, ov, cutoff=cutoff)
    assert a["nalignments"] == len(rows)
    if cutoff == 0.0 and err <= 0.02:
        assert s["marked"] > 0 and 0 < s["nnz"] < 2 * s["edges_kept"]        # real transitive edges get removed
    # a second run on the same context gives the same graph (buffers reused)
    s2 = e.transitive_reduction(cutoff, 1000)
    assert all(s[k] == s2[k] for k in COUNTS)
    e.close()


def test_loaded_overlaps_take_precedence_and_ids_are_global():
    """elba_set_overlaps replaces the context's own alignments as the input; ids are taken as given."""
    rng = np.random.default_rng(5)
    rows, cols, vals = sg.random_overlaps(rng, 60, density=0.3, p_fail=0.1, p_contained=0.0)
    e = elba_amd.Engine(17, 2, 8)
    e.set_overlaps(60, rows, cols, vals)
    _same(e, 60, rows, cols, vals, cutoff=0.0)
    e.set_overlaps(80, rows + 20, cols + 20, vals)           # the same graph on reads 20..79 of a larger set
    st = e.transitive_reduction(0.0, 1000)
    g = e.export_string_graph()
    S, _, _ = po.string_graph(60, rows, cols, vals, cutoff=0.0)
    assert (g["rows"] == S["rows"] + 20).all() and (g["cols"] == S["cols"] + 20).all() and st["nnz"] == S["n"]
    e.close()


@pytest.mark.parametrize("seed", range(4))
def test_gpu_equals_the_literal_python_restatement(seed):
    """Not through the C oracle: the GPU's S against the pure-Python statement-by-statement restatement of src/main.cpp:305-312 and
    src/TransitiveReduction.cpp (tests/string_graph_util.py), which shares no code with either."""
    rng = np.random.default_rng(900 + seed)
    M = int(rng.integers(5, 60))
    rows, cols, vals = sg.random_overlaps(rng, M, density=float(rng.uniform(0.1, 0.6)), p_fail=0.1, p_contained=0.02 if seed % 2 else 0.0, p_nodir=0.05, suffix_range=2500)
    cutoff, fuzz = float(rng.choice([0.0, 0.65])), int(rng.choice([0, 1000]))
    e = elba_amd.Engine(17, 2, 8)
    e.set_overlaps(M, rows, cols, vals)
    st = e.transitive_reduction(cutoff, fuzz)
    g = e.export_string_graph()
    want, wflags, wst = sg.python_string_graph(M, rows, cols, vals, cutoff, fuzz)
    assert [(int(r), int(c)) for r, c in zip(g["rows"], g["cols"])] == [(r, c) for r, c, _ in want]
    for a, (_, _, v) in enumerate(want):
        for f in po.OVERLAP_DTYPE.names:
            if f != "pad":
                assert g["vals"][a][f] == v[f], (a, f)
    assert list(e.export_read_flags(M)) == list(wflags)
    for key in ("bad_reads", "edges_passed", "contained_reads", "edges_kept", "products", "marked", "removed", "nnz", "iterations"):
        assert st[key] == wst[key], key
    e.close()
