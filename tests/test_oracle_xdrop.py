"""CPU: the oracle's x-drop restatement (oracle/elba_oracle.c, f1) against golden vectors produced by the REFERENCE's own
src/XDropAligner.cpp (tests/golden/make_golden.py through oracle/_ref), and live against oracle/_ref where it is built."""
import os

import numpy as np
import pytest

import util
from oracle import pyoracle as po

G = util.GOLDEN


def _vectors(name):
    rows = []
    for line in open(os.path.join(G, name)):
        if line[0] == "#":
            continue
        rows.append([int(x) for x in line.split()])
    return rows


@pytest.mark.parametrize("fa,k", [("small_err", 17), ("small_clean", 17)])
def test_oracle_xdrop_matches_reference_vectors(fa, k):
    seqs = util.read_fasta(os.path.join(G, fa + ".fa"))
    buf, off, lens = po.pack_reads(seqs)
    vec = _vectors("xdrop_%s_k%d.txt" % (fa, k))
    assert len(vec) > 50
    for v in vec:
        i, j, q0, t0, mat, mis, gap, x = v[:8]
        got = po.xdrop(buf[int(off[i]):], int(lens[i]), buf[int(off[j]):], int(lens[j]), q0, t0, k, mat, mis, gap, x)
        assert list(got) == v[8:], (v[:8], got, v[8:])


def test_oracle_xdrop_live_against_reference_build():
    R = po.ref_lib(17)
    if R is None or not hasattr(R, "ref_xdrop"):
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(3)
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(17, 2, 8); o.count_and_build(buf, off, lens); o.spgemm(1)
    B = o.B()
    rows = np.repeat(np.arange(B["M"]), np.diff(B["rowptr"]))
    up = np.nonzero(rows < B["col"])[0]
    for e in rng.choice(up, size=min(300, len(up)), replace=False):
        i, j = int(rows[e]), int(B["col"][e])
        mat, mis, gap, x = int(rng.integers(1, 3)), -int(rng.integers(1, 4)), -int(rng.integers(1, 4)), int(rng.integers(0, 60))
        a = po.xdrop(buf[int(off[i]):], int(lens[i]), buf[int(off[j]):], int(lens[j]), int(B["val"][e]["q0"]), int(B["val"][e]["t0"]), 17, mat, mis, gap, x)
        b = po.ref_xdrop(R, buf[int(off[i]):], int(lens[i]), buf[int(off[j]):], int(lens[j]), int(B["val"][e]["q0"]), int(B["val"][e]["t0"]), mat, mis, gap, x)
        assert a == b, ((i, j, mat, mis, gap, x), a, b)
    # rejected seeds: out of range, the (0,0) rule, not a shared k-mer
    for _ in range(500):
        i, j = (int(v) for v in rng.integers(0, len(seqs), 2))
        q0, t0 = int(rng.integers(-2, lens[i] + 2)), int(rng.integers(-2, lens[j] + 2))
        a = po.xdrop(buf[int(off[i]):], int(lens[i]), buf[int(off[j]):], int(lens[j]), q0, t0, 17)
        b = po.ref_xdrop(R, buf[int(off[i]):], int(lens[i]), buf[int(off[j]):], int(lens[j]), q0, t0)
        assert a == b


def test_overlap_fields_follow_the_classification():
    """Overlap::extend_overlap (src/Overlap.cpp:44-72): direction / suffix / contained flags per OverlapClass."""
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(17, 2, 8); o.count_and_build(buf, off, lens); o.spgemm(1)
    rows, cols, ov, cells = o.align_upper(buf, off, lens)
    assert len(rows) == o.stat("nupper") and cells > 0
    for a in range(len(rows)):
        v = ov[a]
        lq, lt = int(lens[rows[a]]), int(lens[cols[a]])
        assert v["passed"] == (v["kind"] != 0)
        assert v["containedQ"] == (v["kind"] == 1) and v["containedT"] == (v["kind"] == 2)
        if v["kind"] in (3, 4):
            begTr = lt - v["endT"] if v["rc"] else v["begT"]
            endTr = lt - v["begT"] if v["rc"] else v["endT"]
            if v["kind"] == 3:
                assert v["direction"] == (0 if v["rc"] else 1) and v["suffix"] == (lt - endTr) - (lq - v["endQ"])
            else:
                assert v["direction"] == (3 if v["rc"] else 2) and v["suffix"] == begTr - v["begQ"]
        else:
            assert v["direction"] == -1 and v["directionT"] == -1
