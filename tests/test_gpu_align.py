"""-m gpu: x-drop seed-and-extend on the GPU (elba_align_seeds) against the CPU oracle, which is itself pinned to the reference's own
XDropAligner.cpp (tests/test_oracle_xdrop.py).  Bit-exact: every field of every overlap."""
import os

import numpy as np
import pytest

import elba_amd
import gpu_util as gu
import util
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
G = util.GOLDEN


def _compare(e, o, packed, off, lens, params, threads=8):
    mat, mis, gap, x = params
    st = e.align_seeds(mat, mis, gap, x)
    g = e.export_overlaps()
    rows, cols, ov, cells = o.align_upper(packed, off, lens, mat, mis, gap, x, nthreads=threads)
    assert st["nalignments"] == len(rows) == g["n"]
    assert (g["rows"] == rows).all() and (g["cols"] == cols).all()
    for f in ov.dtype.names:
        if f == "pad":
            continue
        bad = np.nonzero(g["vals"][f] != ov[f])[0]
        assert len(bad) == 0, (params, f, len(bad), bad[:5], g["vals"][bad[:5]], ov[bad[:5]])
    assert st["cells"] == cells, (st["cells"], cells)
    assert st["passed"] == int(ov["passed"].sum())
    return st


@pytest.mark.parametrize("name,k,lo,up", [("small_err", 17, 2, 8), ("small_clean", 17, 2, 8), ("small_clean", 31, 3, 12)])
@pytest.mark.parametrize("params", [(1, -1, -1, 15), (1, -2, -3, 30), (2, -3, -2, 7), (1, -1, -1, 0)])
def test_align_matches_oracle_on_golden_reads(name, k, lo, up, params):
    packed, off, lens = po.pack_reads(util.read_fasta(os.path.join(G, name + ".fa")))
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up)
    o = gu.oracle_run(packed, off, lens, k, lo, up)
    _compare(e, o, packed, off, lens, params)
    e.close()


def test_align_long_noisy_reads_default_parameters():
    """8 kb reads with 15 % errors (the bench workload's kind): thousands of antidiagonals per extension, window slides, early x-drops."""
    packed, off, lens, info = elba_amd.synth_reads(11, 150000, 20, 8000, 1500, error_rate=0.15, min_len=1000)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8)
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    a = _compare(e, o, packed, off, lens, (1, -1, -1, 15))
    assert a["nalignments"] > 500 and a["passed"] > 0
    e.close()


@pytest.mark.parametrize("tiers", ["1", "2", "4", "8", "1248", "24"])
def test_align_wide_bands_escalate_and_reach_the_strided_kernel(tiers):
    """x-drop 90 on accurate reads: the band outgrows 64 (128, 256) columns; those extensions are redone on the next tier and, past
    the last one, by the strided kernel — same results whatever the tiers."""
    packed, off, lens, info = elba_amd.synth_reads(12, 60000, 12, 3000, 400, error_rate=0.02, min_len=500)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 40, options={"aln_tiers": int(tiers)})
    o = gu.oracle_run(packed, off, lens, 17, 2, 40)
    a = _compare(e, o, packed, off, lens, (1, -1, -1, 90))
    if tiers == "1":
        assert a["extensions_strided"] > 0
    b = _compare(e, o, packed, off, lens, (1, -1, -1, 900))        # bands of more than 512 columns: beyond every register tier
    assert b["extensions_strided"] > 0
    e.close()


def test_align_requires_reads_and_seed_matrix():
    e = elba_amd.Engine(17, 2, 8)
    with pytest.raises(elba_amd.ElbaError):
        e.align_seeds()
    e.close()
