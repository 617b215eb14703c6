"""-m gpu: BASELINE.json configs[1] at FULL size (the bench workload: 16 893 reads, 138.9 M k-mer instances) checked through
size-independent properties of the domain (the oracle's k-mer stage would need ~30 s of one core at this size) AND, since round 5, every entry of B
against the oracle's product of the GPU-built A (test_every_entry_of_B_equals_the_oracle):
  * pattern symmetry and numshared symmetry (B = A·Aᵀ),  Y = diag + 2·strict-upper,  columns strictly ascending in every row;
  * every stored seed names the same k-mer in both reads, forward or reverse complement (the reference's test.py:57-65);
  * seeds[0] <= seeds[1] in the canonical order when both lie on one k-mer id is not observable, but seeds[0] == seeds[1] iff ... numshared >= 2 always;
  * idempotence: a second run returns the identical matrix;  DCSC export is the transpose walk of the CSR export.
"""
import numpy as np
import pytest

import elba_amd
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    packed, off, lens, info = elba_amd.synth_reads(1, 4_640_000, 30.0, 8240.0, 2000.0, error_rate=0.15, min_len=1000)
    e = elba_amd.Engine(17, 2, 8)
    e.set_reads(packed, off, lens)
    ks = e.count_kmers(); ms = e.create_kmer_matrix(); st = e.create_seed_matrix()
    B = e.export_csr()
    yield dict(e=e, packed=packed, off=off, lens=lens, ks=ks, ms=ms, st=st, B=B)
    e.close()


def test_counts_are_consistent(full):
    ks, ms, st, B = full["ks"], full["ms"], full["st"], full["B"]
    assert ks["nreads"] == 16893 and ks["instances"] == int(np.maximum(full["lens"].astype(np.int64) - 16, 0).sum())
    assert ms["nnz"] == ks["entries"] and ms["ncols"] == ks["reliable"]
    assert st["nnz"] == B["Y"] == int(B["rowptr"][-1])
    assert st["nnz"] == st["nnz_diag"] + 2 * st["nnz_upper"]
    assert st["nnz_before_prune"] >= st["nnz"] and st["products"] >= st["nnz_before_prune"]
    h = full["e"].kmer_histogram()
    assert h[:2].sum() == 0 and h.sum() == ks["reliable"] and int((h * np.arange(len(h))).sum()) == ks["entries"]
    assert st["products"] == int((h * np.arange(len(h)) ** 2).sum())          # P = sum_k c_k^2


def test_every_entry_of_B_equals_the_oracle(full):
    """BASELINE configs[1] WHOLE, entry by entry (round 5): the oracle multiplies the very matrix the GPU built (its columns, copied off the device) on the
    host's cores — 42 M products — and every row pointer, column and seed field of the GPU's B is compared with it."""
    import gpu_util as gu
    assert gu.assert_whole_B_equals_oracle(full["e"], 17, 2, 8, full["st"]) == full["B"]["Y"]


def test_pattern_and_numshared_are_symmetric_and_rows_sorted(full):
    B = full["B"]
    M = B["M"]
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(B["rowptr"]))
    cols = B["col"]
    assert ((np.diff(cols) > 0) | (np.diff(rows) > 0)).all()                 # strictly ascending columns within a row
    key = rows * M + cols
    tkey = cols * M + rows
    order = np.argsort(tkey, kind="stable")
    assert (tkey[order] == key).all()                                         # pattern symmetric (key is sorted)
    assert (B["val"]["numshared"][order] == B["val"]["numshared"]).all()      # numshared(i,j) == numshared(j,i)
    assert (B["val"]["numshared"] >= 2).all()
    # B(j,i) is B(i,j) with the two positions of each seed exchanged — exactly: the canonical seeds are the minimum / maximum over a cross
    # product of positions per shared k-mer
    v, vt = B["val"], B["val"][order]
    assert (v["q0"] == vt["t0"]).all() and (v["t0"] == vt["q0"]).all() and (v["q1"] == vt["t1"]).all() and (v["t1"] == vt["q1"]).all()


def test_sampled_seeds_are_genuine_shared_kmers(full):
    L = po.lib()
    B, packed, off, lens = full["B"], full["packed"], full["off"], full["lens"]
    rows = np.repeat(np.arange(B["M"], dtype=np.int64), np.diff(B["rowptr"]))
    rng = np.random.default_rng(0)
    bad = 0
    for e in rng.choice(B["Y"], size=20000, replace=False):
        i, j, v = int(rows[e]), int(B["col"][e]), B["val"][e]
        for (q, t) in ((v["q0"], v["t0"]), (v["q1"], v["t1"])):
            bad += not L.orc_seed_is_valid(packed.ctypes.data + int(off[i]), int(lens[i]), packed.ctypes.data + int(off[j]), int(lens[j]), int(q), int(t), 17)
    assert bad == 0


def test_second_run_is_identical_and_dcsc_is_the_transpose_walk(full):
    e, B = full["e"], full["B"]
    e.create_seed_matrix()
    B2 = e.export_csr()
    assert (B["rowptr"] == B2["rowptr"]).all() and (B["col"] == B2["col"]).all() and (B["val"] == B2["val"]).all()
    M = B["M"]
    d = e.export_dcsc(0, M, 0, M)
    assert d["nnz"] == B["Y"]
    colidx = np.repeat(d["jc"], np.diff(d["cp"]))
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(B["rowptr"]))
    order = np.lexsort((rows, B["col"]))                                      # CSR entries in column-major order
    assert (colidx == B["col"][order]).all() and (d["ir"] == rows[order]).all() and (d["numx"] == B["val"][order]).all()
