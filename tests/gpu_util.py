"""Helpers shared by the -m gpu parity tests: run the HIP path through the C ABI and compare with the oracle."""
import numpy as np

import elba_amd
from oracle import pyoracle as po


def oracle_run(packed, off, lens, k, lo, up, threads=4):
    o = po.Oracle(k, lo, up)
    o.count_and_build(packed, off, lens)
    o.spgemm(threads)
    return o


def assert_A_equal(gA, oA):
    assert (gA["M"], gA["N"], gA["Z"]) == (oA["M"], oA["N"], oA["Z"])
    if gA["kmers"] is not None:
        assert (gA["kmers"] == oA["kmers"]).all()
        if oA.get("kmers_lo") is not None:
            assert gA.get("kmers_lo") is not None and (gA["kmers_lo"] == oA["kmers_lo"]).all()
        if oA.get("kmers_lo2") is not None:
            assert gA.get("kmers_lo2") is not None and (gA["kmers_lo2"] == oA["kmers_lo2"]).all()
    assert (gA["colptr"] == oA["colptr"]).all()
    assert (gA["csc_read"] == oA["csc_read"].astype(np.int64)).all()
    assert (gA["csc_pos"] == oA["csc_pos"]).all()
    assert (gA["rowptr"] == oA["rowptr"]).all()
    assert (gA["csr_kid"] == oA["csr_kid"].astype(np.int64)).all()
    assert (gA["csr_pos"] == oA["csr_pos"]).all()


def assert_B_equal(gB, oB):
    assert gB["Y"] == oB["Y"], (gB["Y"], oB["Y"])
    assert (gB["rowptr"] == oB["rowptr"]).all()
    assert (gB["col"] == oB["col"].astype(np.int64)).all()
    bad = np.nonzero(gB["val"] != oB["val"])[0]
    assert len(bad) == 0, (len(bad), gB["val"][bad[:5]], oB["val"][bad[:5]])


def assert_stats_equal(st, o):
    assert st["nnz"] == o.stat("Y")
    assert st["products"] == o.stat("P")
    assert st["nnz_before_prune"] == o.stat("Yraw")
    assert st["nnz_diag"] == o.stat("ndiag")
    assert st["nnz_upper"] == o.stat("nupper")
    assert st["max_numshared"] == o.stat("maxshared")


def gpu_full(packed, off, lens, k, lo, up, **kw):
    e = elba_amd.Engine(k, lo, up, **kw)
    e.set_reads(packed, off, lens)
    ks = e.count_kmers()
    ms = e.create_kmer_matrix()
    st = e.create_seed_matrix()
    return e, ks, ms, st
