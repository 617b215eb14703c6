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


def host_copy(dev_ptr, count, dtype):
    """`count` items of `dtype` from a raw device pointer into a fresh numpy array (hipMemcpy through ctypes: plumbing for the checker)."""
    import ctypes
    import torch  # noqa: F401  (the process's HIP runtime: the system libamdhip64.so resolves against the ROCr torch has loaded)
    hip = ctypes.CDLL("libamdhip64.so")
    out = np.empty(int(count), dtype=dtype)
    if count:
        rc = hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(int(dev_ptr)), ctypes.c_size_t(out.nbytes), 2)
        if rc != 0:
            raise RuntimeError("hipMemcpy D2H failed: %d" % rc)
    return out


SEED_DTYPE = np.dtype([("q0", "<u4"), ("t0", "<u4"), ("q1", "<u4"), ("t1", "<u4"), ("numshared", "<i4")])


def assert_whole_B_equals_oracle(e, k, lo, up, st=None, threads=None):
    """EVERY entry of the engine's B against the oracle's, at any size the host holds: the columns of A leave the device as they are (u32 pointers,
    read << 32 | pos — the reference's AT), the oracle derives CSR from them (orc_set_csc), runs create_seed_matrix's region on `threads` host threads
    (orc_spgemm: the literal left fold) and compares row pointers, columns and all five seed fields (orc_compare_B).  What bench.py does for the
    headline matrix (`parity_vs_oracle_full`), as a test.  Returns the number of entries compared."""
    import os
    if threads is None:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        threads = max(1, min(threads, 32))
    v = e.device_view()
    colptr = host_copy(v["a_colptr"], v["N"] + 1, np.uint32)
    csc = host_copy(v["a_csc"], v["Z"], np.uint64)
    o = po.Oracle(k, lo, up)
    o.set_csc(int(v["M"]), int(v["N"]), colptr, csc, threads)
    del colptr, csc
    o.spgemm(threads)
    rowptr = host_copy(v["b_rowptr"], v["M"] + 1, np.int64)
    col = host_copy(v["b_col"], v["Y"], np.uint32)
    val = host_copy(v["b_val"], v["Y"], SEED_DTYPE)
    assert int(v["Y"]) == o.stat("Y"), (int(v["Y"]), o.stat("Y"))
    ndiff = o.compare_B(rowptr, col, val, threads)
    assert ndiff == 0, ndiff
    if st is not None:
        assert_stats_equal(st, o)
    return int(v["Y"])
