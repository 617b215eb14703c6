"""Seeded random cases against the oracle, through the C ABI: matrices handed over as triples (ownership hints, reads that hold a k-mer
twice, columns of 1 .. 100 entries, positions beyond 16 bits, empty and tiny rows) and whole read sets (k = 7 .. 95, LOWER / UPPER incl.
UPPER > 62, error-free to 12 % error, reads of 10 bases to a few thousand).  Bit-exact, cold and warm calls."""
import numpy as np
import pytest

import elba_amd
import gpu_util as gu
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(24))
def test_random_matrix_against_oracle(case):
    rng = np.random.default_rng(1000 + case)
    M = int(rng.choice([5, 64, 300, 3000, 40000]))
    ncol = int(rng.choice([1, 50, 2000, 30000]))
    maxlen = int(rng.choice([2, 3, 8, 20, 64, 100]))
    maxpos = int(rng.choice([100, 5000, 65535, 200000]))
    hot = rng.choice(M, min(M, int(rng.choice([3, 40, 500]))), replace=False)
    rows, cols, vals = [], [], []
    for c in range(ncol):
        n = int(rng.integers(1, maxlen + 1))
        r = np.sort(rng.choice(hot, n, replace=bool(rng.random() < 0.3) or n > len(hot)))
        p = rng.integers(0, maxpos + 1, n)
        for rr in np.unique(r):                          # (read, pos) pairs of a column are distinct, positions ascending per read
            m = r == rr
            p[m] = np.sort(rng.choice(maxpos + 1, int(m.sum()), replace=False))
        rows.append(r); cols.append(np.full(n, c)); vals.append(p)
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
    up = max(2, maxlen)
    o = po.Oracle(17, 2, up)
    o.set_triples(M, ncol, rows, cols, vals)
    o.spgemm(8)
    oB = o.B()
    e = elba_amd.Engine(17, 2, up)
    e.set_kmer_matrix(M, ncol, rows, cols, vals)
    st = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), oB)
    gu.assert_stats_equal(st, o)
    e.set_option("overlap_cold_calls", 1)
    st = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), oB)
    gu.assert_stats_equal(st, o)
    e.close()


@pytest.mark.parametrize("case", range(24))
def test_random_read_set_against_oracle(case):
    rng = np.random.default_rng(5000 + case)
    k = int(rng.choice([7, 11, 15, 17, 21, 25, 31, 33, 45, 63, 65, 95]))
    lo = int(rng.choice([2, 2, 3]))
    up = max(lo, int(rng.choice([lo, 4, 8, 20, 50, 70])))
    genome = int(rng.choice([500, 20000, 300000]))
    depth = float(rng.choice([3, 10, 25]))
    avg = float(rng.choice([60, 400, 3000]))
    err = float(rng.choice([0.0, 0.02, 0.12]))
    packed, off, lens, _ = elba_amd.synth_reads(100 + case, genome, depth, avg, avg / 4, error_rate=err, min_len=max(10, int(avg / 6)))
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up)
    o = gu.oracle_run(packed, off, lens, k, lo, up, threads=8)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()


@pytest.mark.parametrize("M,ncol,L", [(6000, 20000, 40), (20000, 60000, 60)])
def test_dense_matrix_with_thousands_of_partners_per_row(M, ncol, L):
    """Columns of L random reads out of M: every row meets most other rows.  The dense path (smaller row owns) starts on the small table tiers,
    abandons and escalates; the 8192-slot tier and the HBM tables run the general kernel under the same ownership rule.  Equal to the
    oracle, cold and warm."""
    rng = np.random.default_rng(M)
    rows, cols, vals = [], [], []
    for c in range(ncol):
        r = np.sort(rng.choice(M, L, replace=False))
        rows.append(r); cols.append(np.full(L, c)); vals.append(rng.integers(0, 60000, L))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
    o = po.Oracle(17, 2, 64)
    o.set_triples(M, ncol, rows, cols, vals)
    o.spgemm(8)
    oB = o.B()
    e = elba_amd.Engine(17, 2, 64)
    e.set_kmer_matrix(M, ncol, rows, cols, vals)
    st = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), oB)
    gu.assert_stats_equal(st, o)
    assert st["rows_escalated"] > 0 or st["rows_global"] > 0
    e.set_option("overlap_cold_calls", 1)
    st = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), oB)
    gu.assert_stats_equal(st, o)
    e.close()


def test_short_rows_many_cold_calls_agree_with_the_oracle():
    """Rows of a few hundred entries finish so quickly that a workgroup's wavefronts can drift a whole phase apart wherever no barrier holds them
    together.  (A version of the numeric kernel without a barrier at the start of a row let one wavefront read the NEXT row's id — published by a
    faster one — once in ~20 calls on this input: two wavefronts then filled one table with the products of two rows.  Single calls pass such a
    bug nine times out of ten; 120 cold calls do not.)  Every call: nnz, products and the statistics of the oracle; B itself on three of them."""
    packed, off, lens, _ = elba_amd.synth_reads(114, 300000, 25.0, 400.0, 100.0, error_rate=0.02, min_len=66)
    o = gu.oracle_run(packed, off, lens, 25, 2, 70, threads=8)
    oB = o.B()
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 25, 2, 70)
    e.set_option("overlap_cold_calls", 1)
    for it in range(120):
        gu.assert_stats_equal(st, o)
        if it % 40 == 0:
            gu.assert_B_equal(e.export_csr(), oB)
        st = e.create_seed_matrix()
    e.close()


@pytest.mark.parametrize("case", range(20))
def test_random_device_triples_through_the_bucket_kernels(case):
    """elba_set_kmer_matrix_device, both ways: the k-mer stage's bucket kernels (two-level partition by column; "kmer_msd" forces them on a
    matrix this small) against the radix sorts of the whole matrix ("kmer_no_msd") and the oracle.  Shuffled triples; columns of 1 .. 40 entries
    (short ones: hints + inline partners; long ones: the dense path's pairs); reads that hold a k-mer twice; positions up to 14, 16 and 18 bits;
    column counts that are no multiple of a bucket; every column holds an entry (else the bucket path hands over to the sorts: the other test)."""
    import torch
    rng = np.random.default_rng(7000 + case)
    M = int(rng.choice([64, 300, 3000, 40000]))
    ncol = int(rng.choice([37, 500, 4097, 30011, 100003]))
    maxlen = int(rng.choice([1, 2, 3, 8, 12, 20, 40]))      # (1: every column a single entry — 1024 columns in a bucket of 1024 entries)
    maxpos = int(rng.choice([5000, 16383, 65535, 200000]))
    hot = rng.choice(M, min(M, int(rng.choice([3, 40, 500, 4000]))), replace=False)
    lens = rng.integers(1, maxlen + 1, ncol)
    cols = np.repeat(np.arange(ncol, dtype=np.int64), lens)
    rows = hot[rng.integers(0, len(hot), len(cols))].astype(np.int64)
    vals = rng.integers(0, maxpos + 1, len(cols)).astype(np.uint32)
    key = np.unique(np.stack([cols, rows, vals.astype(np.int64)], axis=1), axis=0)      # (a read holds a k-mer at a position once)
    cols, rows, vals = key[:, 0].copy(), key[:, 1].copy(), key[:, 2].astype(np.uint32)
    assert len(np.unique(cols)) == ncol
    Z = len(cols)
    up = max(2, maxlen)
    o = po.Oracle(17, 2, up)
    o.set_triples(M, ncol, rows, cols, vals)
    o.spgemm(8)
    perm = rng.permutation(Z)
    dr = torch.from_numpy(rows[perm]).cuda(); dc = torch.from_numpy(cols[perm]).cuda(); dv = torch.from_numpy(vals[perm].view(np.int32)).cuda()
    out = {}
    for name, opts in (("buckets", {"kmer_msd": 1}), ("sorts", {"kmer_no_msd": 1})):
        e = elba_amd.Engine(17, 2, up, options=opts)
        e.set_kmer_matrix_device(M, ncol, Z, dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
        took_buckets = e.get_stat("triples_path") == 1
        assert took_buckets == (name == "buckets") or (name == "buckets" and Z < 16384), (name, took_buckets, Z)      # (a matrix that fits a bucket or two: nothing to partition)
        st = e.create_seed_matrix()
        gu.assert_B_equal(e.export_csr(), o.B())
        gu.assert_stats_equal(st, o)
        out[name] = e.export_kmer_matrix()
        e.close()
    gu.assert_A_equal(out["buckets"], out["sorts"])
