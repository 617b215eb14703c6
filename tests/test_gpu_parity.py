"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle and the committed golden fixtures.
Bit-exact everywhere (integer / index work)."""
import os

import numpy as np
import pytest

import elba_amd
import gpu_util as gu
import synth
import util
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
G = util.GOLDEN


def _golden_set(name, meta):
    seqs = util.read_fasta(os.path.join(G, name + ".fa"))
    return po.pack_reads(seqs)


@pytest.mark.parametrize("name,idx", [("small_err", 0), ("small_clean", 0), ("small_clean", 1)])
def test_seed_matrix_from_triples_matches_golden_and_oracle(name, idx):
    """create_seed_matrix alone: A handed over as shuffled COO triples (the reference boundary of include/SharedSeeds.hpp:98-99)."""
    m = util.golden_meta()[name][idx]
    k, lo, up = m["k"], m["lower"], m["upper"]
    gk, gr, gp = util.read_triples(os.path.join(G, "%s_k%d_L%d_U%d.triples" % (name, k, lo, up)))
    uk, kid = np.unique(gk, return_inverse=True)
    perm = np.random.default_rng(0).permutation(len(gk))
    e = elba_amd.Engine(k, lo, up)
    ms = e.set_kmer_matrix(m["M"], len(uk), gr[perm], kid[perm], gp[perm])
    assert (ms["nrows"], ms["ncols"], ms["nnz"]) == (m["M"], m["N"], m["Z"])
    st = e.create_seed_matrix()
    assert (st["products"], st["nnz_before_prune"], st["nnz"]) == (m["P"], m["Yraw"], m["Y"])
    gB = e.export_csr()
    golden = util.read_B(os.path.join(G, "%s_k%d_L%d_U%d.B" % (name, k, lo, up)))
    assert (util.b_triplets(dict(M=gB["M"], rowptr=gB["rowptr"], col=gB["col"], val=gB["val"])) == golden).all()
    o = po.Oracle(k, lo, up)
    o.set_triples(m["M"], len(uk), gr, kid, gp)
    o.spgemm(2)
    gu.assert_B_equal(gB, o.B())
    gu.assert_stats_equal(st, o)
    e.close()


@pytest.mark.parametrize("name,idx", [("small_err", 0), ("small_clean", 0), ("small_clean", 1)])
def test_full_pipeline_matches_golden_triples(name, idx):
    """reads -> count -> A: the (kmer, read, pos) triples must equal the ones the reference's KmerOps replay produced."""
    m = util.golden_meta()[name][idx]
    k, lo, up = m["k"], m["lower"], m["upper"]
    packed, off, lens = _golden_set(name, m)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up)
    assert (ks["instances"], ks["reliable"], ks["entries"]) == (m["I"], m["N"], m["Z"])
    gA = e.export_kmer_matrix()
    km, rd, ps = util.triples_from_A(dict(colptr=gA["colptr"], kmers=gA["kmers"], csc_read=gA["csc_read"], csc_pos=gA["csc_pos"]))
    gk, gr, gp = util.read_triples(os.path.join(G, "%s_k%d_L%d_U%d.triples" % (name, k, lo, up)))
    assert (km == gk).all() and (rd == gr).all() and (ps == gp).all()
    o = gu.oracle_run(packed, off, lens, k, lo, up)
    gu.assert_A_equal(gA, o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    assert (e.kmer_histogram() == o.A()["hist"]).all()
    e.close()


@pytest.mark.parametrize("idx", [0, 1])
def test_reference_sample_reads(idx):
    """The reference's bundled reads.fa; figures measured by the survey from the reference's own code."""
    m = util.golden_meta()["reads_ref_appB"][idx]
    seqs = util.read_fasta(os.path.join(G, "reads_ref.fa.gz"))
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, m["k"], m["lower"], m["upper"])
    assert (ks["nreads"], ks["instances"], ks["reliable"], ks["entries"]) == (m["M"], m["I"], m["N"], m["Z"])
    assert (st["products"], st["nnz_before_prune"], st["nnz"], st["nnz_upper"], st["max_numshared"]) == (m["P"], m["Yraw"], m["Y"], m["nupper"], m["maxshared"])
    o = gu.oracle_run(packed, off, lens, m["k"], m["lower"], m["upper"])
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


def test_medium_noisy_reads_full_pipeline():
    packed, off, lens, info = elba_amd.synth_reads(21, 400000, 20, 4000, 800, error_rate=0.12)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8)
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    assert ks["distinct"] == o.stat("ndistinct")
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    assert st["algorithmic_bytes"] == 16 * o.stat("Z") + 8 * (2 * o.stat("M") + o.stat("N") + 3) + 24 * o.stat("Y")
    e.close()


def _concat(sets):
    packed, off, lens, base = [], [], [], 0
    for (p, o, l, _) in sets:
        nb = int(o[-1]) + (int(l[-1]) + 3) // 4 if len(l) else 0
        packed.append(p[:nb]); off.append(o + np.uint64(base)); lens.append(l); base += nb
    return np.concatenate(packed + [np.zeros(16, np.uint8)]), np.concatenate(off), np.concatenate(lens)


def test_every_table_tier_including_hbm_spill():
    """A noisy set with very uneven read lengths (row bounds from ~10^2 to ~10^3.5: every LDS tier) plus an error-free, extremely
    deep set whose per-row partner bound exceeds the largest LDS table (HBM spill path)."""
    noisy = elba_amd.synth_reads(5, 150000, 14, 2500, 1800, error_rate=0.10, min_len=60)
    deep = elba_amd.synth_reads(15, 2000, 250, 100, 10, error_rate=0.0, min_len=60)
    packed, off, lens = _concat([noisy, deep])
    assert len(lens) > 4300
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 600)
    assert st["rows_lds"] > 3000 and st["rows_lds"] + st["rows_global"] == int((np.diff(e.export_kmer_matrix()["rowptr"]) > 0).sum())
    o = gu.oracle_run(packed, off, lens, 17, 2, 600, threads=8)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()


def test_wide_rows_use_block_and_global_sorts():
    """Rows with > 64 and > 4096 surviving partners exercise the LDS-bitonic and HBM-bitonic finalize paths."""
    packed, off, lens, info = elba_amd.synth_reads(6, 200, 2200, 100, 0, error_rate=0.0, min_len=100)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 21, 2, 20000)
    o = gu.oracle_run(packed, off, lens, 21, 2, 20000, threads=8)
    assert np.diff(o.B()["rowptr"]).max() > 4096
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()


def test_optimistic_tables_escalate_and_spill_to_hbm():
    """Collision/overflow stress: rows whose distinct-partner count defeats every optimistic LDS table.  A is handed over as
    triples: `dense` columns each hold ALL reads (every row then has M partners, M / 2 > 3/4 of the largest LDS table -> HBM spill),
    a band of medium columns makes rows that overflow the small tiers only (escalation), singletons make rows that fit at once."""
    M, rng = 14000, np.random.default_rng(12)
    rows, cols, vals = [], [], []
    ncol = 0
    for c in range(2):                                   # two dense columns over rows 0..6999 -> 7000 partners, numshared 2
        r = np.arange(7000); rows.append(r); cols.append(np.full(len(r), ncol)); vals.append(rng.integers(0, 5000, len(r))); ncol += 1
    for c in range(2):                                   # rows 0..13999 in two more dense columns -> 14000 partners: a row accumulates half of
        r = np.arange(M); rows.append(r); cols.append(np.full(len(r), ncol)); vals.append(rng.integers(0, 5000, len(r))); ncol += 1      # them (the other half is mirrored), still more than 3/4 of the largest LDS table
    for b in range(0, 3000, 500):                        # medium: groups of 500 rows sharing 2 columns
        for c in range(2):
            r = 7000 + np.arange(b, b + 500) % 3500; rows.append(r); cols.append(np.full(len(r), ncol)); vals.append(rng.integers(0, 5000, len(r))); ncol += 1
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
    e = elba_amd.Engine(17, 2, 8)
    e.set_kmer_matrix(M, ncol, rows, cols, vals)
    st = e.create_seed_matrix()
    assert st["rows_global"] > 0 and st["rows_escalated"] > 0
    o = po.Oracle(17, 2, 8)
    o.set_triples(M, ncol, rows, cols, vals)
    o.spgemm(8)
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()


def test_staging_overflow_triggers_second_pass_with_identical_result():
    packed, off, lens, info = elba_amd.synth_reads(8, 60000, 15, 2500, 400, error_rate=0.05)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 12, workspace_hint_bytes=24 * 100)
    assert st["passes"] == 2
    o = gu.oracle_run(packed, off, lens, 17, 2, 12)
    gu.assert_B_equal(e.export_csr(), o.B())
    st2 = e.create_seed_matrix()          # workspace is now large enough
    assert st2["passes"] == 1
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


def test_dcsc_export_is_what_pairwise_alignment_walks():
    """src/PairwiseAlignment.cpp:28-56 walk over (jc, cp, ir, numx), whole matrix and a 2x2-grid cell, against the oracle's export."""
    m = util.golden_meta()["small_err"][0]
    packed, off, lens = _golden_set("small_err", m)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, m["k"], m["lower"], m["upper"])
    o = gu.oracle_run(packed, off, lens, m["k"], m["lower"], m["upper"])
    M = m["M"]
    for (r0, r1, c0, c1) in [(0, M, 0, M), (0, M // 2, M // 2, M), (M // 2, M, 0, M // 2), (3, 3, 0, M)]:
        g, w = e.export_dcsc(r0, r1, c0, c1), o.export_dcsc(r0, r1, c0, c1)
        assert g["nnz"] == w["nnz"] and g["nzc"] == w["nzc"]
        for key in ("jc", "cp", "ir", "numx"):
            assert (g[key] == w[key]).all(), key
    full = e.export_dcsc(0, M, 0, M)
    cand = sum(1 for ci in range(full["nzc"]) for x in range(full["cp"][ci], full["cp"][ci + 1]) if full["ir"][x] < full["jc"][ci])
    assert cand == st["nnz_upper"]
    e.close()


def test_edge_cases_empty_short_and_N_reads():
    # no reads at all
    e = elba_amd.Engine(17, 2, 8)
    e.set_reads(np.zeros(16, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    ks = e.count_kmers(); ms = e.create_kmer_matrix(); st = e.create_seed_matrix()
    assert (ks["instances"], ks["reliable"], ms["nnz"], st["nnz"]) == (0, 0, 0, 0)
    assert e.export_csr()["Y"] == 0
    # reads shorter than k, exactly k, all-N (N -> A), lower-case, duplicated reads, a homopolymer (one k-mer many times > UPPER)
    rng = np.random.default_rng(4)
    base = bytes(rng.choice(list(b"ACGT"), 300).tolist())
    seqs = [b"ACGT", base[:17], base[:17], b"N" * 40, b"n" * 40, base.lower(), base, synth.revcomp(base), b"A" * 16, b"", base[100:260], b"T" * 60]
    packed, off, lens = po.pack_reads(seqs)
    for (k, lo, up) in [(17, 2, 8), (17, 2, 200), (5, 2, 50)]:
        e2, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up)
        o = gu.oracle_run(packed, off, lens, k, lo, up)
        gu.assert_A_equal(e2.export_kmer_matrix(), o.A())
        gu.assert_B_equal(e2.export_csr(), o.B())
        gu.assert_stats_equal(st, o)
        e2.close()
    e.close()


def test_stage_order_and_bad_input_errors():
    e = elba_amd.Engine(17, 2, 8)
    with pytest.raises(elba_amd.ElbaError) as ei:
        e.count_kmers()
    assert ei.value.status == 5
    with pytest.raises(elba_amd.ElbaError) as ei:
        e.create_seed_matrix()
    assert ei.value.status == 5
    with pytest.raises(elba_amd.ElbaError) as ei:
        e.set_kmer_matrix(4, 4, [0, 5], [0, 1], [1, 2])
    assert ei.value.status == 1
    e.close()


def test_repeated_calls_are_deterministic():
    packed, off, lens, info = elba_amd.synth_reads(9, 80000, 12, 3000, 500, error_rate=0.1)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8)
    a = e.export_csr()
    for _ in range(3):
        e.create_seed_matrix()
        b = e.export_csr()
        assert (a["rowptr"] == b["rowptr"]).all() and (a["col"] == b["col"]).all() and (a["val"] == b["val"]).all()
    e.close()


def test_pattern_and_counts_do_not_depend_on_kmer_id_order():
    """The reference numbers k-mers by unordered_map iteration order (src/KmerOps.cpp:380-394), this build by value rank.
    Whatever the numbering: same pattern of B, same numshared; only WHICH shared k-mer lands in seeds[0]/[1] may change — and every stored
    seed stays a genuine shared k-mer (the reference's test.py:57-65)."""
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    packed, off, lens = po.pack_reads(seqs)
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    A, oB = o.A(), o.B()
    rows = np.repeat(np.arange(A["M"], dtype=np.int64), np.diff(A["rowptr"]))
    perm = np.random.default_rng(7).permutation(A["N"])                 # a different k-mer numbering
    e = elba_amd.Engine(17, 2, 8)
    e.set_kmer_matrix(A["M"], A["N"], rows, perm[A["csr_kid"]], A["csr_pos"])
    e.create_seed_matrix()
    gB = e.export_csr()
    assert (gB["rowptr"] == oB["rowptr"]).all() and (gB["col"] == oB["col"].astype(np.int64)).all()
    assert (gB["val"]["numshared"] == oB["val"]["numshared"]).all()
    assert (gB["val"] != oB["val"]).any()                                 # the seeds themselves do follow the numbering ...
    L = po.lib()
    brow = np.repeat(np.arange(gB["M"]), np.diff(gB["rowptr"]))
    for x in range(gB["Y"]):                                              # ... and are valid all the same
        i, j, v = int(brow[x]), int(gB["col"][x]), gB["val"][x]
        for q, t in ((v["q0"], v["t0"]), (v["q1"], v["t1"])):
            assert L.orc_seed_is_valid(packed.ctypes.data + int(off[i]), int(lens[i]), packed.ctypes.data + int(off[j]), int(lens[j]), int(q), int(t), 17)
    e.close()


@pytest.mark.parametrize("k,lo,up", [(33, 2, 12), (45, 2, 12), (63, 2, 12), (65, 2, 12), (77, 2, 12), (95, 2, 12)])
def test_two_word_kmers_full_pipeline(k, lo, up):
    """32 < k <= 63 (NLONGS == 2 in the reference, include/Kmer.hpp:95-97): reads -> A -> B equal the oracle's, whose two-word k-mers are
    pinned to the reference's Kmer<2>::GetRepKmers (tests/test_oracle_golden.py).  Reads around byte and word boundaries included."""
    rng = np.random.default_rng(k)
    seqs = util.read_fasta(os.path.join(G, "small_clean.fa")) + util.read_fasta(os.path.join(G, "small_err.fa"))
    seqs += [bytes(rng.choice(list(b"ACGT"), n).tolist()) for n in (k - 1, k, k + 1, k + 2, k + 3, 64, 65, 96, 97, 127, 128, 129)]
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up)
    o = gu.oracle_run(packed, off, lens, k, lo, up)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    a = e.align_seeds()                                             # the seed check of the aligner walks k bases too
    rows, cols, ov, _ = o.align_upper(packed, off, lens)
    g = e.export_overlaps()
    assert (g["rows"] == rows).all() and all((g["vals"][f] == ov[f]).all() for f in ov.dtype.names if f != "pad")
    e.close()


def test_position_carrying_accumulators_equal_the_looked_up_seeds():
    """Reads whose positions fit 16 bits get 64-bit accumulators that carry the positions (no seed-decoding loads); the option "no_pay"
    keeps the 32-bit accumulators + look-ups: same B, bit for bit, on every tier."""
    noisy = elba_amd.synth_reads(51, 150000, 14, 2500, 1800, error_rate=0.10, min_len=60)
    deep = elba_amd.synth_reads(52, 3000, 120, 120, 10, error_rate=0.0, min_len=60)
    packed, off, lens = _concat([noisy, deep])
    o = gu.oracle_run(packed, off, lens, 17, 2, 300, threads=8)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 300)
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()
    e2, ks2, ms2, st2 = gu.gpu_full(packed, off, lens, 17, 2, 300, options={"no_pay": 1})
    gu.assert_B_equal(e2.export_csr(), o.B())
    gu.assert_stats_equal(st2, o)
    e2.close()


def test_reads_longer_than_16_bit_positions_use_the_lookup_path():
    """One read beyond 65 535 bases: positions no longer fit the payload, the matrix keeps the plain formats."""
    long_ = elba_amd.synth_reads(53, 90000, 6, 70000, 4000, error_rate=0.05, min_len=66000)
    short = elba_amd.synth_reads(54, 90000, 6, 3000, 500, error_rate=0.05, min_len=500)
    packed, off, lens = _concat([short, long_])
    assert lens.max() > 65535
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 30)
    o = gu.oracle_run(packed, off, lens, 17, 2, 30, threads=8)
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()


def test_words_one_or_two_bits_short_mark_rare_positions_for_lookup():
    """Many reads (19 id bits) and a few long ones (positions up to 2^15): read id + position need 34 bits.  The gathered word keeps 13
    position bits; the 1 % of entries beyond them are marked and their seeds looked up — B equals the oracle's, seeds included."""
    rng = np.random.default_rng(77)
    M, ncol = 300000, 6000
    rows, cols, vals = [], [], []
    hot = rng.choice(M, 900, replace=False)                  # reads that actually share k-mers (so that pairs share >= 2 of them)
    for c in range(ncol):
        r = np.unique(rng.choice(hot, int(rng.integers(2, 7))))
        rows.append(r); cols.append(np.full(len(r), c))
        v = rng.integers(0, 6000, len(r))
        far = rng.random(len(r)) < 0.01
        v[far] = rng.integers(8191, 30000, int(far.sum()))
        vals.append(v)
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
    assert (vals >= 8191).sum() > 50 and vals.max() < 32768
    e = elba_amd.Engine(17, 2, 8)
    e.set_kmer_matrix(M, ncol, rows, cols, vals)
    st = e.create_seed_matrix()
    o = po.Oracle(17, 2, 8)
    o.set_triples(M, ncol, rows, cols, vals)
    o.spgemm(8)
    B = e.export_csr(); oB = o.B()
    gu.assert_B_equal(B, oB)
    gu.assert_stats_equal(st, o)
    # some surviving seed really sits on a marked position
    assert ((oB["val"]["t0"] >= 8191) | (oB["val"]["t1"] >= 8191)).sum() > 0
    e.close()


@pytest.mark.parametrize("knob", ["no_symmetry", "no_ell", "no_pay", "kmer_pairs", "mir32", "kmer_unfused", "csr_pairs", "no_hints", "emit_plain", "no_sample", "kmer_no_msd"])
def test_alternative_paths_kept_for_ab_runs_give_the_same_matrices(knob):
    """Alternatives kept behind elba_set_option (both triangles accumulated instead of one + mirror, plain CSC columns instead of the
    padded ones, 32-bit accumulators + seed look-ups, (value, payload) pairs through the k-mer sort, 32-byte mirror / staging records,
    per-head column emission, (read, entry) pairs through the CSR sort, no ownership hints in the rows of A, the k-mer emit without its
    fused histogram, no sampled rows on a cold call, the LSD sort instead of the two-level partition): A and B must not change."""
    packed, off, lens, info = elba_amd.synth_reads(61, 200000, 16, 3000, 900, error_rate=0.10, min_len=200)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8, options={knob: 1})
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    st2 = e.create_seed_matrix()                       # steady-state call on the same matrix
    gu.assert_B_equal(e.export_csr(), o.B())
    e.set_option("overlap_cold_calls", 1)              # and a cold one
    st3 = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st3, o)
    e.close()


@pytest.mark.parametrize("which", ["small_err_k17", "reads_ref_default_k31"])
def test_reference_kmer_numbering_through_set_kmer_matrix(which):
    """A numbered the way a one-rank run of the reference numbers its k-mers (tests/golden/*.order: unordered_map
    iteration order replayed on the reference's compiled code) handed over as triples: the GPU's B equals the oracle's under THAT
    numbering bit for bit (seeds = min / max product of that numbering), and its pattern and numshared equal the canonical B's.
    reads_ref_default_k31: the reference's default build (Makefile:1-3: k = 31, L = 15, U = 35) on its bundled reads.fa."""
    if which == "small_err_k17":
        k, lo, up = 17, 2, 8
        rd, ref_ids, ps, canon_ids, N, _ = util.libstdcxx_triples("small_err")
        M = util.golden_meta()["small_err"][0]["M"]
    else:
        k, lo, up = 31, 15, 35
        M, N, rd, ref_ids, ps, canon_ids, _ = util.reference_default_triples()
    e = elba_amd.Engine(k, lo, up)
    e.set_kmer_matrix(M, N, rd, ref_ids, ps)
    st = e.create_seed_matrix()
    gB = e.export_csr()
    o = po.Oracle(k, lo, up); o.set_triples(M, N, rd, ref_ids, ps); o.spgemm(4)
    gu.assert_B_equal(gB, o.B())
    gu.assert_stats_equal(st, o)
    oc = po.Oracle(k, lo, up); oc.set_triples(M, N, rd, canon_ids, ps); oc.spgemm(4)
    Bc = oc.B()
    assert (gB["rowptr"] == Bc["rowptr"]).all() and (gB["col"] == Bc["col"].astype(np.int64)).all() and (gB["val"]["numshared"] == Bc["val"]["numshared"]).all()
    e.close()


@pytest.mark.parametrize("k", [17, 31, 33, 65])
def test_reference_kmer_hash_and_owner_on_the_device(k):
    """Kmer::GetHash (murmur3_x64_128, seed 313, h1) and GetKmerOwner computed on the GPU.  One-word k-mers: the reference's
    own vectors (tests/golden/kmer_vectors_k*.txt: murmur of rep and of fwd, generated by the reference's compiled Kmer<1>::GetHash); multi-word
    k-mers: the oracle's murmur3 (pinned to the reference's HashFuncs vectors, tests/golden/murmur_vectors.txt) over the 16 / 24 key bytes.
    Owners: the reference's double-precision formula (oracle/elba_oracle.c:orc_kmer_owner, src/KmerOps.cpp:352-359) for p in 1..16."""
    import ctypes as C
    L = po.lib()
    e = elba_amd.Engine(k, 2, 8)
    if k <= 31:
        rows = [line.split() for line in open(os.path.join(G, "kmer_vectors_k%d.txt" % k)) if line[0] != "#"]
        km = np.array([int(r[3], 16) for r in rows] + [int(r[1], 16) for r in rows], dtype=np.uint64)
        want = np.array([int(r[4], 16) for r in rows] + [int(r[5], 16) for r in rows], dtype=np.uint64)
        W = 1
    else:
        W = 3 if k > 64 else 2
        km = np.random.default_rng(k).integers(0, 2**63, size=(500, W), dtype=np.int64).astype(np.uint64)
        want = np.zeros(len(km), dtype=np.uint64)
        out = (C.c_uint64 * 2)()
        for i in range(len(km)):
            L.orc_murmur3_x64_128(km[i].tobytes(), 8 * W, 313, out)
            want[i] = out[0]
    for p in (1, 2, 4, 8, 16):
        h, ow = e.kmer_hash_owner(km, p)
        assert (h == want).all()
        assert ow.tolist() == [L.orc_kmer_owner(C.c_uint64(int(x)), p) for x in want.tolist()] and ow.max() < p
    e.close()


@pytest.mark.parametrize("shape", ["pairs", "wide"])
def test_reads_that_hold_a_kmer_twice_and_the_ownership_hints(shape):
    """The rows of A carry a hint per entry: "this row accumulates no pair of this column"  — an entry so marked never
    fetches its column.  The cases the rule has to get right: a read that holds the k-mer twice (it owes itself the cross products), columns
    of two reads (exactly one of them fetches), read ids of equal and of different parity, and long columns.  B, P and the diagonal
    equal the oracle's, on a cold, a warm and a hint-free call."""
    rng = np.random.default_rng(5 if shape == "pairs" else 6)
    M, ncol = 3000, 40000
    rows, cols, vals = [], [], []
    hot = rng.choice(M, 400, replace=False)
    for c in range(ncol):
        n = 2 if shape == "pairs" and c % 4 else int(rng.integers(2, 9 if shape == "pairs" else 40))
        r = rng.choice(hot, n, replace=True)              # with replacement: a read may hold the k-mer more than once
        if c % 7 == 0: r[1] = r[0]                         # ... and every 7th column certainly does
        r = np.sort(r)
        p = rng.integers(0, 5000, n)
        for rr in np.unique(r):                            # (read, pos) pairs of a column are distinct and ascending per read
            m = r == rr
            p[m] = np.sort(rng.choice(5000, int(m.sum()), replace=False))
        rows.append(r); cols.append(np.full(n, c)); vals.append(p)
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
    up = 8 if shape == "pairs" else 40
    o = po.Oracle(17, 2, up)
    o.set_triples(M, ncol, rows, cols, vals)
    o.spgemm(8)
    oB = o.B()
    for opts in ({}, {"no_hints": 1}):
        e = elba_amd.Engine(17, 2, up, options=opts)
        e.set_kmer_matrix(M, ncol, rows, cols, vals)
        st = e.create_seed_matrix()
        gu.assert_B_equal(e.export_csr(), oB)
        gu.assert_stats_equal(st, o)
        e.create_seed_matrix()
        gu.assert_B_equal(e.export_csr(), oB)
        e.close()


@pytest.mark.parametrize("knob", [None, "no_suffix", "no_row_order", ("dense_up", 0), ("dense_up", 2), {"kmer_msd": 1}, {"kmer_msd": 1, "csr_pairs_late": 1}, {"kmer_msd": 1, "msd_no_rank": 1},
                                  {"kmer_msd": 1, "msd_small_cap": 64}])
def test_dense_columns_take_the_path_of_their_own(knob):
    """Deep, nearly error-free reads with a generous UPPER: columns of ~30 reads, hundreds of products per surviving pair.  Such matrices are
    multiplied by the dense path (pairs owned by the smaller row, the owned candidates of a row entry = its column behind it: DESIGN.md §4.1);
    the option "no_suffix" keeps them on the general path.  The dense path names partners by LABEL (rank of the read among the reads sorted by their
    smallest k-mer: reads of one locus get neighbouring labels; "no_row_order" names them by row) and starts its rows on the tier "dense_up" says
    (1: eight wavefronts on a 1024-slot table).  A, B and the statistics equal the oracle's every way, on a cold and a warm call, and a read that
    holds a k-mer twice (a repeat family) is among them.  Through the two-level partition ("kmer_msd" forces it on an input this small) the
    bucket kernels write the CSR build's (read, entry) pairs themselves — the entry carries its column's length and its own place in it, which is
    known where the column lies sorted — ("csr_pairs_late": the CSR build does), and sort a bucket by (column rank, read) ranges
    ("msd_no_rank": by ranges of the value bits, where a long column is one range); "msd_small_cap": the crowded-bucket kernel writes them."""
    packed, off, lens, info = elba_amd.synth_reads(91, 60000, 30, 3000, 600, error_rate=0.01, min_len=500, repeat_families=3, repeat_fraction=0.1, repeat_len=400)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 40, options=(dict(knob) if isinstance(knob, dict) else {knob[0]: knob[1]} if isinstance(knob, tuple) else {knob: 1}) if knob else None)
    if isinstance(knob, dict):
        assert e.get_stat("kmer_path") == 1
    o = gu.oracle_run(packed, off, lens, 17, 2, 40, threads=8)
    assert ms["max_col_nnz"] > 16 if "max_col_nnz" in ms else True
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.set_option("overlap_cold_calls", 1)
    st = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    if knob is None:
        # the same dense matrix handed over as device-resident triples (INTEGRATION.md Option B) gets its labels too
        import torch
        Z, M, N = int(ms["nnz"]), int(ms["nrows"]), int(ms["ncols"])
        dr = torch.empty(Z, dtype=torch.int64, device="cuda"); dc = torch.empty(Z, dtype=torch.int64, device="cuda"); dv = torch.empty(Z, dtype=torch.int32, device="cuda")
        e.export_triples_device(dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
        for opts in ({"kmer_no_msd": 1}, {"kmer_msd": 1}):            # (through the sorts of matrix.hip / through the k-mer stage's bucket kernels)
            e2 = elba_amd.Engine(17, 2, 40, options=opts)
            e2.set_kmer_matrix_device(M, N, Z, dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
            assert e2.get_stat("triples_path") == (1 if "kmer_msd" in opts else 0)
            st2 = e2.create_seed_matrix()
            gu.assert_B_equal(e2.export_csr(), o.B())
            gu.assert_stats_equal(st2, o)
            e2.close()
    if isinstance(knob, dict):
        e.create_kmer_matrix()                              # (the pairs were consumed: rebuilt from the column pointers)
        e.create_seed_matrix()
        gu.assert_A_equal(e.export_kmer_matrix(), o.A())
        gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


@pytest.mark.parametrize("drop", [1, 2, 3])
def test_dropped_index_bits_are_recovered_from_the_reads(drop):
    """A k-mer instance travels through the counting sort as ONE word, value << pb | (instance index >> drop); beyond 2^30 instances at
    k = 17 the lowest index bits do not fit and the kept entries find their instance among the 2^drop candidates by recomputing the
    candidates' k-mers (k_runs_emit).  The option "kmer_drop" forces that path on a small input, low-complexity reads included (neighbouring
    positions that hold the SAME k-mer: the dup-th candidate is the right one)."""
    reads, _ = synth.make_reads(77, 60000, 14, 2500, 700, error=0.08, min_len=100)
    # a few homopolymer / dinucleotide reads: runs of equal canonical k-mers at consecutive positions
    extra = [b"A" * 300, b"AC" * 200, b"T" * 150 + b"G" * 150, b"ACG" * 120] * 3
    seqs = list(reads) + extra
    np.random.default_rng(3).shuffle(seqs)
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 12, options={"kmer_drop": drop})
    o = gu.oracle_run(packed, off, lens, 17, 2, 12, threads=8)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


@pytest.mark.parametrize("opts", [None, {"kmer_msd": 1}, {"kmer_unfused": 1}])
def test_create_kmer_matrix_twice_after_one_count_gives_the_same_matrices(opts):
    """elba_create_kmer_matrix consumes the CSR sort keys the fused column pass left behind (hint bits are ORed into them, the sort
    ping-pongs over them); a second call after ONE elba_count_kmers must rebuild what it needs and give the same A and B."""
    packed, off, lens, info = elba_amd.synth_reads(62, 150000, 14, 3000, 900, error_rate=0.10, min_len=200)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8, options=opts)
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.create_kmer_matrix()
    st2 = e.create_seed_matrix()
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st2, o)
    # elba_release_workspace gives the stages' scratch back: the resident matrices multiply as before, and a count whose sort keys went with the
    # scratch still builds its matrix
    e.release_workspace()
    st3 = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st3, o)
    e.count_kmers(); e.release_workspace(); e.create_kmer_matrix()
    st4 = e.create_seed_matrix()
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st4, o)
    e.close()


@pytest.mark.parametrize("k,lo,up,cap", [(17, 2, 8, 0), (17, 2, 40, 0), (17, 3, 12, 0), (15, 2, 8, 0), (13, 2, 30, 0), (11, 2, 60, 0), (9, 2, 200, 0), (17, 2, 8, 2), (13, 2, 30, 100), (9, 2, 200, 1000)])
def test_two_level_partition_kmer_path_equals_the_oracle(k, lo, up, cap):
    """k <= 17: the k-mer stage counts by a two-level value partition and an LDS count table per bucket (kmer_msd.hip) on inputs of some size;
    the option "kmer_msd" forces it on a small one (most buckets empty or tiny).  Low-complexity reads (runs of one canonical k-mer at
    neighbouring positions, k-mers far beyond UPPER), reads shorter than k and reads around the word boundaries of the packed stream included:
    reliable k-mers, columns, rows, hints-carrying B and every statistic equal the oracle's.  cap > 0: buckets with more entries than that are
    emitted by the crowded-bucket kernel (windows of staged columns) instead of the LDS sort of the small ones — both kernels in one run."""
    reads, _ = synth.make_reads(78 + k, 60000, 14, 2500, 700, error=0.08, min_len=100)
    rng = np.random.default_rng(k)
    extra = [b"A" * 300, b"AC" * 200, b"T" * 150 + b"G" * 150, b"ACG" * 120, b"ACGT" * 90] * 3
    extra += [bytes(rng.choice(list(b"ACGT"), n).tolist()) for n in (k - 1, k, k + 1, k + 2, 31, 32, 33, 63, 64, 65, 3)]
    seqs = list(reads) + extra
    np.random.default_rng(5).shuffle(seqs)
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up, options={"kmer_msd": 1, "msd_small_cap": cap})
    o = gu.oracle_run(packed, off, lens, k, lo, up, threads=8)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    if up <= 12:
        assert e.device_view()["a_csr_format"] == 3           # short columns, general SpGEMM path: the rows carry inline partners (ELBA_CSR_INLINE)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.create_kmer_matrix()                                  # the repeat call rebuilds from the column pointers (the sort keys were consumed)
    st2 = e.create_seed_matrix()
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


@pytest.mark.parametrize("opts", [{"no_inline": 1}, {"no_symmetry": 1}, {"no_pay": 1}, {"no_ell": 1}, {"mir32": 1}, {"no_sample": 1}, {"no_ell_compact": 1}, {"ell_slot_cap": 100}, {"msd_rank": 1}])
def test_inline_partners_against_their_alternatives(opts):
    """The two-level partition path writes the owning row's entry of every two-read column as an inline partner (no column fetch in the
    SpGEMM).  Without them ("no_inline"; both triangles accumulated: every row needs every column), with 32-bit accumulators + look-ups / plain CSC columns (which switch the inline format off), wide staging records, no
    sampled rows, buckets sorted by (column rank, read) ranges as they are when UPPER allows long columns ("msd_rank"): A, B and the statistics
    equal the oracle's, on a first, a warm and a cold call."""
    packed, off, lens, info = elba_amd.synth_reads(63, 200000, 16, 3000, 900, error_rate=0.12, min_len=200)
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8, options=dict(opts, kmer_msd=1))
    fmt = e.device_view()["a_csr_format"]
    assert fmt == (1 if ("no_inline" in opts or "no_pay" in opts or "no_ell" in opts or "no_symmetry" in opts) else 3)
    # gather slots: with inline partners the padded column store holds only the columns some row entry still fetches — unless switched off, or
    # the store is too small for them (test hook: the emit is then repeated without slots)
    slots, ncols = e.device_view()["a_gather_slots"], ms["ncols"]
    assert (slots == 0) if (fmt != 3 or "no_ell_compact" in opts or "ell_slot_cap" in opts) else slots > 0, (slots, ncols)      # (slots are drawn in chunks: on a small matrix more are drawn than it has columns)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    st2 = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), o.B())
    e.set_option("overlap_cold_calls", 1)
    st3 = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st3, o)
    e.close()


@pytest.mark.parametrize("nreads", [500, 1500])
def test_two_level_partition_windows_of_a_crowded_bucket(nreads):
    """A genome of few distinct k-mers at high depth: single buckets of the two-level partition hold thousands of reliable columns and, with
    1500 reads, tens of thousands of entries — more than one staging window per bucket half (kmer_msd.hip: EW), instances beyond what a
    workgroup keeps in registers (k_msd_bucket); with 500 reads 4-8 thousand entries: the widest instantiation of the LDS sort (k_msd_emit_small on 1024 lanes)."""
    rng = np.random.default_rng(11)
    unit = bytes(rng.choice(list(b"ACGT"), 24).tolist())
    seqs = []
    for r in range(nreads):
        # reads built from short mutations of one unit: a few thousand distinct k-mers that share their leading bases
        s = bytearray(unit * 12)
        for _ in range(6):
            s[int(rng.integers(0, len(s)))] = int(rng.choice(list(b"ACGT")))
        seqs.append(bytes(s))
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 250, options={"kmer_msd": 1})
    o = gu.oracle_run(packed, off, lens, 17, 2, 250, threads=8)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


@pytest.mark.parametrize("knob", [None, "csr_pairs", "kmer_msd"])
def test_device_resident_triples_in_any_order(knob):
    """INTEGRATION.md Option B hands A over as COO triples that already sit in HBM (elba_set_kmer_matrix_device).  The triples of a GPU-built A,
    exported on the device and SHUFFLED, rebuild the same matrix — through the one-word sort (k-mer id | read | position in 64 bits), with the
    option "csr_pairs" through the three stable pair sorts, and through the k-mer stage's bucket kernels (two-level partition by column: what a
    matrix of some size takes; "kmer_msd" forces it on this one) — and B equals the oracle's.  An index out of range is refused; a matrix with an
    empty column keeps the sort (the bucket kernels number the columns they find)."""
    import torch
    packed, off, lens, info = elba_amd.synth_reads(23, 50000, 25, 3000, 700, error_rate=0.10, min_len=200)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8)
    o = gu.oracle_run(packed, off, lens, 17, 2, 8, threads=8)
    Z, M, N = int(ms["nnz"]), int(ms["nrows"]), int(ms["ncols"])
    dr = torch.empty(Z, dtype=torch.int64, device="cuda"); dc = torch.empty(Z, dtype=torch.int64, device="cuda"); dv = torch.empty(Z, dtype=torch.int32, device="cuda")
    e.export_triples_device(dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
    perm = torch.randperm(Z, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    dr, dc, dv = dr[perm].contiguous(), dc[perm].contiguous(), dv[perm].contiguous()
    e2 = elba_amd.Engine(17, 2, 8, options={knob: 1} if knob else {"kmer_no_msd": 1})
    m2 = e2.set_kmer_matrix_device(M, N, Z, dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
    assert int(m2["nnz"]) == Z
    assert e2.get_stat("triples_path") == (1 if knob == "kmer_msd" else 0)
    st2 = e2.create_seed_matrix()
    gu.assert_A_equal(e2.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e2.export_csr(), o.B())
    gu.assert_stats_equal(st2, o)
    if knob == "kmer_msd":
        assert e2.device_view()["a_csr_format"] == 3      # inline partners, as from the reads
        # the same triples without the entries of one column: N stays, the column is empty — the sort path builds it, like an engine without the option
        keep = dc != int(dc[0])
        r3, c3, v3 = dr[keep].contiguous(), dc[keep].contiguous(), dv[keep].contiguous()
        e3 = elba_amd.Engine(17, 2, 8, options={"kmer_no_msd": 1})
        for eng in (e2, e3):
            eng.set_kmer_matrix_device(M, N, int(r3.numel()), r3.data_ptr(), c3.data_ptr(), v3.data_ptr())
            eng.create_seed_matrix()
        assert e2.get_stat("triples_path") == 0
        gu.assert_A_equal(e2.export_kmer_matrix(), e3.export_kmer_matrix())
        gu.assert_B_equal(e2.export_csr(), e3.export_csr())
        e3.close()
    dc[Z // 2] = N                                   # a column that does not exist
    with pytest.raises(Exception):
        e2.set_kmer_matrix_device(M, N, Z, dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
    e.close(); e2.close()


def test_duplicate_triples_are_kept_by_the_bucket_kernels():
    """include/elba_amd.h (elba_set_kmer_matrix): duplicates are kept (SumDuplicates = false, src/KmerOps.cpp:400).  Two triples with the same
    (row, column, position) pack to the same word in the bucket kernels' LDS sort; equal words must still get places of their own (they are
    ranked by where the first scatter put them).  Exact duplicates — single ones, runs of four, a whole column doubled — through the bucket
    kernels ("kmer_msd") equal the radix sorts of matrix.hip ("kmer_no_msd") entry for entry, A and B."""
    import torch
    packed, off, lens, info = elba_amd.synth_reads(29, 40000, 25, 3000, 700, error_rate=0.10, min_len=200)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8)
    Z, M, N = int(ms["nnz"]), int(ms["nrows"]), int(ms["ncols"])
    dr = torch.empty(Z, dtype=torch.int64, device="cuda"); dc = torch.empty(Z, dtype=torch.int64, device="cuda"); dv = torch.empty(Z, dtype=torch.int32, device="cuda")
    e.export_triples_device(dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
    g = torch.Generator(device="cuda").manual_seed(11)
    once = torch.randperm(Z, device="cuda", generator=g)[: Z // 50]                 # 2 % of the entries twice
    four = torch.randperm(Z, device="cuda", generator=g)[: Z // 400].repeat(3)      # some of them four times
    col = (dc == int(dc[Z // 3])).nonzero().flatten()                              # one column doubled whole
    idx = torch.cat([torch.arange(Z, device="cuda"), once, four, col])
    idx = idx[torch.randperm(idx.numel(), device="cuda", generator=g)]
    r2, c2, v2 = dr[idx].contiguous(), dc[idx].contiguous(), dv[idx].contiguous()
    Z2 = int(idx.numel())
    out = []
    for knob in ("kmer_msd", "kmer_no_msd"):
        eng = elba_amd.Engine(17, 2, 8, options={knob: 1})
        m2 = eng.set_kmer_matrix_device(M, N, Z2, r2.data_ptr(), c2.data_ptr(), v2.data_ptr())
        assert int(m2["nnz"]) == Z2
        assert eng.get_stat("triples_path") == (1 if knob == "kmer_msd" else 0)
        st2 = eng.create_seed_matrix()
        out.append((eng.export_kmer_matrix(), eng.export_csr(), st2))
        eng.close()
    gu.assert_A_equal(out[0][0], out[1][0])
    gu.assert_B_equal(out[0][1], out[1][1])
    assert out[0][2]["nnz"] == out[1][2]["nnz"] and out[0][2]["products"] == out[1][2]["products"]
    e.close()


@pytest.mark.parametrize("lo,up,opts", [(2, 8, {}), (2, 40, {}), (2, 8, {"msd_small_cap": 300}), (3, 12, {"no_inline": 1})])
def test_value_range_batches_of_the_kmer_stage_equal_the_unbatched_matrix(lo, up, opts):
    """More k-mer instances than a 32-bit place holds are counted in PASSES over ranges of first digits (kmer_msd.hip, "VALUE-RANGE BATCHING": the
    reference batches its exchange so that size is no limit, include/KmerOps.hpp:33-56).  The option "kmer_batch_instances" forces passes on a small
    set: three passes, and one pass per first digit, give the k-mers, both orientations of A and B of the unbatched run bit for bit — general and
    dense matrices, crowded buckets (the windowed kernel), a build without inline partners."""
    packed, off, lens, info = elba_amd.synth_reads(83, 900000, 14, 3000, 800, error_rate=0.06, min_len=100)
    base = dict(opts, kmer_msd=1)
    e0, ks0, ms0, st0 = gu.gpu_full(packed, off, lens, 17, lo, up, options=base)
    assert e0.get_stat("kmer_path") == 1
    A0, B0 = e0.export_kmer_matrix(), e0.export_csr()
    e0.close()
    I = int(ks0["instances"])
    for cap in (I // 3 + 1, 1):                                      # three passes; a pass per first digit (a digit is never split)
        e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, lo, up, options=dict(base, kmer_batch_instances=cap))
        assert e.get_stat("kmer_path") == 1 and e.get_stat("kmer_passes") >= (3 if cap > 1 else 100)
        assert all(ks[f] == ks0[f] for f in ("instances", "distinct", "reliable", "entries"))
        gu.assert_A_equal(e.export_kmer_matrix(), A0)
        gu.assert_B_equal(e.export_csr(), B0)
        assert all(st[f] == st0[f] for f in ("nnz", "products", "nnz_before_prune", "nnz_diag", "nnz_upper", "max_numshared"))
        e.close()


@pytest.mark.parametrize("dk", [0, 1, 2, 4])
def test_more_gather_trips_in_flight_give_the_same_matrix(dk):
    """The option "dk" (gather trips per iteration of the padded-column loop: one, two, four or eight; chosen per matrix by default) selects
    other instantiations of the numeric kernel: B and the statistics must not change."""
    packed, off, lens, info = elba_amd.synth_reads(61, 200000, 16, 3000, 900, error_rate=0.10, min_len=200)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 17, 2, 8, options={"dk": dk})
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.set_option("overlap_cold_calls", 1)
    st = e.create_seed_matrix()
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()


@pytest.mark.parametrize("shape", ["forced_ratio", "sampled_cold", "wide_rows"])
def test_mirror_slabs_of_any_size_give_the_same_matrix(shape):
    """Mirror slabs (spgemm.hip): the numeric kernel writes a staged entry's transposed image straight into its partner row's slab when its ticket
    lies inside it; images beyond the slab (or drawn before a ratio was known) wait for k_mirror, and the finalize reads a row's mirrored
    entries from both places.  Slabs far too small, just right and generous, sized by the test hook, by a cold call's sample of rows and by the
    previous call's measurement: B never changes."""
    if shape == "forced_ratio":
        packed, off, lens, info = elba_amd.synth_reads(71, 300000, 18, 3500, 900, error_rate=0.10, min_len=200)
        k, lo, up = 17, 2, 8
    elif shape == "sampled_cold":      # >= 8192 rows: a cold call computes a sample of rows first and sizes the slabs by what they staged
        packed, off, lens, info = elba_amd.synth_reads(72, 320000, 24, 800, 250, error_rate=0.06, min_len=120)
        k, lo, up = 17, 2, 12
        assert len(lens) >= 8192
    else:                              # rows of B beyond 256 / 1024 / 4096 entries: every finalize kernel reads slabs
        packed, off, lens, info = elba_amd.synth_reads(6, 200, 2200, 100, 0, error_rate=0.0, min_len=100)
        k, lo, up = 21, 2, 20000
    o = gu.oracle_run(packed, off, lens, k, lo, up, threads=8)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up)
    gu.assert_B_equal(e.export_csr(), o.B())
    nmir = (o.stat("Y") - o.stat("ndiag")) // 2
    seen = []
    for cold, q16, pct in [(1, 0, 175), (1, 1, 100), (1, 400, 100), (1, 3000, 100), (1, 1 << 18, 100), (0, 0, 175), (0, 0, 30), (1, 0, 20), (1, 0, 1000)]:
        e.set_option("overlap_cold_calls", cold); e.set_option("slab_q16", q16); e.set_option("slab_pct", pct)
        st = e.create_seed_matrix()
        gu.assert_B_equal(e.export_csr(), o.B())
        gu.assert_stats_equal(st, o)
        seen.append((cold, q16, pct, e.get_stat("overlap_slab_q16"), e.get_stat("overlap_mirror_placed")))
    by = {(c, q, p): (used, placed) for c, q, p, used, placed in seen}
    gen = by[(1, 1 << 18, 100)]                                       # four slab entries per row entry: generous for reads with errors ...
    assert gen[0] > 3000 and gen[1] <= by[(1, 3000, 100)][1] <= by[(1, 400, 100)][1] <= by[(1, 1, 100)][1] and gen[1] < by[(1, 1, 100)][1], seen
    if shape == "forced_ratio":
        assert gen == (1 << 18, 0), seen                              # ... nothing waits for k_mirror (the other shapes: the sample's own rows do / error-free reads stage a dozen entries per row entry)
    assert by[(1, 1, 100)][0] == 1 and 0 < by[(1, 1, 100)][1] <= nmir, seen       # 16 entries per row: most images take the old way
    assert by[(0, 0, 175)][0] > 0, seen                                # a warm call sizes its slabs by the previous call's measurement
    assert by[(0, 0, 30)][1] > by[(0, 0, 175)][1], seen
    if shape == "sampled_cold":
        assert by[(1, 0, 175)][0] > 0 and by[(1, 0, 175)][1] < nmir // 10, seen      # sized by the sample: the sample's own rows and a few per cent of the others miss their slab
        assert by[(1, 0, 1000)][1] < by[(1, 0, 20)][1], seen
    else:
        assert by[(1, 0, 175)] == (0, nmir), seen                      # a cold call on a small matrix has no ratio: no slabs
    e.set_option("no_slab", 1); e.set_option("overlap_cold_calls", 0)
    st = e.create_seed_matrix()
    assert e.get_stat("overlap_slab_q16") == 0
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


@pytest.mark.parametrize("k,lo,up,bits", [(31, 2, 8, 0), (31, 3, 40, 0), (19, 2, 8, 0), (21, 2, 12, 14), (25, 2, 8, 9), (31, 2, 8, 20), (27, 2, 30, 10)])
def test_wide_partition_kmer_path_equals_the_oracle(k, lo, up, bits):
    """19 <= k <= 31 (the reference's default build is k = 31): value + read + position do not fit one word, the instances travel as 16-byte
    records through a two-level partition of their leading value bits and every bucket is sorted in LDS (kmer_msd.hip, k31_*).  Forced on a
    small input ("kmer_msd"; "msd_wide_bits": other splits of the value — few, large buckets; many, empty ones).  Low-complexity reads, reads
    shorter than k and reads around the word boundaries of the packed stream included: A, B and every statistic equal the oracle's."""
    reads, _ = synth.make_reads(178 + k, 60000, 14, 2500, 700, error=0.05, min_len=100)
    rng = np.random.default_rng(k)
    extra = [b"A" * 300, b"AC" * 200, b"T" * 150 + b"G" * 150, b"ACG" * 120, b"ACGT" * 90] * 3
    extra += [bytes(rng.choice(list(b"ACGT"), n).tolist()) for n in (k - 1, k, k + 1, k + 2, 31, 32, 33, 63, 64, 65, 3)]
    seqs = list(reads) + extra
    np.random.default_rng(5).shuffle(seqs)
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, k, lo, up, options={"kmer_msd": 1, "msd_wide_bits": bits})
    assert e.get_stat("kmer_path") == 2
    o = gu.oracle_run(packed, off, lens, k, lo, up, threads=8)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    assert (e.kmer_histogram() == o.A()["hist"]).all()
    e.create_kmer_matrix()                                  # the repeat call rebuilds from the column pointers (the sort keys were consumed)
    e.create_seed_matrix()
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


def test_wide_partition_gives_up_a_bucket_of_too_many_distinct_kmers():
    """Random reads cut into FOUR buckets (msd_wide_bits = 2): every bucket holds more distinct k-mers than k31_count's LDS table takes.  Round 4 sent the
    whole input to the sort of kmer.hip; since round 5 such buckets ALONE are taken out, sorted by k-mer and cut into pseudo-buckets of 2^16 distinct
    k-mers that the k <= 17 bucket kernels count and emit (kmer_msd.hip: k31_gather_crowded ...): the input stays on the partition path, same matrices."""
    reads, _ = synth.make_reads(9, 3700, 5, 600, 100, error=0.10, min_len=100)
    packed, off, lens = po.pack_reads(list(reads))
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 31, 2, 8, options={"kmer_msd": 1, "msd_wide_bits": 2})
    o = gu.oracle_run(packed, off, lens, 31, 2, 8, threads=8)
    assert 3300 * 4 < ks["instances"] < 4000 * 4 and ks["distinct"] > 3200 * 4 and e.get_stat("kmer_path") == 2      # (buckets of < 4096 instances, > 3072 of them distinct)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


@pytest.mark.parametrize("lo,up", [(2, 8), (2, 40)])
def test_wide_partition_keeps_a_satellite_bucket_and_a_homopolymer_on_the_partition_path(lo, up):
    """What a real genome brings (task 3): a SATELLITE — 10^5 distinct 31-mers that share their ten leading bases, three (or more) copies of each: one
    bucket of 3.6 * 10^5 records, 10^5 of them distinct, two pseudo-buckets — plus a HOMOPOLYMER (one k-mer, 24 000 records: walked in chunks by k31_count),
    on top of ordinary reads.  The input stays on the wide partition path (kmer_path == 2), equals the oracle entry for entry — also with UPPER = 40, where
    the matrix is dense and the satellite's columns are kept — and the stage takes at most 1.3 x what it takes without the two."""
    rng = np.random.default_rng(77)
    bp, bo, bl, _ = elba_amd.synth_reads(17, 5_000_000, 24, 2500, 600, error_rate=0.04, min_len=100)      # 120 M instances: the stage takes ~7 ms (the crowded bucket's own kernels ~1.2 ms, whatever the input's size)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    tails = bases[rng.integers(0, 4, size=(100000, 21))]
    sat = [b"A" * 10 + t.tobytes() for t in tails]                  # canonical = forward (it starts with ten A): all of them in the lowest bucket
    copies = [sat[i] for i in rng.integers(0, len(sat), size=60000)]
    extra = sat * 3 + copies + [b"C" * 630] * 40                     # every satellite k-mer 3-6 times; 40 x 600 instances of CCC...C
    k = 31

    def run(seqs):
        nb = int(bo[-1]) + (int(bl[-1]) + 3) // 4                   # bytes of the base reads (DnaBuffer layout: a read starts on a byte)
        if seqs:
            xp, xo, xl = po.pack_reads(seqs)
            packed, off, lens = np.concatenate([bp[:nb], xp]), np.concatenate([bo, xo + np.uint64(nb)]), np.concatenate([bl, xl])
        else:
            packed, off, lens = np.concatenate([bp[:nb], np.zeros(8, dtype=np.uint8)]), bo, bl
        e = elba_amd.Engine(k, lo, up, options={"kmer_msd": 1})
        e.set_reads(packed, off, lens)
        e.count_kmers(); e.create_kmer_matrix()                      # allocations
        best = None
        for _ in range(3):
            ks = e.count_kmers()
            best = ks["ms_total"] if best is None else min(best, ks["ms_total"])
        e.create_kmer_matrix()
        return e, ks, best, (packed, off, lens)

    e0, ks0, t_clean, _ = run([])
    assert e0.get_stat("kmer_path") == 2
    e0.close()
    e, ks, t_sat, (packed, off, lens) = run(extra)
    assert e.get_stat("kmer_path") == 2
    st = e.create_seed_matrix()
    o = gu.oracle_run(packed, off, lens, k, lo, up, threads=8)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    assert ks["reliable"] >= ks0["reliable"] + 99000                 # the satellite's k-mers are kept (3-6 copies each; a handful of the random tails coincide)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    assert t_sat <= 1.3 * t_clean, (t_sat, t_clean)
    e.close()


def test_wide_partition_walks_a_homopolymer_bucket_in_chunks():
    """A bucket far beyond what a workgroup holds in registers (tens of thousands of instances of ONE k-mer: homopolymer runs) is walked in chunks
    — the count table only holds the distinct k-mers — and stays on the wide path: same matrices."""
    reads, _ = synth.make_reads(7, 40000, 10, 2000, 500, error=0.05, min_len=100)
    seqs = list(reads) + [b"A" * 700] * 40 + [b"AC" * 400] * 20
    packed, off, lens = po.pack_reads(seqs)
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 31, 2, 8, options={"kmer_msd": 1})
    assert e.get_stat("kmer_path") == 2
    o = gu.oracle_run(packed, off, lens, 31, 2, 8, threads=8)
    assert (ks["instances"], ks["distinct"], ks["reliable"], ks["entries"]) == (o.stat("I"), o.stat("ndistinct"), o.stat("N"), o.stat("Z"))
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()


def test_reference_default_build_through_the_wide_partition():
    """The reference's bundled reads.fa at its default build (Makefile:1-3: k = 31, L = 15, U = 35) through the wide partition path: the figures
    the survey measured from the reference's own code and the oracle's matrices."""
    m = util.golden_meta()["reads_ref_appB"][1]
    assert (m["k"], m["lower"], m["upper"]) == (31, 15, 35)
    packed, off, lens = po.pack_reads(util.read_fasta(os.path.join(G, "reads_ref.fa.gz")))
    e, ks, ms, st = gu.gpu_full(packed, off, lens, 31, 15, 35, options={"kmer_msd": 1})
    assert e.get_stat("kmer_path") == 2
    assert (ks["nreads"], ks["instances"], ks["reliable"], ks["entries"]) == (m["M"], m["I"], m["N"], m["Z"])
    assert (st["products"], st["nnz_before_prune"], st["nnz"], st["nnz_upper"], st["max_numshared"]) == (m["P"], m["Yraw"], m["Y"], m["nupper"], m["maxshared"])
    o = gu.oracle_run(packed, off, lens, 31, 15, 35)
    gu.assert_A_equal(e.export_kmer_matrix(), o.A())
    gu.assert_B_equal(e.export_csr(), o.B())
    e.close()
