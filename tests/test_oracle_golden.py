"""CPU tests: the oracle (oracle/elba_oracle.c) against the committed golden vectors, which were produced by the
REFERENCE's own compiled code (tests/golden/make_golden.py) or measured from it by the survey."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from oracle import pyoracle as po

G = util.GOLDEN


def _rows(name):
    return [l.split() for l in open(os.path.join(G, name)) if l[0] != "#" and l.strip()]


@pytest.mark.parametrize("k", [17, 31])
def test_kmer_pack_twin_rep_hash(k):
    L = po.lib()
    for s, fwd, twin, rep, h, hf in _rows("kmer_vectors_k%d.txt" % k):
        w = L.orc_kmer_from_ascii(s.encode(), k)
        assert w == int(fwd, 16), s
        assert L.orc_kmer_twin(w, k) == int(twin, 16), s
        assert L.orc_kmer_rep(w, k) == int(rep, 16), s
        assert L.orc_kmer_hash(int(rep, 16)) == int(h, 16), s
        assert L.orc_kmer_hash(w) == int(hf, 16), s
        # twin is an involution; rep is the smaller of the two
        assert L.orc_kmer_twin(L.orc_kmer_twin(w, k), k) == w


def test_encode_vectors():
    L = po.lib()
    for s, hx in _rows("encode_vectors.txt"):
        mem = np.zeros((len(s) + 3) // 4, dtype=np.uint8)
        nb = L.orc_encode_read(s.encode(), len(s), mem.ctypes.data)
        assert nb == len(mem) and mem.tobytes().hex() == hx, s


def test_murmur_vectors():
    L = po.lib()
    for key, h1, h2 in _rows("murmur_vectors.txt"):
        b = b"" if key == "-" else bytes.fromhex(key)
        out = (C.c_uint64 * 2)()
        L.orc_murmur3_x64_128(b, len(b), 313, out)
        assert (out[0], out[1]) == (int(h1, 16), int(h2, 16)), key


@pytest.mark.parametrize("k", [17, 31])
def test_rolling_canonical_kmers(k):
    L = po.lib()
    for s, ks in _rows("read_kmers_k%d.txt" % k):
        buf, off, lens = po.pack_reads([s.encode()])
        out = np.zeros(max(1, len(s)), dtype=np.uint64)
        n = L.orc_read_kmers(buf.ctypes.data, len(s), k, out.ctypes.data)
        exp = [] if ks == "-" else [int(x, 16) for x in ks.split(",")]
        assert n == len(exp) and out[:n].tolist() == exp


def test_owner_formula():
    # src/KmerOps.cpp:352-359; KmerOps.cpp itself is unbuildable here (needs CombBLAS): formula-level check only
    L = po.lib()
    for p in (1, 2, 4, 8, 16):
        assert L.orc_kmer_owner(0, p) == 0
        assert L.orc_kmer_owner((1 << 63), p) == p // 2
        assert L.orc_kmer_owner((1 << 64) - (1 << 20), p) in (p - 1, p)  # double rounding at the very top, as in the reference
    rng = np.random.default_rng(5)
    for h in rng.integers(0, 1 << 63, 200, dtype=np.uint64).tolist():
        for p in (3, 8):
            o = L.orc_kmer_owner(h, p)
            assert o == int(float(h) * p / float((1 << 64) - 1))


def test_semiring_fold_shapes():
    # include/SharedSeeds.hpp:41-52; SURVEY.md App. B: left fold p1..p4 -> {(1,1),(4,4),4}; balanced tree -> {(1,1),(3,3),4}
    L = po.lib()
    p = [L.orc_sr_multiply(i, i) for i in range(1, 5)]
    acc = p[0]
    for x in p[1:]:
        acc = L.orc_sr_add(acc, x)
    assert (acc.q0, acc.t0, acc.q1, acc.t1, acc.numshared) == (1, 1, 4, 4, 4)
    tree = L.orc_sr_add(L.orc_sr_add(p[0], p[1]), L.orc_sr_add(p[2], p[3]))
    assert (tree.q0, tree.t0, tree.q1, tree.t1, tree.numshared) == (1, 1, 3, 3, 4)
    one = L.orc_sr_multiply(7, 9)
    assert (one.q0, one.t0, one.q1, one.t1, one.numshared) == (7, 9, 0, 0, 1)


def _check_set(name, meta):
    k, lo, up = meta["k"], meta["lower"], meta["upper"]
    seqs = util.read_fasta(os.path.join(G, name + ".fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(k, lo, up)
    o.count_and_build(buf, off, lens)
    o.spgemm(1)
    for key in ("M", "I", "N", "Z", "P", "Yraw", "Y"):
        assert o.stat(key) == meta[key], (name, key)
    A = o.A()
    km, rd, ps = util.triples_from_A(A)
    gk, gr, gp = util.read_triples(os.path.join(G, "%s_k%d_L%d_U%d.triples" % (name, k, lo, up)))
    assert (km == gk).all() and (rd == gr).all() and (ps == gp).all()
    gB = util.read_B(os.path.join(G, "%s_k%d_L%d_U%d.B" % (name, k, lo, up)))
    assert (util.b_triplets(o.B()) == gB).all()
    return o, buf, off, lens


def test_small_err_set():
    m = util.golden_meta()["small_err"][0]
    _check_set("small_err", m)


@pytest.mark.parametrize("idx", [0, 1])
def test_small_clean_set(idx):
    m = util.golden_meta()["small_clean"][idx]
    _check_set("small_clean", m)


def _seed_checks(o, buf, off, lens, k):
    """Canonical rule + test.py:57-65 validity + CSR/CSC consistency, straight from A."""
    L = po.lib()
    A, B = o.A(), o.B()
    rows = np.repeat(np.arange(B["M"]), np.diff(B["rowptr"]))
    # brute-force canonical seeds for a sample of entries
    rng = np.random.default_rng(1)
    idx = rng.choice(len(rows), size=min(300, len(rows)), replace=False) if len(rows) else []
    for e in idx:
        i, j, v = int(rows[e]), int(B["col"][e]), B["val"][e]
        prods = []
        for a in range(A["rowptr"][i], A["rowptr"][i + 1]):
            kid, q = int(A["csr_kid"][a]), int(A["csr_pos"][a])
            for f in range(A["colptr"][kid], A["colptr"][kid + 1]):
                if int(A["csc_read"][f]) == j:
                    prods.append((kid, q, int(A["csc_pos"][f])))
        assert len(prods) == v["numshared"] >= 2
        assert (v["q0"], v["t0"]) == min(prods)[1:] and (v["q1"], v["t1"]) == max(prods)[1:]
        for (q, t) in ((v["q0"], v["t0"]), (v["q1"], v["t1"])):
            assert L.orc_seed_is_valid(buf.ctypes.data + int(off[i]), int(lens[i]), buf.ctypes.data + int(off[j]), int(lens[j]), int(q), int(t), k)


def test_seed_rule_and_validity_small():
    m = util.golden_meta()["small_err"][0]
    o, buf, off, lens = _check_set("small_err", m)
    _seed_checks(o, buf, off, lens, m["k"])


@pytest.mark.parametrize("idx", [0, 1])
def test_reference_sample_appB(idx):
    """The reference's bundled reads.fa; expected figures measured by the survey from the reference's own KmerOps.cpp."""
    m = util.golden_meta()["reads_ref_appB"][idx]
    seqs = util.read_fasta(os.path.join(G, "reads_ref.fa.gz"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(m["k"], m["lower"], m["upper"])
    o.count_and_build(buf, off, lens)
    o.spgemm(2)
    for key in ("M", "I", "N", "Z", "P", "Yraw", "Y", "nupper", "maxshared"):
        assert o.stat(key) == m[key], key
    assert o.stat("Y") == o.stat("ndiag") + 2 * o.stat("nupper")  # symmetric pattern
    A = o.A()
    rows = np.repeat(np.arange(A["M"], dtype=np.int64), np.diff(A["rowptr"]))
    key = rows * A["N"] + A["csr_kid"]
    assert len(key) - len(np.unique(key)) == m["dups"]
    if idx == 0:
        _seed_checks(o, buf, off, lens, m["k"])


def test_threads_do_not_change_result():
    m = util.golden_meta()["small_err"][0]
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(m["k"], m["lower"], m["upper"])
    o.count_and_build(buf, off, lens)
    o.spgemm(1); B1 = o.B()
    o.spgemm(4); B4 = o.B()
    assert (B1["rowptr"] == B4["rowptr"]).all() and (B1["col"] == B4["col"]).all() and (B1["val"] == B4["val"]).all()


def test_dcsc_export_matches_consumer_walk():
    """src/PairwiseAlignment.cpp:28-56: walking (jc,cp,ir,numx) must visit exactly B's entries; strict upper = candidates."""
    m = util.golden_meta()["small_err"][0]
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(m["k"], m["lower"], m["upper"])
    o.count_and_build(buf, off, lens); o.spgemm(1)
    B = o.B(); M = B["M"]
    d = o.export_dcsc(0, M, 0, M)
    assert d["nnz"] == B["Y"]
    seen = {}
    for ci in range(d["nzc"]):
        for e in range(d["cp"][ci], d["cp"][ci + 1]):
            seen[(int(d["ir"][e]), int(d["jc"][ci]))] = d["numx"][e]
        assert (np.diff(d["ir"][d["cp"][ci]:d["cp"][ci + 1]]) > 0).all()
    rows = np.repeat(np.arange(M), np.diff(B["rowptr"]))
    assert len(seen) == B["Y"]
    for e in range(B["Y"]):
        assert seen[(int(rows[e]), int(B["col"][e]))] == B["val"][e]
    assert sum(1 for (r, c) in seen if r < c) == o.stat("nupper")
    # a grid-cell block (2x2 grid, cell (0,1)) carries local indices
    h = M // 2
    blk = o.export_dcsc(0, h, h, M)
    for ci in range(blk["nzc"]):
        for e in range(blk["cp"][ci], blk["cp"][ci + 1]):
            assert seen[(int(blk["ir"][e]), int(blk["jc"][ci]) + h)] == blk["numx"][e]


def test_triples_entry_reproduces_B():
    """create_seed_matrix's own input form: A handed over as (row, col, val) triples (src/KmerOps.cpp:380-400)."""
    m = util.golden_meta()["small_err"][0]
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(m["k"], m["lower"], m["upper"])
    o.count_and_build(buf, off, lens); o.spgemm(1)
    A, B = o.A(), o.B()
    rows = np.repeat(np.arange(A["M"], dtype=np.int64), np.diff(A["rowptr"]))
    perm = np.random.default_rng(3).permutation(A["Z"])
    o2 = po.Oracle(m["k"], m["lower"], m["upper"])
    o2.set_triples(A["M"], A["N"], rows[perm], A["csr_kid"].astype(np.int64)[perm], A["csr_pos"][perm])
    o2.spgemm(1)
    B2 = o2.B()
    assert (B["rowptr"] == B2["rowptr"]).all() and (B["col"] == B2["col"]).all() and (B["val"] == B2["val"]).all()


def test_column_hand_over_and_B_comparison():
    """The hand-over bench.py uses for the whole headline matrix: A given as its columns (u32 pointers, read << 32 | pos in (read, pos)
    order), CSR derived by row ranges on several threads; and the entry-by-entry comparison of a foreign B."""
    m = util.golden_meta()["small_err"][0]
    buf, off, lens = po.pack_reads(util.read_fasta(os.path.join(G, "small_err.fa")))
    o = po.Oracle(m["k"], m["lower"], m["upper"])
    o.count_and_build(buf, off, lens); o.spgemm(1)
    A, B = o.A(), o.B()
    csc = (A["csc_read"].astype(np.uint64) << np.uint64(32)) | A["csc_pos"].astype(np.uint64)
    for threads in (1, 3):
        o2 = po.Oracle(m["k"], m["lower"], m["upper"])
        o2.set_csc(A["M"], A["N"], A["colptr"].astype(np.uint32), csc, threads)
        A2 = o2.A()
        for key in ("colptr", "csc_read", "csc_pos", "rowptr", "csr_kid", "csr_pos"):
            assert (A[key] == A2[key]).all(), key
        o2.spgemm(threads)
        assert o2.compare_B(B["rowptr"], B["col"], B["val"], threads) == 0
    v = B["val"].copy(); v["t1"][5] ^= 1
    c = B["col"].copy(); c[7] += 1
    assert o.compare_B(B["rowptr"], B["col"], v, 2) == 1 and o.compare_B(B["rowptr"], c, v, 2) == 2
    assert o.compare_B(B["rowptr"][:-1], B["col"], B["val"]) == -1
    bad = csc.copy(); bad[[0, 1]] = bad[[1, 0]]
    if A["colptr"][1] >= 2 and bad[0] != bad[1]:
        with pytest.raises(RuntimeError):
            po.Oracle(m["k"], m["lower"], m["upper"]).set_csc(A["M"], A["N"], A["colptr"].astype(np.uint32), bad, 1)


@pytest.mark.skipif(po.ref_lib(17) is None, reason="oracle/_ref not built (reference tree absent)")
def test_live_against_reference_primitives():
    """Random cross-check against the reference's compiled Kmer/DnaSeq/HashFuncs (only where oracle/_ref exists)."""
    L = po.lib()
    rng = np.random.default_rng(99)
    for k in (17, 31):
        R = po.ref_lib(k)
        for _ in range(20):
            ln = int(rng.integers(k, 400))
            s = bytes(rng.choice(list(b"ACGTNacgtn"), ln).tolist())
            m1 = np.zeros((ln + 3) // 4 + 8, dtype=np.uint8); m2 = m1.copy()
            R.ref_encode(s, ln, m1.ctypes.data); L.orc_encode_read(s, ln, m2.ctypes.data)
            assert (m1 == m2).all()
            o1 = np.zeros(ln, dtype=np.uint64); o2 = o1.copy()
            n1 = R.ref_kmers(m1.ctypes.data, ln, o1.ctypes.data, 1)
            n2 = L.orc_read_kmers(m2.ctypes.data, ln, k, o2.ctypes.data)
            assert n1 == n2 and (o1 == o2).all()
            for w in o1[:5].tolist():
                x = C.c_uint64(w)
                assert R.ref_kmer_hash(C.byref(x)) == L.orc_kmer_hash(w)


@pytest.mark.parametrize("k", [33, 63, 65, 95])
def test_two_word_kmers_match_reference_vectors(k):
    """32 < k < 96 (NLONGS == 2 and 3): every canonical k-mer of the sample reads, every word, against the reference's Kmer<N>::GetRepKmers."""
    L = po.lib()
    n_checked = 0
    for line in open(os.path.join(G, "read_kmers2_k%d.txt" % k)):
        if line[0] == "#":
            continue
        s, ks = line.split()
        buf, off, lens = po.pack_reads([s.encode()])
        want = [] if ks == "-" else [tuple(int(x, 16) for x in t.split(":")) for t in ks.split(",")]
        assert len(want) == max(0, len(s) - k + 1)
        for p_, w in enumerate(want):
            out = np.zeros(3, dtype=np.uint64)
            L.orc_kmerN_at(buf.ctypes.data, p_, k, out.ctypes.data)
            assert tuple(int(x) for x in out[:len(w)]) == w and all(int(x) == 0 for x in out[len(w):]), (s, p_)
            if k <= 64:
                o2 = np.zeros(2, dtype=np.uint64)
                L.orc_kmer2_at(buf.ctypes.data, p_, k, o2.ctypes.data)
                assert (int(o2[0]), int(o2[1])) == w
            n_checked += 1
    assert n_checked > 200


def test_two_word_counting_is_consistent_with_one_word_counting():
    """The same reads at k = 31 (one word) and k = 33 (two words) go through the same counting code; at k = 33 the reliable set must equal
    a dictionary count over orc_kmer2_at (itself pinned to the reference above)."""
    seqs = util.read_fasta(os.path.join(G, "small_clean.fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(33, 2, 12); o.count_and_build(buf, off, lens)
    L = po.lib()
    cnt = {}
    for r in range(len(lens)):
        for p_ in range(max(0, int(lens[r]) - 33 + 1)):
            out = np.zeros(2, dtype=np.uint64)
            L.orc_kmer2_at(buf.ctypes.data + int(off[r]), p_, 33, out.ctypes.data)
            cnt.setdefault((int(out[0]), int(out[1])), []).append((r, p_))
    rel = sorted(km for km, v in cnt.items() if 2 <= len(v) <= 12)
    A = o.A()
    assert A["N"] == len(rel) and [(int(a), int(b)) for a, b in zip(A["kmers"], A["kmers_lo"])] == rel
    for kid, km in enumerate(rel):
        e0, e1 = int(A["colptr"][kid]), int(A["colptr"][kid + 1])
        assert list(zip(A["csc_read"][e0:e1].tolist(), A["csc_pos"][e0:e1].tolist())) == sorted(cnt[km])


def test_reference_kmer_numbering_replay_matches_the_survey_and_leaves_pattern_and_counts_unchanged():
    """SURVEY.md §8c-3.  tests/golden/*.order hold the k-mer ids of a one-rank reference run (unordered_map iteration order after
    reserve(ceil(HLL)), replayed on the reference's own compiled code).  (a) For the reference's bundled reads.fa the replay's HLL
    estimate, bucket count, keys after pass 1 and N are the figures the survey measured from the reference's own KmerOps.cpp (App. B).
    (b) Under that numbering the oracle's B has the same pattern and the same numshared as under the canonical (value-rank) numbering;
    the seeds follow the numbering, and each is still the min / max product of ITS numbering (checked against a brute-force fold)."""
    order, meta = util.read_order(os.path.join(G, "reads_ref_k17_L2_U8.order"))
    assert len(order) == 14751 and int(meta["buckets"]) == 299951 and int(meta["keys_after_pass1"]) == 136991 and round(meta["hll"]) == 283870
    rd, ref_ids, ps, canon_ids, N, _ = util.libstdcxx_triples("small_err")
    M = util.golden_meta()["small_err"][0]["M"]
    oc = po.Oracle(17, 2, 8); oc.set_triples(M, N, rd, canon_ids, ps); oc.spgemm(1)
    orf = po.Oracle(17, 2, 8); orf.set_triples(M, N, rd, ref_ids, ps); orf.spgemm(1)
    Bc, Br = oc.B(), orf.B()
    assert (Bc["rowptr"] == Br["rowptr"]).all() and (Bc["col"] == Br["col"]).all() and (Bc["val"]["numshared"] == Br["val"]["numshared"]).all()
    assert (Bc["val"] != Br["val"]).any()
    # brute-force fold under the reference numbering: seeds[0] / seeds[1] = products with minimal / maximal (id, posQ, posT)
    cols = {}
    for r, c, p in zip(rd.tolist(), ref_ids.tolist(), ps.tolist()):
        cols.setdefault(c, []).append((r, p))
    best = {}
    for c, ents in cols.items():
        for (i, pq) in ents:
            for (j, pt) in ents:
                key, cand = (i, j), (c, pq, pt)
                lo, hi = best.get(key, (cand, cand))
                best[key] = (min(lo, cand), max(hi, cand))
    rows = np.repeat(np.arange(M), np.diff(Br["rowptr"]))
    for x in range(0, Br["Y"], 7):
        i, j, v = int(rows[x]), int(Br["col"][x]), Br["val"][x]
        lo, hi = best[(i, j)]
        assert (int(v["q0"]), int(v["t0"]), int(v["q1"]), int(v["t1"])) == (lo[1], lo[2], hi[1], hi[2])


def test_reference_default_build_numbering_k31():
    """The same for the reference's DEFAULT build (Makefile:1-3: k = 31, L = 15, U = 35) on its bundled reads.fa: the replay's N and keys after
    pass 1 are the survey's figures from the reference's own KmerOps.cpp (105 754 and 146 244), the reliable set is the oracle's,
    and pattern + numshared + every count do not depend on the numbering (P = 65 606 685, Y = 12 021 as measured there)."""
    M, N, rd, ref_ids, ps, canon_ids, meta = util.reference_default_triples()
    assert N == 105754 and int(meta["keys_after_pass1"]) == 146244 and len(rd) == 2579051
    oc = po.Oracle(31, 15, 35); oc.set_triples(M, N, rd, canon_ids, ps); oc.spgemm(4)
    orf = po.Oracle(31, 15, 35); orf.set_triples(M, N, rd, ref_ids, ps); orf.spgemm(4)
    for o in (oc, orf):
        assert (o.stat("P"), o.stat("Yraw"), o.stat("Y"), o.stat("nupper"), o.stat("maxshared")) == (65606685, 12021, 12021, 5897, 18074)
    Bc, Br = oc.B(), orf.B()
    assert (Bc["rowptr"] == Br["rowptr"]).all() and (Bc["col"] == Br["col"]).all() and (Bc["val"]["numshared"] == Br["val"]["numshared"]).all()
    assert (Bc["val"] != Br["val"]).any()


@pytest.mark.parametrize("k,lo,up,threads", [(17, 2, 8, 2), (17, 2, 8, 7), (31, 3, 30, 4), (33, 2, 8, 3), (65, 2, 8, 5), (9, 2, 60, 8)])
def test_the_oracles_kmer_stage_on_several_threads_equals_the_one_thread_statement(k, lo, up, threads):
    """orc_count_and_build_mt (bench.py's all-cores CPU figure for the k-mer stage: reads split over the threads, 256 value buckets sorted and counted
    independently) against the plain one-thread restatement of a7-a10: every array of A, the count histogram and the counters — including reads
    shorter than k, empty reads and more threads than reads would fill."""
    import elba_amd
    packed, off, lens, info = elba_amd.synth_reads(100 + k, 120000, 15, 1500, 900, error_rate=0.07, min_len=5)
    o1 = po.Oracle(k, lo, up); o1.count_and_build(packed, off, lens)
    o2 = po.Oracle(k, lo, up); o2.count_and_build(packed, off, lens, threads)
    A1, A2 = o1.A(), o2.A()
    for key in A1:
        if isinstance(A1[key], np.ndarray):
            assert np.array_equal(A1[key], A2[key]), key
        else:
            assert A1[key] == A2[key], key
    assert all(o1.stat(s) == o2.stat(s) for s in ("I", "N", "Z", "ndistinct", "M"))
    o1.spgemm(2); o2.spgemm(2)
    assert o1.stat("Y") == o2.stat("Y") and o1.stat("P") == o2.stat("P")
    # a handful of reads on many threads (most threads hold no read)
    o3 = po.Oracle(k, lo, up); o3.count_and_build(packed, off[:5], lens[:5], 16)
    o4 = po.Oracle(k, lo, up); o4.count_and_build(packed, off[:5], lens[:5])
    assert o3.stat("N") == o4.stat("N") and o3.stat("Z") == o4.stat("Z") and np.array_equal(o3.A()["csr_kid"], o4.A()["csr_kid"])
