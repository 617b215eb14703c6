#!/usr/bin/env python3
"""Regenerates tests/golden/* IN THE BUILD CONTAINER (needs /root/reference and `make -C oracle ref`).

Every expected value written here comes from the REFERENCE's own compiled code (oracle/_ref: DnaSeq::compress,
Kmer<1>::{set_kmer,GetTwin,GetRep,GetHash,GetRepKmers}, murmurhash3, Bloom) or from figures the survey measured from
the reference — never from oracle/elba_oracle.c, which these fixtures exist to check.
The B fixtures (pattern + numshared) are derived from the reference-produced A triples by a brute-force Python dict
fold (definition of the semiring product count, include/SharedSeeds.hpp:41-52), independent of the oracle.
"""
import ctypes as C
import gzip
import json
import os
import sys
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402  (pack_reads only: encoder cross-checked below against ref_encode)
import synth  # noqa: E402

REF = "/root/reference"


def ref_pack(R, seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    nb = (lens.astype(np.int64) + 3) // 4
    off = np.zeros(len(seqs), dtype=np.uint64)
    off[1:] = np.cumsum(nb)[:-1]
    buf = np.zeros(int(nb.sum()) + 8, dtype=np.uint8)
    for i, s in enumerate(seqs):
        R.ref_encode(s, len(s), buf.ctypes.data + int(off[i]))
    return buf, off, lens


def kmer_vectors(k, rng, n=64):
    R = po.ref_lib(k)
    rows = []
    specials = [b"A" * k, b"T" * k, b"C" * k, b"G" * k, (b"ACGT" * 9)[:k], (b"TGCA" * 9)[:k], b"N" * k, (b"acgtn" * 8)[:k]]
    for i in range(n):
        s = specials[i] if i < len(specials) else bytes(rng.choice(list(b"ACGT"), k).tolist())
        f, t, r = C.c_uint64(), C.c_uint64(), C.c_uint64()
        R.ref_kmer_from_ascii(s, C.byref(f), C.byref(t), C.byref(r))
        h = R.ref_kmer_hash(C.byref(r))
        hf = R.ref_kmer_hash(C.byref(f))
        rows.append("%s %016x %016x %016x %016x %016x" % (s.decode(), f.value, t.value, r.value, h, hf))
    with open(os.path.join(HERE, "kmer_vectors_k%d.txt" % k), "w") as fo:
        fo.write("# ascii fwd twin rep murmur3_64(rep) murmur3_64(fwd)   [reference Kmer<1>, KMER_SIZE=%d]\n" % k)
        fo.write("\n".join(rows) + "\n")


def encode_vectors(rng):
    R = po.ref_lib(17)
    rows = []
    cases = [b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", b"acgtn", b"NNNNNNN", b"TTTTTTTTT", b"GATTACAGATTACAN", b"tgcaTGCAnN"]
    for ln in (16, 17, 18, 19, 63, 64, 65, 66, 257):
        cases.append(bytes(rng.choice(list(b"ACGTacgtNn"), ln).tolist()))
    for s in cases:
        mem = np.zeros((len(s) + 3) // 4, dtype=np.uint8)
        nb = R.ref_encode(s, len(s), mem.ctypes.data)
        assert nb == len(mem)
        rows.append("%s %s" % (s.decode(), mem.tobytes().hex()))
    with open(os.path.join(HERE, "encode_vectors.txt"), "w") as fo:
        fo.write("# ascii packed_hex   [reference DnaSeq::compress]\n")
        fo.write("\n".join(rows) + "\n")


def murmur_vectors(rng):
    R = po.ref_lib(17)
    rows = []
    for ln in (0, 1, 7, 8, 9, 15, 16, 17, 24, 31, 32, 40):
        key = bytes(rng.integers(0, 256, ln, dtype=np.uint8).tolist())
        out = (C.c_uint64 * 2)()
        R.ref_murmur3_128(key, ln, out)
        rows.append("%s %016x %016x" % (key.hex() or "-", out[0], out[1]))
    with open(os.path.join(HERE, "murmur_vectors.txt"), "w") as fo:
        fo.write("# key_hex h1 h2   [reference murmurhash3_128, seed 313]\n")
        fo.write("\n".join(rows) + "\n")


def read_kmer_vectors(k, rng):
    """Rolling canonical k-mers of whole reads (GetRepKmers), lengths around k and byte boundaries."""
    R = po.ref_lib(k)
    rows = []
    for ln in (k - 1, k, k + 1, k + 2, k + 3, k + 4, 3 * k, 100):
        s = bytes(rng.choice(list(b"ACGT"), ln).tolist())
        mem = np.zeros((ln + 3) // 4 + 8, dtype=np.uint8)
        R.ref_encode(s, ln, mem.ctypes.data)
        out = np.zeros(max(1, ln), dtype=np.uint64)
        n = R.ref_kmers(mem.ctypes.data, ln, out.ctypes.data, 1)
        rows.append("%s %s" % (s.decode(), ",".join("%016x" % v for v in out[:n]) or "-"))
    with open(os.path.join(HERE, "read_kmers_k%d.txt" % k), "w") as fo:
        fo.write("# ascii_read canonical_kmers_hex   [reference Kmer<1>::GetRepKmers, KMER_SIZE=%d]\n" % k)
        fo.write("\n".join(rows) + "\n")


def brute_B(M, reads, kmers, pos):
    """B pattern + numshared straight from the definition: for every pair of A entries in the same column one product."""
    cols = defaultdict(list)
    for km, r, p in zip(kmers.tolist(), reads.tolist(), pos.tolist()):
        cols[km].append((r, p))
    cnt = defaultdict(int)
    P = 0
    for km, ents in cols.items():
        for (i, _) in ents:
            for (j, _) in ents:
                cnt[(i, j)] += 1
                P += 1
    return cnt, P


def small_set(name, seed, k, lower, upper, **kw):
    reads, _ = synth.make_reads(seed, **kw)
    with open(os.path.join(HERE, name + ".fa"), "w") as fo:
        for i, s in enumerate(reads):
            fo.write(">%d\n%s\n" % (i + 1, s.decode()))
    R = po.ref_lib(k)
    buf, off, lens = ref_pack(R, reads)
    buf2, off2, lens2 = po.pack_reads(reads)
    assert (buf == buf2).all() and (off == off2).all()
    cap = int(lens.sum())
    okm = np.zeros(cap, dtype=np.uint64); ord_ = np.zeros(cap, dtype=np.int64); opos = np.zeros(cap, dtype=np.uint32)
    I = int(sum(max(0, int(l) - k + 1) for l in lens))
    res = {}
    variants = {}
    for tag, entries in (("I", I), ("tiny", max(8, I // 64)), ("huge", 8 * I)):
        keys1 = C.c_int64()
        Z = R.ref_replay_count(buf.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens), lower, upper, entries,
                               okm.ctypes.data, ord_.ctypes.data, opos.ctypes.data, cap, C.byref(keys1))
        assert Z >= 0
        variants[tag] = (okm[:Z].copy(), ord_[:Z].copy(), opos[:Z].copy(), keys1.value)
    base = variants["I"]
    for tag, v in variants.items():  # Bloom sizing must not change the result
        assert len(v[0]) == len(base[0]) and (v[0] == base[0]).all() and (v[1] == base[1]).all() and (v[2] == base[2]).all(), tag
    km, rd, ps, keys1 = base
    Z = len(km)
    N = len(np.unique(km))
    cnt, P = brute_B(len(reads), rd, km, ps)
    Yraw = len(cnt)
    kept = sorted((i, j, n) for (i, j), n in cnt.items() if n > 1)
    with open(os.path.join(HERE, "%s_k%d_L%d_U%d.triples" % (name, k, lower, upper)), "w") as fo:
        fo.write("# kmer_hex read pos   sorted   [reference KmerOps two-pass replay on reference Bloom+Kmer]\n")
        for a, b, c in zip(km.tolist(), rd.tolist(), ps.tolist()):
            fo.write("%016x %d %d\n" % (a, b, c))
    with open(os.path.join(HERE, "%s_k%d_L%d_U%d.B" % (name, k, lower, upper)), "w") as fo:
        fo.write("# row col numshared   (after prune numshared<=1; full matrix)\n")
        for i, j, n in kept:
            fo.write("%d %d %d\n" % (i, j, n))
    res.update(dict(k=k, lower=lower, upper=upper, M=len(reads), I=I, N=N, Z=Z, P=P, Yraw=Yraw, Y=len(kept),
                    keys_after_pass1_bloomI=keys1, bloom_variants_checked=list(variants)))
    return res


def xdrop_vectors(name, k, lower, upper, rng):
    """f1: (i, j, seedQ, seedT, mat, mis, gap, xdrop) -> what the reference's own xdrop_aligner + classify_alignment return
    (src/XDropAligner.cpp, through oracle/_ref).  The INPUT seeds are the oracle's seeds[0] of the strict upper triangle of B (any
    shared k-mer would do) plus some perturbed / invalid ones; every expected value comes from the reference's code."""
    import util
    R = po.ref_lib(k)
    seqs = util.read_fasta(os.path.join(HERE, name + ".fa"))
    buf, off, lens = ref_pack(R, seqs)
    o = po.Oracle(k, lower, upper); o.count_and_build(buf, off, lens); o.spgemm(1)
    B = o.B()
    rows = np.repeat(np.arange(B["M"]), np.diff(B["rowptr"]))
    params = [(1, -1, -1, 15), (1, -2, -3, 30), (2, -3, -2, 7), (1, -1, -1, 0), (1, -1, -1, 49)]
    with open(os.path.join(HERE, "xdrop_%s_k%d.txt" % (name, k)), "w") as f:
        f.write("# i j seedQ seedT mat mis gap xdrop -> ret begQ endQ begT endT score rc OverlapClass   (reference: src/XDropAligner.cpp, KMER_SIZE=%d)\n" % k)
        n = 0
        for e in range(B["Y"]):
            i, j = int(rows[e]), int(B["col"][e])
            if i >= j:
                continue
            q0, t0 = int(B["val"][e]["q0"]), int(B["val"][e]["t0"])
            cases = [(q0, t0, params[n % len(params)])]
            if n % 7 == 0:
                cases.append((q0 + int(rng.integers(-3, 4)), t0, params[0]))          # usually not a shared k-mer any more: rejected
            if n % 11 == 0:
                cases.append((int(B["val"][e]["q1"]), int(B["val"][e]["t1"]), params[1]))   # the other stored seed
            for (a, b, (mat, mis, gap, x)) in cases:
                out = po.ref_xdrop(R, buf[int(off[i]):], int(lens[i]), buf[int(off[j]):], int(lens[j]), a, b, mat, mis, gap, x)
                f.write("%d %d %d %d %d %d %d %d %s\n" % (i, j, a, b, mat, mis, gap, x, " ".join(str(v) for v in out)))
            n += 1
    return n


def read_kmer_vectors2(k, rng):
    """Canonical k-mers of whole reads for 32 < k <= 64 (NLONGS == 2): both words of every k-mer, from the reference's Kmer<2>::GetRepKmers."""
    R = po.ref_lib(k)
    nl = R.ref_kmer_nlongs()
    assert nl == (k + 31) // 32 and nl in (2, 3)
    rows = []
    for ln in (k - 1, k, k + 1, k + 2, k + 3, k + 4, 2 * k + 5, 3 * k, 200 if k < 64 else 320):
        s = bytes(rng.choice(list(b"ACGT"), ln).tolist())
        if ln in (200, 320):
            h = ln * 3 // 10
            s = s[:h] + bytes(reversed(s[:h].translate(bytes.maketrans(b"ACGT", b"TGCA")))) + s[2 * h:]     # a reverse-complement repeat: twin == forward cases nearby
        mem = np.zeros((ln + 3) // 4 + 8, dtype=np.uint8)
        R.ref_encode(s, ln, mem.ctypes.data)
        out = np.zeros(nl * max(1, ln), dtype=np.uint64)
        n = R.ref_kmers(mem.ctypes.data, ln, out.ctypes.data, 1)
        rows.append("%s %s" % (s.decode(), ",".join(":".join("%016x" % out[nl * i + w] for w in range(nl)) for i in range(n)) or "-"))
    with open(os.path.join(HERE, "read_kmers2_k%d.txt" % k), "w") as fo:
        fo.write("# ascii_read canonical_kmers_hex (longs[0]:longs[1][:longs[2]])   [reference Kmer<%d>::GetRepKmers, KMER_SIZE=%d]\n" % (nl, k))
        fo.write("\n".join(rows) + "\n")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "kmers2":         # only the two-word k-mer vectors
        rng = np.random.default_rng(20261005)
        read_kmer_vectors2(33, rng); read_kmer_vectors2(63, rng); read_kmer_vectors2(65, rng); read_kmer_vectors2(95, rng)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "xdrop":          # only the x-drop vectors (the other fixtures stay as committed)
        assert po.ref_lib(17) is not None, "run `make -C oracle ref` first"
        rng = np.random.default_rng(20261004)
        print(xdrop_vectors("small_err", 17, 2, 8, rng), xdrop_vectors("small_clean", 17, 2, 8, rng))
        return
    assert os.path.isdir(REF), "needs the reference tree"
    assert po.ref_lib(17) is not None and po.ref_lib(31) is not None, "run `make -C oracle ref` first"
    rng = np.random.default_rng(20261003)
    kmer_vectors(17, rng); kmer_vectors(31, rng)
    encode_vectors(rng); murmur_vectors(rng)
    read_kmer_vectors(17, rng); read_kmer_vectors(31, rng)
    meta = {}
    meta["small_err"] = [small_set("small_err", 11, 17, 2, 8, genome_length=6000, depth=12, avg_len=900, sd_len=200, error=0.08)]
    meta["small_clean"] = [small_set("small_clean", 12, 17, 2, 8, genome_length=5000, depth=5, avg_len=700, sd_len=150, error=0.0, repeats=2),
                           small_set("small_clean", 12, 31, 3, 12, genome_length=5000, depth=5, avg_len=700, sd_len=150, error=0.0, repeats=2)]
    # the reference's bundled sample + the figures the survey measured from the reference's KmerOps.cpp on it
    raw = open(os.path.join(REF, "reads.fa"), "rb").read()
    with gzip.GzipFile(os.path.join(HERE, "reads_ref.fa.gz"), "wb", compresslevel=9, mtime=0) as g:
        g.write(raw)
    meta["reads_ref_appB"] = [
        dict(k=17, lower=2, upper=8, M=227, I=3321268, N=14751, Z=51086, dups=133, P=236778, Yraw=2550, Y=2479, nupper=1130, maxshared=3145,
             note="SURVEY.md App. B prints diag=220, inconsistent with its own Y=2479=diag+2*1130 -> diag=219"),
        dict(k=31, lower=15, upper=35, M=227, I=3318090, N=105754, Z=2579051, dups=1, P=65606685, Yraw=12021, Y=12021, ndiag=227, nupper=5897, maxshared=18074),
    ]
    json.dump(meta, open(os.path.join(HERE, "golden_meta.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
