"""Shared helpers for the tests: FASTA/golden readers and the oracle-side reference computations."""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    seqs, cur = [], []
    with op(path, "rb") as f:
        for line in f:
            if line[:1] == b">":
                if cur:
                    seqs.append(b"".join(cur)); cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return seqs


def golden_meta():
    return json.load(open(os.path.join(GOLDEN, "golden_meta.json")))


def read_triples(path):
    km, rd, ps = [], [], []
    for line in open(path):
        if line[0] == "#":
            continue
        a, b, c = line.split()
        km.append(int(a, 16)); rd.append(int(b)); ps.append(int(c))
    return np.array(km, dtype=np.uint64), np.array(rd, dtype=np.int64), np.array(ps, dtype=np.uint32)


def read_B(path):
    rows = [tuple(int(x) for x in line.split()) for line in open(path) if line[0] != "#"]
    return np.array(rows, dtype=np.int64).reshape(-1, 3)


def triples_from_A(A):
    """(kmer value, read, pos) of every CSC entry, in CSC order (== sorted by (kmer, read, pos))."""
    cnt = np.diff(A["colptr"])
    km = np.repeat(A["kmers"], cnt)
    return km, A["csc_read"].astype(np.int64), A["csc_pos"]


def b_triplets(B):
    rows = np.repeat(np.arange(B["M"], dtype=np.int64), np.diff(B["rowptr"]))
    return np.stack([rows, B["col"].astype(np.int64), B["val"]["numshared"].astype(np.int64)], axis=1)


def read_order(path):
    """tests/golden/*.order: (k-mers by reference k-mer id, header fields) — see tests/golden/make_order_golden.py."""
    op = (lambda: gzip.open(path, "rt")) if path.endswith(".gz") else (lambda: open(path))
    head = op().readline()
    meta = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in head.split() if "=" in kv}
    km = np.array([int(line, 16) for line in op() if line[0] != "#"], dtype=np.uint64)
    return km, meta


def libstdcxx_triples(name):
    """The golden A triples of `name` renumbered the way a one-rank reference run numbers its k-mers:
    returns (M-agnostic) reads, k-mer ids under the reference order, positions, the canonical (value-rank) ids, N."""
    gk, gr, gp = read_triples(os.path.join(GOLDEN, "%s_k17_L2_U8.triples" % name))
    order, meta = read_order(os.path.join(GOLDEN, "%s_k17_L2_U8.order" % name))
    uk, canon = np.unique(gk, return_inverse=True)
    assert len(order) == len(uk) and (np.sort(order) == uk).all()          # the reference's map holds exactly the reliable k-mers
    id_of_value = {int(v): i for i, v in enumerate(order.tolist())}
    ref_ids = np.array([id_of_value[int(v)] for v in gk.tolist()], dtype=np.int64)
    return gr, ref_ids, gp, canon.astype(np.int64), len(uk), meta


def reference_default_triples():
    """The reference's DEFAULT build (Makefile:1-3: k = 31, L = 15, U = 35) on its bundled reads.fa, numbered the way a one-rank run of the
    reference numbers its k-mers (tests/golden/reads_ref_k31_L15_U35.order.gz): (M, N, reads, reference ids, positions, canonical ids, header).
    The entries come from the oracle's count (pinned to SURVEY.md App. B's N = 105 754, Z = 2 579 051 for this set)."""
    from oracle import pyoracle as po
    buf, off, lens = po.pack_reads(read_fasta(os.path.join(GOLDEN, "reads_ref.fa.gz")))
    o = po.Oracle(31, 15, 35)
    o.count_and_build(buf, off, lens)
    A = o.A()
    order, meta = read_order(os.path.join(GOLDEN, "reads_ref_k31_L15_U35.order.gz"))
    assert len(order) == A["N"] and (np.sort(order) == A["kmers"]).all()          # the reference's map holds exactly the reliable k-mers
    ref_of_canon = np.empty(A["N"], dtype=np.int64)
    ref_of_canon[np.searchsorted(A["kmers"], order)] = np.arange(A["N"], dtype=np.int64)
    canon = np.repeat(np.arange(A["N"], dtype=np.int64), np.diff(A["colptr"]))
    return A["M"], A["N"], A["csc_read"].astype(np.int64), ref_of_canon[canon], A["csc_pos"], canon, meta
