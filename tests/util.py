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
