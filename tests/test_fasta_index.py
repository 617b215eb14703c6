"""Host side of the FASTA ingest (elba_amd/fasta.py): .fai records and chunk bounds as the reference's FastaIndex computes them
(src/FastaIndex.cpp:15-23, :222-224, :256-283) — checked against the reference's bundled sample and a plain Python re-read."""
import gzip
import os

import numpy as np

import util
from elba_amd import fasta

G = util.GOLDEN


def _extract(chunk, start, rec):
    """the per-record loop of FastaIndex::getmydna (src/FastaIndex.cpp:256-283): `bases` characters per line, one newline byte skipped"""
    out, loc, remain = [], int(rec["pos"]) - start, int(rec["len"])
    while remain > 0:
        cnt = min(int(rec["bases"]), remain)
        out.append(chunk[loc:loc + cnt]); remain -= cnt; loc += cnt + 1
    return b"".join(out)


def test_fai_of_the_reference_sample_and_rank_chunks(tmp_path):
    p = str(tmp_path / "reads.fa")
    with gzip.open(os.path.join(G, "reads_ref.fa.gz"), "rb") as f, open(p, "wb") as o:
        o.write(f.read())
    fasta.write_fai(p)
    names, recs = fasta.read_fai(p + ".fai")
    seqs = util.read_fasta(p)
    assert len(recs) == len(seqs) == 227 and [int(r["len"]) for r in recs] == [len(s) for s in seqs]
    size = os.path.getsize(p)
    for lo, hi in [(0, 227), (0, 60), (60, 150), (150, 227), (226, 227)]:
        chunk, start = fasta.load_chunk(p, recs[lo:hi])
        s0, e0 = fasta.chunk_bounds(recs[lo:hi], size)
        assert start == s0 == int(recs[lo]["pos"]) and len(chunk) == e0 - s0
        for r in range(lo, hi):
            assert _extract(chunk, start, recs[r]) == seqs[r]


def test_fai_line_format(tmp_path):
    p = str(tmp_path / "x.fa")
    open(p, "wb").write(b">a desc\nACGTAC\nGT\n>b\nTTTT\n")
    fasta.write_fai(p)
    assert open(p + ".fai").read().split("\n")[:2] == ["a\t8\t8\t6\t7", "b\t4\t21\t4\t5"]
    names, recs = fasta.read_fai(p + ".fai")
    assert names == ["a", "b"] and recs.dtype == fasta.FAI_DTYPE and int(recs[1]["pos"]) == 21
