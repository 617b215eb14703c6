"""-m gpu: FASTA chunk -> 2-bit DnaBuffer on the device (elba_set_reads_fasta) against the oracle's encoder and the reference's own
DnaSeq::compress vectors (tests/golden/encode_vectors.txt)."""
import os

import numpy as np
import pytest

import elba_amd
from elba_amd import fasta
import gpu_util as gu
import util
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
G = util.GOLDEN


def _write_fasta(path, seqs, width):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">read%d some description\n" % i)
            if width <= 0:
                f.write(s + b"\n")
            else:
                for a in range(0, len(s), width):
                    f.write(s[a:a + width] + b"\n")


def _ingest_and_compare(path, seqs, lo=0, hi=None):
    fasta.write_fai(path)
    names, recs = fasta.read_fai(path + ".fai")
    assert len(recs) == len(seqs) and [int(r["len"]) for r in recs] == [len(s) for s in seqs]
    hi = len(seqs) if hi is None else hi
    chunk, start = fasta.load_chunk(path, recs[lo:hi])
    e = elba_amd.Engine(17, 2, 8)
    st = e.set_reads_fasta(chunk, start, recs[lo:hi], first_global_id=lo)
    want, woff, wlen = po.pack_reads(seqs[lo:hi])
    assert st["nreads"] == hi - lo and st["bases"] == int(wlen.sum())
    got, goff, glen = e.export_reads(hi - lo, st["packed_bytes"])
    assert (goff == woff).all() and (glen == wlen).all()
    assert (got[:st["packed_bytes"]] == want[:st["packed_bytes"]]).all()
    return e, (want, woff, wlen)


def test_reference_compress_vectors_through_the_gpu_encoder(tmp_path):
    """every ASCII string of the reference-generated encode vectors, as one FASTA record each (lengths 1.., N/n, lower case)"""
    vec = [line.split() for line in open(os.path.join(G, "encode_vectors.txt")) if line[0] != "#"]
    seqs = [v[0].encode() for v in vec]
    p = str(tmp_path / "vec.fa")
    _write_fasta(p, seqs, 0)
    fasta.write_fai(p)
    names, recs = fasta.read_fai(p + ".fai")
    chunk, start = fasta.load_chunk(p, recs)
    e = elba_amd.Engine(17, 2, 8)
    st = e.set_reads_fasta(chunk, start, recs)
    got, off, ln = e.export_reads(len(seqs), st["packed_bytes"])
    for i, v in enumerate(vec):
        nb = (len(v[0]) + 3) // 4
        assert got[int(off[i]):int(off[i]) + nb].tobytes().hex() == v[1], (v[0], v[1])
    e.close()


@pytest.mark.parametrize("width", [0, 60, 61, 7, 1])
def test_wrapped_fasta_any_line_width_and_rank_chunks(tmp_path, width):
    rng = np.random.default_rng(width + 1)
    alphabet = np.frombuffer(b"ACGTacgtNnXR", dtype=np.uint8)       # incl. characters outside the code table (code 4 is ORed in, src/DnaSeq.cpp:18-24)
    seqs = [alphabet[rng.integers(0, len(alphabet) if i % 3 == 0 else 8, int(rng.integers(1, 700)))].tobytes() for i in range(60)]
    p = str(tmp_path / ("w%d.fa" % width))
    _write_fasta(p, seqs, width)
    e, _ = _ingest_and_compare(p, seqs)
    e.close()
    # a rank's share: records [17, 43) from its own chunk of the file (src/FastaIndex.cpp:222-241)
    e, _ = _ingest_and_compare(p, seqs, 17, 43)
    e.close()


def test_pipeline_from_fasta_equals_pipeline_from_packed_reads(tmp_path):
    import shutil
    seqs = util.read_fasta(os.path.join(G, "small_err.fa"))
    p = str(tmp_path / "small_err.fa")
    shutil.copy(os.path.join(G, "small_err.fa"), p)
    e, (packed, off, lens) = _ingest_and_compare(p, seqs)
    ks = e.count_kmers(); e.create_kmer_matrix(); st = e.create_seed_matrix()
    o = gu.oracle_run(packed, off, lens, 17, 2, 8)
    gu.assert_B_equal(e.export_csr(), o.B())
    gu.assert_stats_equal(st, o)
    e.close()
