#!/usr/bin/env python3
"""Regenerates tests/golden/*.order (k = 17 / L 2 / U 8, and the reference's default build 31 / 15 / 35) IN THE BUILD CONTAINER (needs /root/reference, `make -C oracle ref_order`, and
LD_LIBRARY_PATH=/usr/lib/x86_64-linux-gnu:/opt/conda/lib for the image's libmpi).

SURVEY.md §8c-3: line `id` of an .order file is the packed canonical k-mer that gets k-mer id `id` in a ONE-RANK run of the reference —
the iteration order of its std::unordered_map after reserve(ceil(HLL estimate)), replayed on the reference's own compiled Kmer /
HashFuncs (murmur3) / Bloom / HyperLogLog code (oracle/ref_shim_order.cpp).  The header line records the HyperLogLog estimate, the
map's bucket count and the keys after pass 1; for the reference's bundled reads.fa they equal what the survey measured from the
reference's own KmerOps.cpp (283 870, 299 951, 136 991, N = 14 751)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (pack_reads only)

import gzip

for k, lo, up, sets in ((17, 2, 8, (("reads_ref", "reads_ref.fa.gz"), ("small_err", "small_err.fa"))),
                        (31, 15, 35, (("reads_ref", "reads_ref.fa.gz"),))):      # (31, 15, 35): the reference's default build (Makefile:1-3) on its bundled reads
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libelbaref_order_k%d.so" % k))
    L.ref_replay_order.restype = C.c_int64
    for name, fasta in sets:
        seqs = util.read_fasta(os.path.join(HERE, fasta))
        packed, off, lens = po.pack_reads(seqs)
        cap = 1 << 22
        out = np.zeros(cap, np.uint64); est = C.c_double(); bc = C.c_int64(); k1 = C.c_int64()
        n = L.ref_replay_order(C.c_void_p(packed.ctypes.data), C.c_void_p(off.ctypes.data), C.c_void_p(lens.ctypes.data), C.c_int64(len(lens)),
                               C.c_void_p(out.ctypes.data), C.c_int64(cap), C.byref(est), C.byref(bc), C.byref(k1))
        assert n >= 0
        path = os.path.join(HERE, "%s_k%d_L%d_U%d.order" % (name, k, lo, up))
        big = n > 50000                                  # the k = 31 numbering of reads.fa has 105 754 lines: kept gzipped
        with (gzip.open(path + ".gz", "wt", compresslevel=9) if big else open(path, "w")) as fo:
            fo.write("# kmer_hex by k-mer id   [reference unordered_map iteration order, 1 rank]   N=%d hll=%.6f buckets=%d keys_after_pass1=%d\n" % (n, est.value, bc.value, k1.value))
            for v in out[:n].tolist():
                fo.write("%016x\n" % v)
        print(name, k, n, est.value, bc.value, k1.value)
