#!/usr/bin/env python3
"""Regenerates tests/golden/*_k17_L2_U8.order IN THE BUILD CONTAINER (needs /root/reference, `make -C oracle ref_order`, and
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

L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libelbaref_order_k17.so"))
L.ref_replay_order.restype = C.c_int64
for name, fasta in (("reads_ref", "reads_ref.fa.gz"), ("small_err", "small_err.fa")):
    seqs = util.read_fasta(os.path.join(HERE, fasta))
    packed, off, lens = po.pack_reads(seqs)
    cap = 1 << 22
    out = np.zeros(cap, np.uint64); est = C.c_double(); bc = C.c_int64(); k1 = C.c_int64()
    n = L.ref_replay_order(C.c_void_p(packed.ctypes.data), C.c_void_p(off.ctypes.data), C.c_void_p(lens.ctypes.data), C.c_int64(len(lens)),
                           C.c_void_p(out.ctypes.data), C.c_int64(cap), C.byref(est), C.byref(bc), C.byref(k1))
    assert n >= 0
    with open(os.path.join(HERE, "%s_k17_L2_U8.order" % name), "w") as fo:
        fo.write("# kmer_hex by k-mer id   [reference unordered_map iteration order, 1 rank]   N=%d hll=%.6f buckets=%d keys_after_pass1=%d\n" % (n, est.value, bc.value, k1.value))
        for v in out[:n].tolist():
            fo.write("%016x\n" % v)
    print(name, n, est.value, bc.value, k1.value)
