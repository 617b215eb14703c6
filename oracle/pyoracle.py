"""ctypes bindings for the CPU oracle (oracle/elba_oracle.c) and the reference-primitive shim (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (elba_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(HERE, "_build", "libelba_oracle.so")


class Seed(C.Structure):
    _fields_ = [("q0", C.c_uint32), ("t0", C.c_uint32), ("q1", C.c_uint32), ("t1", C.c_uint32), ("numshared", C.c_int32)]


SEED_DTYPE = np.dtype([("q0", "<u4"), ("t0", "<u4"), ("q1", "<u4"), ("t1", "<u4"), ("numshared", "<i4")])
# orc_overlap_t == elba_overlap_t (include/elba_amd.h): the fields of the reference's Overlap that extend_overlap fills (src/Overlap.cpp:24-73)
OVERLAP_DTYPE = np.dtype([("begQ", "<i4"), ("begT", "<i4"), ("endQ", "<i4"), ("endT", "<i4"), ("score", "<i4"), ("suffix", "<i4"), ("suffixT", "<i4"),
                          ("direction", "i1"), ("directionT", "i1"), ("rc", "u1"), ("passed", "u1"), ("containedQ", "u1"), ("containedT", "u1"), ("kind", "u1"), ("pad", "u1")])
XSEED_DTYPE = np.dtype([("begQ", "<i4"), ("endQ", "<i4"), ("begT", "<i4"), ("endT", "<i4"), ("score", "<i4"), ("rc", "<i4")])


def build(force=False):
    if force or not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "elba_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_ORACLE_SO)
        L.orc_encode_read.restype = C.c_size_t
        L.orc_encode_read.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_kmer_from_ascii.restype = C.c_uint64
        L.orc_kmer_from_ascii.argtypes = [C.c_char_p, C.c_int]
        L.orc_kmer_twin.restype = C.c_uint64
        L.orc_kmer_twin.argtypes = [C.c_uint64, C.c_int]
        L.orc_kmer_rep.restype = C.c_uint64
        L.orc_kmer_rep.argtypes = [C.c_uint64, C.c_int]
        L.orc_kmer_extend.restype = C.c_uint64
        L.orc_kmer_extend.argtypes = [C.c_uint64, C.c_int, C.c_int]
        L.orc_kmer_hash.restype = C.c_uint64
        L.orc_kmer_hash.argtypes = [C.c_uint64]
        L.orc_murmur3_x64_128.restype = None
        L.orc_murmur3_x64_128.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_kmer_owner.restype = C.c_int
        L.orc_kmer_owner.argtypes = [C.c_uint64, C.c_int]
        L.orc_kmerN_at.restype = None
        L.orc_kmerN_at.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_kmer2_at.restype = None
        L.orc_kmer2_at.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_read_kmers.restype = C.c_int64
        L.orc_read_kmers.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_enumerate_classes.restype = C.c_int64
        L.orc_enumerate_classes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_sr_multiply.restype = Seed
        L.orc_sr_multiply.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_sr_add.restype = Seed
        L.orc_sr_add.argtypes = [Seed, Seed]
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_destroy.restype = None
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_count_and_build.restype = C.c_int
        L.orc_count_and_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_count_and_build_mt.restype = C.c_int
        L.orc_count_and_build_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_set_triples.restype = C.c_int
        L.orc_set_triples.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_csc.restype = C.c_int
        L.orc_set_csc.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_compare_B.restype = C.c_int64
        L.orc_compare_B.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_spgemm.restype = C.c_int
        L.orc_spgemm.argtypes = [C.c_void_p, C.c_int]
        L.orc_export_dcsc.restype = C.c_int
        L.orc_export_dcsc.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 6
        L.orc_free_ptr.restype = None
        L.orc_free_ptr.argtypes = [C.c_void_p]
        L.orc_seed_is_valid.restype = C.c_int
        L.orc_seed_is_valid.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_xdrop_aligner.restype = C.c_int
        L.orc_xdrop_aligner.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_classify_alignment.restype = C.c_int
        L.orc_classify_alignment.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_overlap_extend.restype = None
        L.orc_overlap_extend.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_string_graph.restype = C.c_int64
        L.orc_string_graph.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_align_upper.restype = C.c_int64
        L.orc_align_upper.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_get_i64.restype = C.c_int64
        L.orc_get_i64.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_ptr.restype = C.c_void_p
        L.orc_get_ptr.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


def pack_reads(seqs):
    """ASCII reads -> (buf u8, byte_off u64, lens u32) in DnaBuffer layout (src/DnaBuffer.cpp:22-29) via the oracle encoder."""
    L = lib()
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    nb = (lens.astype(np.int64) + 3) // 4
    off = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum(nb)[:-1]
    buf = np.zeros(int(nb.sum()) + 8, dtype=np.uint8)
    for i, s in enumerate(seqs):
        b = s if isinstance(s, bytes) else s.encode()
        L.orc_encode_read(b, len(b), buf.ctypes.data + int(off[i]))
    return buf, off, lens


def enumerate_classes(buf, off, lens, k, class_ids, nthreads=8):
    """Every canonical k-mer instance whose value modulo 4096 is one of `class_ids`, over ALL reads, on `nthreads` host threads (ctypes calls
    release the GIL): arrays (value right-aligned, read, pos), ordered by read then position."""
    from concurrent.futures import ThreadPoolExecutor
    L = lib()
    buf = np.ascontiguousarray(buf, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.uint64); lens = np.ascontiguousarray(lens, dtype=np.uint32)
    bitmap = np.zeros(64, dtype=np.uint64)
    for c in class_ids:
        bitmap[int(c) >> 6] |= np.uint64(1) << np.uint64(int(c) & 63)
    n = len(lens)
    cuts = np.linspace(0, n, max(1, nthreads * 4) + 1).astype(np.int64)

    def work(a, b):
        cnt = L.orc_enumerate_classes(buf.ctypes.data, off.ctypes.data, lens.ctypes.data, int(a), int(b), k, bitmap.ctypes.data, None, None, None, 0)
        v = np.empty(cnt, dtype=np.uint64); r = np.empty(cnt, dtype=np.uint32); p = np.empty(cnt, dtype=np.uint32)
        got = L.orc_enumerate_classes(buf.ctypes.data, off.ctypes.data, lens.ctypes.data, int(a), int(b), k, bitmap.ctypes.data, v.ctypes.data, r.ctypes.data, p.ctypes.data, cnt)
        assert got == cnt
        return v, r, p

    with ThreadPoolExecutor(nthreads) as ex:
        parts = list(ex.map(lambda ab: work(*ab), zip(cuts[:-1], cuts[1:])))
    return np.concatenate([x[0] for x in parts]), np.concatenate([x[1] for x in parts]), np.concatenate([x[2] for x in parts])


class Oracle:
    """Reads -> reliable k-mers -> A -> B with the CPU restatement."""

    STAT = dict(M=0, I=1, N=2, Z=3, ndistinct=4, P=5, Yraw=6, Y=7, ndiag=8, nupper=9, maxshared=10)

    def __init__(self, k, lower, upper):
        self.L = lib()
        self.h = self.L.orc_create(k, lower, upper)
        if not self.h:
            raise ValueError("invalid (k, lower, upper)")
        self.k, self.lower, self.upper = k, lower, upper

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def stat(self, name):
        return int(self.L.orc_get_i64(self.h, self.STAT[name]))

    def count_and_build(self, buf, off, lens, nthreads=1):
        """reads -> A (both orientations); nthreads > 1: the same result on that many host threads (orc_count_and_build_mt)"""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        if nthreads > 1:
            rc = self.L.orc_count_and_build_mt(self.h, buf.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens), int(nthreads))
        else:
            rc = self.L.orc_count_and_build(self.h, buf.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens))
        if rc:
            raise RuntimeError("orc_count_and_build failed: %d" % rc)

    def set_triples(self, M, N, rows, cols, vals):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.uint32)
        rc = self.L.orc_set_triples(self.h, M, N, len(rows), rows.ctypes.data, cols.ctypes.data, vals.ctypes.data)
        if rc:
            raise RuntimeError("orc_set_triples failed: %d" % rc)

    def set_csc(self, M, N, colptr_u32, csc_u64, nthreads=1):
        """A from its columns in the reference's AT order (colptr u32[N+1], entries read << 32 | pos in (read, pos) order); CSR is derived."""
        colptr_u32 = np.ascontiguousarray(colptr_u32, dtype=np.uint32); csc_u64 = np.ascontiguousarray(csc_u64, dtype=np.uint64)
        assert len(colptr_u32) == N + 1
        rc = self.L.orc_set_csc(self.h, M, N, len(csc_u64), colptr_u32.ctypes.data, csc_u64.ctypes.data, int(nthreads))
        if rc:
            raise RuntimeError("orc_set_csc failed: %d" % rc)

    def compare_B(self, rowptr_i64, col_u32, val_seed, nthreads=1):
        """rows + entries in which a CSR B (i64 row pointers, u32 columns, 20-byte values) differs from the oracle's; -1: other shape"""
        rowptr_i64 = np.ascontiguousarray(rowptr_i64, dtype=np.int64); col_u32 = np.ascontiguousarray(col_u32, dtype=np.uint32)
        assert val_seed.dtype.itemsize == 20 and val_seed.flags["C_CONTIGUOUS"]
        return int(self.L.orc_compare_B(self.h, len(rowptr_i64) - 1, len(col_u32), rowptr_i64.ctypes.data, col_u32.ctypes.data, val_seed.ctypes.data, int(nthreads)))

    def spgemm(self, nthreads=1):
        rc = self.L.orc_spgemm(self.h, nthreads)
        if rc:
            raise RuntimeError("orc_spgemm failed: %d" % rc)

    def A(self):
        M, N, Z = self.stat("M"), self.stat("N"), self.stat("Z")
        g = lambda w, n, dt: _arr(self.L.orc_get_ptr(self.h, w), n, dt)
        return dict(M=M, N=N, Z=Z,
                    kmers=g(0, N, np.uint64), kmers_lo=(g(11, N, np.uint64) if self.k > 32 else None), kmers_lo2=(g(12, N, np.uint64) if self.k > 64 else None), colptr=g(1, N + 1, np.int64), csc_read=g(2, Z, np.uint32), csc_pos=g(3, Z, np.uint32),
                    rowptr=g(4, M + 1, np.int64), csr_kid=g(5, Z, np.uint32), csr_pos=g(6, Z, np.uint32),
                    hist=g(7, self.upper + 2, np.int64))

    def B(self):
        M, Y = self.stat("M"), self.stat("Y")
        g = lambda w, n, dt: _arr(self.L.orc_get_ptr(self.h, w), n, dt)
        return dict(M=M, Y=Y, rowptr=g(8, M + 1, np.int64), col=g(9, Y, np.uint32), val=g(10, Y, SEED_DTYPE))

    def align_upper(self, buf, off, lens, mat=1, mis=-1, gap=-1, dropoff=15, nthreads=1, stride=1):
        """PairwiseAlignment on one rank (src/PairwiseAlignment.cpp:28-95): x-drop from seeds[0] of every stored B(i,j), i < j.
        Returns (rows, cols, overlaps[OVERLAP_DTYPE], DP cells computed).  stride > 1 aligns only every stride-th pair (the others stay zero)."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.uint64); lens = np.ascontiguousarray(lens, dtype=np.uint32)
        cap = self.stat("nupper") + 1
        rows = np.zeros(cap, dtype=np.int64); cols = np.zeros(cap, dtype=np.int64); out = np.zeros(cap, dtype=OVERLAP_DTYPE)
        cells = C.c_int64()
        n = self.L.orc_align_upper(self.h, buf.ctypes.data, off.ctypes.data, lens.ctypes.data, mat, mis, gap, dropoff, nthreads, stride,
                                   rows.ctypes.data, cols.ctypes.data, out.ctypes.data, cap, C.byref(cells))
        if n < 0:
            raise RuntimeError("orc_align_upper: capacity")
        return rows[:n], cols[:n], out[:n], cells.value

    def export_dcsc(self, row_lo, row_hi, col_lo, col_hi):
        nnz, nzc = C.c_int64(), C.c_int64()
        jc, cp, ir, numx = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.L.orc_export_dcsc(self.h, row_lo, row_hi, col_lo, col_hi, C.byref(nnz), C.byref(nzc), C.byref(jc), C.byref(cp), C.byref(ir), C.byref(numx))
        out = dict(nnz=nnz.value, nzc=nzc.value, jc=_arr(jc.value, nzc.value, np.int64), cp=_arr(cp.value, nzc.value + 1, np.int64),
                   ir=_arr(ir.value, nnz.value, np.int64), numx=_arr(numx.value, nnz.value, SEED_DTYPE))
        for p in (jc, cp, ir, numx):
            self.L.orc_free_ptr(p)
        return out


STRING_STATS = ("bad_reads", "edges_passed", "contained_reads", "edges_kept", "products", "nnzN", "marked", "removed", "nnz", "iterations")


def string_graph(nreads, rows, cols, vals, cutoff=0.65, fuzz=1000):
    """src/main.cpp:305-312: bad-read and contained-read removal + TransitiveReduction on the aligned pairs (rows < cols).
    Returns dict(rows, cols, vals) of S in the order parallel_write_paf walks it (columns, then rows), read flags (bit 0 bad, bit 1 contained), stats."""
    L = lib()
    rows = np.ascontiguousarray(rows, dtype=np.int64); cols = np.ascontiguousarray(cols, dtype=np.int64); vals = np.ascontiguousarray(vals, dtype=OVERLAP_DTYPE)
    n = len(rows)
    cap = 2 * n + 1
    orow = np.zeros(cap, dtype=np.int64); ocol = np.zeros(cap, dtype=np.int64); oval = np.zeros(cap, dtype=OVERLAP_DTYPE)
    flags = np.zeros(nreads + 1, dtype=np.uint8); st = np.zeros(10, dtype=np.int64)
    ns = L.orc_string_graph(nreads, n, rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, cutoff, fuzz, orow.ctypes.data, ocol.ctypes.data, oval.ctypes.data, cap,
                            flags.ctypes.data, st.ctypes.data)
    if ns < 0:
        raise ValueError("orc_string_graph: bad input (rows must be < cols, ids within [0, nreads))")
    return dict(n=int(ns), rows=orow[:ns], cols=ocol[:ns], vals=oval[:ns]), flags[:nreads], dict(zip(STRING_STATS, (int(v) for v in st)))


def ref_lib(k):
    """The reference's own primitives compiled from /root/reference (oracle/_ref); None when not built."""
    p = os.path.join(HERE, "_ref", "libelbaref_k%d.so" % k)
    if not os.path.exists(p):
        return None
    R = C.CDLL(p)
    R.ref_encode.restype = C.c_size_t
    R.ref_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    R.ref_kmers.restype = C.c_int64
    R.ref_kmers.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    R.ref_kmer_from_ascii.restype = None
    R.ref_kmer_from_ascii.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_kmer_hash.restype = C.c_uint64
    R.ref_kmer_hash.argtypes = [C.c_void_p]
    R.ref_murmur3_128.restype = None
    R.ref_murmur3_128.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    R.ref_bloom_second_sightings.restype = C.c_int64
    R.ref_bloom_second_sightings.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    R.ref_xdrop.restype = None
    R.ref_xdrop.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    R.ref_replay_count.restype = C.c_int64
    R.ref_replay_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return R


def xdrop(qmem, qlen, tmem, tlen, begQ, begT, k, mat=1, mis=-1, gap=-1, dropoff=15):
    """orc_xdrop_aligner + orc_classify_alignment on one pair -> (ret, begQ, endQ, begT, endT, score, rc, kind)."""
    L = lib()
    r = np.zeros(1, dtype=XSEED_DTYPE)
    qmem = np.ascontiguousarray(qmem, dtype=np.uint8); tmem = np.ascontiguousarray(tmem, dtype=np.uint8)
    ret = L.orc_xdrop_aligner(qmem.ctypes.data, qlen, tmem.ctypes.data, tlen, begQ, begT, k, mat, mis, gap, dropoff, r.ctypes.data, None)
    kind = L.orc_classify_alignment(r.ctypes.data, qlen, tlen)
    x = r[0]
    return (int(ret), int(x["begQ"]), int(x["endQ"]), int(x["begT"]), int(x["endT"]), int(x["score"]), int(x["rc"]), int(kind))


def ref_xdrop(R, qmem, qlen, tmem, tlen, begQ, begT, mat=1, mis=-1, gap=-1, dropoff=15):
    """The reference's xdrop_aligner + classify_alignment (oracle/_ref) -> same tuple as xdrop()."""
    out = np.zeros(8, dtype=np.int32)
    qmem = np.ascontiguousarray(qmem, dtype=np.uint8); tmem = np.ascontiguousarray(tmem, dtype=np.uint8)
    R.ref_xdrop(qmem.ctypes.data, qlen, tmem.ctypes.data, tlen, begQ, begT, mat, mis, gap, dropoff, out.ctypes.data)
    return tuple(int(v) for v in out)
