"""ctypes binding of include/elba_amd.h and include/elba_synth.h (no torch types cross this boundary)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # ELBA_AMD_LIB: another build of the same library (A/B runs of kernel variants); never a different implementation
    return os.environ.get("ELBA_AMD_LIB") or os.path.join(_HERE, "lib", "libelba_amd.so")


def numeric_source_fingerprint():
    """sha256 (16 hex digits) of the sources that decide what the SpGEMM's numeric kernels do and what they read: the kernels themselves and
    the files that write the format of A (inline partners, hints, padded-column strides).  A rocprof counter summary under profiles/ is valid
    for exactly these sources: bench.py quotes `roofline.traffic` from profiles/traffic.json only while this matches the value stored there.
    None when the sources are not beside the library (an install without csrc/): nothing is quoted then."""
    import hashlib
    h = hashlib.sha256()
    try:
        for name in ("spgemm.hip", "spgemm_direct.hpp", "spgemm_table.hpp", "matrix.hip", "kmer_msd.hip", "kmer.hip", "common.hpp"):
            with open(os.path.join(_HERE, "csrc", name), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


class ElbaError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("elba status %d: %s" % (status, text))
        self.status = status


class Seed(C.Structure):
    _fields_ = [("q0", C.c_uint32), ("t0", C.c_uint32), ("q1", C.c_uint32), ("t1", C.c_uint32), ("numshared", C.c_int32)]


SEED_DTYPE = np.dtype([("q0", "<u4"), ("t0", "<u4"), ("q1", "<u4"), ("t1", "<u4"), ("numshared", "<i4")])


class Cfg(C.Structure):
    _fields_ = [("k", C.c_int32), ("lower", C.c_int32), ("upper", C.c_int32), ("device", C.c_int32),
                ("workspace_hint_bytes", C.c_int64), ("flags", C.c_int32), ("timing_stride", C.c_int32)]


class KmerStats(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("instances", C.c_int64), ("distinct", C.c_int64), ("reliable", C.c_int64), ("entries", C.c_int64),
                ("ms_total", C.c_float), ("ms_count", C.c_float), ("ms_lookup", C.c_float), ("ms_sort", C.c_float)]


class MatrixStats(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("max_row_nnz", C.c_int64), ("ms_total", C.c_float)]


class OverlapStats(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("products", C.c_int64), ("nnz_before_prune", C.c_int64), ("nnz", C.c_int64), ("nnz_diag", C.c_int64),
                ("nnz_upper", C.c_int64), ("max_numshared", C.c_int64), ("rows_lds", C.c_int64), ("rows_global", C.c_int64), ("rows_escalated", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("passes", C.c_int32), ("timed", C.c_int32),
                ("ms_total", C.c_float), ("ms_symbolic", C.c_float), ("ms_numeric", C.c_float), ("ms_finalize", C.c_float)]


class IngestStats(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("bases", C.c_int64), ("packed_bytes", C.c_int64), ("chunk_bytes", C.c_int64), ("ms_total", C.c_float), ("ms_encode", C.c_float)]


class AlignStats(C.Structure):
    _fields_ = [("nalignments", C.c_int64), ("seeds_rejected", C.c_int64), ("passed", C.c_int64), ("contained", C.c_int64),
                ("extensions_strided", C.c_int64), ("cells", C.c_int64), ("ms_total", C.c_float), ("ms_extend", C.c_float)]


class StringStats(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("nedges", C.c_int64), ("bad_reads", C.c_int64), ("edges_passed", C.c_int64), ("contained_reads", C.c_int64),
                ("edges_kept", C.c_int64), ("products", C.c_int64), ("marked", C.c_int64), ("removed", C.c_int64), ("nnz", C.c_int64),
                ("iterations", C.c_int32), ("reserved", C.c_int32), ("ms_total", C.c_float), ("ms_minplus", C.c_float)]


class Overlaps(C.Structure):
    _fields_ = [("n", C.c_int64), ("rows", C.c_void_p), ("cols", C.c_void_p), ("vals", C.c_void_p)]


# elba_overlap_t (include/elba_amd.h): the fields Overlap::extend_overlap fills (src/Overlap.cpp:24-73)
OVERLAP_DTYPE = np.dtype([("begQ", "<i4"), ("begT", "<i4"), ("endQ", "<i4"), ("endT", "<i4"), ("score", "<i4"), ("suffix", "<i4"), ("suffixT", "<i4"),
                          ("direction", "i1"), ("directionT", "i1"), ("rc", "u1"), ("passed", "u1"), ("containedQ", "u1"), ("containedT", "u1"), ("kind", "u1"), ("reserved", "u1")])


class Dcsc(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("nzc", C.c_int64),
                ("jc", C.c_void_p), ("cp", C.c_void_p), ("ir", C.c_void_p), ("numx", C.c_void_p)]


class Csr(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


class KmerMatrix(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("kmers", C.c_void_p), ("kmers_lo", C.c_void_p), ("kmers_lo2", C.c_void_p), ("colptr", C.c_void_p),
                ("csc_row", C.c_void_p), ("csc_val", C.c_void_p), ("rowptr", C.c_void_p), ("csr_col", C.c_void_p), ("csr_val", C.c_void_p)]


class DeviceView(C.Structure):
    _fields_ = [("M", C.c_int64), ("N", C.c_int64), ("Z", C.c_int64), ("Y", C.c_int64),
                ("a_rowptr", C.c_void_p), ("a_csr", C.c_void_p), ("a_colptr", C.c_void_p), ("a_csc", C.c_void_p),
                ("b_rowptr", C.c_void_p), ("b_col", C.c_void_p), ("b_val", C.c_void_p), ("stream", C.c_void_p),
                ("a_csr_format", C.c_uint32), ("a_csr_pos_mask", C.c_uint32), ("a_kmers", C.c_void_p),
                ("a_gather_slots", C.c_int64), ("a_slot_kid", C.c_void_p)]


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_length", C.c_int64), ("depth", C.c_double), ("avg_len", C.c_double), ("sd_len", C.c_double),
                ("min_len", C.c_int64), ("error_rate", C.c_double), ("repeat_families", C.c_int32), ("repeat_fraction", C.c_double),
                ("repeat_len", C.c_int64), ("first_read", C.c_int64), ("num_reads", C.c_int64)]


class SynthReads(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("total_reads", C.c_int64), ("packed_bytes", C.c_int64), ("total_bases", C.c_int64),
                ("packed", C.c_void_p), ("byte_off", C.c_void_p), ("len", C.c_void_p), ("genome_pos", C.c_void_p), ("strand", C.c_void_p)]


EXPORTED_SYMBOLS = [
    "elba_abi_version", "elba_strerror", "elba_last_error", "elba_ctx_create", "elba_ctx_destroy", "elba_set_reads", "elba_set_reads_device",
    "elba_count_kmers", "elba_create_kmer_matrix", "elba_set_kmer_matrix", "elba_create_seed_matrix", "elba_export_dcsc", "elba_free_dcsc",
    "elba_export_csr", "elba_free_csr", "elba_export_kmer_matrix", "elba_free_kmer_matrix", "elba_kmer_histogram", "elba_get_device_view", "elba_set_option",
    "elba_align_seeds", "elba_export_overlaps", "elba_free_overlaps", "elba_set_overlaps", "elba_transitive_reduction", "elba_export_string_graph", "elba_export_read_flags", "elba_set_reads_fasta", "elba_export_reads", "elba_dist_set_all_reads",
    "elba_synth_num_reads", "elba_synth_generate", "elba_synth_free",
    "elba_kmer_hash_owner", "elba_dist_value_histogram", "elba_dist_set_owner_ranges", "elba_dist_set_kmer_id_base", "elba_dist_count_owners", "elba_dist_fill_send", "elba_dist_packed_format", "elba_dist_fill_send_packed", "elba_dist_unpack_records", "elba_dist_count_records", "elba_dist_get_reliable_kmers", "elba_dist_copy_reliable_kmers",
    "elba_dist_set_global_kmers", "elba_dist_panel_counts", "elba_dist_panel_fill", "elba_dist_panel_counts_win", "elba_dist_panel_fill_win", "elba_dist_set_panel",
    "elba_seed_matrix_begin", "elba_seed_matrix_fill", "elba_seed_matrix_end", "elba_set_stream", "elba_seed_matrix_send", "elba_seed_matrix_recv", "elba_set_kmer_matrix_device", "elba_export_triples_device", "elba_get_stat", "elba_release_workspace",
]

_lib = None


def load_library():
    """Loads libelba_amd.so or raises: the product path never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as the system one).  Whichever is loaded first
    # serves both; when the system one came first, torch found no GPU afterwards (measured on the GPU box: "No HIP GPUs are available").
    # The binding hands device memory to torch tensors (tests, bench.py, the multi-GPU driver), so torch's copy is loaded first when there is one.
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(hip):
                C.CDLL(hip, mode=C.RTLD_GLOBAL)
    except Exception:      # noqa: BLE001 — without torch the system runtime is the only one
        pass
    if not os.path.exists(p):
        raise ElbaError(-1, "libelba_amd.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C elba_amd/csrc`" % p)
    L = C.CDLL(p)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.elba_abi_version.restype = i32
    L.elba_strerror.restype = C.c_char_p; L.elba_strerror.argtypes = [i32]
    L.elba_last_error.restype = C.c_char_p; L.elba_last_error.argtypes = [vp]
    L.elba_ctx_create.restype = i32; L.elba_ctx_create.argtypes = [C.POINTER(vp), C.POINTER(Cfg)]
    L.elba_ctx_destroy.restype = None; L.elba_ctx_destroy.argtypes = [vp]
    L.elba_set_reads.restype = i32; L.elba_set_reads.argtypes = [vp, vp, vp, vp, i64, i64]
    L.elba_set_reads_device.restype = i32; L.elba_set_reads_device.argtypes = [vp, vp, i64, vp, vp, i64, i64]
    L.elba_set_reads_fasta.restype = i32; L.elba_set_reads_fasta.argtypes = [vp, vp, i64, C.c_uint64, vp, i64, i64, C.POINTER(IngestStats)]
    L.elba_export_reads.restype = i32; L.elba_export_reads.argtypes = [vp, vp, i64, vp, vp, i64]
    L.elba_count_kmers.restype = i32; L.elba_count_kmers.argtypes = [vp, C.POINTER(KmerStats)]
    L.elba_create_kmer_matrix.restype = i32; L.elba_create_kmer_matrix.argtypes = [vp, C.POINTER(MatrixStats)]
    L.elba_set_kmer_matrix.restype = i32; L.elba_set_kmer_matrix.argtypes = [vp, i64, i64, i64, vp, vp, vp, C.POINTER(MatrixStats)]
    L.elba_set_kmer_matrix_device.restype = i32; L.elba_set_kmer_matrix_device.argtypes = [vp, i64, i64, i64, vp, vp, vp, C.POINTER(MatrixStats)]
    L.elba_export_triples_device.restype = i32; L.elba_export_triples_device.argtypes = [vp, vp, vp, vp]
    L.elba_create_seed_matrix.restype = i32; L.elba_create_seed_matrix.argtypes = [vp, C.POINTER(OverlapStats)]
    L.elba_align_seeds.restype = i32; L.elba_align_seeds.argtypes = [vp, i32, i32, i32, i32, C.POINTER(AlignStats)]
    L.elba_export_overlaps.restype = i32; L.elba_export_overlaps.argtypes = [vp, C.POINTER(Overlaps)]
    L.elba_free_overlaps.restype = None; L.elba_free_overlaps.argtypes = [C.POINTER(Overlaps)]
    L.elba_set_overlaps.restype = i32; L.elba_set_overlaps.argtypes = [vp, i64, vp, vp, vp, i64]
    L.elba_transitive_reduction.restype = i32; L.elba_transitive_reduction.argtypes = [vp, C.c_double, i32, C.POINTER(StringStats)]
    L.elba_export_string_graph.restype = i32; L.elba_export_string_graph.argtypes = [vp, C.POINTER(Overlaps)]
    L.elba_export_read_flags.restype = i32; L.elba_export_read_flags.argtypes = [vp, vp, i64]
    L.elba_export_dcsc.restype = i32; L.elba_export_dcsc.argtypes = [vp, i64, i64, i64, i64, C.POINTER(Dcsc)]
    L.elba_free_dcsc.restype = None; L.elba_free_dcsc.argtypes = [C.POINTER(Dcsc)]
    L.elba_export_csr.restype = i32; L.elba_export_csr.argtypes = [vp, i64, i64, C.POINTER(Csr)]
    L.elba_free_csr.restype = None; L.elba_free_csr.argtypes = [C.POINTER(Csr)]
    L.elba_export_kmer_matrix.restype = i32; L.elba_export_kmer_matrix.argtypes = [vp, C.POINTER(KmerMatrix)]
    L.elba_free_kmer_matrix.restype = None; L.elba_free_kmer_matrix.argtypes = [C.POINTER(KmerMatrix)]
    L.elba_kmer_histogram.restype = i32; L.elba_kmer_histogram.argtypes = [vp, vp, i64]
    L.elba_get_device_view.restype = i32; L.elba_get_device_view.argtypes = [vp, C.POINTER(DeviceView)]
    L.elba_set_option.restype = i32; L.elba_set_option.argtypes = [vp, C.c_char_p, i64]
    L.elba_get_stat.restype = i32; L.elba_get_stat.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.elba_release_workspace.restype = i32; L.elba_release_workspace.argtypes = [vp]
    L.elba_kmer_hash_owner.restype = i32; L.elba_kmer_hash_owner.argtypes = [vp, vp, i64, i32, vp, vp]
    L.elba_synth_num_reads.restype = i64; L.elba_synth_num_reads.argtypes = [C.POINTER(SynthCfg)]
    L.elba_synth_generate.restype = i32; L.elba_synth_generate.argtypes = [C.POINTER(SynthCfg), C.POINTER(SynthReads)]
    L.elba_synth_free.restype = None; L.elba_synth_free.argtypes = [C.POINTER(SynthReads)]
    _lib = L
    return L


def _copy(ptr, n, dtype):
    dt = np.dtype(dtype)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


def _stats(s):
    return {f[0]: getattr(s, f[0]) for f in s._fields_}


def synth_reads(seed, genome_length, depth, avg_len, sd_len, error_rate=0.0, min_len=100, repeat_families=0, repeat_fraction=0.0,
                repeat_len=0, first_read=0, num_reads=-1):
    """Generates (packed u8, byte_off u64, len u32, info) with the native generator (host code; no GPU needed)."""
    L = load_library()
    cfg = SynthCfg(seed, genome_length, depth, avg_len, sd_len, min_len, error_rate, repeat_families, repeat_fraction, repeat_len, first_read, num_reads)
    out = SynthReads()
    rc = L.elba_synth_generate(C.byref(cfg), C.byref(out))
    if rc:
        raise ElbaError(rc, "elba_synth_generate failed")
    try:
        packed = _copy(out.packed, out.packed_bytes + 16, np.uint8)
        off = _copy(out.byte_off, out.nreads, np.uint64)
        ln = _copy(out.len, out.nreads, np.uint32)
        info = dict(nreads=out.nreads, total_reads=out.total_reads, total_bases=out.total_bases,
                    genome_pos=_copy(out.genome_pos, out.nreads, np.int64), strand=_copy(out.strand, out.nreads, np.uint8))
    finally:
        L.elba_synth_free(C.byref(out))
    return packed, off, ln, info


class Engine:
    """One context on one GPU.  Method names follow the reference's free functions (include/KmerOps.hpp:24-31,
    include/SharedSeeds.hpp:98-99); each is a single C-ABI call."""

    def __init__(self, k, lower, upper, device=0, workspace_hint_bytes=0, flags=0, timing_stride=0, options=None):
        self.L = load_library()
        self.h = C.c_void_p()
        cfg = Cfg(k, lower, upper, device, workspace_hint_bytes, flags, timing_stride)
        rc = self.L.elba_ctx_create(C.byref(self.h), C.byref(cfg))
        if rc:
            self.h = C.c_void_p()
            raise ElbaError(rc, self.L.elba_strerror(rc).decode())
        self.k, self.lower, self.upper, self.device = k, lower, upper, device
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.elba_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise ElbaError(rc, "%s: %s" % (self.L.elba_strerror(rc).decode(), self.L.elba_last_error(self.h).decode()))

    # --- inputs ---
    def set_reads(self, packed, byte_off, lens, first_global_id=0):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        byte_off = np.ascontiguousarray(byte_off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self._check(self.L.elba_set_reads(self.h, packed.ctypes.data, byte_off.ctypes.data, lens.ctypes.data, len(lens), first_global_id))

    def set_reads_device(self, d_packed, packed_bytes, d_byte_off, d_len, nreads, first_global_id=0):
        self._check(self.L.elba_set_reads_device(self.h, d_packed, packed_bytes, d_byte_off, d_len, nreads, first_global_id))

    def set_reads_fasta(self, chunk, chunk_file_offset, recs, first_global_id=0):
        """FastaIndex::getmydna (src/FastaIndex.cpp:191-290) with the 2-bit encoding on the GPU: `chunk` = raw FASTA bytes from
        `chunk_file_offset` on, `recs` = this rank's .fai records (elba_amd.fasta.FAI_DTYPE)."""
        buf = np.frombuffer(chunk, dtype=np.uint8) if not isinstance(chunk, np.ndarray) else np.ascontiguousarray(chunk, dtype=np.uint8)
        recs = np.ascontiguousarray(recs)
        assert recs.dtype.itemsize == 24
        st = IngestStats()
        self._check(self.L.elba_set_reads_fasta(self.h, buf.ctypes.data if buf.size else None, buf.size, chunk_file_offset, recs.ctypes.data if recs.size else None,
                                                len(recs), first_global_id, C.byref(st)))
        return _stats(st)

    def export_reads(self, nreads, packed_bytes):
        packed = np.zeros(packed_bytes + 16, dtype=np.uint8); off = np.zeros(nreads, dtype=np.uint64); ln = np.zeros(nreads, dtype=np.uint32)
        self._check(self.L.elba_export_reads(self.h, packed.ctypes.data, packed_bytes, off.ctypes.data, ln.ctypes.data, nreads))
        return packed, off, ln

    def set_kmer_matrix(self, nrows, ncols, rows, cols, vals):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.uint32)
        st = MatrixStats()
        self._check(self.L.elba_set_kmer_matrix(self.h, nrows, ncols, len(rows), rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, C.byref(st)))
        return _stats(st)

    def set_kmer_matrix_device(self, nrows, ncols, nnz, d_rows, d_cols, d_vals):
        """The triples resident in HBM (device pointers: int64 rows, int64 cols, uint32 vals)."""
        st = MatrixStats()
        self._check(self.L.elba_set_kmer_matrix_device(self.h, nrows, ncols, nnz, d_rows, d_cols, d_vals, C.byref(st)))
        return _stats(st)

    def export_triples_device(self, d_rows, d_cols, d_vals):
        self._check(self.L.elba_export_triples_device(self.h, d_rows, d_cols, d_vals))

    def kmer_hash_owner(self, kmers, nprocs):
        """Kmer::GetHash and GetKmerOwner of the reference, computed on the device: kmers = n x W packed words."""
        km = np.ascontiguousarray(kmers, dtype=np.uint64)
        W = 3 if self.k > 64 else (2 if self.k > 32 else 1)
        n = km.size // W
        h = np.zeros(n, dtype=np.uint64); ow = np.zeros(n, dtype=np.int32)
        self._check(self.L.elba_kmer_hash_owner(self.h, km.ctypes.data if n else None, n, int(nprocs), h.ctypes.data, ow.ctypes.data))
        return h, ow

    def set_option(self, name, value):
        self._check(self.L.elba_set_option(self.h, name.encode(), int(value)))

    def get_stat(self, name):
        """a diagnostic counter of the last stage call, by name (elba_get_stat)"""
        v = C.c_int64(0)
        self._check(self.L.elba_get_stat(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def release_workspace(self):
        """the stage calls' scratch memory back to the device; matrices, reads and the multiplication's buffers stay (elba_release_workspace)"""
        self._check(self.L.elba_release_workspace(self.h))

    # --- stages ---
    def count_kmers(self):
        """get_kmer_count_map_keys + get_kmer_count_map_values."""
        st = KmerStats()
        self._check(self.L.elba_count_kmers(self.h, C.byref(st)))
        return _stats(st)

    def create_kmer_matrix(self):
        st = MatrixStats()
        self._check(self.L.elba_create_kmer_matrix(self.h, C.byref(st)))
        return _stats(st)

    def create_seed_matrix(self):
        st = OverlapStats()
        self._check(self.L.elba_create_seed_matrix(self.h, C.byref(st)))
        return _stats(st)

    def align_seeds(self, mat=1, mis=-1, gap=-1, dropoff=15):
        """PairwiseAlignment (src/PairwiseAlignment.cpp:5-106) on one rank: x-drop from seeds[0] of every stored B(i,j), i < j."""
        st = AlignStats()
        self._check(self.L.elba_align_seeds(self.h, mat, mis, gap, dropoff, C.byref(st)))
        return _stats(st)

    def export_overlaps(self):
        o = Overlaps()
        self._check(self.L.elba_export_overlaps(self.h, C.byref(o)))
        try:
            return dict(n=o.n, rows=_copy(o.rows, o.n, np.int64), cols=_copy(o.cols, o.n, np.int64), vals=_copy(o.vals, o.n, OVERLAP_DTYPE))
        finally:
            self.L.elba_free_overlaps(C.byref(o))

    # --- string graph (src/main.cpp:305-312) ---
    def set_overlaps(self, nreads, rows, cols, vals):
        """Load aligned pairs (rows < cols, ascending in (row, col)) instead of using this engine's own alignments."""
        rows = np.ascontiguousarray(rows, dtype=np.int64); cols = np.ascontiguousarray(cols, dtype=np.int64); vals = np.ascontiguousarray(vals, dtype=OVERLAP_DTYPE)
        if not (len(rows) == len(cols) == len(vals)):
            raise ValueError("set_overlaps: rows, cols, vals differ in length")
        self._check(self.L.elba_set_overlaps(self.h, int(nreads), rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, len(rows)))

    def transitive_reduction(self, bad_read_cutoff=0.65, fuzz=1000):
        """find_bad_reads / find_contained_reads + prunes (src/main.cpp:305-311) and TransitiveReduction (src/TransitiveReduction.cpp:3-90)."""
        st = StringStats()
        self._check(self.L.elba_transitive_reduction(self.h, float(bad_read_cutoff), int(fuzz), C.byref(st)))
        d = _stats(st)
        d.pop("reserved", None)
        return d

    def export_string_graph(self):
        """Entries of S in the order parallel_write_paf walks them (columns ascending, rows ascending within a column)."""
        o = Overlaps()
        self._check(self.L.elba_export_string_graph(self.h, C.byref(o)))
        try:
            return dict(n=o.n, rows=_copy(o.rows, o.n, np.int64), cols=_copy(o.cols, o.n, np.int64), vals=_copy(o.vals, o.n, OVERLAP_DTYPE))
        finally:
            self.L.elba_free_overlaps(C.byref(o))

    def export_read_flags(self, nreads):
        """One byte per read: bit 0 = bad read, bit 1 = contained read."""
        f = np.zeros(int(nreads), dtype=np.uint8)
        self._check(self.L.elba_export_read_flags(self.h, f.ctypes.data, int(nreads)))
        return f

    # --- outputs ---
    def export_csr(self, row_lo=0, row_hi=None):
        if row_hi is None:
            row_hi = self.device_view()["M"]
        o = Csr()
        self._check(self.L.elba_export_csr(self.h, row_lo, row_hi, C.byref(o)))
        try:
            return dict(M=o.nrows, Y=o.nnz, rowptr=_copy(o.rowptr, o.nrows + 1, np.int64), col=_copy(o.col, o.nnz, np.int64), val=_copy(o.val, o.nnz, SEED_DTYPE))
        finally:
            self.L.elba_free_csr(C.byref(o))

    def export_dcsc(self, row_lo, row_hi, col_lo, col_hi):
        o = Dcsc()
        self._check(self.L.elba_export_dcsc(self.h, row_lo, row_hi, col_lo, col_hi, C.byref(o)))
        try:
            return dict(nnz=o.nnz, nzc=o.nzc, jc=_copy(o.jc, o.nzc, np.int64), cp=_copy(o.cp, o.nzc + 1, np.int64),
                        ir=_copy(o.ir, o.nnz, np.int64), numx=_copy(o.numx, o.nnz, SEED_DTYPE))
        finally:
            self.L.elba_free_dcsc(C.byref(o))

    def export_kmer_matrix(self):
        o = KmerMatrix()
        self._check(self.L.elba_export_kmer_matrix(self.h, C.byref(o)))
        try:
            return dict(M=o.nrows, N=o.ncols, Z=o.nnz, kmers=_copy(o.kmers, o.ncols, np.uint64) if o.kmers else None,
                        kmers_lo=_copy(o.kmers_lo, o.ncols, np.uint64) if o.kmers_lo else None,
                        kmers_lo2=_copy(o.kmers_lo2, o.ncols, np.uint64) if o.kmers_lo2 else None,
                        colptr=_copy(o.colptr, o.ncols + 1, np.int64), csc_read=_copy(o.csc_row, o.nnz, np.int64), csc_pos=_copy(o.csc_val, o.nnz, np.uint32),
                        rowptr=_copy(o.rowptr, o.nrows + 1, np.int64), csr_kid=_copy(o.csr_col, o.nnz, np.int64), csr_pos=_copy(o.csr_val, o.nnz, np.uint32))
        finally:
            self.L.elba_free_kmer_matrix(C.byref(o))

    def kmer_histogram(self, n=None):
        n = n or self.upper + 2
        h = np.zeros(n, dtype=np.int64)
        self._check(self.L.elba_kmer_histogram(self.h, h.ctypes.data, n))
        return h

    def device_view(self):
        v = DeviceView()
        self._check(self.L.elba_get_device_view(self.h, C.byref(v)))
        return _stats(v)
