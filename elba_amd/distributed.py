"""Multi-GPU driver of the overlap engine: 1D read-row shards x hash-owned k-mer columns.

One process per GPU.  The heavy lifting is in libelba_amd.so (elba_dist_* entry points); this module only sequences the stages and
issues the collectives through torch.distributed — backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests, where a
test-defined backend stands in for the HIP library.  Collectives used, all over the whole world:

    all_to_all_single   #1  k-mer instances (16 B records) to the k-mer's owner rank          (src/KmerOps.cpp:117-151, :244-274)
    all_reduce + all_gather value histogram (4096 bins) -> owners by value range; owners' counts -> global k-mer ids (src/KmerOps.cpp:371-375 Exscan)
    all_to_all_single   #2  every column, whole, to each rank that owns one of its reads        (SpParMat ctor / Transpose redistribution)
    all_reduce              scalars only (counts, timings)

There is no bulk all-reduce anywhere; all-to-all on the 8-GPU xGMI mesh drives all 7 links of a GPU at once.
After exchange #2 every rank holds complete columns for all k-mers of its reads, so the SpGEMM itself (create_seed_matrix) needs no
communication: the reference's SUMMA broadcasts inside "creating seed matrix (spgemm)" have no counterpart here by construction.
"""
import ctypes as C

import numpy as np

from . import capi


def partition_by_bases(lens, nranks):
    """Contiguous read partition balanced by bases: the greedy rule of src/FastaIndex.cpp:47-94 (rank i takes reads until the next one
    would push it over total/nranks; the last rank takes the remainder).  Returns bounds[nranks+1]."""
    lens = np.asarray(lens, dtype=np.int64)
    total, n = int(lens.sum()), len(lens)
    avg = total / nranks
    bounds, r = [0], 0
    for _ in range(nranks - 1):
        sofar = 0
        while r < n and sofar + int(lens[r]) < avg:
            sofar += int(lens[r]); r += 1
        bounds.append(r)
    bounds.append(n)
    return np.array(bounds, dtype=np.int64)


OWNER_BINS = 4096          # ELBA_OWNER_BINS of include/elba_amd.h


def owner_ranges_from_histogram(hist, nranks):
    """Boundaries (exclusive upper bins) of `nranks` contiguous value ranges holding about the same number of k-mer instances each:
    rank r ends at the first bin where the running count reaches (r + 1) / nranks of the total.  Deterministic in `hist`."""
    cum = np.cumsum(np.asarray(hist, dtype=np.int64))
    total = int(cum[-1]) if len(cum) else 0
    upper = np.empty(nranks, dtype=np.int64)
    for r in range(nranks):
        target = -(-total * (r + 1) // nranks)
        upper[r] = int(np.searchsorted(cum, target, side="left")) + 1 if total else (len(cum) * (r + 1)) // nranks
    upper = np.minimum(np.maximum.accumulate(upper), len(cum))
    upper[-1] = len(cum)
    return upper.astype(np.uint32)


def kmer_words(k):
    """64-bit words per k-mer (NLONGS, include/Kmer.hpp:95-97): a record of exchange #1 is that many words + one of (read << 32 | pos)."""
    return 3 if k > 64 else (2 if k > 32 else 1)


class HipBackend:
    """The product backend: every method is one C-ABI call on device buffers owned by torch tensors."""

    def __init__(self, k, lower, upper, device, timing_stride=0):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.e = capi.Engine(k, lower, upper, device=device, timing_stride=timing_stride)
        self.L, self.h = self.e.L, self.e.h
        L = self.L
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.elba_dist_value_histogram.restype = i32; L.elba_dist_value_histogram.argtypes = [vp, vp, i64]
        L.elba_dist_set_owner_ranges.restype = i32; L.elba_dist_set_owner_ranges.argtypes = [vp, i32, vp]
        L.elba_dist_set_kmer_id_base.restype = i32; L.elba_dist_set_kmer_id_base.argtypes = [vp, i64, i64]
        L.elba_dist_count_owners.restype = i32; L.elba_dist_count_owners.argtypes = [vp, i32, vp]
        L.elba_dist_fill_send.restype = i32; L.elba_dist_fill_send.argtypes = [vp, i32, vp, vp]
        L.elba_dist_count_records.restype = i32; L.elba_dist_count_records.argtypes = [vp, vp, i64, C.POINTER(capi.KmerStats)]
        L.elba_dist_packed_format.restype = i32; L.elba_dist_packed_format.argtypes = [vp, i32, vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.elba_dist_fill_send_packed.restype = i32; L.elba_dist_fill_send_packed.argtypes = [vp, i32, vp, vp]
        L.elba_dist_unpack_records.restype = i32; L.elba_dist_unpack_records.argtypes = [vp, i32, i32, vp, vp, vp]
        L.elba_dist_get_reliable_kmers.restype = i32; L.elba_dist_get_reliable_kmers.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.elba_dist_copy_reliable_kmers.restype = i32; L.elba_dist_copy_reliable_kmers.argtypes = [vp, vp, i64]
        L.elba_dist_set_global_kmers.restype = i32; L.elba_dist_set_global_kmers.argtypes = [vp, vp, i64]
        L.elba_dist_panel_counts.restype = i32; L.elba_dist_panel_counts.argtypes = [vp, i32, vp, vp]
        L.elba_dist_panel_fill.restype = i32; L.elba_dist_panel_fill.argtypes = [vp, i32, vp, vp, vp]
        L.elba_dist_panel_counts_win.restype = i32; L.elba_dist_panel_counts_win.argtypes = [vp, i32, vp, vp, vp, vp]
        L.elba_dist_panel_fill_win.restype = i32; L.elba_dist_panel_fill_win.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.elba_seed_matrix_begin.restype = i32; L.elba_seed_matrix_begin.argtypes = [vp, i32, vp, vp]
        L.elba_seed_matrix_fill.restype = i32; L.elba_seed_matrix_fill.argtypes = [vp, vp, vp]
        L.elba_seed_matrix_end.restype = i32; L.elba_seed_matrix_end.argtypes = [vp, vp, i64, C.POINTER(capi.OverlapStats)]
        L.elba_set_stream.restype = i32; L.elba_set_stream.argtypes = [vp, vp]
        L.elba_seed_matrix_send.restype = i32; L.elba_seed_matrix_send.argtypes = [vp, i32, vp, vp, i64]
        L.elba_seed_matrix_recv.restype = i32; L.elba_seed_matrix_recv.argtypes = [vp, vp, i64, C.POINTER(capi.OverlapStats), C.POINTER(i64)]
        L.elba_dist_set_panel.restype = i32; L.elba_dist_set_panel.argtypes = [vp, vp, i64, i64, i64, i64, i64, C.POINTER(capi.MatrixStats)]
        L.elba_dist_set_all_reads.restype = i32; L.elba_dist_set_all_reads.argtypes = [vp, vp, i64, vp, vp, i64]

    def empty_records(self, n, width=2):
        return self.torch.empty((max(int(n), 0), int(width)), dtype=self.torch.int64, device=self.dev)

    def empty_words(self, n):
        return self.torch.empty((max(int(n), 0),), dtype=self.torch.int64, device=self.dev)

    def set_reads(self, packed, off, lens, first_global_id):
        self.e.set_reads(packed, off, lens, first_global_id)

    def set_all_reads(self, packed_words, byte_off, lens):
        """Every read of the run (replicated): packed 2-bit bytes as an int64 word tensor on the device, global byte offsets / lengths on the host."""
        torch = self.torch
        off = torch.from_numpy(np.ascontiguousarray(byte_off, dtype=np.uint64).view(np.int64)).to(self.dev)
        ln = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.uint32).view(np.int32)).to(self.dev)
        self.e._check(self.L.elba_dist_set_all_reads(self.h, packed_words.data_ptr() if packed_words.numel() else None, packed_words.numel() * 8,
                                                     off.data_ptr() if len(lens) else None, ln.data_ptr() if len(lens) else None, len(lens)))

    def align_seeds(self, mat, mis, gap, dropoff):
        return self.e.align_seeds(mat, mis, gap, dropoff)

    def export_overlaps(self):
        return self.e.export_overlaps()

    def set_overlaps(self, nreads, rows, cols, vals):
        self.e.set_overlaps(nreads, rows, cols, vals)

    def transitive_reduction(self, bad_read_cutoff, fuzz):
        return self.e.transitive_reduction(bad_read_cutoff, fuzz)

    def export_string_graph(self):
        return self.e.export_string_graph()

    def export_read_flags(self, nreads):
        return self.e.export_read_flags(nreads)

    def value_histogram(self):
        out = np.zeros(OWNER_BINS, dtype=np.uint64)
        self.e._check(self.L.elba_dist_value_histogram(self.h, out.ctypes.data, OWNER_BINS))
        return out.astype(np.int64)

    def set_owner_ranges(self, upper_bins):
        u = np.ascontiguousarray(upper_bins, dtype=np.uint32)
        self.e._check(self.L.elba_dist_set_owner_ranges(self.h, len(u), u.ctypes.data))

    def set_kmer_id_base(self, base, nall):
        self.e._check(self.L.elba_dist_set_kmer_id_base(self.h, int(base), int(nall)))
        self._rec = None

    def count_owners(self, nranks):
        out = np.zeros(nranks, dtype=np.uint64)
        self.e._check(self.L.elba_dist_count_owners(self.h, nranks, out.ctypes.data))
        return out.astype(np.int64)

    def fill_send(self, nranks, send, offsets):
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.e._check(self.L.elba_dist_fill_send(self.h, nranks, send.data_ptr() if send.numel() else None, off.ctypes.data))

    def packed_format(self, nranks, bounds, all_lens):
        """exchange #1 with 8-byte records (elba_dist_packed_format): (value bits, index bits) if an instance fits 64 bits, else None — the same on every rank"""
        b = np.ascontiguousarray(bounds, dtype=np.int64); l = np.ascontiguousarray(all_lens, dtype=np.uint32)
        fits, vb, ib = C.c_int(0), C.c_int(0), C.c_int(0)
        self.e._check(self.L.elba_dist_packed_format(self.h, nranks, b.ctypes.data, l.ctypes.data if len(l) else None, C.byref(fits), C.byref(vb), C.byref(ib)))
        return (int(vb.value), int(ib.value)) if fits.value else None

    def fill_send_packed(self, nranks, send, offsets):
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.e._check(self.L.elba_dist_fill_send_packed(self.h, nranks, send.data_ptr() if send.numel() else None, off.ctypes.data))

    def unpack_records(self, nranks, rank, packed, recv_counts, out=None):
        """8-byte records -> (k-mer, global read << 32 | pos); `packed` = the peers' segments one after the other, recv_counts[p] records from peer p.
        out: where the records go (a slice of a larger record tensor: the chunked exchange unpacks round by round); returns `out`."""
        rc = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        if out is None:
            out = self.empty_records(int(rc.sum()), 2)
        self.e._check(self.L.elba_dist_unpack_records(self.h, nranks, rank, packed.data_ptr() if packed.numel() else None, rc.ctypes.data, out.data_ptr() if out.numel() else None))
        return out

    def count_records(self, rec):
        self._rec = rec                                   # borrowed by the library until set_global_kmers
        st = capi.KmerStats()
        self.e._check(self.L.elba_dist_count_records(self.h, rec.data_ptr() if rec.numel() else None, rec.shape[0], C.byref(st)))
        return capi._stats(st)

    def reliable_kmers(self, n):
        """n k-mers of kmer_words(k) interleaved words each."""
        out = self.empty_words(n * kmer_words(self.e.k))
        self.e._check(self.L.elba_dist_copy_reliable_kmers(self.h, out.data_ptr() if n else None, n))
        return out

    def set_global_kmers(self, allk):
        nall = allk.numel() // kmer_words(self.e.k)
        self.e._check(self.L.elba_dist_set_global_kmers(self.h, allk.data_ptr() if allk.numel() else None, nall))
        self._rec = None

    def panel_counts(self, nranks, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        out = np.zeros(nranks, dtype=np.uint64)
        self.e._check(self.L.elba_dist_panel_counts(self.h, nranks, b.ctypes.data, out.ctypes.data))
        return out.astype(np.int64)

    def panel_fill(self, nranks, bounds, send, offsets):
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.e._check(self.L.elba_dist_panel_fill(self.h, nranks, b.ctypes.data, send.data_ptr() if send.numel() else None, off.ctypes.data))

    def panel_counts_win(self, nranks, bounds, win_lo, win_hi):
        b = np.ascontiguousarray(bounds, dtype=np.uint64); lo = np.ascontiguousarray(win_lo, dtype=np.uint64); hi = np.ascontiguousarray(win_hi, dtype=np.uint64)
        out = np.zeros(nranks, dtype=np.uint64)
        self.e._check(self.L.elba_dist_panel_counts_win(self.h, nranks, b.ctypes.data, lo.ctypes.data, hi.ctypes.data, out.ctypes.data))
        return out.astype(np.int64)

    def panel_fill_win(self, nranks, bounds, win_lo, win_hi, send, offsets):
        b = np.ascontiguousarray(bounds, dtype=np.uint64); lo = np.ascontiguousarray(win_lo, dtype=np.uint64); hi = np.ascontiguousarray(win_hi, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.e._check(self.L.elba_dist_panel_fill_win(self.h, nranks, b.ctypes.data, lo.ctypes.data, hi.ctypes.data, send.data_ptr() if send.numel() else None, off.ctypes.data))

    def set_panel(self, rec, m_total, n_total, row_lo, row_hi):
        st = capi.MatrixStats()
        self.e._check(self.L.elba_dist_set_panel(self.h, rec.data_ptr() if rec.numel() else None, rec.shape[0], m_total, n_total, row_lo, row_hi, C.byref(st)))
        return capi._stats(st)

    def create_seed_matrix(self):
        return self.e.create_seed_matrix()

    def set_option(self, name, value):
        self.e.set_option(name, value)

    # sharded call with mirror exchange (elba_seed_matrix_begin / _fill / _end)
    def seed_begin(self, nranks, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        out = np.zeros(nranks, dtype=np.uint64)
        self.e._check(self.L.elba_seed_matrix_begin(self.h, nranks, b.ctypes.data, out.ctypes.data))
        return out.astype(np.int64)

    def seed_fill(self, send, offsets):
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.e._check(self.L.elba_seed_matrix_fill(self.h, send.data_ptr() if send.numel() else None, off.ctypes.data))

    def seed_end(self, recv):
        st = capi.OverlapStats()
        self.e._check(self.L.elba_seed_matrix_end(self.h, recv.data_ptr() if recv.numel() else None, recv.shape[0], C.byref(st)))
        return capi._stats(st)

    # the same with fixed-size slots and ONE host synchronisation per step (elba_set_stream / elba_seed_matrix_send / _recv)
    def use_current_stream(self):
        """The library launches on torch's current stream from now on: its kernels and torch's collectives are ordered by the stream."""
        self.e._check(self.L.elba_set_stream(self.h, C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)))
        self.shares_stream = True

    def seed_send(self, nranks, bounds, send, slot):
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        self.e._check(self.L.elba_seed_matrix_send(self.h, nranks, b.ctypes.data, send.data_ptr(), int(slot)))

    def seed_recv(self, recv, slot):
        """-> (stats, slot) when the step is complete, (None, slot needed) when every rank has to repeat it."""
        st = capi.OverlapStats()
        need = C.c_int64(0)
        rc = self.L.elba_seed_matrix_recv(self.h, recv.data_ptr(), int(slot), C.byref(st), C.byref(need))
        if rc == 8:                                   # ELBA_ERR_RETRY
            return None, int(need.value)
        self.e._check(rc)
        return capi._stats(st), int(need.value)

    def export_csr(self, row_lo, row_hi):
        return self.e.export_csr(row_lo, row_hi)

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)


class HostStagedDist:
    """`torch.distributed` over gloo for DEVICE tensors: every collective goes through host copies.  Not the product transport (that is RCCL,
    backend "nccl") — a rehearsal transport: RCCL refuses two ranks on one device, so the only way to run the N > 1 driver (`bench.py --gpus N`,
    the sequence of collectives of src/KmerOps.cpp:117-151,244-274,371-375) as N separate PROCESSES on a one-GPU box is to carry its collectives
    over the host.  Same call signatures as the torch.distributed functions the driver uses; `.cpu()` waits for the stream, so the ordering the
    shared stream gives RCCL holds here too."""

    def __init__(self, dist):
        self.d = dist
        self.ReduceOp = dist.ReduceOp

    def barrier(self):
        self.d.barrier()

    def all_reduce(self, t, op=None):
        c = t.cpu()
        self.d.all_reduce(c, op=op if op is not None else self.d.ReduceOp.SUM)
        t.copy_(c)

    def all_gather(self, outs, t):
        co = [o.cpu() for o in outs]
        self.d.all_gather(co, t.cpu())
        for o, c in zip(outs, co):
            o.copy_(c)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        co = out.cpu()
        self.d.all_to_all_single(co, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)
        out.copy_(co)


class DistributedOverlap:
    MAX_RECORDS_PER_PEER = 1 << 25          # 512 MiB of 16-byte records per peer and all-to-all round

    def __init__(self, k, lower, upper, device=0, rank=0, world=1, dist=None, backend=None, timing_stride=0):
        self.k, self.lower, self.upper = k, lower, upper
        self.rank, self.world, self.dist = rank, world, dist
        self.be = backend if backend is not None else HipBackend(k, lower, upper, device, timing_stride)
        self.nlocal = 0
        self.bounds = None
        self.row_batches = 1
        self.force_exchange = False       # world of one: run the step through the exchange machinery all the same (self-test, bench.py under ELBA_FORCE_DIST)
        self.time_phases = False          # record (send, exchange, recv) device times of every step with the fixed-slot exchange (bench.py)
        self.phase_ms = None
        self.panel_records = 0
        self.inline_partners = None       # inline partners in the panel's rows (a third fewer column fetches): None = whenever the step will use the mirror exchange
        self._panel_inline = False
        self._cur_block = 0
        self.packed_exchange = True       # exchange #1 in 8-byte records where they fit (False: always (k-mer, read << 32 | pos) — A/B, tests)
        self.exchange_chunks = 0          # rounds of the packed exchange #1 (0: four once a peer's message reaches 4 M records, else one; 1: never chunked; n > 1: n rounds)
        self.exchange_rounds = 1          # (what the last build used)
        self.exchange_format = None

    # ---- inputs -----------------------------------------------------------------------------------------------------
    def set_reads(self, packed, off, lens, first_global_id, bounds):
        """This rank's shard of the read set (DnaBuffer layout) and the global partition bounds[world+1]."""
        self.bounds = np.asarray(bounds, dtype=np.int64)
        assert len(self.bounds) == self.world + 1 and int(self.bounds[self.rank]) == first_global_id
        self.nlocal = len(lens)
        self._reads = (np.asarray(packed, dtype=np.uint8), np.asarray(off, dtype=np.uint64), np.asarray(lens, dtype=np.uint32))   # kept for the alignment stage's all-gather
        self.be.set_reads(packed, off, lens, first_global_id)

    def generate_and_set_reads(self, w, weak=True):
        """Synthetic shard: rank r generates reads [bounds[r], bounds[r+1]) of one read set drawn from a genome that grows with the
        world size (weak scaling: per-GPU reads fixed)."""
        genome = w["genome"] * (self.world if weak else 1)
        rep = w.get("repeats", (0, 0.0, 0))      # (families, fraction of the genome, length: the dense workloads — the same read set as the one-GPU run's)
        cfg = capi.SynthCfg(w["seed"], genome, w["depth"], w["avg_len"], w["sd_len"], w["min_len"], w["error"], rep[0], rep[1], rep[2], 0, -1)
        total = int(capi.load_library().elba_synth_num_reads(C.byref(cfg)))
        per = (total + self.world - 1) // self.world
        bounds = np.minimum(np.arange(self.world + 1, dtype=np.int64) * per, total)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        packed, off, lens, info = capi.synth_reads(w["seed"], genome, w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                                   repeat_families=rep[0], repeat_fraction=rep[1], repeat_len=rep[2], first_read=lo, num_reads=hi - lo)
        self.set_reads(packed, off, lens, lo, bounds)
        info["total_reads"] = total
        return info

    def _all_read_lengths(self):
        """the lengths of ALL reads, in global read order (one all-gather of the shards' u32 lengths, padded to the longest shard)"""
        lens = np.asarray(self._reads[2], dtype=np.uint32)
        if self.world == 1:
            return lens
        torch = self.be.torch
        per = np.diff(self.bounds).astype(np.int64)
        mx = int(max(per.max(initial=0), 1))
        pad = torch.zeros(mx, dtype=torch.int32, device=self.be.dev)
        if len(lens):
            pad[:len(lens)] = torch.from_numpy(lens.view(np.int32).copy()).to(self.be.dev)
        outs = [torch.zeros(mx, dtype=torch.int32, device=self.be.dev) for _ in range(self.world)]
        self.dist.all_gather(outs, pad)
        return np.concatenate([o[:int(n)].cpu().numpy().view(np.uint32) for o, n in zip(outs, per)]) if int(per.sum()) else np.zeros(0, np.uint32)

    def _release_cached(self):
        """Exchange buffers are torch tensors of tens of GB (config 5: 40 GB of instances per rank); freed ones stay in torch's caching
        allocator, where the library's own hipMalloc cannot reach them — hand them back to the device."""
        t = getattr(self.be, "torch", None)
        if t is not None and t.cuda.is_available():
            t.cuda.empty_cache()

    # ---- collectives ------------------------------------------------------------------------------------------------
    def _exchange_counts(self, counts, with_max=False):
        """counts[p] = records this rank sends to rank p -> what it receives from every rank.  with_max: every rank also learns the largest
        single message of the whole exchange (each rank sends its own largest beside the count), so that the ranks agree on the number of
        all-to-all rounds without another collective."""
        c = np.asarray(counts, dtype=np.int64)
        if not with_max:
            t = self.be.torch.tensor(c, device=self.be.dev)
            r = self.be.torch.empty_like(t)
            self.dist.all_to_all_single(r, t)
            return r.cpu().numpy()                      # .cpu() waits for the collective
        t = self.be.torch.tensor(np.stack([c, np.full_like(c, c.max(initial=0))], axis=1).reshape(-1), device=self.be.dev)
        r = self.be.torch.empty_like(t)
        self.dist.all_to_all_single(r, t)
        r = r.cpu().numpy().reshape(-1, 2)
        return r[:, 0].copy(), int(max(r[:, 1].max(initial=0), c.max(initial=0)))

    def _all_to_all_records(self, send, send_counts, recv_counts, largest=None):
        torch = self.be.torch
        sc = np.asarray(send_counts, dtype=np.int64); rc = np.asarray(recv_counts, dtype=np.int64)
        width = int(send.shape[1])
        recv = self.be.empty_records(int(rc.sum()), width)
        # Batched like the reference's BatchState (include/KmerOps.hpp:33-56, MAX_ALLTOALL_MEM): at most MAX_RECORDS_PER_PEER
        # records per peer and round.  Measured on MI355X / RCCL 2.26.6 (scratch test in profiles/r01_notes.md): a single
        # all_to_all_single message of >= ~2 GiB per peer delivers only its first GiB, silently.
        CH = max(1, self.MAX_RECORDS_PER_PEER * 2 // width)          # the cap is in bytes per peer: wider records, fewer of them
        rounds = int(max(1, -(-int(max(sc.max(initial=0), rc.max(initial=0)) if largest is None else largest) // CH)))
        if self.world > 1 and largest is None:      # (largest: the biggest message of the whole exchange is known to every rank already)
            t = torch.tensor([rounds], dtype=torch.int64, device=self.be.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            rounds = int(t.item())
        soff = np.concatenate([[0], np.cumsum(sc)]); roff = np.concatenate([[0], np.cumsum(rc)])
        if rounds == 1:
            self.dist.all_to_all_single(recv, send, output_split_sizes=[int(x) for x in rc], input_split_sizes=[int(x) for x in sc])
        else:
            for r in range(rounds):
                s_lo = np.minimum(r * CH, sc); s_hi = np.minimum((r + 1) * CH, sc)
                r_lo = np.minimum(r * CH, rc); r_hi = np.minimum((r + 1) * CH, rc)
                part = torch.cat([send[int(soff[p] + s_lo[p]):int(soff[p] + s_hi[p])] for p in range(self.world)])
                got = self.be.empty_records(int((r_hi - r_lo).sum()), width)
                self.dist.all_to_all_single(got, part, output_split_sizes=[int(x) for x in (r_hi - r_lo)], input_split_sizes=[int(x) for x in (s_hi - s_lo)])
                pos = 0
                for p in range(self.world):
                    n = int(r_hi[p] - r_lo[p])
                    recv[int(roff[p] + r_lo[p]):int(roff[p] + r_hi[p])] = got[pos:pos + n]
                    pos += n
        # the collective runs on torch's communication stream, the library on its own HIP stream: make the hand-over explicit
        self.be.synchronize()
        return recv

    def _exchange_packed_chunked(self, send, send_counts, recv_counts, chunks):
        """Exchange #1 in 8-byte records, in ROUNDS: round r + 1's all-to-all is posted (async_op: RCCL runs it on its own stream) before round r's records are
        unpacked by the library on ITS stream — the unpack (and the copy-free placement of its output in the final record buffer) overlaps the transfer.
        The sender's packing is not chunked (elba_dist_fill_send_packed fills the whole send buffer: one kernel pass over the reads).  VERDICT r4, task 6-ii.
        Rounds: at least `chunks`, and as many as keep a peer's message under MAX_RECORDS_PER_PEER bytes.  Returns the (k-mer, read << 32 | pos) records."""
        torch = self.be.torch
        W = self.world
        sc = np.asarray(send_counts, dtype=np.int64); rc = np.asarray(recv_counts, dtype=np.int64)
        CH = max(1, self.MAX_RECORDS_PER_PEER * 2)                    # 8-byte records per peer and round under the byte cap (the cap is 16 x MAX_RECORDS_PER_PEER bytes)
        rounds = int(max(chunks, -(-int(max(sc.max(initial=0), rc.max(initial=0))) // CH), 1))
        if W > 1:
            t = torch.tensor([rounds], dtype=torch.int64, device=self.be.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            rounds = int(t.item())
        soff = np.concatenate([[0], np.cumsum(sc)]); roff = np.concatenate([[0], np.cumsum(rc)])
        cut = lambda c, r: (c * r) // rounds                            # round r of a peer's records: [cut(c, r), cut(c, r + 1))
        recv = self.be.empty_records(int(rc.sum()), 2)
        self.exchange_rounds = rounds

        def post(r):
            s_lo, s_hi = cut(sc, r), cut(sc, r + 1)
            r_lo, r_hi = cut(rc, r), cut(rc, r + 1)
            part = torch.cat([send[int(soff[p] + s_lo[p]):int(soff[p] + s_hi[p])] for p in range(W)]) if rounds > 1 else send
            got = self.be.empty_records(int((r_hi - r_lo).sum()), 1)
            kw = dict(output_split_sizes=[int(x) for x in (r_hi - r_lo)], input_split_sizes=[int(x) for x in (s_hi - s_lo)])
            work = None
            try:
                work = self.dist.all_to_all_single(got, part, async_op=True, **kw)
            except TypeError:                                           # (a transport without async collectives: the host-staged rehearsal, the in-process test group)
                self.dist.all_to_all_single(got, part, **kw)
            return work, got, part, (r_hi - r_lo)

        pending = post(0)
        at = 0
        for r in range(rounds):
            work, got, part, counts = pending
            pending = post(r + 1) if r + 1 < rounds else None           # posted BEFORE this round is waited for and unpacked
            if work is not None and hasattr(work, "wait"):
                work.wait()                                             # torch's current stream waits for round r ...
                if hasattr(torch, "cuda") and torch.cuda.is_available() and str(self.be.dev).startswith("cuda"):
                    torch.cuda.current_stream().synchronize()           # ... and the host for that stream alone: the next round stays in flight on RCCL's
            else:
                self.be.synchronize()
            n = int(counts.sum())
            self.be.unpack_records(W, self.rank, got, counts, out=recv[at:at + n])      # (returns when the library's stream has finished: `got` may go)
            at += n
            del got, part
        return recv

    def _all_gather_words(self, local, n):
        torch = self.be.torch
        ns = [torch.zeros(1, dtype=torch.int64, device=self.be.dev) for _ in range(self.world)]
        self.dist.all_gather(ns, torch.tensor([n], dtype=torch.int64, device=self.be.dev))
        ns = [int(x.item()) for x in ns]
        mx = max(ns + [1])
        pad = self.be.empty_words(mx)
        pad.zero_()
        if n:
            pad[:n] = local
        outs = [self.be.empty_words(mx) for _ in range(self.world)]
        self.dist.all_gather(outs, pad)
        allw = torch.cat([o[:m] for o, m in zip(outs, ns)]) if sum(ns) else self.be.empty_words(0)
        self.be.synchronize()                       # torch streams -> library stream hand-over
        return allw, ns

    # ---- stages -----------------------------------------------------------------------------------------------------
    def build_kmer_matrix(self, row_batches=1):
        """get_kmer_count_map_keys/values + create_kmer_matrix + Transpose, distributed.  Returns (kmer stats, matrix stats).
        row_batches > 1: the owners are set up and the caller loads the panels row block by row block (load_row_block)."""
        W = self.world
        torch = self.be.torch
        # owners: value ranges balanced on the all-reduced histogram of the instances (4096 bins: 32 KB per rank)
        if W > 1:
            h = torch.from_numpy(self.be.value_histogram()).to(self.be.dev)
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            self.be.set_owner_ranges(owner_ranges_from_histogram(h.cpu().numpy(), W))
        # exchange #1: instances to owners
        sc = self.be.count_owners(W)
        rc = self._exchange_counts(sc)
        kw = kmer_words(self.k)
        # one-word k-mers travel as 8-byte records where (value inside the owner's range, instance index in the sender's reads) fit 64 bits (round 5;
        # include/elba_amd.h: elba_dist_packed_format) — every rank needs every rank's read lengths for that: one all-gather of 4 bytes per read
        fmt = None
        if kw == 1 and self.packed_exchange and hasattr(self.be, "packed_format"):
            fmt = self.be.packed_format(W, self.bounds, self._all_read_lengths())
        self.exchange_format = "8-byte records (%d value bits + %d index bits)" % fmt if fmt else "%d-byte records" % (8 * (kw + 1))
        if fmt:
            send = self.be.empty_records(int(sc.sum()), 1)
            self.be.fill_send_packed(W, send, np.concatenate([[0], np.cumsum(sc)[:-1]]))
            if self.exchange_chunks > 1 or (self.exchange_chunks == 0 and W > 1 and int(sc.max(initial=0)) >= (1 << 22)):
                # rounds of the all-to-all overlapped with the unpack of the round before (four rounds once a peer's message reaches 32 MB; "exchange_chunks" sets it)
                recv = self._exchange_packed_chunked(send, sc, rc, self.exchange_chunks if self.exchange_chunks > 1 else 4)
                del send
            else:
                got = self._all_to_all_records(send, sc, rc)
                del send
                self._release_cached()
                recv = self.be.unpack_records(W, self.rank, got, rc)
                del got
        else:
            send = self.be.empty_records(int(sc.sum()), kw + 1)
            self.be.fill_send(W, send, np.concatenate([[0], np.cumsum(sc)[:-1]]))
            recv = self._all_to_all_records(send, sc, rc)
            del send
        self._release_cached()
        ks = self.be.count_records(recv)
        # global k-mer ids: the owners hold ascending value ranges, so id = exclusive scan of the owners' counts + local index
        # (src/KmerOps.cpp:371-375: MPI_Exscan of the local map sizes) — one all-gather of W integers, no k-mer leaves its owner
        nloc = int(ks["reliable"])
        cnt = [torch.zeros(1, dtype=torch.int64, device=self.be.dev) for _ in range(W)]
        self.dist.all_gather(cnt, torch.tensor([nloc], dtype=torch.int64, device=self.be.dev))
        ns = [int(x.item()) for x in cnt]
        n_total = int(sum(ns))
        self.be.set_kmer_id_base(int(sum(ns[:self.rank])), n_total)
        del recv
        self._release_cached()
        self.n_total = n_total
        ks = dict(ks)
        ks["instances"] = int(sc.sum())          # instances enumerated from THIS rank's reads
        ks["nreads"] = self.nlocal
        self.exchange_bytes = dict(instances=int(sc.sum()) * (8 if fmt else 8 * (kw + 1)), panels=0, instance_format=self.exchange_format)
        if row_batches > 1:
            self.row_batches = row_batches
            return ks, None                      # the caller walks the row blocks: load_row_block(t), create_seed_matrix(), export_csr()
        self.row_batches = 1
        ms = self.load_row_block(0)
        return ks, ms

    def row_block(self, rank, t):
        """Rows [lo, hi) of rank `rank` in row block t of self.row_batches (equal read counts; the last block takes the remainder)."""
        lo, hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        per = -(-(hi - lo) // self.row_batches) if hi > lo else 0
        return min(hi, lo + t * per), min(hi, lo + (t + 1) * per)

    def load_row_block(self, t):
        """Exchange #2 for row block t of every rank (all ranks call this together): every column, whole, to each rank that has one of its
        reads in that rank's block; then the block's panel becomes the context's A.  One block = the whole shard unless build_kmer_matrix was
        given row_batches > 1 (a shard whose full panel would not fit: dense columns reach nearly every rank; the reference batches its
        exchange too, include/KmerOps.hpp:33-56)."""
        W = self.world
        if self.row_batches == 1:
            pc = self.be.panel_counts(W, self.bounds)
        else:
            wl = np.array([self.row_block(r, t)[0] for r in range(W)], dtype=np.int64); wh = np.array([self.row_block(r, t)[1] for r in range(W)], dtype=np.int64)
            pc = self.be.panel_counts_win(W, self.bounds, wl, wh)
        prc = self._exchange_counts(pc)
        send = self.be.empty_records(int(pc.sum()))
        if self.row_batches == 1:
            self.be.panel_fill(W, self.bounds, send, np.concatenate([[0], np.cumsum(pc)[:-1]]))
        else:
            self.be.panel_fill_win(W, self.bounds, wl, wh, send, np.concatenate([[0], np.cumsum(pc)[:-1]]))
        panel = self._all_to_all_records(send, pc, prc)
        del send
        self._release_cached()
        m_total = int(self.bounds[-1])
        # inline partners follow the pair-ownership rule of the mirror exchange (the parity rule over ALL rows): the panel gets them when that is how
        # it will be multiplied; a call without the exchange on such a panel reloads it without them (create_seed_matrix)
        want_inl = self.inline_partners if self.inline_partners is not None else ((W > 1 or self.force_exchange) and self.row_batches == 1 and hasattr(self.be, "seed_begin"))
        if hasattr(self.be, "set_option"):
            self.be.set_option("panel_inline", 1 if want_inl else 0)
            self._panel_inline = bool(want_inl)
        self._cur_block = t
        self.block = self.row_block(self.rank, t) if self.row_batches > 1 else (int(self.bounds[self.rank]), int(self.bounds[self.rank + 1]))
        ms = dict(self.be.set_panel(panel, m_total, self.n_total, self.block[0], self.block[1]))
        del panel
        self._release_cached()
        ms["panel_records"] = int(prc.sum())
        self.panel_records = int(prc.sum())
        self._slot = 0                    # a new panel: the slot size of the mirror exchange is guessed afresh
        ms["nnz"] = int(prc.sum())
        self.exchange_bytes["panels"] += int(pc.sum()) * 16
        return ms

    def create_seed_matrix(self, exchange=None):
        """This rank's rows of B.  With more than one rank (and unless exchange=False) every pair of rows that live on two ranks is
        accumulated by ONE of them and its mirror image travels to the other in one all-to-all of 32-byte records (elba_seed_matrix_begin /
        _fill / _end): each rank then does 1/world of the one-GPU work instead of computing every cross-rank pair twice.  Without the
        exchange the call has no communication at all."""
        if exchange is None:
            exchange = (self.world > 1 or self.force_exchange) and self.row_batches == 1 and hasattr(self.be, "seed_begin")
        if not exchange:
            if self._panel_inline:                  # (every rank takes this branch together: the reload is a collective)
                self.inline_partners = False
                self.load_row_block(self._cur_block)
            return self.be.create_seed_matrix()
        if exchange != "counted" and hasattr(self.be, "seed_send"):
            return self._create_seed_matrix_slots()
        W = self.world
        sc = self.be.seed_begin(W, self.bounds)
        rc, largest = self._exchange_counts(sc, with_max=True)
        send = self.be.empty_records(int(sc.sum()), 4)
        self.be.seed_fill(send, np.concatenate([[0], np.cumsum(sc)[:-1]]))
        recv = self._all_to_all_records(send, sc, rc, largest=largest)
        del send
        self.mirror_bytes = int(sc.sum()) * 32
        return self.be.seed_end(recv)

    def _create_seed_matrix_slots(self):
        """The step with ONE host synchronisation.  The mirror images travel in fixed-size slots (equal splits: no counts cross the host, no
        buffer is sized from a count), the library and the collective run on one stream, and what the counted variant checks in between —
        did everything fit? — is checked once at the end, by every rank alike (a flag in every slot's header).  The slot size starts from
        an upper-bound guess that all ranks compute from the same numbers; a repeated step grows it, a completed one sets it to what the step
        needed (+ 1/8: the same figure on every rank, it travels in the headers)."""
        W, torch = self.world, self.be.torch
        if not getattr(self.be, "shares_stream", False):
            self.be.use_current_stream()
        if getattr(self, "_slot", 0) == 0:
            # a first guess every rank agrees on: the mirror images a rank sends are about nnz(B of the shard) / 2, nnz(B) stays below a quarter of the
            # panel's entries on every read set seen, and they spread over W - 1 peers (twice that as margin; too small a guess costs one repeated
            # step, which tells every rank the size that was needed); all-reduced so that the largest shard decides
            g = torch.tensor([max(int(self.panel_records) // (8 * (W - 1)), 1 << 12) if W > 1 else 1 << 12], dtype=torch.int64, device=self.be.dev)
            if W > 1:
                self.dist.all_reduce(g, op=self.dist.ReduceOp.MAX)
            self._slot = int(g.item()) + 1
        for attempt in range(6):
            slot = self._slot
            if getattr(self, "_slot_bufs", (0, None, None))[0] != slot:
                self._slot_bufs = (slot, self.be.empty_records(W * slot, 4), self.be.empty_records(W * slot, 4))
            _, send, recv = self._slot_bufs
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.time_phases else None
            if ev: ev[0].record()
            self.be.seed_send(W, self.bounds, send, slot)
            if ev: ev[1].record()
            self.dist.all_to_all_single(recv, send)
            if ev: ev[2].record()
            st, need = self.be.seed_recv(recv, slot)
            if ev:
                ev[3].record(); ev[3].synchronize()
                self.phase_ms = dict(send=ev[0].elapsed_time(ev[1]), exchange=ev[1].elapsed_time(ev[2]), recv=ev[2].elapsed_time(ev[3]))
            if st is not None:
                self._slot = max(need, 2)            # every rank got the same figure: the next step's slots are as large as this one needed (+ 1/8)
                self.mirror_bytes = W * slot * 32
                return st
            self._slot = max(self._slot, need)
        raise RuntimeError("create_seed_matrix: the mirror exchange did not settle")

    # ---- the step after the path: x-drop alignment of the candidate pairs, sharded ------------------------------------------
    def align_seeds(self, mat=1, mis=-1, gap=-1, dropoff=15):
        """PairwiseAlignment across ranks.  The reads are replicated with ONE all-gather (2 bits per base: the whole read set is small
        next to a GPU's HBM; the reference moves row/column read blocks of its 2D grid instead, src/DistributedFastaData.cpp), then
        every rank aligns its share of the stored pairs of ITS rows of B with no further communication: pair {i < j} is taken by the
        rank of row i when i + j is even and by the rank of row j when it is odd (both ranks store the pair; the seeds of the mirrored
        entry are swapped back), always as (query i, target j) — the union over ranks is exactly the one-rank result."""
        packed, off, lens = self._reads
        # the caller's byte offsets may be padded, aligned or reordered (elba_set_reads accepts any layout): the block that travels is
        # re-packed densely — read r at the sum of the byte counts before it — which is the layout the receivers rebuild below
        rb = (lens.astype(np.int64) + 3) // 4
        dense = np.concatenate([[0], np.cumsum(rb)]).astype(np.int64) if len(lens) else np.zeros(1, np.int64)
        nb = int(dense[-1])
        nwords = (nb + 7) // 8
        buf = np.zeros(nwords * 8, dtype=np.uint8)
        if len(lens) and np.array_equal(off.astype(np.int64), dense[:-1]):
            buf[:nb] = packed[:nb]
        else:
            for r in range(len(lens)):
                buf[int(dense[r]):int(dense[r + 1])] = packed[int(off[r]):int(off[r]) + int(rb[r])]
        torch = self.be.torch
        local = torch.from_numpy(buf.view(np.int64).copy()).to(self.be.dev)
        allw, ns = self._all_gather_words(local, nwords)                          # packed bytes of every rank, 8-byte aligned blocks
        lens_t = torch.from_numpy(lens.astype(np.int64)).to(self.be.dev)
        alll, nl = self._all_gather_words(lens_t, len(lens))
        all_lens = alll.cpu().numpy().astype(np.uint32)
        # global byte offsets: rank r's block starts behind the blocks of the ranks before it; inside a block reads keep their offsets
        nbytes = (all_lens.astype(np.int64) + 3) // 4
        all_off = np.zeros(len(all_lens), dtype=np.uint64)
        base, at = 0, 0
        for r in range(self.world):
            n = nl[r]
            if n:
                loc = np.concatenate([[0], np.cumsum(nbytes[at:at + n])[:-1]])
                all_off[at:at + n] = (base + loc).astype(np.uint64)
            at += n; base += ns[r] * 8
        self.be.set_all_reads(allw, all_off, all_lens)
        return self.be.align_seeds(mat, mis, gap, dropoff)

    def export_overlaps(self):
        return self.be.export_overlaps()

    # ---- and the step after that: the string graph -------------------------------------------------------------------------
    def transitive_reduction(self, bad_read_cutoff=0.65, fuzz=1000):
        """src/main.cpp:305-312 across ranks.  The aligned pairs are small next to everything before them (52 bytes per pair): every
        rank gathers the others' shares with ONE all-gather, merges them into (row, col) order and runs the whole reduction itself —
        replicas, no further exchange (the reference runs a distributed SpGEMM and several distributed element-wise passes here).
        Every rank ends up holding all of S; export_string_graph(local=True) cuts out the rows of this rank's reads."""
        torch = self.be.torch
        g = self.be.export_overlaps()
        n = int(g["n"])
        rec = np.zeros((n, 7), dtype=np.int64)                    # row, col, 36 bytes of Overlap padded to 40
        if n:
            rec[:, 0] = g["rows"]; rec[:, 1] = g["cols"]
            v = np.zeros((n, 40), dtype=np.uint8); v[:, :36] = np.ascontiguousarray(g["vals"]).view(np.uint8).reshape(n, 36)
            rec[:, 2:] = v.view(np.int64).reshape(n, 5)
        local = torch.from_numpy(rec.reshape(-1).copy()).to(self.be.dev)
        allw, ns = self._all_gather_words(local, n * 7)
        a = allw.cpu().numpy().reshape(-1, 7)
        rows, cols = a[:, 0].copy(), a[:, 1].copy()
        vals = np.ascontiguousarray(a[:, 2:]).view(np.uint8).reshape(-1, 40)[:, :36].copy().view(capi.OVERLAP_DTYPE).reshape(-1)
        order = np.lexsort((cols, rows))
        m_total = int(self.bounds[-1])
        self.be.set_overlaps(m_total, rows[order], cols[order], vals[order])
        return self.be.transitive_reduction(bad_read_cutoff, fuzz)

    def export_string_graph(self, local=False):
        S = self.be.export_string_graph()
        if local:
            keep = (S["rows"] >= int(self.bounds[self.rank])) & (S["rows"] < int(self.bounds[self.rank + 1]))
            S = dict(n=int(keep.sum()), rows=S["rows"][keep], cols=S["cols"][keep], vals=S["vals"][keep])
        return S

    def export_csr(self):
        """This rank's rows of B (global column ids) — of the current row block when the shard is walked in blocks."""
        return self.be.export_csr(self.block[0], self.block[1])
