"""Test-side helpers for the distributed driver (elba_amd/distributed.py):

  * ThreadedGroup — an in-process communicator with torch.distributed's call signatures, so that W ranks can be run as W threads on
    ONE GPU (the -m gpu tests) exercising the real HIP building blocks with real exchange patterns;
  * NumpyBackend — a CPU stand-in for the HIP library's elba_dist_* calls, built on the ORACLE's k-mer enumeration and SpGEMM, so that
    the driver's sequencing and its collectives can be tested under gloo with world_size 2 on a machine without GPUs.

Both live under tests/ on purpose: the product (elba_amd/) never imports the oracle and has no CPU compute path.
"""
import threading

import numpy as np
import torch

from oracle import pyoracle as po


class ThreadedGroup:
    """W ranks = W threads of one process.  handle(rank) returns an object with all_to_all_single / all_gather / all_reduce /
    barrier that behave like torch.distributed's for that rank."""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def handle(self, rank):
        return _Handle(self, rank)


class _Handle:
    class ReduceOp:
        SUM, MAX = "sum", "max"

    def __init__(self, g, rank):
        self.g, self.rank = g, rank

    def barrier(self):
        self.g.bar.wait()

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None):
        g, W = self.g, self.g.world
        if input_split_sizes is None:
            n = input.shape[0] // W
            input_split_sizes = [n] * W
            output_split_sizes = [n] * W
        offs = np.concatenate([[0], np.cumsum(input_split_sizes)]).astype(np.int64)
        if input.is_cuda:
            torch.cuda.synchronize()
        g.slots[self.rank] = [input[int(offs[d]):int(offs[d + 1])] for d in range(W)]
        g.bar.wait()
        pos = 0
        for src in range(W):
            piece = g.slots[src][self.rank]
            assert piece.shape[0] == output_split_sizes[src], (piece.shape, output_split_sizes, src, self.rank)
            output[pos:pos + piece.shape[0]] = piece
            pos += piece.shape[0]
        if output.is_cuda:
            torch.cuda.synchronize()
        g.bar.wait()

    def all_gather(self, outs, t):
        g = self.g
        if t.is_cuda:
            torch.cuda.synchronize()
        g.slots[self.rank] = t
        g.bar.wait()
        for src in range(g.world):
            outs[src].copy_(g.slots[src])
        if t.is_cuda:
            torch.cuda.synchronize()
        g.bar.wait()

    def all_reduce(self, t, op="sum"):
        g = self.g
        g.slots[self.rank] = t.clone()
        g.bar.wait()
        vals = torch.stack([g.slots[s] for s in range(g.world)])
        res = vals.sum(0) if op == "sum" else vals.max(0).values
        g.bar.wait()
        t.copy_(res)


def run_ranks(world, fn):
    """Runs fn(rank, handle) on `world` threads; re-raises the first exception."""
    g = ThreadedGroup(world)
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r, g.handle(r))
        except BaseException as e:          # noqa: BLE001
            err[r] = e
            g.bar.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


_M1, _M2 = np.uint64(0xff51afd7ed558ccd), np.uint64(0xc4ceb9fe1a85ec53)


def _mix64(k):
    k = k.astype(np.uint64).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33); k *= _M1; k ^= k >> np.uint64(33); k *= _M2; k ^= k >> np.uint64(33)
    return k


class NumpyBackend:
    """CPU stand-in for HipBackend (same method names and tensor conventions), for the gloo tests only."""

    def __init__(self, k, lower, upper):
        self.k, self.lower, self.upper = k, lower, upper
        self.torch = torch
        self.dev = torch.device("cpu")

    def empty_records(self, n, width=2):
        return torch.zeros((max(int(n), 0), int(width)), dtype=torch.int64)

    def empty_words(self, n):
        return torch.zeros((max(int(n), 0),), dtype=torch.int64)

    @property
    def kw(self):
        return 3 if self.k > 64 else (2 if self.k > 32 else 1)

    def set_reads(self, packed, off, lens, first_global_id):
        L = po.lib()
        if self.kw > 1:                                    # multi-word k-mers: one oracle call per position (test sizes only)
            rows, rp = [], []
            w = np.zeros(3, dtype=np.uint64)
            for r in range(len(lens)):
                for p in range(max(0, int(lens[r]) - self.k + 1)):
                    L.orc_kmerN_at(packed.ctypes.data + int(off[r]), p, self.k, w.ctypes.data)
                    rows.append(w[:self.kw].copy()); rp.append(((first_global_id + r) << 32) | p)
            self.km = np.array(rows, dtype=np.uint64).reshape(-1, self.kw)
            self.rp = np.array(rp, dtype=np.uint64)
            return
        kms, rds, pss = [], [], []
        for r in range(len(lens)):
            out = np.zeros(max(1, int(lens[r])), dtype=np.uint64)
            n = L.orc_read_kmers(packed.ctypes.data + int(off[r]), int(lens[r]), self.k, out.ctypes.data)
            kms.append(out[:n]); rds.append(np.full(n, first_global_id + r, dtype=np.uint64)); pss.append(np.arange(n, dtype=np.uint64))
        self.km = np.concatenate(kms) if kms else np.zeros(0, np.uint64)
        self.rp = ((np.concatenate(rds) << np.uint64(32)) | np.concatenate(pss)) if kms else np.zeros(0, np.uint64)

    def _first_word(self):
        return self.km[:, 0] if self.km.ndim == 2 else self.km

    def value_histogram(self):
        return np.bincount((self._first_word() >> np.uint64(52)).astype(np.int64), minlength=4096).astype(np.int64)

    def set_owner_ranges(self, upper_bins):
        self.range_upper = np.asarray(upper_bins, dtype=np.int64)

    def _owner(self, km, W):
        if W == 1:
            return np.zeros(len(km), dtype=np.int64)
        first = km[:, 0] if km.ndim == 2 else km
        return np.searchsorted(self.range_upper, (first >> np.uint64(52)).astype(np.int64), side="right").astype(np.int64)

    def count_owners(self, W):
        self.own = self._owner(self.km, W)
        return np.bincount(self.own, minlength=W).astype(np.int64)

    def fill_send(self, W, send, offsets):
        order = np.argsort(self.own, kind="stable")
        km = self.km[order] if self.km.ndim == 2 else self.km[order][:, None]
        rec = np.concatenate([km, self.rp[order][:, None]], axis=1).view(np.int64)
        send.copy_(torch.from_numpy(rec.copy()))

    # exchange #1 with 8-byte records (include/elba_amd.h: elba_dist_packed_format): the same format, restated in numpy
    def packed_format(self, W, bounds, all_lens):
        if self.kw != 1 or getattr(self, "no_packed", False):
            return None
        k2 = 2 * self.k
        self._pb = np.asarray(bounds, dtype=np.int64)
        n = np.maximum(np.asarray(all_lens, dtype=np.int64) - self.k + 1, 0)
        self._poff = [np.concatenate([[0], np.cumsum(n[self._pb[r]:self._pb[r + 1]])]).astype(np.uint64) for r in range(W)]
        maxI = max(int(o[-1]) for o in self._poff)
        up = np.asarray(self.range_upper, dtype=np.int64) if W > 1 else np.array([4096], dtype=np.int64)
        lo_bins = np.concatenate([[0], up[:-1]])
        self._plo = (lo_bins.astype(np.uint64) << np.uint64(k2 - 12))
        maxw = int(((up - lo_bins).max()) << (k2 - 12))
        bits = lambda x: max(1, int(x).bit_length())
        vb, ib = bits(max(maxw - 1, 0)), bits(max(maxI - 1, 0))
        if vb + ib > 64:
            return None
        self._pib = ib
        return vb, ib

    def fill_send_packed(self, W, send, offsets):
        order = np.argsort(self.own, kind="stable")
        val = (self.km >> np.uint64(64 - 2 * self.k))[order]
        g = np.arange(len(self.km), dtype=np.uint64)[order]          # set_reads enumerated the instances in (read, pos) order: the index is the place
        rec = ((val - self._plo[self.own[order]]) << np.uint64(self._pib)) | g
        send.copy_(torch.from_numpy(rec.view(np.int64).copy()).reshape(-1, 1))

    def unpack_records(self, W, rank, packed, recv_counts, out=None):
        a = packed.numpy().reshape(-1).view(np.uint64)
        dest = out
        out = np.zeros((len(a), 2), dtype=np.uint64)
        at = 0
        for p in range(W):
            n = int(recv_counts[p])
            w = a[at:at + n]
            g = w & np.uint64((1 << self._pib) - 1)
            val = (w >> np.uint64(self._pib)) + self._plo[rank]
            r = np.searchsorted(self._poff[p], g, side="right") - 1
            out[at:at + n, 0] = val << np.uint64(64 - 2 * self.k)
            out[at:at + n, 1] = ((np.uint64(self._pb[p]) + r.astype(np.uint64)) << np.uint64(32)) | (g - self._poff[p][r])
            at += n
        res = torch.from_numpy(out.view(np.int64).copy())
        if dest is not None:
            dest.copy_(res)
            return dest
        return res

    def count_records(self, rec):
        a = rec.numpy().view(np.uint64)
        if self.kw > 1:
            kw = self.kw
            rp = a[:, kw]
            order = np.lexsort([rp] + [a[:, w] for w in range(kw - 1, -1, -1)])      # by words (first word most significant), then (read, pos)
            a = a[order]
            keys = a[:, :kw]
            new = np.ones(len(a), dtype=bool)
            if len(a) > 1:
                new[1:] = (keys[1:] != keys[:-1]).any(axis=1)
            start = np.flatnonzero(new)
            cnt = np.diff(np.append(start, len(a)))
            keep = (cnt >= self.lower) & (cnt <= self.upper)
            self.rel = keys[start[keep]]
            self.cols = [a[s:s + c, kw] for s, c in zip(start[keep], cnt[keep])]
            return dict(nreads=0, instances=len(a), distinct=len(start), reliable=int(keep.sum()), entries=int(cnt[keep].sum()), ms_total=0.0, ms_count=0.0, ms_lookup=0.0, ms_sort=0.0)
        km, rp = a[:, 0], a[:, 1]
        order = np.lexsort((rp, km))
        km, rp = km[order], rp[order]
        uk, start, cnt = np.unique(km, return_index=True, return_counts=True)
        keep = (cnt >= self.lower) & (cnt <= self.upper)
        self.rel = uk[keep]
        self.cols = [rp[s:s + c] for s, c in zip(start[keep], cnt[keep])]
        return dict(nreads=0, instances=len(km), distinct=len(uk), reliable=int(keep.sum()), entries=int(cnt[keep].sum()), ms_total=0.0, ms_count=0.0, ms_lookup=0.0, ms_sort=0.0)

    def reliable_kmers(self, n):
        return torch.from_numpy(np.ascontiguousarray(self.rel).reshape(-1).view(np.int64).copy())

    def set_kmer_id_base(self, base, nall):
        self.gid = (np.uint64(base) + np.arange(len(self.rel), dtype=np.uint64)).astype(np.uint64)

    def set_global_kmers(self, allk):
        if self.kw > 1:
            rows = allk.numpy().view(np.uint64).reshape(-1, self.kw)
            order = np.lexsort([rows[:, w] for w in range(self.kw - 1, -1, -1)])
            rank = {tuple(int(x) for x in rows[i]): j for j, i in enumerate(order)}
            self.gid = np.array([rank[tuple(int(x) for x in r)] for r in self.rel], dtype=np.uint64)
            return
        s = np.sort(allk.numpy().view(np.uint64))
        self.gid = np.searchsorted(s, self.rel).astype(np.uint64)

    def _dests(self, bounds, col):
        return np.unique(np.searchsorted(np.asarray(bounds, dtype=np.uint64), col >> np.uint64(32), side="right") - 1)

    def _dests_win(self, bounds, wl, wh, col):
        reads = col >> np.uint64(32)
        r = np.searchsorted(np.asarray(bounds, dtype=np.uint64), reads, side="right") - 1
        ok = (reads >= np.asarray(wl, dtype=np.uint64)[r]) & (reads < np.asarray(wh, dtype=np.uint64)[r])
        return np.unique(r[ok])

    def panel_counts_win(self, W, bounds, wl, wh):
        out = np.zeros(W, dtype=np.int64)
        for col in self.cols:
            for d in self._dests_win(bounds, wl, wh, col):
                out[d] += len(col)
        return out

    def panel_fill_win(self, W, bounds, wl, wh, send, offsets):
        per = [[] for _ in range(W)]
        for g, col in zip(self.gid, self.cols):
            for d in self._dests_win(bounds, wl, wh, col):
                per[d].append(np.stack([np.full(len(col), g, dtype=np.uint64), col], axis=1))
        flat = [np.concatenate(p) if p else np.zeros((0, 2), np.uint64) for p in per]
        rec = np.concatenate(flat) if flat else np.zeros((0, 2), np.uint64)
        send.copy_(torch.from_numpy(rec.view(np.int64).copy()))

    def panel_counts(self, W, bounds):
        out = np.zeros(W, dtype=np.int64)
        for col in self.cols:
            for d in self._dests(bounds, col):
                out[d] += len(col)
        return out

    def panel_fill(self, W, bounds, send, offsets):
        per = [[] for _ in range(W)]
        for g, col in zip(self.gid, self.cols):
            for d in self._dests(bounds, col):
                per[d].append(np.stack([np.full(len(col), g, dtype=np.uint64), col], axis=1))
        flat = [np.concatenate(p) if p else np.zeros((0, 2), np.uint64) for p in per]
        rec = np.concatenate(flat) if flat else np.zeros((0, 2), np.uint64)
        send.copy_(torch.from_numpy(rec.view(np.int64).copy()))

    def set_panel(self, rec, m_total, n_total, row_lo, row_hi):
        a = rec.numpy().view(np.uint64)
        self.o = po.Oracle(self.k, self.lower, self.upper)
        self.o.set_triples(m_total, n_total, (a[:, 1] >> np.uint64(32)).astype(np.int64), a[:, 0].astype(np.int64), (a[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        self.win = (row_lo, row_hi)
        return dict(nrows=m_total, ncols=n_total, nnz=len(a), max_row_nnz=0, ms_total=0.0)

    def create_seed_matrix(self):
        self.o.spgemm(1)
        B = self.o.B()
        lo, hi = self.win
        e0, e1 = int(B["rowptr"][lo]), int(B["rowptr"][hi])
        self.B = dict(M=hi - lo, Y=e1 - e0, rowptr=B["rowptr"][lo:hi + 1] - e0, col=B["col"][e0:e1].astype(np.int64), val=B["val"][e0:e1])
        return dict(nnz=e1 - e0, products=0, algorithmic_bytes=0, ms_total=0.0, ms_numeric=0.0, ms_symbolic=0.0, ms_finalize=0.0)

    def export_csr(self, row_lo, row_hi):
        return self.B

    # ---- stand-in for elba_seed_matrix_begin / _fill / _end: ONE rank accumulates a cross-rank pair, the other receives its mirror image ----
    @staticmethod
    def _owns(i, j):
        return (j < i) if ((i ^ j) & 1) else (j > i)

    def seed_begin(self, W, bounds):
        self.o.spgemm(1)
        B = self.o.B()
        lo, hi = self.win
        self._kept, self._remote = {}, [[] for _ in range(W)]
        b = np.asarray(bounds, dtype=np.int64)
        for i in range(lo, hi):
            for e in range(int(B["rowptr"][i]), int(B["rowptr"][i + 1])):
                j, v = int(B["col"][e]), B["val"][e]
                if j != i and not self._owns(i, j):
                    continue
                self._kept.setdefault(i, []).append((j, v))
                if j != i:
                    mv = np.zeros(1, dtype=po.SEED_DTYPE)[0]
                    mv["q0"], mv["t0"], mv["q1"], mv["t1"], mv["numshared"] = v["t0"], v["q0"], v["t1"], v["q1"], v["numshared"]
                    if lo <= j < hi:
                        self._kept.setdefault(j, []).append((i, mv.copy()))
                    else:
                        d = int(np.searchsorted(b, j, side="right") - 1)
                        self._remote[d].append((j, i, int(mv["q0"]), int(mv["t0"]), int(mv["q1"]), int(mv["t1"]), int(mv["numshared"]), 0))
        return np.array([len(x) for x in self._remote], dtype=np.int64)

    def seed_fill(self, send, offsets):
        flat = [r for per in self._remote for r in per]
        rec = np.array(flat, dtype=np.uint32).reshape(-1, 8) if flat else np.zeros((0, 8), np.uint32)
        send.copy_(torch.from_numpy(rec.view(np.int64).reshape(-1, 4).copy()))

    def seed_end(self, recv):
        lo, hi = self.win
        r = recv.numpy().view(np.uint32).reshape(-1, 8)
        for (j, i, q0, t0, q1, t1, n, _) in r.tolist():
            assert lo <= j < hi
            mv = np.zeros(1, dtype=po.SEED_DTYPE)[0]
            mv["q0"], mv["t0"], mv["q1"], mv["t1"], mv["numshared"] = q0, t0, q1, t1, n
            self._kept.setdefault(j, []).append((i, mv.copy()))
        rowptr, col, val = [0], [], []
        for i in range(lo, hi):
            ents = sorted(self._kept.get(i, []), key=lambda x: x[0])
            col += [e[0] for e in ents]; val += [e[1] for e in ents]
            rowptr.append(len(col))
        self.B = dict(M=hi - lo, Y=len(col), rowptr=np.array(rowptr, dtype=np.int64), col=np.array(col, dtype=np.int64),
                      val=np.array(val, dtype=po.SEED_DTYPE) if val else np.zeros(0, dtype=po.SEED_DTYPE))
        return dict(nnz=len(col), products=0, algorithmic_bytes=0, ms_total=0.0, ms_numeric=0.0, ms_symbolic=0.0, ms_finalize=0.0)

    def set_all_reads(self, packed_words, byte_off, lens):
        self.all_reads = (packed_words.numpy().view(np.uint8).copy(), np.asarray(byte_off, dtype=np.uint64), np.asarray(lens, dtype=np.uint32))

    def align_seeds(self, mat, mis, gap, dropoff):
        """The rank's share of the pairs (i + j parity rule of elba_align_seeds on a row shard), aligned with the oracle's x-drop."""
        buf, off, lens = self.all_reads
        lo, hi = self.win
        B = self.B
        rows_l = np.repeat(np.arange(lo, hi), np.diff(B["rowptr"]))
        out_r, out_c, out_v = [], [], []
        L = po.lib()
        for e in range(B["Y"]):
            i, j = int(rows_l[e]), int(B["col"][e])
            if i == j or ((i + j) & 1) != (0 if j > i else 1):
                continue
            v = B["val"][e]
            qi, tj = (i, j) if j > i else (j, i)
            q0, t0 = (int(v["q0"]), int(v["t0"])) if j > i else (int(v["t0"]), int(v["q0"]))
            o = np.zeros(1, dtype=po.OVERLAP_DTYPE)
            L.orc_overlap_extend(buf.ctypes.data + int(off[qi]), int(lens[qi]), buf.ctypes.data + int(off[tj]), int(lens[tj]), q0, t0, self.k, mat, mis, gap, dropoff, o.ctypes.data, None)
            out_r.append(qi); out_c.append(tj); out_v.append(o[0])
        self.ov = dict(n=len(out_r), rows=np.array(out_r, dtype=np.int64), cols=np.array(out_c, dtype=np.int64),
                       vals=np.array(out_v, dtype=po.OVERLAP_DTYPE) if out_v else np.zeros(0, dtype=po.OVERLAP_DTYPE))
        return dict(nalignments=len(out_r))

    def export_overlaps(self):
        return self.ov

    def set_overlaps(self, nreads, rows, cols, vals):
        self.edges = (nreads, rows, cols, vals)

    def transitive_reduction(self, bad_read_cutoff, fuzz):
        self.S, self.flags, st = po.string_graph(*self.edges, cutoff=bad_read_cutoff, fuzz=fuzz)
        return st

    def export_string_graph(self):
        return self.S

    def synchronize(self):
        pass


def stitch_rows(parts):
    """Concatenate per-rank row blocks of B into one CSR."""
    rowptr, col, val, base = [np.zeros(1, np.int64)], [], [], 0
    for b in parts:
        rowptr.append(b["rowptr"][1:] + base)
        col.append(b["col"]); val.append(b["val"])
        base += int(b["rowptr"][-1])
    return dict(rowptr=np.concatenate(rowptr), col=np.concatenate(col), val=np.concatenate(val), Y=base)
