"""CPU tests of the multi-GPU driver (elba_amd/distributed.py): its sequencing and collectives under torch.distributed `gloo` with
world_size 2 (real processes), and under the in-process ThreadedGroup with 3 ranks.  The local compute of each rank is the test-side
NumpyBackend (oracle-based); what is under test is the driver: partition, owner exchange, global k-mer ids, column panels, row windows.
The stitched per-rank rows of B must equal the single-process oracle's B on the whole read set, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import dist_sim
import elba_amd
from elba_amd.distributed import DistributedOverlap, partition_by_bases
from oracle import pyoracle as po

K, LO, UP = 17, 2, 8


def _reads():
    return elba_amd.synth_reads(77, 40000, 10, 2500, 600, error_rate=0.03, min_len=100)


def _shard(packed, off, lens, lo, hi):
    b0 = int(off[lo]) if lo < len(off) else 0
    b1 = int(off[hi - 1]) + (int(lens[hi - 1]) + 3) // 4 if hi > lo else b0
    return np.concatenate([packed[b0:b1], np.zeros(16, np.uint8)]), (off[lo:hi] - np.uint64(b0)), lens[lo:hi]


def _expected():
    packed, off, lens, _ = _reads()
    o = po.Oracle(K, LO, UP)
    o.count_and_build(packed, off, lens)
    o.spgemm(2)
    return o


def _run_rank(rank, world, dist, backend, align=False):
    packed, off, lens, _ = _reads()
    bounds = partition_by_bases(lens, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sp, so, sl = _shard(packed, off, lens, lo, hi)
    d = DistributedOverlap(K, LO, UP, rank=rank, world=world, dist=dist, backend=backend)
    d.exchange_chunks = 2            # (exchange #1 in two rounds: under torch.distributed the second one is posted with async_op before the first is unpacked)
    d.set_reads(sp, so, sl, lo, bounds)
    ks, ms = d.build_kmer_matrix()
    assert d.exchange_rounds >= 2    # (more where a test lowers MAX_RECORDS_PER_PEER)
    st = d.create_seed_matrix()
    if align:
        d.align_seeds()
        ov = d.export_overlaps()
        sst = d.transitive_reduction(0.65, 1000)
        return d.export_csr(), ks, ms, st, ov, (sst, d.export_string_graph(), d.export_string_graph(local=True))
    return d.export_csr(), ks, ms, st


def test_partition_rule_matches_reference_greedy():
    # src/FastaIndex.cpp:47-94: take reads while the next one keeps the rank under total/p; the last rank takes the rest
    assert partition_by_bases([10, 20, 30, 40, 50, 5, 5], 3).tolist() == [0, 2, 3, 7]
    assert partition_by_bases([5] * 8, 4).tolist() == [0, 1, 2, 3, 8]          # strict '<' leaves the remainder to the last rank
    assert partition_by_bases([7, 7, 7], 1).tolist() == [0, 3]
    b = partition_by_bases(np.random.default_rng(1).integers(100, 5000, 1000), 8)
    assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) > 0).all()


def test_three_ranks_in_process_equal_single_process_oracle():
    o = _expected()
    parts = dist_sim.run_ranks(3, lambda r, h: _run_rank(r, 3, h, dist_sim.NumpyBackend(K, LO, UP)))
    B = dist_sim.stitch_rows([p[0] for p in parts])
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
    assert sum(p[1]["reliable"] for p in parts) == o.stat("N")          # owners partition the reliable k-mers
    assert sum(p[1]["entries"] for p in parts) == o.stat("Z")
    assert sum(p[1]["instances"] for p in parts) == o.stat("I")


def _run_rank_blocks(rank, world, handle, backend, nblocks):
    """The shard walked in row blocks: the owners are set up once, every block gets its own panel exchange, SpGEMM and rows of B."""
    packed, off, lens, _ = _reads()
    bounds = partition_by_bases(lens, world)
    a, b = int(bounds[rank]), int(bounds[rank + 1])
    sp, so, sl = _shard(packed, off, lens, a, b)
    d = DistributedOverlap(K, LO, UP, rank=rank, world=world, dist=handle, backend=backend)
    d.set_reads(sp, so, sl, a, bounds)
    ks, ms = d.build_kmer_matrix(row_batches=nblocks)
    assert ms is None
    rows = []
    for t in range(nblocks):
        d.load_row_block(t)
        d.create_seed_matrix()
        rows.append(d.export_csr())
    return dist_sim.stitch_rows(rows)


def test_row_block_batching_gives_the_same_rows():
    """A shard whose panel would not fit is walked block by block (elba_dist_panel_*_win): 2 ranks x 3 blocks must stitch to the oracle's B."""
    o = _expected()
    parts = dist_sim.run_ranks(2, lambda r, h: _run_rank_blocks(r, 2, h, dist_sim.NumpyBackend(K, LO, UP), 3))
    B = dist_sim.stitch_rows(parts)
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()


def test_without_the_mirror_exchange_the_rows_are_the_same():
    """create_seed_matrix(exchange=False): every rank accumulates its cross-rank pairs itself (no communication inside the call)."""
    o = _expected()

    def body(rank, h):
        packed, off, lens, _ = _reads()
        bounds = partition_by_bases(lens, 2)
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        sp, so, sl = _shard(packed, off, lens, a, b)
        d = DistributedOverlap(K, LO, UP, rank=rank, world=2, dist=h, backend=dist_sim.NumpyBackend(K, LO, UP))
        d.set_reads(sp, so, sl, a, bounds)
        d.build_kmer_matrix()
        d.create_seed_matrix(exchange=False)
        return d.export_csr()

    B = dist_sim.stitch_rows(dist_sim.run_ranks(2, body))
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()


def test_batched_all_to_all_rounds_give_the_same_result(monkeypatch):
    """The exchange is cut into rounds of at most MAX_RECORDS_PER_PEER records per peer (the reference batches its all-to-all too,
    include/KmerOps.hpp:33-56); force many rounds."""
    monkeypatch.setattr(DistributedOverlap, "MAX_RECORDS_PER_PEER", 4099)
    o = _expected()
    parts = dist_sim.run_ranks(2, lambda r, h: _run_rank(r, 2, h, dist_sim.NumpyBackend(K, LO, UP)))
    B = dist_sim.stitch_rows([p[0] for p in parts])
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()


def _check_alignment_union(parts):
    """driver logic of the sharded alignment stage (reads replicated by all-gather, i + j parity shares): union == one-rank oracle"""
    packed, off, lens, _ = _reads()
    o = _expected()
    want_r, want_c, want_v, _ = o.align_upper(packed, off, lens, nthreads=2)
    rows = np.concatenate([p["rows"] for p in parts]); cols = np.concatenate([p["cols"] for p in parts]); vals = np.concatenate([p["vals"] for p in parts])
    order = np.lexsort((cols, rows))
    assert len(rows) == len(want_r) and (rows[order] == want_r).all() and (cols[order] == want_c).all()
    for f in want_v.dtype.names:
        if f != "pad":
            assert (vals[order][f] == want_v[f]).all(), f


def _check_string_graphs(parts):
    """every rank gathered all shares and reduced the whole graph: each holds the one-rank S; the local cuts partition it"""
    packed, off, lens, _ = _reads()
    o = _expected()
    want_r, want_c, want_v, _ = o.align_upper(packed, off, lens, nthreads=2)
    S, _, st = po.string_graph(len(lens), want_r, want_c, want_v, cutoff=0.65, fuzz=1000)
    assert S["n"] > 0
    for sst, whole, local in parts:
        assert sst["nnz"] == S["n"] and (whole["rows"] == S["rows"]).all() and (whole["cols"] == S["cols"]).all()
        for f in S["vals"].dtype.names:
            if f != "pad":
                assert (whole["vals"][f] == S["vals"][f]).all(), f
    assert sum(local["n"] for _, _, local in parts) == S["n"]
    assert sorted(zip(np.concatenate([l["rows"] for _, _, l in parts]).tolist(), np.concatenate([l["cols"] for _, _, l in parts]).tolist())) == sorted(zip(S["rows"].tolist(), S["cols"].tolist()))


def test_three_ranks_alignment_shares_cover_every_pair_once():
    parts = dist_sim.run_ranks(3, lambda r, h: _run_rank(r, 3, h, dist_sim.NumpyBackend(K, LO, UP), align=True))
    _check_alignment_union([p[4] for p in parts])
    assert all(p[4]["n"] > 0 for p in parts)
    _check_string_graphs([p[5] for p in parts])


def _gloo_worker(rank, world, port, outdir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, ks, ms, st, ov, (sst, S, Sl) = _run_rank(rank, world, dist, dist_sim.NumpyBackend(K, LO, UP), align=True)
        np.savez(os.path.join(outdir, "r%d.npz" % rank), rowptr=B["rowptr"], col=B["col"], val=B["val"], reliable=ks["reliable"],
                 arows=ov["rows"], acols=ov["cols"], avals=ov["vals"], snnz=sst["nnz"], srows=S["rows"], scols=S["cols"], svals=S["vals"],
                 lrows=Sl["rows"], lcols=Sl["cols"], lvals=Sl["vals"])
    finally:
        dist.destroy_process_group()


def test_two_processes_over_gloo_equal_single_process_oracle(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(2)]
    B = dist_sim.stitch_rows([dict(rowptr=p["rowptr"], col=p["col"], val=p["val"]) for p in parts])
    o = _expected()
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
    assert sum(int(p["reliable"]) for p in parts) == o.stat("N")
    _check_alignment_union([dict(rows=p["arows"], cols=p["acols"], vals=p["avals"]) for p in parts])
    _check_string_graphs([(dict(nnz=int(p["snnz"])), dict(rows=p["srows"], cols=p["scols"], vals=p["svals"]),
                           dict(n=len(p["lrows"]), rows=p["lrows"], cols=p["lcols"], vals=p["lvals"])) for p in parts])


@pytest.mark.parametrize("k", [33, 65])
def test_multi_word_kmers_through_the_exchange(k):
    """k > 31: a record of exchange #1 carries two or three k-mer words + (read, pos); owners hash every word; global ids = rank in the
    lexicographically sorted union.  Two in-process ranks, driver logic only (NumpyBackend)."""
    packed, off, lens, _ = elba_amd.synth_reads(91, 6000, 8, 700, 150, error_rate=0.01, min_len=120)
    o = po.Oracle(k, 2, 10)
    o.count_and_build(packed, off, lens)
    o.spgemm(2)

    def body(rank, h):
        bounds = partition_by_bases(lens, 2)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        sp, so, sl = _shard(packed, off, lens, lo, hi)
        d = DistributedOverlap(k, 2, 10, rank=rank, world=2, dist=h, backend=dist_sim.NumpyBackend(k, 2, 10))
        d.set_reads(sp, so, sl, lo, bounds)
        ks, ms = d.build_kmer_matrix()
        d.create_seed_matrix()
        return d.export_csr(), ks

    parts = dist_sim.run_ranks(2, body)
    B = dist_sim.stitch_rows([p[0] for p in parts])
    oB = o.B()
    assert o.stat("Y") > 50
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
    assert sum(p[1]["reliable"] for p in parts) == o.stat("N") and sum(p[1]["entries"] for p in parts) == o.stat("Z")


def test_the_ranks_generate_shares_of_the_one_process_read_set_repeat_families_included():
    """bench.py --gpus N: rank r generates reads [bounds[r], bounds[r+1]) of the workload's read set (DistributedOverlap.generate_and_set_reads).
    The shares, concatenated, are the read set the one-GPU run generates — with the repeat families of the dense workloads (round 4: they
    were dropped, the N > 1 run of a dense workload built another read set and its self-validating counts said so)."""
    w = dict(seed=4, genome=60000, depth=12.0, avg_len=2000.0, sd_len=300.0, min_len=500, error=0.01, repeats=(3, 0.1, 400))
    packed, off, lens, _ = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                                repeat_families=3, repeat_fraction=0.1, repeat_len=400)
    plain = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"])
    assert not (len(plain[0]) == len(packed) and (plain[0] == packed).all())          # (the families change the reads)

    class Capture:
        """a backend that only keeps what set_reads hands it"""
        def set_reads(self, p, o, l, first):
            self.got = (np.array(p), np.array(o), np.array(l), first)

    seqs = []
    for rank in range(3):
        d = DistributedOverlap(K, LO, UP, rank=rank, world=3, dist=None, backend=Capture())
        info = d.generate_and_set_reads(w, weak=False)
        assert info["total_reads"] == len(lens)
        p, o, l, first = d.be.got
        assert first == int(d.bounds[rank])
        for i in range(len(l)):
            seqs.append(bytes(p[int(o[i]):int(o[i]) + (int(l[i]) + 3) // 4]))
    whole = [bytes(packed[int(off[i]):int(off[i]) + (int(lens[i]) + 3) // 4]) for i in range(len(lens))]
    assert seqs == whole


@pytest.mark.parametrize("world", [2, 3, 8])
def test_exchange_one_in_8_byte_records_equals_the_16_byte_exchange(world):
    """Exchange #1 (round 5): an instance travels as (value inside its owner's range) << index bits | instance index in the sender's reads — 8 bytes,
    the source rank known from the receive segment, every rank's read lengths all-gathered — and the owner turns it back into the 16-byte record
    (k-mer, global read << 32 | pos).  The driver with the packed exchange equals the driver without it and the one-process oracle, rows, counters and
    all; the bytes a rank sends are halved; with EIGHT ranks (the target machine: value ranges, panel windows and the ownership rule with eight
    participants) as with two and three."""
    o = _expected()

    def run(r, h, packed, chunks=1):
        rp, ro, rl, _ = _reads()
        bounds = partition_by_bases(rl, world)
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        sp, so, sl = _shard(rp, ro, rl, lo, hi)
        d = DistributedOverlap(K, LO, UP, rank=r, world=world, dist=h, backend=dist_sim.NumpyBackend(K, LO, UP))
        d.packed_exchange = packed
        d.exchange_chunks = chunks
        d.set_reads(sp, so, sl, lo, bounds)
        ks, ms = d.build_kmer_matrix()
        d.create_seed_matrix()
        assert d.exchange_rounds == (chunks if packed else 1)
        return d.export_csr(), ks, dict(d.exchange_bytes)

    a = dist_sim.run_ranks(world, lambda r, h: run(r, h, True))
    b = dist_sim.run_ranks(world, lambda r, h: run(r, h, False))
    c = dist_sim.run_ranks(world, lambda r, h: run(r, h, True, chunks=3))      # the packed exchange in three rounds, each unpacked while the next one travels
    Ba, Bb, Bc, oB = dist_sim.stitch_rows([p[0] for p in a]), dist_sim.stitch_rows([p[0] for p in b]), dist_sim.stitch_rows([p[0] for p in c]), o.B()
    for pa, pc in zip(a, c):
        assert all(pa[1][f] == pc[1][f] for f in ("reliable", "entries", "instances", "distinct")) and pa[2] == pc[2]
    for B in (Ba, Bb, Bc):
        assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
    for pa, pb in zip(a, b):
        assert all(pa[1][f] == pb[1][f] for f in ("reliable", "entries", "instances", "distinct"))
        assert pa[2]["instance_format"].startswith("8-byte") and pb[2]["instance_format"].startswith("16-byte")
        assert 2 * pa[2]["instances"] == pb[2]["instances"] and pa[2]["panels"] == pb[2]["panels"]
    assert sum(p[1]["reliable"] for p in a) == o.stat("N") and sum(p[1]["entries"] for p in a) == o.stat("Z")
