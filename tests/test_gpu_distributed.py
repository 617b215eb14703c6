"""-m gpu: the distributed building blocks of the HIP library (elba_dist_*) driven by elba_amd/distributed.py with W ranks run as W
threads on the one GPU of the test box (ThreadedGroup).  The exchange patterns, owner hashing, global k-mer ids, panels and row windows
are the real ones; only the transport is in-process.  Stitched rows of B must equal the oracle's B on the whole read set."""
import numpy as np
import pytest

import dist_sim
import elba_amd
from elba_amd.distributed import DistributedOverlap, HipBackend, partition_by_bases
from oracle import pyoracle as po
from test_distributed_cpu import _shard

pytestmark = pytest.mark.gpu


def _run(world, reads, k, lo, up, align=False):
    packed, off, lens, _ = reads
    bounds = partition_by_bases(lens, world)

    def body(rank, h):
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        sp, so, sl = _shard(packed, off, lens, a, b)
        d = DistributedOverlap(k, lo, up, device=0, rank=rank, world=world, dist=h, backend=HipBackend(k, lo, up, 0))
        d.set_reads(sp, so, sl, a, bounds)
        ks, ms = d.build_kmer_matrix()
        st = d.create_seed_matrix()
        out = (d.export_csr(), ks, ms, st)
        if align:
            a = d.align_seeds()
            out = out + (d.export_overlaps(), a)
            sst = d.transitive_reduction()
            out = out + ((sst, d.export_string_graph(), d.export_string_graph(local=True)),)
        d.be.e.close()
        return out

    return dist_sim.run_ranks(world, body)


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_overlap_equals_oracle(world):
    reads = elba_amd.synth_reads(31, 300000, 15, 4000, 900, error_rate=0.10, min_len=200)
    o = po.Oracle(17, 2, 8)
    o.count_and_build(*reads[:3])
    o.spgemm(4)
    parts = _run(world, reads, 17, 2, 8)
    B = dist_sim.stitch_rows([p[0] for p in parts])
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all()
    assert (B["val"] == oB["val"]).all()
    assert sum(p[1]["reliable"] for p in parts) == o.stat("N")
    assert sum(p[1]["entries"] for p in parts) == o.stat("Z")
    assert sum(p[1]["instances"] for p in parts) == o.stat("I")
    assert sum(p[3]["nnz"] for p in parts) == o.stat("Y")
    assert sum(p[3]["products"] for p in parts) == o.stat("P")


def test_uneven_shards_and_empty_rank():
    """Fewer long reads than ranks can balance: some rank ends up with very few reads; results must not change."""
    reads = elba_amd.synth_reads(32, 20000, 12, 6000, 3000, error_rate=0.05, min_len=50)
    o = po.Oracle(17, 2, 12)
    o.count_and_build(*reads[:3])
    o.spgemm(2)
    parts = _run(3, reads, 17, 2, 12)
    B = dist_sim.stitch_rows([p[0] for p in parts])
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()


def merge_overlaps(parts):
    """Union of the ranks' alignment shares, in the one-rank order (rows ascending, columns ascending)."""
    rows = np.concatenate([p["rows"] for p in parts]); cols = np.concatenate([p["cols"] for p in parts]); vals = np.concatenate([p["vals"] for p in parts])
    order = np.lexsort((cols, rows))
    return rows[order], cols[order], vals[order]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_alignment_equals_one_rank_oracle(world):
    """Every candidate pair is aligned exactly once across the ranks (i + j parity rule), as (query = smaller id, target = larger id),
    from the seed the one-rank run would use: the union equals the oracle's PairwiseAlignment on the whole read set, field by field."""
    reads = elba_amd.synth_reads(33, 120000, 12, 3000, 700, error_rate=0.10, min_len=200)
    o = po.Oracle(17, 2, 8)
    o.count_and_build(*reads[:3])
    o.spgemm(4)
    want_r, want_c, want_v, _ = o.align_upper(reads[0], reads[1], reads[2], nthreads=8)
    parts = _run(world, reads, 17, 2, 8, align=True)
    rows, cols, vals = merge_overlaps([p[4] for p in parts])
    assert len(rows) == len(want_r) and (rows == want_r).all() and (cols == want_c).all()
    for f in want_v.dtype.names:
        if f != "pad":
            assert (vals[f] == want_v[f]).all(), f
    shares = [p[5]["nalignments"] for p in parts]
    assert sum(shares) == len(want_r) and min(shares) > 0.5 * max(shares)          # balanced without any exchange


def test_string_graph_from_gathered_shares_equals_one_rank_oracle():
    """Every rank gathers the others' aligned pairs (one all-gather), merges them and reduces the whole graph on its GPU context: each
    holds the one-rank S of the oracle; the local cuts (rows of the rank's reads) partition it."""
    reads = elba_amd.synth_reads(34, 100000, 12, 3000, 700, error_rate=0.02, min_len=300)
    o = po.Oracle(17, 2, 12)
    o.count_and_build(*reads[:3])
    o.spgemm(4)
    want_r, want_c, want_v, _ = o.align_upper(reads[0], reads[1], reads[2], nthreads=8)
    S, _, st = po.string_graph(len(reads[2]), want_r, want_c, want_v)
    assert S["n"] > 0 and st["marked"] > 0
    parts = _run(3, reads, 17, 2, 12, align=True)
    for p in parts:
        sst, whole, local = p[6]
        assert all(sst[k] == st[k] for k in ("bad_reads", "contained_reads", "edges_kept", "products", "marked", "removed", "nnz"))
        assert (whole["rows"] == S["rows"]).all() and (whole["cols"] == S["cols"]).all()
        for f in S["vals"].dtype.names:
            if f != "pad":
                assert (whole["vals"][f] == S["vals"][f]).all(), f
    assert sum(p[6][2]["n"] for p in parts) == S["n"]


@pytest.mark.parametrize("k,world", [(33, 2), (63, 3), (65, 2), (95, 3)])
def test_multi_word_kmers_sharded_overlap_equals_oracle(k, world):
    """k > 31 across ranks: records of two / three k-mer words + (read, pos), every word in the owner hash, the owner's multi-word sort
    (index permutation, last word first), global ids by lexicographic rank in the gathered union."""
    reads = elba_amd.synth_reads(35, 120000, 10, 2500, 600, error_rate=0.02, min_len=200)
    o = po.Oracle(k, 2, 12)
    o.count_and_build(*reads[:3])
    o.spgemm(4)
    assert o.stat("Y") > 500
    parts = _run(world, reads, k, 2, 12)
    B = dist_sim.stitch_rows([p[0] for p in parts])
    oB = o.B()
    assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
    assert sum(p[1]["reliable"] for p in parts) == o.stat("N") and sum(p[1]["entries"] for p in parts) == o.stat("Z")
    assert sum(p[1]["instances"] for p in parts) == o.stat("I")


def test_row_block_batching_on_the_gpu():
    """Row-block batching with the real HIP building blocks (elba_dist_panel_counts_win / _fill_win, the owner's columns kept apart from the
    panels that overwrite the context's A): 2 ranks x 4 blocks and 3 ranks x 2 blocks stitch to the oracle's B."""
    reads = elba_amd.synth_reads(35, 250000, 15, 4000, 900, error_rate=0.10, min_len=200)
    packed, off, lens, _ = reads
    o = po.Oracle(17, 2, 8); o.count_and_build(packed, off, lens); o.spgemm(4)
    oB = o.B()
    for world, nblocks in ((2, 4), (3, 2)):
        bounds = partition_by_bases(lens, world)

        def body(rank, h):
            a, b = int(bounds[rank]), int(bounds[rank + 1])
            sp, so, sl = _shard(packed, off, lens, a, b)
            d = DistributedOverlap(17, 2, 8, device=0, rank=rank, world=world, dist=h, backend=HipBackend(17, 2, 8, 0))
            d.set_reads(sp, so, sl, a, bounds)
            ks, ms = d.build_kmer_matrix(row_batches=nblocks)
            rows, prod = [], 0
            for t in range(nblocks):
                d.load_row_block(t)
                st = d.create_seed_matrix()
                prod += st["products"]
                rows.append(d.export_csr())
            d.be.e.close()
            return dist_sim.stitch_rows(rows), prod

        parts = dist_sim.run_ranks(world, body)
        B = dist_sim.stitch_rows([p[0] for p in parts])
        assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
        assert sum(p[1] for p in parts) == o.stat("P")


@pytest.mark.parametrize("world", [2, 3])
def test_with_and_without_the_mirror_exchange(world):
    """Default at world > 1: every cross-rank pair is accumulated by ONE rank and its mirror image sent to the other (elba_seed_matrix_begin /
    _fill / _end + one all-to-all); exchange=False: both ranks accumulate it, no communication.  Same rows of B either way (the oracle's),
    and with the exchange the ranks' products add up to the ONE-GPU symmetric schedule's, not to twice the cross-rank part."""
    reads = elba_amd.synth_reads(36, 250000, 15, 4000, 900, error_rate=0.10, min_len=200)
    packed, off, lens, _ = reads
    o = po.Oracle(17, 2, 8); o.count_and_build(packed, off, lens); o.spgemm(4)
    oB = o.B()
    bounds = partition_by_bases(lens, world)
    # True: fixed-size slots, one host synchronisation per step (elba_seed_matrix_send / _recv); "tiny": the same from a slot of 4 records — every
    # rank is told to repeat the step with the size that was needed; "counted": begin / fill / end with counts crossing the host
    for exchange in (True, "tiny", "counted", False):
        def body(rank, h):
            a, b = int(bounds[rank]), int(bounds[rank + 1])
            sp, so, sl = _shard(packed, off, lens, a, b)
            d = DistributedOverlap(17, 2, 8, device=0, rank=rank, world=world, dist=h, backend=HipBackend(17, 2, 8, 0))
            d.set_reads(sp, so, sl, a, bounds)
            d.build_kmer_matrix()
            if exchange == "tiny":
                d._slot = 4
            st = d.create_seed_matrix(exchange=True if exchange == "tiny" else exchange)
            st2 = d.create_seed_matrix(exchange=True if exchange == "tiny" else exchange)          # a second call on the same panel
            out = (d.export_csr(), st, st2)
            d.be.e.close()
            return out

        parts = dist_sim.run_ranks(world, body)
        B = dist_sim.stitch_rows([p[0] for p in parts])
        assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
        assert sum(p[1]["nnz"] for p in parts) == o.stat("Y") == sum(p[2]["nnz"] for p in parts)
        assert sum(p[1]["nnz_upper"] for p in parts) == o.stat("nupper") and sum(p[1]["nnz_diag"] for p in parts) == o.stat("ndiag")
        assert sum(p[1]["products"] for p in parts) == o.stat("P")
        if exchange:
            assert sum(p[1]["nnz_before_prune"] for p in parts) == o.stat("Yraw")


@pytest.mark.parametrize("world,nblocks", [(2, 1), (2, 3), (3, 2)])
def test_dense_columns_on_shards_and_row_blocks_take_the_dense_path(world, nblocks):
    """Deep, nearly error-free reads with a generous UPPER (columns of ~30 reads: BASELINE config 5's shape) on row shards and on row blocks of
    shards.  A context with a row window stores its padded columns rotated — the window's reads first — so that "the smaller row owns a pair of
    the window, partners outside it are kept" still means "the candidates of a row entry are the column's slots behind its own": the dense
    kernel runs there too (a_csr_format says so), and the stitched rows equal the oracle's B, seeds included.  nblocks == 1: whole shards,
    with the mirror exchange (general kernel on the same dense-format matrix) and without it (dense kernel)."""
    reads = elba_amd.synth_reads(91, 60000, 30, 3000, 600, error_rate=0.01, min_len=500, repeat_families=3, repeat_fraction=0.1, repeat_len=400)
    packed, off, lens, _ = reads
    o = po.Oracle(17, 2, 40); o.count_and_build(packed, off, lens); o.spgemm(8)
    oB = o.B()
    bounds = partition_by_bases(lens, world)
    for exchange in ((True, False) if nblocks == 1 else (False,)):
        def body(rank, h):
            a, b = int(bounds[rank]), int(bounds[rank + 1])
            sp, so, sl = _shard(packed, off, lens, a, b)
            d = DistributedOverlap(17, 2, 40, device=0, rank=rank, world=world, dist=h, backend=HipBackend(17, 2, 40, 0))
            d.set_reads(sp, so, sl, a, bounds)
            if nblocks == 1:
                d.build_kmer_matrix()
                fmt = d.be.e.device_view()["a_csr_format"]
                st = d.create_seed_matrix(exchange=exchange)
                out = (d.export_csr(), st["products"], fmt, st["nnz"])
            else:
                d.build_kmer_matrix(row_batches=nblocks)
                rows, prod, fmt, nnz = [], 0, 2, 0
                for t in range(nblocks):
                    d.load_row_block(t)
                    fmt = min(fmt, d.be.e.device_view()["a_csr_format"])
                    st = d.create_seed_matrix()
                    prod += st["products"]; nnz += st["nnz"]
                    rows.append(d.export_csr())
                out = (dist_sim.stitch_rows(rows), prod, fmt, nnz)
            d.be.e.close()
            return out

        parts = dist_sim.run_ranks(world, body)
        assert all(p[2] == 2 for p in parts)                   # ELBA_CSR_DENSE on every shard / row block
        B = dist_sim.stitch_rows([p[0] for p in parts])
        assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all()
        bad = np.nonzero(B["val"] != oB["val"])[0]
        assert len(bad) == 0, (len(bad), B["val"][bad[:4]], oB["val"][bad[:4]])
        assert sum(p[3] for p in parts) == o.stat("Y")
        assert sum(p[1] for p in parts) == o.stat("P")


def test_mid_size_sharded_build_equals_the_one_gpu_matrix():
    """BASELINE config 3 at 1/8 of its genome (25 k reads of 10 kb, 250 M k-mer instances — the one-GPU run takes the two-level partition path,
    the owners of the sharded run sort) on 4 ranks with the mirror exchange: the stitched rows equal the one-GPU B bit for bit, the ranks'
    products add up to the one-GPU schedule's and every rank does about a quarter of it."""
    reads = elba_amd.synth_reads(2, 66_700_000 // 8, 30.0, 10000.0, 1500.0, error_rate=0.15, min_len=1000)
    packed, off, lens, _ = reads
    e = elba_amd.Engine(17, 2, 8, options={"kmer_msd": 1}); e.set_reads(packed, off, lens); e.count_kmers(); e.create_kmer_matrix(); st = e.create_seed_matrix()
    B1 = e.export_csr(); e.close()
    world = 4
    bounds = partition_by_bases(lens, world)

    def body(rank, h):
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        sp, so, sl = _shard(packed, off, lens, a, b)
        d = DistributedOverlap(17, 2, 8, device=0, rank=rank, world=world, dist=h, backend=HipBackend(17, 2, 8, 0))
        d.set_reads(sp, so, sl, a, bounds)
        d.build_kmer_matrix()
        s2 = d.create_seed_matrix()
        out = (d.export_csr(), s2)
        d.be.e.close()
        return out

    parts = dist_sim.run_ranks(world, body)
    B = dist_sim.stitch_rows([p[0] for p in parts])
    assert B["Y"] == B1["Y"] == st["nnz"] and (B["rowptr"] == B1["rowptr"]).all() and (B["col"] == B1["col"].astype(np.int64)).all() and (B["val"] == B1["val"]).all()
    prods = [p[1]["products"] for p in parts]
    assert sum(prods) == st["products"] and max(prods) < 1.35 * min(prods)


def test_a_panel_with_inline_partners_is_multiplied_with_the_exchange_only():
    """The panel of a shard carries inline partners when the step will use the mirror exchange (option "panel_inline", set by the driver): they follow
    the parity rule over ALL rows, which is the exchange's ownership rule.  elba_create_seed_matrix on such a windowed matrix would apply another rule
    (partners outside the window are kept) and is refused; the driver's create_seed_matrix(exchange=False) reloads the panel without them — both
    give the oracle's rows."""
    world = 2
    packed, off, lens, _ = elba_amd.synth_reads(37, 200000, 15, 4000, 900, error_rate=0.10, min_len=200)
    o = po.Oracle(17, 2, 8); o.count_and_build(packed, off, lens); o.spgemm(4)
    oB = o.B()
    bounds = partition_by_bases(lens, world)

    def body(rank, h):
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        sp, so, sl = _shard(packed, off, lens, a, b)
        d = DistributedOverlap(17, 2, 8, device=0, rank=rank, world=world, dist=h, backend=HipBackend(17, 2, 8, 0))
        d.set_reads(sp, so, sl, a, bounds)
        d.build_kmer_matrix()
        assert d._panel_inline
        refused = False
        try:
            d.be.e.create_seed_matrix()                 # straight to the library: no exchange on a windowed matrix with inline partners
        except elba_amd.ElbaError:
            refused = True
        st = d.create_seed_matrix()                     # with the exchange
        B1 = d.export_csr()
        st2 = d.create_seed_matrix(exchange=False)      # the driver reloads the panel without inline partners
        assert not d._panel_inline
        out = (B1, d.export_csr(), refused, st["nnz"], st2["nnz"])
        d.be.e.close()
        return out

    parts = dist_sim.run_ranks(world, body)
    assert all(p[2] for p in parts)
    for which in (0, 1):
        B = dist_sim.stitch_rows([p[which] for p in parts])
        assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
    assert sum(p[3] for p in parts) == o.stat("Y") == sum(p[4] for p in parts)


def test_eight_ranks_with_8_byte_exchange_records_equal_the_one_gpu_matrix():
    """The target machine has EIGHT GPUs: eight ranks as threads on the one GPU of the test box — value-range owners, panel windows,
    the pair-ownership rule and the mirror exchange with eight participants — on BASELINE config 3 at 1/16 of its genome.  Exchange #1 travels as 8-byte
    records (round 5: value inside the owner's range | instance index in the sender's reads; elba_dist_packed_format) — half the bytes of the
    (k-mer, read << 32 | pos) records, which the same driver with packed_exchange = False still sends: both equal the one-GPU B bit for bit."""
    reads = elba_amd.synth_reads(2, 66_700_000 // 16, 30.0, 10000.0, 1500.0, error_rate=0.15, min_len=1000)
    packed, off, lens, _ = reads
    e = elba_amd.Engine(17, 2, 8, options={"kmer_msd": 1}); e.set_reads(packed, off, lens); ks1 = e.count_kmers(); e.create_kmer_matrix(); st = e.create_seed_matrix()
    B1 = e.export_csr(); e.close()
    world = 8
    bounds = partition_by_bases(lens, world)

    def body(rank, h, packed_records):
        a, b = int(bounds[rank]), int(bounds[rank + 1])
        sp, so, sl = _shard(packed, off, lens, a, b)
        d = DistributedOverlap(17, 2, 8, device=0, rank=rank, world=world, dist=h, backend=HipBackend(17, 2, 8, 0))
        d.packed_exchange = packed_records
        d.exchange_chunks = 3            # (the packed exchange in three rounds, each unpacked into its place in the record buffer while the next one travels)
        d.set_reads(sp, so, sl, a, bounds)
        ks, ms = d.build_kmer_matrix()
        assert d.exchange_rounds == (3 if packed_records else 1)
        s2 = d.create_seed_matrix()
        out = (d.export_csr(), s2, ks, dict(d.exchange_bytes))
        d.be.e.close()
        return out

    for packed_records in (True, False):
        parts = dist_sim.run_ranks(world, lambda r, h: body(r, h, packed_records))
        B = dist_sim.stitch_rows([p[0] for p in parts])
        assert B["Y"] == B1["Y"] == st["nnz"] and (B["rowptr"] == B1["rowptr"]).all() and (B["col"] == B1["col"].astype(np.int64)).all() and (B["val"] == B1["val"]).all()
        assert sum(p[1]["products"] for p in parts) == st["products"]
        assert sum(p[2]["instances"] for p in parts) == ks1["instances"] and sum(p[2]["entries"] for p in parts) == ks1["entries"] and sum(p[2]["reliable"] for p in parts) == ks1["reliable"]
        for p in parts:
            assert p[3]["instance_format"].startswith("8-byte" if packed_records else "16-byte")
            assert p[3]["instances"] == p[2]["instances"] * (8 if packed_records else 16)
