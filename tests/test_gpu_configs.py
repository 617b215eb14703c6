"""-m gpu: BASELINE.json configs 3, 4 and 5 (restated per SURVEY.md §8d) on ONE MI355X, through the C ABI.

At these sizes the oracle cannot rerun the whole pipeline in seconds, so every configuration is judged by
  * size-independent properties: Y = diag + 2 * upper, P = sum_k c_k^2, columns ascending, pattern and numshared symmetric, mirrored
    seeds, idempotence of a second (cold) call, every sampled seed a genuine shared k-mer (the reference's test.py:57-65);
  * ORACLE EQUALITY ON SAMPLED ROWS: for a few hundred random rows the entries of A they touch (the rows themselves and every column they
    meet, whole) are fetched from the device, handed to the oracle as triples, and the oracle's rows of B must equal the GPU's bit for
    bit — pattern, numshared and both seeds.  (The sub-matrix contains every column of a sampled row completely, so the oracle's fold
    over it is that row's fold over all of A.)
  config 3  200 100 reads x 10 kb, 66.7 Mb genome, 30x, 15 % error, U = 8: as written, whole, on one GPU
  config 4  C. elegans-HiFi-like, 100 Mb genome, 40x, 0.5 % error, U = 4: as written, whole, on one GPU (266 666 reads of 15 kb, 4.0 G k-mer
            instances: just inside one context's 32-bit instance index, two index bits dropped from the sort words and recovered)
  config 5  20 repeat families, 1 % error, U = 35: 1/25 of the genome (80 k reads, 10.8 G products): the dense / spill stress at a size one GPU holds;
            and at 1/8 — ONE GPU's share of the 8-GPU run, 250 000 reads — whole and through row-block batching (the last test but three)
"""
import os

import numpy as np
import pytest

import elba_amd
from oracle import pyoracle as po
import gpu_util as gu

pytestmark = pytest.mark.gpu

CONFIGS = {
    "config3-200k-long-reads": dict(seed=2, genome=66_700_000, depth=30.0, avg=10000.0, sd=1500.0, err=0.15, min_len=1000, k=17, L=2, U=8, rep=(0, 0.0, 0), nsample=200),
    "config4-celegans-hifi": dict(seed=3, genome=100_000_000, depth=40.0, avg=15000.0, sd=2000.0, err=0.005, min_len=1000, k=17, L=2, U=4, rep=(0, 0.0, 0), nsample=300),
    "config5-dense-repeats-25th": dict(seed=4, genome=20_000_000, depth=40.0, avg=10000.0, sd=1000.0, err=0.01, min_len=1000, k=17, L=2, U=35, rep=(20, 0.05, 5000), nsample=60),
}


class _DevArray:
    """A device pointer of the library as something torch can wrap without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _device_tensors(e):
    import torch
    v = e.device_view()
    t = lambda ptr, n, ts: torch.as_tensor(_DevArray(ptr, n, ts), device="cuda")      # noqa: E731
    return dict(M=v["M"], N=v["N"], Z=v["Z"],
                rowptr=t(v["a_rowptr"], v["M"] + 1, "<i4"), csr=t(v["a_csr"], v["Z"], "<i8"),
                colptr=t(v["a_colptr"], v["N"] + 1, "<i4"), csc=t(v["a_csc"], v["Z"], "<i8"))


def _rows_from_columns(d, rows):
    """(k-mer id, pos) of every entry of the given rows, from the COLUMN side of A (a_colptr / a_csc), as {row: (kids, pos)} in (k-mer id, pos)
    order.  The row side may hold inline partners (ELBA_CSR_INLINE: an entry that carries its column's other read instead of the k-mer id)."""
    import torch
    dev = d["csc"].device
    rt = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    csc = d["csc"]
    hit = torch.zeros(d["M"], dtype=torch.bool, device=dev); hit[rt] = True
    z = torch.nonzero(hit[(csc >> 32) & 0xFFFFFFFF]).reshape(-1)
    cp = (d["colptr"].to(torch.int64) & 0xFFFFFFFF)
    kid = torch.searchsorted(cp, z, right=True) - 1
    ent = csc[z]
    r = ((ent >> 32) & 0xFFFFFFFF).cpu().numpy(); pos = (ent & 0xFFFFFFFF).cpu().numpy(); kid = kid.cpu().numpy()
    out = {}
    order = np.lexsort((pos, kid, r))
    r, pos, kid = r[order], pos[order], kid[order]
    cuts = np.searchsorted(r, np.asarray(rows))
    ends = np.searchsorted(r, np.asarray(rows), side="right")
    for row, a, b in zip(rows, cuts, ends):
        out[int(row)] = (kid[a:b].astype(np.int64), pos[a:b].astype(np.int64))
    return out


def _check_inline_entries(d, v, row, ent, kids, pos):
    """An entry with bit 63 set (ELBA_CSR_INLINE) must be an entry whose row accumulates exactly ONE pair of its column (smaller read when the ids'
    sum is even, else the larger) and occurs in it once, and carry exactly that partner entry; every other entry is kid << 32 | hint << 30 | pos."""
    import torch
    cp = (d["colptr"].to(torch.int64) & 0xFFFFFFFF)
    inl = (ent >> 63) & 1
    hi = (ent >> 32) & 0x7FFFFFFF
    lo_ = ent & 0xFFFFFFFF
    plain = inl == 0
    ids = hi.copy()
    if v["a_gather_slots"]:             # gather slots: an entry that fetches its column (no inline partner, hint == 0) names the column's slot; the slot names the k-mer
        gather = plain & (((lo_ >> 30) & 3) == 0)
        sk = torch.as_tensor(_DevArray(v["a_slot_kid"], v["a_gather_slots"], "<i4"), device="cuda")
        assert (hi[gather] < v["a_gather_slots"]).all(), row
        ids[gather] = (sk[torch.from_numpy(hi[gather].astype(np.int64)).to(sk.device)].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    assert (ids[plain] == kids[plain]).all() and ((lo_[plain] & int(v["a_csr_pos_mask"])) == pos[plain]).all(), row
    if inl.any():
        ks = torch.from_numpy(kids[inl == 1]).to(cp.device)
        c0, c1 = cp[ks].cpu().numpy(), cp[ks + 1].cpu().numpy()
        jh = hi[inl == 1]
        j = np.where(jh > (row >> 1), 2 * jh + (row & 1), np.where(jh < (row >> 1), 2 * jh + ((row & 1) ^ 1), row ^ 1))
        posq, post = lo_[inl == 1] & 0xFFFF, lo_[inl == 1] >> 16
        for a, b, jj, pq, pt in zip(c0, c1, j, posq, post):
            col = d["csc"][int(a):int(b)].cpu().numpy()
            r, p_ = (col >> 32) & 0xFFFFFFFF, col & 0xFFFFFFFF
            mine = r == row
            owned = (~mine) & np.where(((row ^ r) & 1) == 1, r < row, r > row)      # owns_pair: the larger row when the ids' sum is odd, else the smaller
            assert mine.sum() == 1 and owned.sum() == 1, (row, r)
            assert r[owned][0] == jj and p_[owned][0] == pt and p_[mine][0] == pq, (row, r, jj)
    return int(inl.sum())


def _sampled_rows_equal_oracle(e, B, k, lo, up, nsample, seed=0, d=None):
    """See the module docstring.  Returns the number of rows compared."""
    import torch
    d = d or _device_tensors(e)
    dev = d["csr"].device
    M = d["M"]
    rp = d["rowptr"].to(torch.int64) & 0xFFFFFFFF
    nnz = (rp[1:] - rp[:-1]).cpu().numpy()
    rng = np.random.default_rng(seed)
    cand = np.nonzero(nnz > 0)[0]
    rows = np.sort(rng.choice(cand, size=min(nsample, len(cand)), replace=False))
    rpc = rp.cpu().numpy()
    if e.device_view()["a_csr_format"] == 3:                            # inline partners: the columns a row meets are named by the column side
        rc = _rows_from_columns(d, rows)
        kids = torch.unique(torch.from_numpy(np.concatenate([rc[int(r)][0] for r in rows])).to(dev))
    else:
        idx = np.concatenate([np.arange(rpc[r], rpc[r + 1]) for r in rows])
        ent = d["csr"][torch.from_numpy(idx).to(dev)]
        kids = torch.unique((ent >> 32) & 0xFFFFFFFF)                   # every column a sampled row meets, ascending
    cp = d["colptr"].to(torch.int64) & 0xFFFFFFFF
    c0, c1 = cp[kids], cp[kids + 1]
    lens_ = (c1 - c0)
    tot = int(lens_.sum().item())
    starts = torch.cumsum(lens_, 0) - lens_
    which = torch.repeat_interleave(torch.arange(len(kids), device=dev), lens_)
    within = torch.arange(tot, device=dev) - starts[which]
    cent = d["csc"][c0[which] + within]
    t_rows = ((cent >> 32) & 0xFFFFFFFF).cpu().numpy().astype(np.int64)
    t_pos = (cent & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)
    t_cols = which.cpu().numpy().astype(np.int64)                     # relabelled 0..len(kids)-1 in ascending k-mer id: the canonical order is kept
    o = po.Oracle(k, lo, up)
    o.set_triples(M, len(kids), t_rows, t_cols, t_pos)
    o.spgemm(8)
    oB = o.B()
    for r in rows:
        g0, g1 = int(B["rowptr"][r]), int(B["rowptr"][r + 1])
        w0, w1 = int(oB["rowptr"][r]), int(oB["rowptr"][r + 1])
        assert g1 - g0 == w1 - w0, (r, g1 - g0, w1 - w0)
        assert (B["col"][g0:g1] == oB["col"][w0:w1].astype(np.int64)).all(), r
        assert (B["val"][g0:g1] == oB["val"][w0:w1]).all(), r
    return len(rows)


def _A_equals_oracle_on_value_classes(e, packed, off, lens, k, lo, up, nclasses=64, nrows=200, seed=0):
    """A itself at full size (the k-mer stage: 2-4 G instances, the two-level partition, LDS count tables, LDS sorts).  (1) 64 of the 4096
    value classes (k-mer value modulo 4096): the oracle's enumerator (orc_read_kmers' loop) walks ALL reads on the host cores and keeps the
    instances of those classes; counting them exactly gives the reliable k-mers of the classes and their (read, pos) lists — the GPU's columns
    for exactly those k-mers must be the same: the same k-mers (none missing, none of the unreliable ones present), the same entries in the
    same order.  (2) 200 rows of CSR(A) re-derived from their reads: every position's canonical k-mer looked up among the GPU's reliable
    k-mers gives the row's (k-mer id, pos) list."""
    import torch
    rng = np.random.default_rng(seed)
    classes = np.sort(rng.choice(4096, size=nclasses, replace=False))
    vals, rd, ps = po.enumerate_classes(packed, off, lens, k, classes, nthreads=16)
    order = np.lexsort((ps, rd, vals))
    vals, rd, ps = vals[order], rd[order], ps[order]
    head = np.concatenate([[True], vals[1:] != vals[:-1]])
    starts = np.nonzero(head)[0]
    counts = np.diff(np.concatenate([starts, [len(vals)]]))
    rel = (counts >= lo) & (counts <= up)
    want_kmers = vals[starts[rel]]
    keep = np.repeat(rel, counts)
    want_read, want_pos = rd[keep].astype(np.int64), ps[keep].astype(np.int64)
    want_len = counts[rel]
    # the GPU's side: the columns of the same classes
    v = e.device_view()
    d = _device_tensors(e)
    dev = d["csc"].device
    kmers = torch.as_tensor(_DevArray(v["a_kmers"], v["N"], "<i8"), device="cuda")
    gvals = (kmers >> (64 - 2 * k)) & ((1 << (2 * k)) - 1)
    cls = torch.zeros(4096, dtype=torch.bool, device=dev); cls[torch.from_numpy(classes).to(dev)] = True
    kids = torch.nonzero(cls[gvals & 4095]).reshape(-1)
    got_kmers = gvals[kids].cpu().numpy().astype(np.uint64)
    assert len(got_kmers) == len(want_kmers) and (got_kmers == want_kmers).all(), (len(got_kmers), len(want_kmers))
    cp = d["colptr"].to(torch.int64) & 0xFFFFFFFF
    c0, c1 = cp[kids], cp[kids + 1]
    glen = (c1 - c0)
    assert (glen.cpu().numpy() == want_len).all()
    tot = int(glen.sum().item())
    st0 = torch.cumsum(glen, 0) - glen
    which = torch.repeat_interleave(torch.arange(len(kids), device=dev), glen)
    cent = d["csc"][c0[which] + (torch.arange(tot, device=dev) - st0[which])]
    assert ((((cent >> 32) & 0xFFFFFFFF).cpu().numpy() == want_read).all() and ((cent & 0xFFFFFFFF).cpu().numpy() == want_pos).all())
    # rows of CSR(A) from their reads
    allk = gvals.cpu().numpy().astype(np.uint64)                      # ascending: k-mer id = rank of the value
    rp = (d["rowptr"].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    L = po.lib()
    pmask = int(v["a_csr_pos_mask"])
    rows = np.sort(rng.choice(len(lens), size=min(nrows, len(lens)), replace=False))
    ninline = 0
    buf = np.zeros(int(lens.max()) + 8, dtype=np.uint64)
    for r in rows:
        n = L.orc_read_kmers(packed.ctypes.data + int(off[r]), int(lens[r]), k, buf.ctypes.data)
        kv = buf[:n] >> np.uint64(64 - 2 * k)
        at = np.searchsorted(allk, kv)
        hit = (at < len(allk)) & (allk[np.minimum(at, len(allk) - 1)] == kv)
        exp_kid, exp_pos = at[hit].astype(np.int64), np.nonzero(hit)[0].astype(np.int64)
        o2 = np.lexsort((exp_pos, exp_kid))
        ent = d["csr"][int(rp[r]):int(rp[r + 1])].cpu().numpy()
        assert len(ent) == len(exp_kid), (r, len(ent), len(exp_kid))
        if v["a_csr_format"] == 3:
            ninline += _check_inline_entries(d, v, int(r), ent, exp_kid[o2], exp_pos[o2])
        else:
            assert (((ent >> 32) & 0xFFFFFFFF) == exp_kid[o2]).all() and ((ent & pmask) == exp_pos[o2]).all(), r
    if v["a_csr_format"] == 3:
        assert ninline > 0
    return len(want_kmers), len(rows)


def _properties(e, B, st, ks, packed, off, lens, k, upper):
    assert st["nnz"] == B["Y"] == int(B["rowptr"][-1])
    assert st["nnz"] == st["nnz_diag"] + 2 * st["nnz_upper"]
    h = e.kmer_histogram(upper + 2)
    assert int((h * np.arange(len(h)) ** 2).sum()) == st["products"] and int(h.sum()) == ks["reliable"] and int((h * np.arange(len(h))).sum()) == ks["entries"]
    M = B["M"]
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(B["rowptr"])); cols = B["col"]
    assert ((np.diff(cols) > 0) | (np.diff(rows) > 0)).all()                      # columns strictly ascending in every row
    key = rows * M + cols; tkey = cols * M + rows
    order = np.argsort(tkey, kind="stable")
    assert (tkey[order] == key).all()                                             # pattern symmetric
    v = B["val"]; vt = v[order]
    assert (v["numshared"] == vt["numshared"]).all() and (v["numshared"] >= 2).all()
    assert (v["q0"] == vt["t0"]).all() and (v["t0"] == vt["q0"]).all() and (v["q1"] == vt["t1"]).all() and (v["t1"] == vt["q1"]).all()      # B(j,i) = B(i,j) with positions exchanged
    L = po.lib(); rng = np.random.default_rng(0); bad = 0
    for x in rng.choice(B["Y"], size=min(5000, B["Y"]), replace=False):
        i, j, s = int(rows[x]), int(cols[x]), v[x]
        for (q, t) in ((s["q0"], s["t0"]), (s["q1"], s["t1"])):
            bad += not L.orc_seed_is_valid(packed.ctypes.data + int(off[i]), int(lens[i]), packed.ctypes.data + int(off[j]), int(lens[j]), int(q), int(t), k)
    assert bad == 0


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_baseline_config_properties_and_sampled_rows(name):
    w = CONFIGS[name]
    packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg"], w["sd"], error_rate=w["err"], min_len=w["min_len"],
                                                   repeat_families=w["rep"][0], repeat_fraction=w["rep"][1], repeat_len=w["rep"][2])
    k = w["k"]
    e = elba_amd.Engine(k, w["L"], w["U"])
    e.set_reads(packed, off, lens)
    ks = e.count_kmers(); ms = e.create_kmer_matrix()
    assert ks["instances"] == int(np.maximum(lens.astype(np.int64) - k + 1, 0).sum()) and ms["nnz"] == ks["entries"] and ms["ncols"] == ks["reliable"]
    st = e.create_seed_matrix()
    B = e.export_csr()
    _properties(e, B, st, ks, packed, off, lens, k, w["U"])
    nk, nr = _A_equals_oracle_on_value_classes(e, packed, off, lens, k, w["L"], w["U"])
    assert nk > 1000 and nr >= 100
    assert _sampled_rows_equal_oracle(e, B, k, w["L"], w["U"], w["nsample"]) >= min(50, w["nsample"])
    # ... and EVERY entry of B against the oracle's product of the same A (round 5: what bench.py does for the headline matrix — 98.7 M entries on
    # config 3, 10.8 G products on config 5 at 1/25 — as a test; the oracle's left fold on every host core the box gives this process)
    assert gu.assert_whole_B_equals_oracle(e, k, w["L"], w["U"], st) == B["Y"]
    # a second call, cold (nothing remembered), returns the identical matrix
    e.set_option("overlap_cold_calls", 1)
    st2 = e.create_seed_matrix()
    assert all(st2[f] == st[f] for f in ("nnz", "products", "nnz_before_prune", "nnz_diag", "nnz_upper", "max_numshared"))
    B2 = e.export_csr(0, min(B["M"], 20000))
    n2 = int(B2["rowptr"][-1])
    assert (B2["rowptr"] == B["rowptr"][:len(B2["rowptr"])]).all() and (B2["col"] == B["col"][:n2]).all() and (B2["val"] == B["val"][:n2]).all()
    e.close()


def test_config5_one_gpus_share_whole_and_through_row_blocks():
    """BASELINE configs[4] (2 M reads, 20 repeat families, U = 35, 8 GPUs) at ONE GPU's share of it: an eighth of the genome, 250 000 reads,
    2.5 G k-mer instances, nnz(A) ~ 1.1 G — the size row-block batching and the dense path were built for (VERDICT r3 task 4).
      (1) whole, in one context: properties, A against the oracle on value classes, sampled rows of B against the oracle;
      (2) the way a rank of the 8-GPU run walks its rows (elba_amd/distributed.py: build_kmer_matrix(row_batches=b) -> load_row_block(t) ->
          create_seed_matrix per block; the panel exchange and the owner-side count run with world = 1): the union of the blocks' rows
          equals the whole matrix bit for bit."""
    import dist_sim
    from elba_amd.distributed import DistributedOverlap, HipBackend
    w = dict(seed=4, genome=62_500_000, depth=40.0, avg=10000.0, sd=1000.0, err=0.01, min_len=1000, k=17, L=2, U=35, rep=(20, 0.05, 5000))
    packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg"], w["sd"], error_rate=w["err"], min_len=w["min_len"],
                                                   repeat_families=w["rep"][0], repeat_fraction=w["rep"][1], repeat_len=w["rep"][2])
    assert 240_000 <= len(lens) <= 260_000
    k = w["k"]
    e = elba_amd.Engine(k, w["L"], w["U"])
    e.set_reads(packed, off, lens)
    ks = e.count_kmers(); ms = e.create_kmer_matrix()
    assert ks["instances"] == int(np.maximum(lens.astype(np.int64) - k + 1, 0).sum()) and ms["nnz"] == ks["entries"] and ms["ncols"] == ks["reliable"]
    assert e.device_view()["a_csr_format"] == 2                       # ELBA_CSR_DENSE: the dense path
    st = e.create_seed_matrix()
    B = e.export_csr()
    _properties(e, B, st, ks, packed, off, lens, k, w["U"])
    nk, nr = _A_equals_oracle_on_value_classes(e, packed, off, lens, k, w["L"], w["U"], nclasses=16, nrows=100)
    assert nk > 1000 and nr >= 100
    assert _sampled_rows_equal_oracle(e, B, k, w["L"], w["U"], 40) >= 40
    whole = dict(products=st["products"], nnz=st["nnz"])
    e.close()
    nblocks = 4
    bounds = np.array([0, len(lens)], dtype=np.int64)

    def body(rank, h):
        d = DistributedOverlap(k, w["L"], w["U"], device=0, rank=0, world=1, dist=h, backend=HipBackend(k, w["L"], w["U"], 0))
        d.set_reads(packed, off, lens, 0, bounds)
        d.build_kmer_matrix(row_batches=nblocks)
        rows, fmt, nnz = [], 2, 0
        for t in range(nblocks):
            d.load_row_block(t)
            fmt = min(fmt, d.be.e.device_view()["a_csr_format"])
            s2 = d.create_seed_matrix()
            nnz += s2["nnz"]
            rows.append(d.export_csr())
        d.be.e.close()
        return dist_sim.stitch_rows(rows), fmt, nnz

    Bb, fmt, nnz = dist_sim.run_ranks(1, body)[0]
    assert fmt == 2 and nnz == whole["nnz"]                           # every row block took the dense path
    assert Bb["Y"] == B["Y"] and (Bb["rowptr"] == B["rowptr"]).all() and (Bb["col"] == B["col"]).all()
    bad = np.nonzero(Bb["val"] != B["val"])[0]
    assert len(bad) == 0, (len(bad), Bb["val"][bad[:4]], B["val"][bad[:4]])


def test_distributed_driver_over_rccl_world_of_one():
    """elba_amd/distributed.py end to end on the real collectives (torch.distributed backend "nccl" = RCCL), world size 1 on this box:
    both all-to-alls, the all-gather, the panel, the row window — B must equal the oracle's."""
    import torch
    import torch.distributed as dist
    from elba_amd.distributed import DistributedOverlap
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        reads = elba_amd.synth_reads(33, 400000, 15, 4000, 900, error_rate=0.10, min_len=200)
        packed, off, lens, _ = reads
        d = DistributedOverlap(17, 2, 8, device=0, rank=0, world=1, dist=dist)
        d.exchange_chunks = 3            # (exchange #1 in three rounds over RCCL itself: each round's all-to-all is posted with async_op before the round before it is unpacked)
        d.set_reads(packed, off, lens, 0, np.array([0, len(lens)], dtype=np.int64))
        ks, ms = d.build_kmer_matrix()
        assert d.exchange_rounds == 3 and d.exchange_format.startswith("8-byte")
        st = d.create_seed_matrix()
        B = d.export_csr()
        o = po.Oracle(17, 2, 8); o.count_and_build(packed, off, lens); o.spgemm(4)
        oB = o.B()
        assert B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"].astype(np.int64)).all() and (B["val"] == oB["val"]).all()
        assert (ks["reliable"], ks["entries"], st["products"], st["nnz_before_prune"]) == (o.stat("N"), o.stat("Z"), o.stat("P"), o.stat("Yraw"))
        al = d.align_seeds()
        rows, cols, ov, _ = o.align_upper(packed, off, lens)
        g = d.export_overlaps()
        assert al["nalignments"] == len(rows) and (g["rows"] == rows).all() and (g["cols"] == cols).all()
        assert all((g["vals"][f] == ov[f]).all() for f in ov.dtype.names if f != "pad")
        d.be.e.close()
    finally:
        dist.destroy_process_group()


def test_text_formats_from_gpu_output(tmp_path):
    """SURVEY.md §8f-4 driven by the GPU's own B and overlaps: B.mtx in the reference's SharedSeeds notation, the dump its test.py reads
    (and test.py's seed check on it), PAF lines — each compared with the same writer fed from the oracle."""
    import util
    from elba_amd import formats as fm
    seqs = util.read_fasta(os.path.join(util.GOLDEN, "small_err.fa"))
    packed, off, lens = po.pack_reads(seqs)
    e = elba_amd.Engine(17, 2, 8)
    e.set_reads(packed, off, lens); e.count_kmers(); e.create_kmer_matrix(); e.create_seed_matrix()
    o = po.Oracle(17, 2, 8); o.count_and_build(packed, off, lens); o.spgemm(1)
    B, oB = e.export_csr(), o.B()
    M = B["M"]
    pg, pw = str(tmp_path / "g.mtx"), str(tmp_path / "o.mtx")
    fm.write_seed_matrix_mm(pg, e.export_dcsc(0, M, 0, M), M)
    fm.write_seed_matrix_mm(pw, o.export_dcsc(0, M, 0, M), M)
    assert open(pg).read() == open(pw).read() and os.path.getsize(pg) > 1000
    dg, dw = str(tmp_path / "g.dump"), str(tmp_path / "o.dump")
    fm.write_testpy_dump(dg, dict(M=M, Y=B["Y"], rowptr=B["rowptr"], col=B["col"], val=B["val"]))
    fm.write_testpy_dump(dw, oB)
    assert open(dg).read() == open(dw).read()
    correct, incorrect = fm.check_seed_dump(dg, [s.upper() for s in seqs], 17)
    assert incorrect == 0 and correct > 0
    e.align_seeds()
    g = e.export_overlaps()
    rows, cols, ov, _ = o.align_upper(packed, off, lens)
    names = ["read%d" % r for r in range(M)]
    fg, fw = str(tmp_path / "g.paf"), str(tmp_path / "o.paf")
    fm.write_paf(fg, g, names, lens)
    fm.write_paf(fw, dict(n=len(rows), rows=rows, cols=cols, vals=ov), names, lens)
    assert open(fg).read() == open(fw).read() and g["n"] > 0
    e.close()


def test_new_reads_invalidate_results_built_from_the_old_ones():
    """after elba_set_reads the stages downstream of the OLD reads must refuse to run (ELBA_ERR_STATE = 5), not answer from stale state."""
    a = elba_amd.synth_reads(71, 60000, 10, 3000, 300, error_rate=0.05)
    b = elba_amd.synth_reads(72, 60000, 10, 3000, 300, error_rate=0.05)
    e = elba_amd.Engine(17, 2, 8)
    e.set_reads(*a[:3]); e.count_kmers(); e.create_kmer_matrix(); e.create_seed_matrix(); e.align_seeds()
    n = min(len(a[2]), len(b[2]))
    e.set_reads(b[0], b[1][:n], b[2][:n])
    for call in (e.align_seeds, e.export_csr, e.create_seed_matrix, e.create_kmer_matrix, e.export_overlaps):
        with pytest.raises(elba_amd.ElbaError) as ei:
            call()
        assert ei.value.status == 5, call
    e.count_kmers(); e.create_kmer_matrix(); st = e.create_seed_matrix()
    o = po.Oracle(17, 2, 8); o.count_and_build(b[0], b[1][:n], b[2][:n]); o.spgemm(2)
    assert st["nnz"] == o.stat("Y")
    e.close()
