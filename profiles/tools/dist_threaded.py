#!/usr/bin/env python3
"""W ranks of the sharded build as W threads on ONE GPU (in-process transport, tests/dist_sim.py): a rehearsal of the multi-GPU step for
rocprofv3 — the cross-rank kernels (k_remote_mirror_slots, k_ingest_remote_slots, k_place_remote) only run with more than one rank.
usage: dist_threaded.py [W=4] [genome divisor=2] [dense]     (config 3 / divisor; `dense`: config 5's shape at 1/(25 * divisor), row blocks)"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np          # noqa: E402
import elba_amd             # noqa: E402
import dist_sim             # noqa: E402
from elba_amd.distributed import DistributedOverlap, HipBackend, partition_by_bases      # noqa: E402
from test_distributed_cpu import _shard      # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
div = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dense = len(sys.argv) > 3 and sys.argv[3] == "dense"
nblocks = int(sys.argv[4]) if len(sys.argv) > 4 else 3
if dense:
    reads = elba_amd.synth_reads(4, 20_000_000 // div, 40.0, 10000.0, 1000.0, error_rate=0.01, min_len=1000, repeat_families=20, repeat_fraction=0.05, repeat_len=5000)
    k, lo, up = 17, 2, 35
else:
    reads = elba_amd.synth_reads(2, 66_700_000 // div, 30.0, 10000.0, 1500.0, error_rate=0.15, min_len=1000)
    k, lo, up = 17, 2, 8
packed, off, lens, _ = reads
print("reads", len(lens), flush=True)
e = elba_amd.Engine(k, lo, up); e.set_reads(packed, off, lens); e.count_kmers(); e.create_kmer_matrix()
e.set_option("overlap_cold_calls", 1)
st = e.create_seed_matrix(); st = e.create_seed_matrix()
B1 = e.export_csr(); e.close()
print("one GPU: nnz %d products %d  cold call %.2f ms (numeric %.2f, finalize %.2f)" % (st["nnz"], st["products"], st["ms_total"], st["ms_numeric"], st["ms_finalize"]), flush=True)
bounds = partition_by_bases(lens, W)


def body(rank, h):
    a, b = int(bounds[rank]), int(bounds[rank + 1])
    sp, so, sl = _shard(packed, off, lens, a, b)
    d = DistributedOverlap(k, lo, up, device=0, rank=rank, world=W, dist=h, backend=HipBackend(k, lo, up, 0))
    d.time_phases = True
    d.set_reads(sp, so, sl, a, bounds)
    t0 = time.time()
    if dense:
        d.build_kmer_matrix(row_batches=nblocks)
        rows, tot, fmt = [], dict(nnz=0, products=0, ms_total=0.0, ms_numeric=0.0, ms_finalize=0.0), 2
        t1 = time.time()
        for t in range(nblocks):
            d.load_row_block(t)
            fmt = min(fmt, d.be.e.device_view()["a_csr_format"])
            s2 = d.create_seed_matrix()
            for key in tot: tot[key] += s2[key]
            rows.append(d.export_csr())
        out = (dist_sim.stitch_rows(rows), tot, t1 - t0, time.time() - t1, dict(dense_format=fmt == 2))
    else:
        d.build_kmer_matrix(); t1 = time.time()
        d.create_seed_matrix()
        s2 = d.create_seed_matrix()
        out = (d.export_csr(), s2, t1 - t0, time.time() - t1, d.phase_ms)
    d.be.e.close()
    return out


parts = dist_sim.run_ranks(W, body)
B = dist_sim.stitch_rows([p[0] for p in parts])
ok = B["Y"] == B1["Y"] and (B["rowptr"] == B1["rowptr"]).all() and (B["col"] == B1["col"].astype(np.int64)).all() and (B["val"] == B1["val"]).all()
print("world", W, "Y", B["Y"], "equal to the one-GPU B:", bool(ok), flush=True)
for r, p in enumerate(parts):
    print("rank %d: nnz %d products %d | library ms: total %.2f numeric %.2f finalize %.2f | phases %s" % (r, p[1]["nnz"], p[1]["products"], p[1]["ms_total"], p[1]["ms_numeric"], p[1]["ms_finalize"], p[4]), flush=True)
assert ok
