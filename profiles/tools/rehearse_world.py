#!/usr/bin/env python3
"""REHEARSAL of the sharded build at any world size on ONE GPU: the ranks run as threads of this process (tests/dist_sim.py: the collectives are the
in-process ThreadedGroup; a GPU box admits at most six processes on its card, so eight ranks cannot be processes), every rank drives the real HIP
building blocks through elba_amd/distributed.py.  NOT a scaling measurement — what it checks is that N participants reproduce the one-GPU counts
(k-mer instances, nnz(A), products, nnz(B)) and what they would put on the wire.
usage: python3 profiles/tools/rehearse_world.py WORLD WORKLOAD [GENOME_DIV] > profiles/r05_rehearsal_worldN_<workload>.json"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench, dist_sim, elba_amd
from elba_amd.distributed import DistributedOverlap, HipBackend

world, name = int(sys.argv[1]), sys.argv[2]
div = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w = dict(bench.WORKLOADS[name]); w["genome"] = w["genome"] // div
k, lo, up = w["k"], w["lower"], w["upper"]
# the one-GPU build of the same read set: what every world must reproduce
rep = w.get("repeats", (0, 0.0, 0))
packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                               repeat_families=rep[0], repeat_fraction=rep[1], repeat_len=rep[2])
e = elba_amd.Engine(k, lo, up); e.set_reads(packed, off, lens)
ks = e.count_kmers(); ms = e.create_kmer_matrix(); st = e.create_seed_matrix()
one = (int(ks["instances"]), int(ms["nnz"]), int(st["products"]), int(st["nnz"]))
e.close(); del packed, off, lens
torch.cuda.empty_cache()
if div == 1 and name in bench.EXPECTED_COUNTS:
    assert one == tuple(bench.EXPECTED_COUNTS[name]), (one, bench.EXPECTED_COUNTS[name])

def body(rank, h):
    d = DistributedOverlap(k, lo, up, device=0, rank=rank, world=world, dist=h, backend=HipBackend(k, lo, up, 0))
    d.generate_and_set_reads(w, weak=False)
    t0 = time.perf_counter(); ks2, ms2 = d.build_kmer_matrix(); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); s2 = d.create_seed_matrix(); t_step = time.perf_counter() - t0
    out = dict(instances=int(ks2["instances"]), entries=int(ks2["entries"]), products=int(s2["products"]), nnz=int(s2["nnz"]), exchange=dict(d.exchange_bytes),
               panel_records=int(d.panel_records), build_s=round(t_build, 3), step_s=round(t_step, 3), exchange_rounds=int(d.exchange_rounds))
    d.be.e.close()
    return out

parts = dist_sim.run_ranks(world, body)
got = (sum(p["instances"] for p in parts), sum(p["entries"] for p in parts), sum(p["products"] for p in parts), sum(p["nnz"] for p in parts))
print(json.dumps({"rehearsal": "%d ranks as THREADS of one process on one MI355X (in-process collectives): not a scaling measurement" % world,
                  "workload": name, "genome_div": div, "world": world, "counts_one_gpu": list(one), "counts": list(got), "counts_match_one_gpu": bool(got == one),
                  "exchange_bytes_per_rank": {"instances": [p["exchange"]["instances"] for p in parts], "format": parts[0]["exchange"].get("instance_format"),
                                              "instances_as_16_byte_records": [p["instances"] * 16 for p in parts], "panels": [p["exchange"]["panels"] for p in parts]},
                  "per_rank": parts}))
sys.exit(0 if got == one else 3)
