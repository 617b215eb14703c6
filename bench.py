#!/usr/bin/env python3
"""bench.py — overlap-matrix build throughput (candidate-overlap nnz/s of B = A·Aᵀ, SharedSeeds semiring) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one process per
GPU with torch.distributed.run.  One JSON line on rank 0.

A "step" is one pass of the hot path's headline region over the synthetic input: device-resident A (CSR + CSC) -> device-resident
pruned CSR B (reference timer "creating seed matrix (spgemm)", src/main.cpp:280-282), all of it: LDS-hash numeric kernels (they gather
every partner entry, accumulate every pair and count the diagonal: A's device format holds per-entry and per-row arrays only, nothing
per product and nothing of B), row-pointer scan, mirror pass, per-row column sort, and the host synchronisation the C ABI performs.
For N > 1 every rank computes its own rows of B from its rows of A and the column panel it received while A was built (RCCL
all-to-all, reference timers "creating k-mer matrix" / "copying and transposing"): the step has no data-path collective.
Inputs are resident in HBM when the timed region starts.  The k-mer stage that builds A on the GPU is run (and reported) before the
timed region; `cold_call_ms` is one call on a freshly rebuilt A (no tier prior, no cached queues, output capacity unknown).

Workload at N = 1: BASELINE.json configs[1] restated per SURVEY.md §8d-2 ("ecsample30x-like": 16 890 reads, 4.64 Mb genome, 30x,
len N(8240, 2000) >= 1000, 15 % sub/ins/del error, k=17, L=2, U=8, seed 1).  At N > 1 the same per-GPU read count is kept and the
genome grows with N (weak scaling): reads are sharded by contiguous row blocks, k-mer columns by hash owner.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (reads per GPU, genome bases per GPU, depth, avg_len, sd_len, min_len, error, k, L, U, seed)
    "ecsample30x-like": dict(genome=4_640_000, depth=30.0, avg_len=8240.0, sd_len=2000.0, min_len=1000, error=0.15, k=17, lower=2, upper=8, seed=1),
    # BASELINE.json configs[2] restated per SURVEY.md §8d-3 (the whole set on ONE GPU when run with --gpus 1; parity-test/scale case, not the bench line)
    "200k-long-reads": dict(genome=66_700_000, depth=30.0, avg_len=10000.0, sd_len=1500.0, min_len=1000, error=0.15, k=17, lower=2, upper=8, seed=2),
    "plumbing-135": dict(genome=100_000, depth=13.5, avg_len=10000.0, sd_len=1000.0, min_len=1000, error=0.0, k=17, lower=2, upper=8, seed=313),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ecsample30x-like", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timing-stride", type=int, default=4, help="the library records its phase events (kernel_ms of the roofline) on every N-th step only: "
                    "an event record costs ~5 us of stream time; kernel_ms is the mean over the steps that were measured")
    ap.add_argument("--no-align", action="store_true", help="skip the ingest and x-drop alignment stages that run once after the timed region (N = 1)")
    ap.add_argument("--align-sharded", action="store_true", help="N > 1: also run the sharded alignment stage (reads replicated by one all-gather)")
    ap.add_argument("--dbg", type=int, default=0, help="diagnostic kernel ablations (results are wrong; never for reporting)")
    args = ap.parse_args()

    import torch
    import elba_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = bool(os.environ.get("ELBA_FORCE_DIST"))      # exercise the multi-GPU driver with world_size 1 (self-test on a 1-GPU box)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w = WORKLOADS[args.workload]
    k, lo, up = w["k"], w["lower"], w["upper"]

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1 and not force_dist:
        from elba_amd.capi import Engine
        t0 = time.time()
        packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"])
        t_gen = time.time() - t0
        eng = Engine(k, lo, up, device=local_rank, flags=args.dbg, timing_stride=args.timing_stride)
        # inputs resident in HBM before anything is timed
        d_packed = torch.from_numpy(packed).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda(); d_len = torch.from_numpy(lens.view(np.int32)).cuda()
        eng.set_reads_device(d_packed.data_ptr(), int(packed.size) - 16, d_off.data_ptr(), d_len.data_ptr(), len(lens))
        torch.cuda.synchronize()
        t0 = time.time(); ks = eng.count_kmers(); ms = eng.create_kmer_matrix(); torch.cuda.synchronize(); t_kmer_wall = time.time() - t0
        runner = eng
        step = eng.create_seed_matrix
        extra_cfg = {}
    else:
        from elba_amd.distributed import DistributedOverlap
        runner = DistributedOverlap(k, lo, up, device=local_rank, rank=rank, world=world, dist=dist, timing_stride=args.timing_stride)
        t0 = time.time()
        info = runner.generate_and_set_reads(w, weak=True)
        t_gen = time.time() - t0
        t0 = time.time(); ks, ms = runner.build_kmer_matrix(); torch.cuda.synchronize(); t_kmer_wall = time.time() - t0
        step = runner.create_seed_matrix
        extra_cfg = {"partition": "1D read rows x hash-owned k-mer columns", "exchange": "RCCL all_to_all_single (column panels)"}

    for _ in range(args.warmup):
        st = step()
    barrier_sync()
    t0 = time.perf_counter()
    acc = dict(ms_total=0.0, ms_numeric=0.0, ms_symbolic=0.0, ms_finalize=0.0)
    ntimed = 0
    for _ in range(args.steps):
        st = step()
        if st.get("timed", 1):
            ntimed += 1
            for key in acc:
                acc[key] += st[key]
    barrier_sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([st["nnz"], st["products"], st["algorithmic_bytes"], ks["instances"], ms["nnz"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        Y, P, abytes, I_tot, Z_tot = [int(x) for x in tot.tolist()]
    else:
        Y, P, abytes, I_tot, Z_tot = st["nnz"], st["products"], st["algorithmic_bytes"], ks["instances"], ms["nnz"]
    steps = max(1, args.steps)
    ms_step = dt / steps * 1e3
    for key in acc:
        acc[key] /= max(1, ntimed)

    # roofline of the dominant kernel (k_spgemm_rows, all table tiers: they jointly process every row once per step).
    # achieved = algorithmic bytes of this rank's rows / HIP-event duration of those launches on the library's stream.
    peak_gbs = 8000.0
    my_bytes = st["algorithmic_bytes"]
    achieved = my_bytes / (acc["ms_numeric"] * 1e-3) / 1e9 if acc["ms_numeric"] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == args.workload and tj.get("n_gpus", 1) == world:
                traffic = tj.get("hbm_bytes_per_step_dominant_kernel")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_spgemm_rows (LDS-hash numeric, all tiers)", "achieved": round(achieved, 3), "peak": peak_gbs, "unit": "GB/s",
                "frac": round(achieved / peak_gbs, 6), "traffic": traffic,
                "algorithmic_bytes_per_step": my_bytes, "bytes_per_nnz": round(my_bytes / max(1, st["nnz"]), 2),
                "kernel_ms": round(acc["ms_numeric"], 4), "kernel_ms_measured_on_steps": ntimed, "region_ms_device": round(acc["ms_total"], 4),
                "frac_whole_region": round(my_bytes / (acc["ms_total"] * 1e-3) / 1e9 / peak_gbs, 6) if acc["ms_total"] > 0 else 0.0,
                "expanded_stream_bytes": 8 * st["products"] + 8 * ms["nnz"] + 24 * st["nnz"]}

    # the resource that actually binds (profiles/r01_notes.md): L2 requests of the numeric kernels (PMC, collected offline like `traffic`) over
    # the kernel time measured live, against the random-gather request ceiling measured on this machine class (profiles/microbench/gather.hip)
    rpath = os.path.join(ROOT, "profiles", "requests.json")
    if os.path.exists(rpath) and acc["ms_numeric"] > 0:
        try:
            rj = json.load(open(rpath))
            if rj.get("workload") == args.workload and rj.get("n_gpus", 1) == world:
                rate = rj["tcc_requests_per_step"] / (acc["ms_numeric"] * 1e-3) / 1e9
                roofline["l2_requests"] = {"per_step": rj["tcc_requests_per_step"], "achieved_G_per_s": round(rate, 2), "measured_ceiling_G_per_s": rj["ceiling_G_requests_per_s"],
                                           "frac_of_ceiling": round(rate / max(rj["ceiling_G_requests_per_s"]), 4)}
        except Exception:
            pass

    # one COLD call: A rebuilt (which forgets the tier prior, the cached queues and the output capacity measured by earlier calls), then
    # a single elba_create_seed_matrix — what a caller that multiplies each matrix once pays (buffers stay allocated)
    cold_ms = None
    if world == 1 and not force_dist and not args.dbg:
        ms2 = eng.create_kmer_matrix(); torch.cuda.synchronize()
        t0 = time.perf_counter(); st_cold = eng.create_seed_matrix(); torch.cuda.synchronize(); cold_ms = (time.perf_counter() - t0) * 1e3
        assert st_cold["nnz"] == st["nnz"]

    # the step before the path: the same reads as FASTA text (one line per record), encoded on the GPU by a second context
    ingest = None
    aux_errors = {}

    def guarded(name, fn):                     # the auxiliary stages never take the headline line down with them
        try:
            return fn()
        except Exception as ex:               # noqa: BLE001
            aux_errors[name] = "%s: %s" % (type(ex).__name__, ex)
            return None

    def run_ingest():
        from elba_amd import fasta as efa
        letters = np.frombuffer(b"ACGT", dtype=np.uint8)
        nb = (lens.astype(np.int64) + 3) // 4
        recs = np.zeros(len(lens), dtype=efa.FAI_DTYPE)
        parts = []; at = 0
        for r in range(len(lens)):
            by = packed[int(off[r]):int(off[r]) + int(nb[r])]
            codes = np.stack([(by >> 6) & 3, (by >> 4) & 3, (by >> 2) & 3, by & 3], axis=1).reshape(-1)[:int(lens[r])]
            hdr = b">%d\n" % r
            parts.append(hdr); at += len(hdr)
            recs[r] = (int(lens[r]), at, int(lens[r]))
            parts.append(letters[codes].tobytes()); parts.append(b"\n"); at += int(lens[r]) + 1
        chunk = b"".join(parts)
        e2 = Engine(k, lo, up, device=local_rank)
        e2.set_reads_fasta(chunk, 0, recs)                     # warm-up (allocations)
        ist = e2.set_reads_fasta(chunk, 0, recs)
        gp, goff, glen = e2.export_reads(len(lens), ist["packed_bytes"])
        same = bool((gp[:ist["packed_bytes"]] == packed[:ist["packed_bytes"]]).all() and (glen == lens).all())
        io_bytes = ist["chunk_bytes"] + ist["packed_bytes"]
        res = {"fasta_bytes": int(ist["chunk_bytes"]), "bases": int(ist["bases"]), "ms_total_with_h2d": round(ist["ms_total"], 3), "ms_encode_kernel": round(ist["ms_encode"], 4),
               "kernel_GBps_read_plus_write": round(io_bytes / max(1e-9, ist["ms_encode"] * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(io_bytes / max(1e-9, ist["ms_encode"] * 1e-3) / 1e9 / peak_gbs, 4),
               "equals_input_reads": same}
        e2.close()
        return res

    if world == 1 and not force_dist and not args.dbg and not args.no_align:
        ingest = guarded("ingest_stage", run_ingest)

    # the step after the path: x-drop seed-and-extend of every candidate pair, once, outside the timed region
    align = None

    def run_align():
        al = eng.align_seeds()
        return {"alignments": int(al["nalignments"]), "passed": int(al["passed"]), "contained": int(al["contained"]), "seeds_rejected": int(al["seeds_rejected"]),
                "extensions_strided": int(al["extensions_strided"]), "cells": int(al["cells"]), "ms": round(al["ms_total"], 3), "ms_extend": round(al["ms_extend"], 3),
                "gcups": round(al["cells"] / max(1e-9, al["ms_extend"] * 1e-3) / 1e9, 3), "alignments_per_s": round(al["nalignments"] / max(1e-9, al["ms_total"] * 1e-3), 1),
                "params": {"mat": 1, "mis": -1, "gap": -1, "xdrop": 15}}

    if world == 1 and not force_dist and not args.dbg and not args.no_align:
        align = guarded("align_stage", run_align)

    # ... and the step after that: bad / contained read removal + transitive reduction of the aligned pairs.  On the
    # bench workload (15 % errors) the reference's default x-drop lets ~17 % of the alignments pass and find_bad_reads (cutoff 0.65) then
    # discards nearly every read — the graph is empty (reported below as it comes out).  The stage is therefore also run on a companion
    # set of accurate reads (0.5 % errors, the regime the default alignment parameters produce a string graph in), through the whole
    # chain reads -> B -> alignments -> S on a context of its own.
    string_graph = None
    sg_eng = None
    SG_CFG = {"genome": 1500000, "depth": 20.0, "avg_len": 8000, "sd_len": 1500, "error": 0.005, "k": 17, "lower": 8, "upper": 30, "seed": 7}

    def sg_numbers(sg_):
        sym_bytes = 2 * sg_["edges_kept"] * 16 + (sg_["nreads"] + 1) * 4      # the symmetrised R the masked product reads: 16-byte entries + row pointers
        return {"reads": int(sg_["nreads"]), "aligned_pairs": int(sg_["nedges"]), "bad_reads": int(sg_["bad_reads"]), "contained_reads": int(sg_["contained_reads"]),
                "edges_kept": int(sg_["edges_kept"]), "products": int(sg_["products"]), "marked": int(sg_["marked"]), "removed": int(sg_["removed"]), "nnz": int(sg_["nnz"]),
                "ms": round(sg_["ms_total"], 4), "ms_minplus_kernel": round(sg_["ms_minplus"], 4),
                "products_per_s": round(sg_["products"] / max(1e-9, sg_["ms_minplus"] * 1e-3), 1) if sg_["ms_minplus"] > 0 else None,
                "kernel_frac_of_hbm_peak": round(sym_bytes / (sg_["ms_minplus"] * 1e-3) / 1e9 / peak_gbs, 5) if sg_["ms_minplus"] > 0 else None}

    def run_string_graph():
        nonlocal sg_eng
        eng.transitive_reduction()                         # warm-up (allocations)
        res = {"params": {"bad_read_cutoff": 0.65, "fuzz": 1000}, "on_bench_workload": sg_numbers(eng.transitive_reduction())}
        c_ = SG_CFG
        sp, so, sl, _ = elba_amd.synth_reads(c_["seed"], c_["genome"], c_["depth"], c_["avg_len"], c_["sd_len"], error_rate=c_["error"], min_len=1000)
        sg_eng = Engine(c_["k"], c_["lower"], c_["upper"], device=local_rank)
        sg_eng.set_reads(sp, so, sl)
        sg_eng.count_kmers(); sg_eng.create_kmer_matrix(); ov_ = sg_eng.create_seed_matrix()
        al_ = sg_eng.align_seeds()
        sg_eng.transitive_reduction()
        res["accurate_reads"] = dict(sg_numbers(sg_eng.transitive_reduction()), config=c_, overlap_nnz=int(ov_["nnz"]), alignments_passed=int(al_["passed"]), align_ms=round(al_["ms_total"], 3))
        return res

    if align is not None:
        string_graph = guarded("string_graph_stage", run_string_graph)

    # N > 1: opt-in (--align-sharded).  The stage has collectives of its own (one all-gather of the reads); it is covered by the gloo /
    # threaded tests, and the default multi-GPU line stays the SpGEMM step alone.
    if (world > 1 or force_dist) and not args.dbg and args.align_sharded:
        # sharded alignment: reads replicated with one all-gather, every rank aligns its share of the pairs of its rows (no data-path collective afterwards)
        t0 = time.perf_counter(); al = runner.align_seeds(); barrier_sync(); t_al = time.perf_counter() - t0
        tot = torch.tensor([al["nalignments"], al["cells"], al["passed"]], dtype=torch.int64, device="cuda")
        mx = torch.tensor([al["ms_total"], t_al * 1e3], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        na, nc, npass = [int(x) for x in tot.tolist()]
        align = {"alignments": na, "passed": npass, "cells": nc, "ms_slowest_rank_device": round(float(mx[0]), 3), "ms_wall_with_read_allgather": round(float(mx[1]), 3),
                 "gcups": round(nc / max(1e-9, float(mx[0]) * 1e-3) / 1e9, 3), "sharing": "pair {i<j}: rank of row i if i+j even, of row j if odd; reads replicated by all_gather"}

    cpu = None
    parity = None
    kmer_ref = None
    distributions = None
    if rank == 0 and world == 1 and not force_dist and not args.no_cpu_baseline:
        from oracle import pyoracle as po   # checker + CPU baseline only
        A = runner.export_kmer_matrix()
        o = po.Oracle(k, lo, up)
        rows = np.repeat(np.arange(A["M"], dtype=np.int64), np.diff(A["rowptr"]))
        o.set_triples(A["M"], A["N"], rows, A["csr_kid"], A["csr_pos"])
        t0 = time.perf_counter(); o.spgemm(1); t1 = time.perf_counter() - t0
        ncores = os.cpu_count() or 1
        t0 = time.perf_counter(); o.spgemm(ncores); tn = time.perf_counter() - t0
        cpu = {"value": round(o.stat("Y") / t1, 1), "unit": "overlap nnz/s", "cores": 1, "kind": "port",
               "sample": "the full workload's SpGEMM region (same A, P=%d products) once, oracle/elba_oracle.c orc_spgemm, gcc -O3" % o.stat("P"),
               "seconds": round(t1, 4), "all_cores": {"value": round(o.stat("Y") / tn, 1), "cores": ncores, "seconds": round(tn, 4)}}
        B = runner.export_csr(); oB = o.B()
        # value distributions of the run: k-mer multiplicities, nnz per row of A, partners per row of B, products per output entry
        try:
            def _q(v):
                v = np.asarray(v, dtype=np.int64)
                return {"min": int(v.min()), "p50": int(np.percentile(v, 50)), "p90": int(np.percentile(v, 90)), "p99": int(np.percentile(v, 99)), "max": int(v.max()), "mean": round(float(v.mean()), 2)} if len(v) else None
            hist = eng.kmer_histogram()
            distributions = {"kmer_multiplicity_histogram": {str(c_): int(n_) for c_, n_ in enumerate(hist) if n_},
                             "row_nnz_A": _q(np.diff(A["rowptr"])), "partners_per_row_B": _q(np.diff(B["rowptr"])),
                             "numshared": _q(B["val"]["numshared"]), "products_per_overlap_nnz": round(st["products"] / max(1, st["nnz"]), 2)}
        except Exception as ex:               # noqa: BLE001
            distributions = {"error": "%s: %s" % (type(ex).__name__, ex)}
        parity = bool(B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"]).all() and (B["val"] == oB["val"]).all())
        # the reference's OWN k-mer stage on one host core, on a bounded sample of the reads (oracle/_ref: Kmer::GetRepKmers, and the two-pass
        # Bloom + map control flow of src/KmerOps.cpp replayed on the reference's Bloom / Kmer code) — a reported baseline, nothing more
        kmer_ref = None
        R = po.ref_lib(k)
        if R is not None:
            ns = min(len(lens), 400)
            sb = int(off[ns - 1]) + (int(lens[ns - 1]) + 3) // 4
            sp = np.concatenate([packed[:sb], np.zeros(16, np.uint8)]); so = off[:ns].copy(); sl = lens[:ns].copy()
            inst = int(np.maximum(sl.astype(np.int64) - k + 1, 0).sum())
            outk = np.zeros(int(sl.max()) + 8, dtype=np.uint64)
            t0 = time.perf_counter()
            for r in range(ns):
                R.ref_kmers(sp.ctypes.data + int(so[r]), int(sl[r]), outk.ctypes.data, 1)
            te = time.perf_counter() - t0
            cap = inst + 8
            ok_, or_, op_ = np.zeros(cap, np.uint64), np.zeros(cap, np.int64), np.zeros(cap, np.uint32)
            import ctypes as C
            k1 = C.c_int64()
            t0 = time.perf_counter()
            zz = R.ref_replay_count(sp.ctypes.data, so.ctypes.data, sl.ctypes.data, ns, lo, up, max(1, inst // 3), ok_.ctypes.data, or_.ctypes.data, op_.ctypes.data, cap, C.byref(k1))
            tc = time.perf_counter() - t0
            kmer_ref = {"kind": "reference", "cores": 1, "sample": "the first %d reads (%d k-mer instances)" % (ns, inst),
                        "enumerate_instances_per_s": round(inst / te, 1), "count_two_pass_instances_per_s": round(inst / tc, 1), "entries_on_sample": int(zz)}
        if align is not None:
            # CPU side of the alignment stage: the oracle's x-drop (pinned to the reference's XDropAligner.cpp) on every stride-th pair, one core
            stride = max(1, align["alignments"] // 400)
            t0 = time.perf_counter(); rws, cls, ov, ccells = o.align_upper(packed, off, lens, nthreads=1, stride=stride); ta = time.perf_counter() - t0
            g = runner.export_overlaps()
            idx = np.arange(0, len(rws), stride)
            same = bool((g["rows"] == rws).all() and (g["cols"] == cls).all() and all((g["vals"][f][idx] == ov[f][idx]).all() for f in ov.dtype.names if f != "pad"))
            align["cpu_baseline"] = {"gcups": round(ccells / ta / 1e9, 4), "cores": 1, "kind": "port", "sample": "every %d-th candidate pair (%d pairs, %d cells)" % (stride, len(idx), ccells), "seconds": round(ta, 3)}
            align["parity_vs_oracle_on_sample"] = same
            # the reference's OWN x-drop (src/XDropAligner.cpp compiled from /root/reference into oracle/_ref, which travels prebuilt) on the same sample
            R = po.ref_lib(k)
            if R is not None and hasattr(R, "ref_xdrop"):
                t0 = time.perf_counter(); ok = True
                for a in idx:
                    i_, j_ = int(rws[a]), int(cls[a])
                    e0 = int(B["rowptr"][i_]) + int(np.searchsorted(B["col"][int(B["rowptr"][i_]):int(B["rowptr"][i_ + 1])], j_))
                    r_ = po.ref_xdrop(R, packed[int(off[i_]):], int(lens[i_]), packed[int(off[j_]):], int(lens[j_]), int(B["val"][e0]["q0"]), int(B["val"][e0]["t0"]))
                    v_ = g["vals"][a]
                    ok = ok and (r_[1], r_[2], r_[3], r_[4], r_[5], r_[6], r_[7]) == (int(v_["begQ"]), int(v_["endQ"]), int(v_["begT"]), int(v_["endT"]), int(v_["score"]), int(v_["rc"]), int(v_["kind"]))
                tr = time.perf_counter() - t0
                align["cpu_baseline_reference"] = {"gcups": round(ccells / tr / 1e9, 4), "cores": 1, "kind": "reference", "seconds": round(tr, 3),
                                                   "what": "the reference's xdrop_aligner + classify_alignment (g++ -O2) on the same sample, called through ctypes"}
                align["parity_vs_reference_on_sample"] = bool(ok)

        if string_graph is not None and align is not None:
            # CPU side: the oracle runs the reference's statements literally (full min-plus SpGEMM, loop included) on the GPU's own aligned pairs
            def sg_check(e_, nreads_, into):
                g_ = e_.export_overlaps()
                t0 = time.perf_counter(); S_, fl_, sst = po.string_graph(nreads_, g_["rows"], g_["cols"], g_["vals"]); ts = time.perf_counter() - t0
                gs = e_.export_string_graph()
                into["parity_vs_oracle"] = bool(gs["n"] == S_["n"] and (gs["rows"] == S_["rows"]).all() and (gs["cols"] == S_["cols"]).all()
                                                and all((gs["vals"][f] == S_["vals"][f]).all() for f in S_["vals"].dtype.names if f != "pad")
                                                and (e_.export_read_flags(nreads_) == fl_).all()
                                                and all(into[k_] == sst[k_] for k_ in ("bad_reads", "contained_reads", "edges_kept", "products", "marked", "removed", "nnz")))
                into["cpu_baseline"] = {"products_per_s": round(sst["products"] / ts, 1), "cores": 1, "kind": "port", "seconds": round(ts, 4),
                                        "sample": "the whole stage once (oracle/elba_oracle.c orc_string_graph: prunes + full R(x)R + compare, %d loop passes)" % sst["iterations"]}
            sg_check(eng, len(lens), string_graph["on_bench_workload"])
            if sg_eng is not None and "accurate_reads" in string_graph:
                sg_check(sg_eng, string_graph["accurate_reads"]["reads"], string_graph["accurate_reads"])

    if rank == 0:
        out = {
            "metric": "overlap nnz/sec (A·Aᵀ SpGEMM, SharedSeeds semiring, after Prune(numshared<=1))",
            "value": round(Y / (dt / steps), 1), "unit": "overlap nnz/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": dict({"workload": args.workload, "reads": int(info["total_reads"]) if "total_reads" in info else None, "k": k, "lower": lo, "upper": up,
                            "genome": w["genome"] * world, "depth": w["depth"], "error": w["error"], "kmer_instances": I_tot, "nnz_A": Z_tot,
                            "products": P, "overlap_nnz": Y, "algorithmic_bytes": abytes}, **extra_cfg),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_vs_oracle": parity,
            "distributions": distributions,
            "cold_call_ms": None if cold_ms is None else round(cold_ms, 4),
            "aux_errors": aux_errors or None,
            "ingest_stage": ingest,
            "align_stage": align,
            "string_graph_stage": string_graph,
            "kmer_stage": {"device_ms": round(ks["ms_total"] + ms["ms_total"], 3), "matrix_build_ms": round(ms["ms_total"], 3), "wall_ms": round(t_kmer_wall * 1e3, 3),
                           "instances_per_s": round(ks["instances"] / max(1e-9, t_kmer_wall), 1), "count_ms": round(ks["ms_count"], 3), "select_sort_ms": round(ks["ms_sort"], 3),
                           "cpu_baseline_reference": kmer_ref},
            "phases_ms": {key: round(v, 4) for key, v in acc.items()},
            "tiers": {key: int(st[key]) for key in ("rows_lds", "rows_global", "rows_escalated", "nnz_before_prune", "passes") if key in st},
            "gen_s": round(t_gen, 2),
        }
        try:                                     # RCCL prints its version banner through C stdio: flush it first so the JSON line is last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
