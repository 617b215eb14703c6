#!/usr/bin/env python3
"""bench.py — overlap-matrix build throughput (candidate-overlap nnz/s of B = A·Aᵀ, SharedSeeds semiring) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one process per
GPU with torch.distributed.run.  One JSON line on rank 0.

A "step" is ONE pass of the hot path's headline region over the synthetic input, the region SURVEY.md §8d defines: device-resident
A (CSR + the k-mer columns, exactly as the k-mer stage leaves them) -> device-resident pruned CSR B (reference timer "creating seed
matrix (spgemm)", src/main.cpp:280-282 around src/SharedSeeds.cpp:4-10).  There is no execution plan behind it: no per-row schedule,
no descriptors, no row order, no product counts are built with A (elba_amd/csrc/spgemm_direct.hpp).  What A's rows do carry, from the
k-mer stage that builds them (timed in `kmer_stage`, at no measurable cost: every entry sees its whole column there anyway), is two
hint bits per entry — "under the parity rule that assigns every pair of rows to one of them, this row accumulates no pair of this
column": such an entry (40 % of them on this workload) does not fetch its column.  Every timed step runs COLD
(elba_set_option "overlap_cold_calls"): it forgets what earlier calls on the same matrix learned (the distinct-partner ratio that picks
the starting table tiers, which tiers and column sorts received rows), so that each step is what a caller that multiplies a matrix once
pays, as ELBA does.  Buffers stay allocated.  The steady state (hints kept between calls) is reported as a secondary key.

Workload at N = 1: BASELINE.json configs[2] restated per SURVEY.md §8d-3 ("200k-long-reads": 200 100 reads of ~10 kb, 66.7 Mb genome,
30x, 15 % sub/ins/del error, k=17, L=2, U=8, seed 2) — the configuration the metric's scaling is quoted on, and it fits one GPU.
At N > 1 the SAME read set is sharded by contiguous row blocks over the ranks (strong scaling); k-mer instances go to the owner of
their value range and come back as column panels (two RCCL all-to-alls, a 32 KB all-reduce and a scalar all-gather while A is built).
The step then has ONE collective: every pair of rows that live on two ranks is accumulated by one of them and its mirrored entry (32 bytes)
is sent to the other — each rank does 1/N of the one-GPU work instead of computing every cross-rank pair twice.
`--workload ecsample30x-like` is BASELINE.json configs[1] (16 893 reads), kept as a parity-test / profiling case.
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
    "200k-long-reads": dict(genome=66_700_000, depth=30.0, avg_len=10000.0, sd_len=1500.0, min_len=1000, error=0.15, k=17, lower=2, upper=8, seed=2),
    "ecsample30x-like": dict(genome=4_640_000, depth=30.0, avg_len=8240.0, sd_len=2000.0, min_len=1000, error=0.15, k=17, lower=2, upper=8, seed=1),
    "plumbing-135": dict(genome=100_000, depth=13.5, avg_len=10000.0, sd_len=1000.0, min_len=1000, error=0.0, k=17, lower=2, upper=8, seed=313),
    # profiling cases (single GPU; not the driver's line): BASELINE configs[3] at half the genome (the whole set — 4.0 G instances — runs too:
    # tests/test_gpu_configs.py) and configs[4] at 1/25 of it
    "hifi-half": dict(genome=50_000_000, depth=40.0, avg_len=15000.0, sd_len=2000.0, min_len=1000, error=0.005, k=17, lower=2, upper=4, seed=3),
    # the reference's DEFAULT build (Makefile:1-3: k = 31, L = 15, U = 35; both of its run recipes use -k 31: script/job.haswell.hifi-celegans.40x.128n:18,
    # script/corigpu-job-ecoli.sh:22) on config-4-like HiFi reads, half the genome (2.0 G k-mer instances)
    "hifi-k31": dict(genome=50_000_000, depth=40.0, avg_len=15000.0, sd_len=2000.0, min_len=1000, error=0.005, k=31, lower=15, upper=35, seed=3),
    "dense-repeats-8th": dict(genome=62_500_000, depth=40.0, avg_len=10000.0, sd_len=1000.0, min_len=1000, error=0.01, k=17, lower=2, upper=35, seed=4, repeats=(20, 0.05, 5000)),      # ONE GPU's share of configs[4] at N = 8 (250 000 reads)
    "dense-repeats-25th": dict(genome=20_000_000, depth=40.0, avg_len=10000.0, sd_len=1000.0, min_len=1000, error=0.01, k=17, lower=2, upper=35, seed=4, repeats=(20, 0.05, 5000)),
}
PEAK_GBS = 8000.0          # HBM3E peak, MI355X_MICROARCH.md

# What the one-GPU build of each workload counts (k-mer instances, nnz(A), semiring products, nnz(B)): properties of the generated read set, the
# same for every number of ranks.  A run on N > 1 GPUs must reproduce them after its all-reduce — the first hardware run of the sharded
# path validates itself; a mismatch ends the run with a non-zero exit code.  (ELBA_BENCH_EXPECT="I,Z,P,Y" overrides: tests.)
EXPECTED_COUNTS = {
    "200k-long-reads": (1996913231, 534826215, 1370686611, 98693580),
    "ecsample30x-like": (138894522, 14180071, 42458349, 1359357),
    "hifi-half": (1996504200, 10175381, 20959215, 1316021),
    "dense-repeats-25th": (798652841, 366045468, 10848555572, 8388812),
    "dense-repeats-8th": (2496166557, 1153437658, 33856055912, 41427794),
    "hifi-k31": (1994637538, 903534856, 27911412714, 10219284),
}


def host_copy(dev_ptr, count, dtype):
    """`count` items of `dtype` from a raw device pointer into a fresh numpy array (hipMemcpy through ctypes: plumbing for the checker)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    out = np.empty(int(count), dtype=dtype)
    if count:
        rc = hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(int(dev_ptr)), ctypes.c_size_t(out.nbytes), 2)
        if rc != 0:
            raise RuntimeError("hipMemcpy D2H failed: %d" % rc)
    return out


def host_memory_available():
    """bytes this process may still allocate: the smaller of MemAvailable and the cgroup's limit minus its usage"""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read().strip())
            left = int(lim) - used
            avail = left if avail is None else min(avail, left)
    except (OSError, ValueError):
        pass
    return avail


def bytes_kmer_stage(bases, I, N, Z):
    """read the packed reads once, materialise the 16-byte instance stream once (write + read), write CSC(A) with k-mer values."""
    return bases // 4 + 32 * I + 8 * (N + 1) + 16 * Z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="200k-long-reads", choices=sorted(WORKLOADS))
    ap.add_argument("--steady-steps", type=int, default=10, help="untimed-by-the-contract steps with the hints of earlier calls kept (secondary figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accounting", action="store_true", help="skip spgemm_prep / from_triples (two more contexts)")
    ap.add_argument("--cpu-sample-div", type=int, default=0, help="CPU baseline, one-core leg: the same generator with the genome divided by this (0: chosen for ~10-30 s of CPU work)")
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the all-cores CPU baseline on the WHOLE workload matrix (and the entry-by-entry comparison of B with it)")
    ap.add_argument("--timing-stride", type=int, default=4, help="the library records its phase events (kernel_ms of the roofline) on every N-th step only: "
                    "an event record costs ~5 us of stream time; kernel_ms is the mean over the steps that were measured")
    ap.add_argument("--aux", action="store_true", help="N = 1: also run the stages on either side of the path once (FASTA ingest, x-drop alignment, string graph); "
                    "minutes on the 200 k-read set")
    ap.add_argument("--weak", action="store_true", help="N > 1: keep the per-GPU read count and grow the genome with N instead of sharding one read set")
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
    # ELBA_DIST_BACKEND=gloo: the N > 1 driver as N processes on ONE GPU, collectives carried over the host (elba_amd.distributed.HostStagedDist) —
    # a rehearsal of this file's multi-rank path on a one-GPU box (RCCL refuses two ranks on one device); the line then says so in "transport"
    backend = os.environ.get("ELBA_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = bool(os.environ.get("ELBA_FORCE_DIST"))      # exercise the multi-GPU driver with world_size 1 (self-test on a 1-GPU box)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            from elba_amd.distributed import HostStagedDist
            dist.init_process_group(backend, rank=rank, world_size=world)
            dist = HostStagedDist(dist)

    w = WORKLOADS[args.workload]
    k, lo, up = w["k"], w["lower"], w["upper"]
    single = world == 1 and not force_dist

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if single:
        from elba_amd.capi import Engine
        t0 = time.time()
        rep = w.get("repeats", (0, 0.0, 0))
        packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                                       repeat_families=rep[0], repeat_fraction=rep[1], repeat_len=rep[2])
        t_gen = time.time() - t0
        # (A/B runs: ELBA_BENCH_OPTIONS="name=value,name=value" sets library options before A is built — never a result-changing switch)
        bench_opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("ELBA_BENCH_OPTIONS", "").split(",") if "=" in kv}
        eng = Engine(k, lo, up, device=local_rank, timing_stride=args.timing_stride, options=bench_opts or None)
        # inputs resident in HBM before anything is timed
        d_packed = torch.from_numpy(packed).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda(); d_len = torch.from_numpy(lens.view(np.int32)).cuda()
        eng.set_reads_device(d_packed.data_ptr(), int(packed.size) - 16, d_off.data_ptr(), d_len.data_ptr(), len(lens))
        torch.cuda.synchronize()
        eng.count_kmers(); eng.create_kmer_matrix(); torch.cuda.synchronize()          # first pass: allocations
        t0 = time.perf_counter(); ks = eng.count_kmers(); ms = eng.create_kmer_matrix(); torch.cuda.synchronize(); t_kmer_wall = time.perf_counter() - t0
        runner = eng
        bases_local = int(lens.astype(np.int64).sum())
        extra_cfg = {}
    else:
        from elba_amd.distributed import DistributedOverlap
        runner = DistributedOverlap(k, lo, up, device=local_rank, rank=rank, world=world, dist=dist, timing_stride=args.timing_stride)
        runner.time_phases = True                 # (send, all-to-all, recv) device times of every step, per rank
        runner.force_exchange = force_dist
        t0 = time.time()
        info = runner.generate_and_set_reads(w, weak=args.weak)
        t_gen = time.time() - t0
        barrier_sync()
        t0 = time.perf_counter(); ks, ms = runner.build_kmer_matrix(); barrier_sync(); t_kmer_wall = time.perf_counter() - t0
        eng = runner.be.e
        bases_local = int(runner._reads[2].astype(np.int64).sum())
        extra_cfg = {"partition": "1D read rows x value-range-owned k-mer columns", "exchange": "RCCL all_to_all_single x2 (+ a 32 KB all_reduce and a scalar all_gather) while A is built; inside the step ONE all_to_all_single of mirrored entries (32 B each): a pair of rows on two ranks is accumulated by one of them",
                     "exchange_bytes_this_rank": getattr(runner, "exchange_bytes", None)}
    step = runner.create_seed_matrix

    # ---- the contract's timed region: K cold steps ------------------------------------------------------------------------------
    eng.set_option("overlap_cold_calls", 1)
    for _ in range(args.warmup):
        st = step()
    barrier_sync()
    t0 = time.perf_counter()
    acc = dict(ms_total=0.0, ms_numeric=0.0, ms_symbolic=0.0, ms_finalize=0.0)
    ntimed = 0
    rank_phase = dict(send=0.0, exchange=0.0, recv=0.0)
    for _ in range(args.steps):
        st = step()
        if st.get("timed", 1):
            ntimed += 1
            for key in acc:
                acc[key] += st[key]
        if not single and getattr(runner, "phase_ms", None):
            for key in rank_phase:
                rank_phase[key] += runner.phase_ms[key] / max(1, args.steps)
    barrier_sync()
    dt = time.perf_counter() - t0
    st_cold = st
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # (nnz(A): every column is counted by the rank that owns its k-mer — a rank's panel holds whole columns, shared with other ranks' panels)
        tot = torch.tensor([st["nnz"], st["products"], st["algorithmic_bytes"], ks["instances"], ks.get("entries", ms["nnz"]), bases_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        Y, P, abytes, I_tot, Z_tot, bases_tot = [int(x) for x in tot.tolist()]
        # every rank's step phases (device time on its stream): queue numeric + group the mirror images | all-to-all | merge + finalize + the one host wait
        ph = torch.tensor([rank_phase["send"], rank_phase["exchange"], rank_phase["recv"], float(st["nnz"]), float(st["products"])], dtype=torch.float64, device="cuda")
        allph = [torch.zeros_like(ph) for _ in range(world)]
        dist.all_gather(allph, ph)
        rank_phases = [dict(rank=r, send_ms=round(float(x[0]), 4), exchange_ms=round(float(x[1]), 4), recv_ms=round(float(x[2]), 4), nnz=int(x[3]), products=int(x[4])) for r, x in enumerate(allph)]
    else:
        rank_phases = None
        Y, P, abytes, I_tot, Z_tot, bases_tot = st["nnz"], st["products"], st["algorithmic_bytes"], ks["instances"], ms["nnz"], bases_local
    # mirror slabs: how the last cold step's mirrored entries reached their rows
    mirror = None
    if hasattr(eng, "get_stat"):
        q16 = eng.get_stat("overlap_slab_q16")
        mirror = {"slab_entries_per_row_entry_q16": q16, "placed_by_k_mirror": eng.get_stat("overlap_mirror_placed"),
                  "what": "mirrored entries go straight from the numeric kernel to their row's slab (sized by the cold call's sample of rows); the rest waits for k_mirror" if q16 else "no slabs in this step"}
    # the counts every run of this workload must reproduce, on any number of ranks
    expect = EXPECTED_COUNTS.get(args.workload) if not args.weak else None
    if os.environ.get("ELBA_BENCH_EXPECT"):
        expect = tuple(int(x) for x in os.environ["ELBA_BENCH_EXPECT"].split(","))
    got_counts = (int(I_tot), int(Z_tot), int(P), int(Y))
    counts_ok = None if expect is None else bool(got_counts == tuple(expect))
    steps = max(1, args.steps)
    ms_step = dt / steps * 1e3
    for key in acc:
        acc[key] /= max(1, ntimed)

    # ---- secondary: the steady state (the hints of earlier calls kept: starting tiers from the measured ratio, unused tiers not launched)
    eng.set_option("overlap_cold_calls", 0)
    steady = None
    if args.steady_steps > 0:
        for _ in range(2):
            step()
        barrier_sync()
        t0 = time.perf_counter()
        sacc, sn = 0.0, 0
        for _ in range(args.steady_steps):
            s2 = step()
            if s2.get("timed", 1):
                sacc += s2["ms_numeric"]; sn += 1
        barrier_sync()
        sdt = (time.perf_counter() - t0) / args.steady_steps
        if dist is not None:
            tt = torch.tensor([sdt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sdt = float(tt.item())
        steady = {"ms_per_step": round(sdt * 1e3, 4), "value": round(Y / sdt, 1), "kernel_ms": round(sacc / max(1, sn), 4),
                  "what": "same region, hints of earlier calls on the same matrix kept (never reached by a caller that multiplies each matrix once)"}

    # ---- roofline of the dominant kernel (k_spgemm_direct, all table tiers: they jointly process every row once per step) -----------
    # achieved = algorithmic bytes of this rank's rows (16 Z + 8 (2M + N + 3) + 24 Y) / HIP-event duration of those
    # launches on the library's stream, measured live on every timing_stride-th timed step.
    my_bytes = st_cold["algorithmic_bytes"]
    # Headline = the WHOLE region: the formula's bytes (both orientations of A read, all of B written) belong to the whole launch sequence — the numeric
    # kernels do not write the 24 Y bytes of B, the finalize kernels do — so bytes and time cover the same kernels.  The
    # numeric kernels alone are reported beside it (`numeric_kernels`): against the same bytes, and against the HBM bytes they were MEASURED to move.
    achieved = my_bytes / (acc["ms_total"] * 1e-3) / 1e9 if acc["ms_total"] > 0 else 0.0
    achieved_num = my_bytes / (acc["ms_numeric"] * 1e-3) / 1e9 if acc["ms_numeric"] > 0 else 0.0
    traffic = None
    traffic_note = None
    tj = None
    fingerprint = elba_amd.capi.numeric_source_fingerprint()
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == args.workload and tj.get("n_gpus", 1) == world:
                if fingerprint is not None and tj.get("numeric_source_sha16") == fingerprint:
                    traffic = tj.get("hbm_bytes_per_step_dominant_kernel")
                    traffic_note = tj.get("note")
                else:
                    traffic_note = "profiles/traffic.json was measured on other kernel sources (%s, now %s): not quoted" % (tj.get("numeric_source_sha16"), fingerprint)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "the launch sequence of one cold call: k_spgemm_direct (plan-free LDS-hash numeric, all tiers; dominant) + classify + row pointers + finalize",
                "achieved": round(achieved, 3), "peak": PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note,
                "algorithmic_bytes_per_step": my_bytes, "bytes_per_nnz": round(my_bytes / max(1, st_cold["nnz"]), 2),
                "kernel_ms": round(acc["ms_numeric"], 4), "kernel_ms_measured_on_steps": ntimed, "region_ms_device": round(acc["ms_total"], 4),
                "frac_whole_region": round(achieved / PEAK_GBS, 6),
                "frac_whole_region_wall": round(abytes / (dt / steps) / 1e9 / PEAK_GBS / world, 6),
                "numeric_kernels": {"kernel_ms": round(acc["ms_numeric"], 4), "frac_of_algorithmic_bytes": round(achieved_num / PEAK_GBS, 6),
                                    "frac_of_measured_traffic": round(traffic / (acc["ms_numeric"] * 1e-3) / 1e9 / PEAK_GBS, 6) if (traffic and acc["ms_numeric"] > 0) else None,
                                    "note": "the dominant kernels alone: the whole step's algorithmic bytes over their time overstates them (they do not write B); the measured figure is what they move"},
                "expanded_stream_bytes": 8 * st_cold["products"] + 8 * ms["nnz"] + 24 * st_cold["nnz"]}
    # the resource that binds the kernel is the rate of 64-byte line requests (profiles/r02_notes.md): PMC request counts (offline, like `traffic`)
    rpath = os.path.join(ROOT, "profiles", "requests.json")
    if os.path.exists(rpath) and acc["ms_numeric"] > 0:
        try:
            rj = json.load(open(rpath))
            if rj.get("workload") == args.workload and rj.get("n_gpus", 1) == world and fingerprint is not None and rj.get("numeric_source_sha16") == fingerprint:
                rate = rj["tcc_requests_per_step"] / (acc["ms_numeric"] * 1e-3) / 1e9
                roofline["l2_requests"] = {"per_step": rj["tcc_requests_per_step"], "achieved_G_per_s": round(rate, 2), "measured_ceiling_G_per_s": rj["ceiling_G_requests_per_s"],
                                           "frac_of_ceiling": round(rate / max(rj["ceiling_G_requests_per_s"]), 4)}
        except Exception:
            pass

    # ---- k-mer stage (reads -> A) and end to end (reads -> B): SURVEY.md §8d secondary figures ----------------------------------------
    kb = bytes_kmer_stage(bases_tot, I_tot, int(ms["ncols"]) if single else 0, Z_tot)
    if dist is not None:
        tk = torch.tensor([t_kmer_wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        t_kmer_wall = float(tk.item())
    kdev = (ks["ms_total"] + ms["ms_total"]) if single else None
    kmer_stage = {"wall_ms": round(t_kmer_wall * 1e3, 3), "device_ms": None if kdev is None else round(kdev, 3),
                  "count_ms": round(ks.get("ms_count", 0.0), 3), "runs_to_columns_ms": round(ks.get("ms_sort", 0.0), 3), "matrix_build_ms": round(ms["ms_total"], 3),
                  "instances": I_tot, "instances_per_s": round(I_tot / max(1e-9, t_kmer_wall), 1),
                  "roofline": {"bound": "hbm", "bytes": kb, "formula": "bases/4 + 32 I + 8 (N+1) + 16 Z", "unit": "GB/s", "peak": PEAK_GBS,
                               "achieved": round(kb / max(1e-9, t_kmer_wall) / 1e9 / world, 3), "frac": round(kb / max(1e-9, t_kmer_wall) / 1e9 / PEAK_GBS / world, 6),
                               "timed": "host clock around elba_count_kmers + elba_create_kmer_matrix (second pass, buffers allocated)" if single else
                                        "host clock around the distributed build of A incl. the three collectives, max over ranks"}}
    e2e_s = t_kmer_wall + dt / steps
    end_to_end = {"ms": round(e2e_s * 1e3, 3), "what": "packed reads on the device -> B: k-mer stage + one cold SpGEMM step (reference timers src/main.cpp:191-282)",
                  "overlap_nnz_per_s": round(Y / e2e_s, 1), "bytes": kb + abytes, "frac": round((kb + abytes) / e2e_s / 1e9 / PEAK_GBS / world, 6)}

    # ---- honest accounting of what the SpGEMM region leans on (N = 1) ------------------------------------------------------------------------
    # (a) the k-mer stage writes two things for the product's sake alone: the padded column store and the ownership bits of the row entries.
    #     Their cost = this k-mer stage minus the same stage built without them ("no_ell", "no_hints"), on a context of its own.
    # (b) what a replacement of create_seed_matrix(A, AT) alone pays (option B): A arrives as device-resident triples,
    #     elba_set_kmer_matrix_device rebuilds CSR + columns + hints, then ONE cold call.
    # what the HBM-table tier costs when every row is FORCED onto it: two cold calls with "tune7" = 6 (k_classify_direct queues every row on the
    # spill tier: tables in HBM instead of LDS), the second one timed; same B by count
    hbm_forced = None
    if single and rank == 0 and not args.no_accounting:
        eng.set_option("tune7", 6); eng.set_option("overlap_cold_calls", 1)
        try:
            eng.create_seed_matrix(); torch.cuda.synchronize()
            t0 = time.perf_counter(); sf = eng.create_seed_matrix(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
            hbm_forced = {"ms_cold_call": round(tf * 1e3, 3), "rows_global": int(sf.get("rows_global", -1)), "rows_lds": int(sf.get("rows_lds", -1)),
                          "same_nnz_and_products": bool(sf["nnz"] == st_cold["nnz"] and sf["products"] == st_cold["products"]),
                          "vs_lds_tiers": round(tf * 1e3 / max(1e-9, ms_step), 2), "what": "option tune7 = 6: every row starts on the HBM-table tier (the sample's rows excepted)"}
        except elba_amd.capi.ElbaError as ex:
            hbm_forced = {"failed": str(ex)}
        eng.set_option("tune7", 0); eng.set_option("overlap_cold_calls", 0)
        eng.create_seed_matrix(); torch.cuda.synchronize()      # (the context holds the LDS-tier result again: exports and the CPU comparison below read it)
    prep = None
    from_triples = None
    if single and rank == 0 and not args.no_accounting and args.workload.startswith("dense-repeats-8"):
        prep = {"skipped": "two more contexts of this size do not fit beside the timed one"}
    elif single and rank == 0 and not args.no_accounting:
        # (round 5) measured by EVENTS inside the k-mer stage: with the option "measure_prep" elba_count_kmers runs its emit kernels a second time without the
        # hint bits, inline partners, gather slots and padded columns (they are the only kernels that write them), both runs bracketed by events on the
        # library's stream: elba_get_stat("spgemm_prep_us") = the difference of two ~11 ms device times (round 4 subtracted two ~50 ms host-clock runs of
        # two different contexts: 2.4 ms in one run, 4.2 in the next).  Matrices built by other kernels (small inputs: the sort path) keep the subtraction.
        eng.release_workspace()      # (the timed context's sort / partition scratch — 64 GB on config 3 — makes room for the contexts below)
        e2 = Engine(k, lo, up, device=local_rank, options={"measure_prep": 1})
        e2.set_reads_device(d_packed.data_ptr(), int(packed.size) - 16, d_off.data_ptr(), d_len.data_ptr(), len(lens))
        prep_runs = []
        try:
            e2.count_kmers(); torch.cuda.synchronize()
            for _ in range(3):
                e2.count_kmers(); torch.cuda.synchronize()
                prep_runs.append((e2.get_stat("spgemm_prep_us"), e2.get_stat("emit_us")))
        except elba_amd.capi.ElbaError as ex:      # (a second context of this size beside the timed one: reported, not hidden)
            prep_runs = [(-2, str(ex))]
        e2.close()
        if prep_runs[0][0] == -2:
            prep_ms = None
        elif all(p_[0] >= 0 for p_ in prep_runs):
            prep_ms = sorted(p_[0] for p_ in prep_runs)[1] / 1e3
            prep_what = ("device events inside elba_count_kmers: the emit kernels as built (%.3f ms) minus the same kernels without hint bits, inline partners, gather slots "
                         "and padded columns; median of three runs %s us" % (sorted(p_[1] for p_ in prep_runs)[1] / 1e3, [p_[0] for p_ in prep_runs]))
        else:
            e2 = Engine(k, lo, up, device=local_rank, options={"no_ell": 1, "no_hints": 1})
            e2.set_reads_device(d_packed.data_ptr(), int(packed.size) - 16, d_off.data_ptr(), d_len.data_ptr(), len(lens))
            e2.count_kmers(); e2.create_kmer_matrix(); torch.cuda.synchronize()
            t0 = time.perf_counter(); e2.count_kmers(); e2.create_kmer_matrix(); torch.cuda.synchronize(); t_plain = time.perf_counter() - t0
            e2.close()
            prep_ms = max(0.0, (t_kmer_wall - t_plain) * 1e3)
            prep_what = "padded column store + ownership hint bits: k-mer stage with them (%.3f ms) minus without (%.3f ms), host clock (this matrix was not built by the bucket kernels)" % (t_kmer_wall * 1e3, t_plain * 1e3)
        if prep_ms is None:
            prep = {"skipped": "the second context did not fit beside the timed one: " + prep_runs[0][1]}
        else:
          prep = {"spgemm_prep_ms": round(prep_ms, 3), "what": prep_what,
                "frac_incl_prep": round(my_bytes / ((acc["ms_numeric"] + prep_ms) * 1e-3) / 1e9 / PEAK_GBS, 6) if acc["ms_numeric"] > 0 else None,
                "frac_whole_region_incl_prep": round(my_bytes / ((acc["ms_total"] + prep_ms) * 1e-3) / 1e9 / PEAK_GBS, 6) if acc["ms_total"] > 0 else None}
        eng.release_workspace()      # (the timed context's sort / partition scratch — 64 GB on config 3: the second context below decides by the free memory whether its padded column store fits)
        Zt = int(ms["nnz"])
        d_rows = torch.empty(Zt, dtype=torch.int64, device="cuda"); d_cols = torch.empty(Zt, dtype=torch.int64, device="cuda"); d_vals = torch.empty(Zt, dtype=torch.int32, device="cuda")
        eng.export_triples_device(d_rows.data_ptr(), d_cols.data_ptr(), d_vals.data_ptr())
        e3 = Engine(k, lo, up, device=local_rank)
        e3.set_kmer_matrix_device(int(ms["nrows"]), int(ms["ncols"]), Zt, d_rows.data_ptr(), d_cols.data_ptr(), d_vals.data_ptr()); e3.create_seed_matrix(); torch.cuda.synchronize()      # allocations
        e3.set_option("overlap_cold_calls", 1)
        t0 = time.perf_counter(); m3 = e3.set_kmer_matrix_device(int(ms["nrows"]), int(ms["ncols"]), Zt, d_rows.data_ptr(), d_cols.data_ptr(), d_vals.data_ptr()); torch.cuda.synchronize(); t_set = time.perf_counter() - t0
        t0 = time.perf_counter(); s3 = e3.create_seed_matrix(); torch.cuda.synchronize(); t_call = time.perf_counter() - t0
        from_triples = {"set_kmer_matrix_device_ms": round(t_set * 1e3, 3), "first_cold_call_ms": round(t_call * 1e3, 3), "ms": round((t_set + t_call) * 1e3, 3),
                        "overlap_nnz_per_s": round(s3["nnz"] / (t_set + t_call), 1), "same_nnz_and_products": bool(s3["nnz"] == st_cold["nnz"] and s3["products"] == st_cold["products"]),
                        "path": "bucket kernels of the k-mer stage" if e3.get_stat("triples_path") == 1 else "radix sorts of the whole matrix",
                        "padded_columns": bool(e3.get_stat("padded_columns")),
                        "what": "device-resident COO triples of A (int64, int64, uint32) -> CSR + k-mer columns + hints + padded columns, then one cold elba_create_seed_matrix"}
        e3.close()
        del d_rows, d_cols, d_vals
    # the reference's five stage timers (src/main.cpp:193,226,260,274,282) and what runs under each label here
    ref_timers = {"collecting distinct k-mers": round(ks.get("ms_count", 0.0), 3), "counting recording k-mer seeds": round(ks.get("ms_sort", 0.0), 3),
                  "creating k-mer matrix": round(ms["ms_total"] if ms else 0.0, 3), "copying and transposing k-mer matrix": 0.0,
                  "creating seed matrix (spgemm)": round(ms_step, 4),
                  "note": "device ms of this rank; labels 1-2 = the two halves of elba_count_kmers (value partition | bucket count + columns), label 4 is empty: both orientations of A leave the k-mer stage together"}

    # ---- CPU baseline + parity on a bounded sample of the same workload (rank 0, N = 1) ------------------------------------------------
    cpu = None
    parity = None
    if rank == 0 and single and not args.no_cpu_baseline:
        from oracle import pyoracle as po   # checker + CPU baseline only
        # the sample: the same generator (lengths, depth, error, k, L, U), genome divided so that the oracle's SpGEMM is ~10-30 s of CPU work
        div = args.cpu_sample_div or max(1, int(round(P / 1.5e8)))
        if div == 1:
            sp, so, sl = packed, off, lens
            es = eng
        else:
            sp, so, sl, _ = elba_amd.synth_reads(w["seed"], max(20000, w["genome"] // div), w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                                 repeat_families=w.get("repeats", (0, 0.0, 0))[0], repeat_fraction=w.get("repeats", (0, 0.0, 0))[1], repeat_len=w.get("repeats", (0, 0.0, 0))[2])
            es = Engine(k, lo, up, device=local_rank)
            es.set_reads(sp, so, sl); es.count_kmers(); es.create_kmer_matrix()
        tg = time.perf_counter(); sst = es.create_seed_matrix(); tg = (time.perf_counter() - tg) * 1e3
        if not sst["ms_total"]: sst["ms_total"] = tg      # (a call the context did not time with events: host clock)
        A = es.export_kmer_matrix()
        o = po.Oracle(k, lo, up)
        rows = np.repeat(np.arange(A["M"], dtype=np.int64), np.diff(A["rowptr"]))
        o.set_triples(A["M"], A["N"], rows, A["csr_kid"], A["csr_pos"])
        t0 = time.perf_counter(); o.spgemm(1); t1 = time.perf_counter() - t0
        ncores = min(os.cpu_count() or 1, 64)
        t0 = time.perf_counter(); o.spgemm(ncores); tn = time.perf_counter() - t0
        B = es.export_csr(); oB = o.B()
        parity = bool(B["Y"] == oB["Y"] and (B["rowptr"] == oB["rowptr"]).all() and (B["col"] == oB["col"]).all() and (B["val"] == oB["val"]).all()
                      and sst["products"] == o.stat("P") and sst["nnz_before_prune"] == o.stat("Yraw"))
        one_core = {"value": round(o.stat("Y") / t1, 1), "unit": "overlap nnz/s", "cores": 1,
                    "sample": "NOT the headline matrix: the same generator with the genome divided by %d (%d reads, nnz(A) = %d, P = %d products, Y = %d): the SpGEMM region once, "
                              "oracle/elba_oracle.c orc_spgemm, gcc -O3" % (div, A["M"], A["Z"], o.stat("P"), o.stat("Y")),
                    "seconds": round(t1, 4), "all_cores_on_that_sample": {"value": round(o.stat("Y") / tn, 1), "cores": ncores, "seconds": round(tn, 4)},
                    "gpu_on_the_same_sample": {"ms_cold_call": round(sst["ms_total"], 4), "value": round(sst["nnz"] / max(1e-9, sst["ms_total"] * 1e-3), 1)}}
        cpu = dict(one_core, kind="port")
        if es is not eng:
            es.close()
        del o, A, B, oB, rows
        # ---- the WHOLE workload matrix on all host cores, and every entry of the GPU's B against it (VERDICT r3 task 7) ----------------------
        # The timed matrix itself: its columns leave the device as they are (u32 pointers, read << 32 | pos — the reference's AT), the oracle
        # derives CSR, runs create_seed_matrix's region once on every core this process may use, and compares B entry by entry.
        full = None
        if not args.no_cpu_full:
            need = 40 * int(ms["nnz"]) + 120 * int(st_cold["nnz"]) + (4 << 30)      # host bytes: A twice + the oracle's B twice + the GPU's B
            avail = host_memory_available()
            if avail is not None and avail < need:
                full = {"skipped": "host memory: %d MB available, ~%d MB needed" % (avail >> 20, need >> 20)}
            else:
                try:
                    ncall = len(os.sched_getaffinity(0))
                except AttributeError:
                    ncall = os.cpu_count() or 1
                ncall = max(1, min(ncall, 64))
                torch.cuda.synchronize()
                eng.set_option("overlap_cold_calls", 1)
                stf = eng.create_seed_matrix()                                          # the B that is compared: one more cold call on the timed matrix
                v = eng.device_view()
                t0 = time.perf_counter()
                colptr = host_copy(v["a_colptr"], v["N"] + 1, np.uint32); csc = host_copy(v["a_csc"], v["Z"], np.uint64)
                of = po.Oracle(k, lo, up)
                of.set_csc(int(v["M"]), int(v["N"]), colptr, csc, ncall)
                t_load = time.perf_counter() - t0
                del colptr, csc
                t0 = time.perf_counter(); of.spgemm(ncall); t_all = time.perf_counter() - t0
                t0 = time.perf_counter()
                g_rowptr = host_copy(v["b_rowptr"], v["M"] + 1, np.int64); g_col = host_copy(v["b_col"], v["Y"], np.uint32)
                g_val = host_copy(v["b_val"], v["Y"], np.dtype([("q0", "<u4"), ("t0", "<u4"), ("q1", "<u4"), ("t1", "<u4"), ("numshared", "<i4")]))
                ndiff = of.compare_B(g_rowptr, g_col, g_val, ncall)
                t_cmp = time.perf_counter() - t0
                same_counts = bool(stf["nnz"] == of.stat("Y") and stf["products"] == of.stat("P") and stf["nnz_before_prune"] == of.stat("Yraw")
                                   and stf["nnz_upper"] == of.stat("nupper") and stf["max_numshared"] == of.stat("maxshared"))
                full = {"value": round(of.stat("Y") / t_all, 1), "unit": "overlap nnz/s", "cores": ncall, "seconds": round(t_all, 3),
                        "sample": "the WHOLE %s matrix (%d reads, nnz(A) = %d, P = %d products, Y = %d): the SpGEMM region once on %d host threads (OpenMP, dynamic rows), "
                                  "oracle/elba_oracle.c orc_spgemm, gcc -O3" % (args.workload, v["M"], v["Z"], of.stat("P"), of.stat("Y"), ncall),
                        "load_seconds": round(t_load, 3), "compare_seconds": round(t_cmp, 3),
                        "parity_vs_oracle_full": bool(ndiff == 0 and same_counts), "entries_that_differ": int(ndiff)}
                del of, g_rowptr, g_col, g_val
        if full is not None and "value" in full:
            cpu = dict(full, kind="port", one_core_on_a_sample=one_core)
        elif full is not None:
            cpu["whole_matrix"] = full
        # the k-mer stage and the whole reads -> B region on the host (reference timers src/main.cpp:191-282): the oracle's count_and_build is a
        # scalar port (1 core), timed on a third of the SpGEMM sample's genome so that it stays within ~10 s
        kp, ko, kl, _ = elba_amd.synth_reads(w["seed"], max(20000, w["genome"] // (div * 3)), w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                             repeat_families=w.get("repeats", (0, 0.0, 0))[0], repeat_fraction=w.get("repeats", (0, 0.0, 0))[1], repeat_len=w.get("repeats", (0, 0.0, 0))[2])
        o2 = po.Oracle(k, lo, up)
        t0 = time.perf_counter(); o2.count_and_build(kp, ko, kl); tk1 = time.perf_counter() - t0
        t0 = time.perf_counter(); o2.spgemm(1); ts1 = time.perf_counter() - t0
        one_k = {"instances_per_s": round(o2.stat("I") / tk1, 1), "seconds": round(tk1, 3), "cores": 1, "instances": int(o2.stat("I")),
                 "sample": "genome divided by %d (%d reads)" % (div * 3, len(kl))}
        one_e = {"overlap_nnz_per_s": round(o2.stat("Y") / (tk1 + ts1), 1), "seconds": round(tk1 + ts1, 3), "cores": 1, "overlap_nnz": int(o2.stat("Y"))}
        del o2, kp, ko, kl
        # ... and on every core this process may use (round 5: orc_count_and_build_mt — reads split over the threads, 256 value buckets sorted and counted
        # independently, the same result bit for bit: tests/test_oracle_golden.py), on the SpGEMM sample's reads (the genome divided by `div`): 64 bytes of
        # host memory per k-mer instance
        try:
            ncall = len(os.sched_getaffinity(0))
        except AttributeError:
            ncall = os.cpu_count() or 1
        ncall = max(1, min(ncall, 64))
        I_s = int(np.maximum(sl.astype(np.int64) - k + 1, 0).sum())
        avail = host_memory_available()
        if ncall > 1 and (avail is None or avail > 80 * I_s + (2 << 30)):
            o3 = po.Oracle(k, lo, up)
            t0 = time.perf_counter(); o3.count_and_build(sp, so, sl, ncall); tkn = time.perf_counter() - t0
            t0 = time.perf_counter(); o3.spgemm(ncall); tsn = time.perf_counter() - t0
            cpu["kmer_stage"] = {"instances_per_s": round(o3.stat("I") / tkn, 1), "seconds": round(tkn, 3), "cores": ncall, "instances": int(o3.stat("I")),
                                 "sample": "genome divided by %d (%d reads): oracle/elba_oracle.c orc_count_and_build_mt, gcc -O3 -fopenmp" % (div, len(sl)),
                                 "gpu_instances_per_s": kmer_stage["instances_per_s"], "one_core": one_k}
            cpu["end_to_end"] = {"overlap_nnz_per_s": round(o3.stat("Y") / (tkn + tsn), 1), "seconds": round(tkn + tsn, 3), "cores": ncall, "overlap_nnz": int(o3.stat("Y")),
                                 "gpu_overlap_nnz_per_s": end_to_end["overlap_nnz_per_s"], "one_core": one_e}
            del o3
        else:
            cpu["kmer_stage"] = dict(one_k, gpu_instances_per_s=kmer_stage["instances_per_s"], limitation="one core only: %s" % ("one core available" if ncall <= 1 else "host memory"))
            cpu["end_to_end"] = dict(one_e, gpu_overlap_nnz_per_s=end_to_end["overlap_nnz_per_s"])

    aux = None
    if args.aux and single and rank == 0:
        aux = run_aux(eng, packed, off, lens, k, lo, up, local_rank)

    if rank == 0:
        out = {
            "metric": "overlap nnz/sec (A·Aᵀ SpGEMM, SharedSeeds semiring, after Prune(numshared<=1))",
            "value": round(Y / (dt / steps), 1), "unit": "overlap nnz/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak" if (args.weak and world > 1) else "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "transport": None if dist is None else {
                "backend": "RCCL (torch.distributed backend nccl)" if backend == "nccl" else "REHEARSAL: %s over host copies, %d processes on %d device(s) — not a scaling measurement" % (backend, world, torch.cuda.device_count()),
                "world": world, "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None,
                "per_rank_overlap_nnz": [r["nnz"] for r in rank_phases] if rank_phases else None},
            "config": dict({"workload": args.workload, "reads": int(info["total_reads"]) if "total_reads" in info else int(info["nreads"]), "k": k, "lower": lo, "upper": up,
                            "genome": w["genome"] * (world if args.weak else 1), "depth": w["depth"], "error": w["error"], "kmer_instances": I_tot, "nnz_A": Z_tot,
                            "products": P, "overlap_nnz": Y, "algorithmic_bytes": abytes,
                            "step": "cold call: CSR + k-mer columns of A -> pruned CSR B, no plan, nothing remembered from earlier calls"}, **extra_cfg),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_vs_oracle_on_cpu_sample": parity,
            "parity_vs_oracle_full": (cpu or {}).get("parity_vs_oracle_full"),
            "counts_match_one_gpu": counts_ok, "counts": dict(zip(("kmer_instances", "nnz_A", "products", "overlap_nnz"), got_counts)),
            "counts_expected": None if expect is None else list(expect),
            "steady_state": steady,
            "kmer_stage": kmer_stage,
            "end_to_end": end_to_end,
            "spgemm_prep": prep,
            "from_triples": from_triples,
            "reference_stage_timers_ms": ref_timers,
            "rank_phases_ms": rank_phases,
            "phases_ms": {key: round(v, 4) for key, v in acc.items()},
            "mirror": mirror,
            "hbm_tier_forced": hbm_forced,
            "tiers": {key: int(st_cold[key]) for key in ("rows_lds", "rows_global", "rows_escalated", "nnz_before_prune", "passes") if key in st_cold},
            "aux_stages": aux,
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
        (dist.d if backend != "nccl" else dist).destroy_process_group()
    if counts_ok is False:
        # every rank holds the all-reduced counts: all of them leave with the same code
        if rank == 0:
            print("bench.py: counts %s differ from the one-GPU build's %s" % (got_counts, tuple(expect)), file=sys.stderr, flush=True)
        sys.exit(3)


def run_aux(eng, packed, off, lens, k, lo, up, device):
    """The stages on either side of the path, once each, outside every timed region: FASTA chunk -> 2-bit reads, x-drop
    alignment of every candidate pair, bad / contained read removal + transitive reduction.  Never takes the headline line down."""
    import elba_amd
    from elba_amd.capi import Engine
    out, errors = {}, {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as ex:               # noqa: BLE001
            errors[name] = "%s: %s" % (type(ex).__name__, ex)

    def ingest():
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
        e2 = Engine(k, lo, up, device=device)
        e2.set_reads_fasta(chunk, 0, recs)                     # warm-up (allocations)
        ist = e2.set_reads_fasta(chunk, 0, recs)
        gp, goff, glen = e2.export_reads(len(lens), ist["packed_bytes"])
        same = bool((gp[:ist["packed_bytes"]] == packed[:ist["packed_bytes"]]).all() and (glen == lens).all())
        io_bytes = ist["chunk_bytes"] + ist["packed_bytes"]
        e2.close()
        return {"fasta_bytes": int(ist["chunk_bytes"]), "ms_total_with_h2d": round(ist["ms_total"], 3), "ms_encode_kernel": round(ist["ms_encode"], 4),
                "frac_of_hbm_peak": round(io_bytes / max(1e-9, ist["ms_encode"] * 1e-3) / 1e9 / PEAK_GBS, 4), "equals_input_reads": same}

    def align():
        al = eng.align_seeds()
        return {"alignments": int(al["nalignments"]), "passed": int(al["passed"]), "cells": int(al["cells"]), "ms": round(al["ms_total"], 3),
                "gcups": round(al["cells"] / max(1e-9, al["ms_extend"] * 1e-3) / 1e9, 3)}

    def string_graph():
        eng.transitive_reduction()
        sg = eng.transitive_reduction()
        return {key: (round(v, 4) if isinstance(v, float) else int(v)) for key, v in sg.items()}

    guarded("ingest_stage", ingest)
    guarded("align_stage", align)
    if "align_stage" in out:
        guarded("string_graph_stage", string_graph)
    if errors:
        out["errors"] = errors
    return out


if __name__ == "__main__":
    main()
