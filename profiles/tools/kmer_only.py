#!/usr/bin/env python3
"""The k-mer stage of a bench.py workload alone (elba_count_kmers twice, nothing behind it) — for rocprofv3 kernel traces of TIMING EXPERIMENTS whose
results are wrong by construction (ELBA_X_* builds): nothing downstream ever reads what they leave.
usage: [ELBA_AMD_LIB=...] [ELBA_BENCH_OPTIONS=a=1,b=2] python3 profiles/tools/kmer_only.py [workload] [--matrix]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, numpy as np
import elba_amd
from elba_amd.capi import Engine
import bench
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "200k-long-reads"
w = bench.WORKLOADS[name]
rep = w.get("repeats", (0, 0.0, 0))
packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg_len"], w["sd_len"], error_rate=w["error"], min_len=w["min_len"],
                                               repeat_families=rep[0], repeat_fraction=rep[1], repeat_len=rep[2])
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("ELBA_BENCH_OPTIONS", "").split(",") if "=" in kv}
eng = Engine(w["k"], w["lower"], w["upper"], device=0, options=opts or None)
d_packed = torch.from_numpy(packed).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda(); d_len = torch.from_numpy(lens.view(np.int32)).cuda()
eng.set_reads_device(d_packed.data_ptr(), int(packed.size) - 16, d_off.data_ptr(), d_len.data_ptr(), len(lens))
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ks = eng.count_kmers()
    if "--matrix" in sys.argv: eng.create_kmer_matrix()
    torch.cuda.synchronize()
    print("pass", it, "wall_ms %.3f" % ((time.perf_counter() - t0) * 1e3), {k: ks[k] for k in ("ms_total", "ms_count", "ms_sort", "reliable", "entries")})
