"""Determinism stress (MI355X): many cold overlap calls on several shapes — short reads with long plain-CSC columns, a config-2-like set, a dense
set, tiny reads, long reads, the table-overflow matrix (escalation, HBM tier), a dense matrix with thousands of partners per row — every call must
reproduce the first one (statistics every call, a hash of B every tenth).  How the row-id race of round 3 was narrowed down; run after any change
to the numeric kernel's synchronisation:  python3 profiles/tools/stress_determinism.py"""
import sys, os, hashlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, elba_amd
def bhash(B):
    h = hashlib.sha1(); h.update(np.ascontiguousarray(B["rowptr"]).tobytes()); h.update(np.ascontiguousarray(B["col"]).tobytes()); h.update(np.ascontiguousarray(B["val"]).tobytes()); return h.hexdigest()[:12]
cases = [
 ("short reads, CSC columns (U=70)", dict(seed=114, genome=300000, depth=25.0, avg=400.0, sd=100.0, err=0.02, minlen=66), (25, 2, 70), 150),
 ("config-2 like", dict(seed=1, genome=1500000, depth=30.0, avg=8000.0, sd=2000.0, err=0.15, minlen=1000), (17, 2, 8), 60),
 ("dense small", dict(seed=91, genome=200000, depth=35.0, avg=3000.0, sd=600.0, err=0.01, minlen=500), (17, 2, 40), 60),
 ("tiny reads", dict(seed=7, genome=50000, depth=20.0, avg=120.0, sd=30.0, err=0.05, minlen=30), (15, 2, 12), 150),
 ("long reads few rows", dict(seed=9, genome=400000, depth=40.0, avg=20000.0, sd=3000.0, err=0.10, minlen=5000), (17, 2, 8), 60),
]
for name, w, (k, lo, up), reps in cases:
    packed, off, lens, _ = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg"], w["sd"], error_rate=w["err"], min_len=w["minlen"])
    e = elba_amd.Engine(k, lo, up); e.set_reads(packed, off, lens); e.count_kmers(); ms = e.create_kmer_matrix()
    e.set_option("overlap_cold_calls", 1)
    ref = None; refh = None; bad = 0
    for it in range(reps):
        st = e.create_seed_matrix()
        key = (st["nnz"], st["products"], st["nnz_upper"], st["max_numshared"], st["nnz_before_prune"], st["nnz_diag"])
        if ref is None: ref = key
        if key != ref: bad += 1; print("   call", it, key, "!=", ref, flush=True)
        if it % 10 == 0:
            h = bhash(e.export_csr())
            if refh is None: refh = h
            if h != refh: bad += 1; print("   call", it, "B hash differs", flush=True)
    print("%-36s reads %6d nnz(A) %9d Y %9d : %d calls, %d deviations  tiers lds/global/escalated %d/%d/%d" % (name, len(lens), ms["nnz"], ref[0], reps, bad, st["rows_lds"], st["rows_global"], st["rows_escalated"]), flush=True)
    e.close()

# ---- escalation / HBM tier / dense escalation
M, rng = 14000, np.random.default_rng(12)
rows, cols, vals = [], [], []
ncol = 0
for c in range(2):
    r = np.arange(7000); rows.append(r); cols.append(np.full(len(r), ncol)); vals.append(rng.integers(0, 5000, len(r))); ncol += 1
for c in range(2):
    r = np.arange(M); rows.append(r); cols.append(np.full(len(r), ncol)); vals.append(rng.integers(0, 5000, len(r))); ncol += 1
for b in range(0, 3000, 500):
    for c in range(2):
        r = 7000 + np.arange(b, b + 500) % 3500; rows.append(r); cols.append(np.full(len(r), ncol)); vals.append(rng.integers(0, 5000, len(r))); ncol += 1
rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
for opts in ({}, {"no_pay": 1}):
    e = elba_amd.Engine(17, 2, 8, options=opts); e.set_kmer_matrix(M, ncol, rows, cols, vals); e.set_option("overlap_cold_calls", 1)
    ref = None; bad = 0
    for it in range(30):
        st = e.create_seed_matrix()
        key = (st["nnz"], st["products"], st["nnz_upper"], st["max_numshared"], st["nnz_before_prune"], st["nnz_diag"], st["rows_global"])
        ref = ref or key
        if key != ref: bad += 1; print("  call", it, key, ref)
    print(opts, "30 calls", bad, "deviations", ref, "escalated", st["rows_escalated"], flush=True); e.close()
# dense matrix with escalation (thousands of partners)
M2, ncol2, L = 6000, 20000, 40
rng = np.random.default_rng(M2); rows, cols, vals = [], [], []
for c in range(ncol2):
    r = np.sort(rng.choice(M2, L, replace=False)); rows.append(r); cols.append(np.full(L, c)); vals.append(rng.integers(0, 60000, L))
rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals).astype(np.uint32)
e = elba_amd.Engine(17, 2, 64); e.set_kmer_matrix(M2, ncol2, rows, cols, vals); e.set_option("overlap_cold_calls", 1)
ref = None; bad = 0
for it in range(20):
    st = e.create_seed_matrix()
    key = (st["nnz"], st["products"], st["nnz_upper"], st["max_numshared"], st["nnz_before_prune"])
    ref = ref or key
    if key != ref: bad += 1; print("  call", it, key, ref)
print("dense escalation 20 calls", bad, "deviations", ref, st["rows_escalated"], st["rows_global"]); e.close()
