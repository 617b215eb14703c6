"""Full-size property check of the 200 k-read workload (BASELINE config 3 on ONE GPU): run by hand on the GPU box (about 45 s),
not collected by pytest.  usage: python tests/fullsize_200k_check.py"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, elba_amd
from oracle import pyoracle as po
t=time.time()
packed, off, lens, info = elba_amd.synth_reads(2, 66_700_000, 30.0, 10000.0, 1500.0, error_rate=0.15, min_len=1000)
print("gen", time.time()-t, info["nreads"], flush=True)
e = elba_amd.Engine(17, 2, 8); e.set_reads(packed, off, lens)
ks = e.count_kmers(); ms = e.create_kmer_matrix(); st = e.create_seed_matrix()
print(ks, ms, st, flush=True)
assert st["nnz"] == st["nnz_diag"] + 2 * st["nnz_upper"]
h = e.kmer_histogram(); assert int((h*np.arange(len(h))**2).sum()) == st["products"] and h.sum() == ks["reliable"]
B = e.export_csr(); M = B["M"]
rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(B["rowptr"])); cols = B["col"]
assert ((np.diff(cols) > 0) | (np.diff(rows) > 0)).all()
key = rows * M + cols; tkey = cols * M + rows
order = np.argsort(tkey, kind="stable")
assert (tkey[order] == key).all(), "pattern not symmetric"
assert (B["val"]["numshared"][order] == B["val"]["numshared"]).all()
L = po.lib(); rng = np.random.default_rng(0); bad = 0
for x in rng.choice(B["Y"], size=20000, replace=False):
    i, j, v = int(rows[x]), int(cols[x]), B["val"][x]
    for (q, tt) in ((v["q0"], v["t0"]), (v["q1"], v["t1"])):
        bad += not L.orc_seed_is_valid(packed.ctypes.data + int(off[i]), int(lens[i]), packed.ctypes.data + int(off[j]), int(lens[j]), int(q), int(tt), 17)
print("bad seeds", bad, "OK" if bad == 0 else "FAIL", flush=True)
