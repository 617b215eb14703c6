"""Full-size property checks of the larger BASELINE.json configurations restated on ONE GPU (run by hand on the GPU box; not collected
by pytest: each takes from 20 s to a few minutes).  usage: python tests/fullsize_check.py CONFIG [--align]

  200k-long-reads      config 3 as written: 200 k reads x 10 kb, 30x, 15 % error, k=17, L=2, U=8 (I = 2.0 G instances)
  celegans-hifi-half   config 4 at half the genome (50 Mb, 40x, N(15000, 2000), 0.5 % error, k=17, L=2, U=4: I = 2.0 G) — the whole 100 Mb set
                       has 4 G instances, past the 32-bit instance index of one context: it needs the 8-GPU path the config names
  dense-repeats-20th   config 5 at 1/25 of the genome (20 Mb with 5 % of it in 20 repeat families, 40x, 10 kb, 1 % error, U=35): the
                       LDS-overflow / spill stress at a size one GPU holds

Checked: Y = diag + 2 * upper; sum_k c_k^2 = products; columns ascending; pattern and numshared symmetric; 20 000 sampled seeds are genuine
shared k-mers (the reference's test.py:57-65); with --align: the GPU x-drop of 300 sampled pairs equals the oracle's.
"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, elba_amd
from oracle import pyoracle as po

CONFIGS = {
    "200k-long-reads": dict(seed=2, genome=66_700_000, depth=30.0, avg=10000.0, sd=1500.0, err=0.15, min_len=1000, k=17, L=2, U=8, rep=(0, 0.0, 0)),
    "celegans-hifi-half": dict(seed=3, genome=50_000_000, depth=40.0, avg=15000.0, sd=2000.0, err=0.005, min_len=1000, k=17, L=2, U=4, rep=(0, 0.0, 0)),
    "dense-repeats-20th": dict(seed=4, genome=20_000_000, depth=40.0, avg=10000.0, sd=1000.0, err=0.01, min_len=1000, k=17, L=2, U=35, rep=(20, 0.05, 5000)),
}
name = sys.argv[1] if len(sys.argv) > 1 else "200k-long-reads"
w = CONFIGS[name]
t = time.time()
packed, off, lens, info = elba_amd.synth_reads(w["seed"], w["genome"], w["depth"], w["avg"], w["sd"], error_rate=w["err"], min_len=w["min_len"],
                                               repeat_families=w["rep"][0], repeat_fraction=w["rep"][1], repeat_len=w["rep"][2])
print(name, "gen %.1f s" % (time.time() - t), info["nreads"], "reads", flush=True)
k = w["k"]
e = elba_amd.Engine(k, w["L"], w["U"]); e.set_reads(packed, off, lens)
ks = e.count_kmers(); ms = e.create_kmer_matrix(); st = e.create_seed_matrix()
for _ in range(3):
    st = e.create_seed_matrix()
print(ks, ms, st, flush=True)
print("steady-state step: %.3f ms total, %.3f numeric, %.3f finalize -> %.2f G nnz/s; roofline frac (numeric) %.3f" % (
    st["ms_total"], st["ms_numeric"], st["ms_finalize"], st["nnz"] / st["ms_total"] / 1e6, st["algorithmic_bytes"] / (st["ms_numeric"] * 1e-3) / 8e12), flush=True)
assert st["nnz"] == st["nnz_diag"] + 2 * st["nnz_upper"]
h = e.kmer_histogram(w["U"] + 2); assert int((h * np.arange(len(h)) ** 2).sum()) == st["products"] and h.sum() == ks["reliable"]
B = e.export_csr(); M = B["M"]
rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(B["rowptr"])); cols = B["col"]
assert ((np.diff(cols) > 0) | (np.diff(rows) > 0)).all()
key = rows * M + cols; tkey = cols * M + rows
order = np.argsort(tkey, kind="stable")
assert (tkey[order] == key).all(), "pattern not symmetric"
assert (B["val"]["numshared"][order] == B["val"]["numshared"]).all()
L = po.lib(); rng = np.random.default_rng(0); bad = 0
for x in rng.choice(B["Y"], size=min(20000, B["Y"]), replace=False):
    i, j, v = int(rows[x]), int(cols[x]), B["val"][x]
    for (q, tt) in ((v["q0"], v["t0"]), (v["q1"], v["t1"])):
        bad += not L.orc_seed_is_valid(packed.ctypes.data + int(off[i]), int(lens[i]), packed.ctypes.data + int(off[j]), int(lens[j]), int(q), int(tt), k)
print("bad seeds", bad, "OK" if bad == 0 else "FAIL", flush=True)
assert bad == 0
if "--align" in sys.argv:
    a = e.align_seeds(); g = e.export_overlaps()
    print(a, "-> %.1f GCUPS" % (a["cells"] / (a["ms_extend"] * 1e-3) / 1e9), flush=True)
    pick = rng.choice(g["n"], size=min(300, g["n"]), replace=False); nbad = 0
    for x in pick:
        i, j = int(g["rows"][x]), int(g["cols"][x])
        eidx = int(B["rowptr"][i]) + int(np.searchsorted(cols[int(B["rowptr"][i]):int(B["rowptr"][i + 1])], j))
        want = po.xdrop(packed[int(off[i]):], int(lens[i]), packed[int(off[j]):], int(lens[j]), int(B["val"][eidx]["q0"]), int(B["val"][eidx]["t0"]), k)
        v = g["vals"][x]
        got = (int(v["score"]) if v["score"] != -1 or v["endQ"] else -1, int(v["begQ"]), int(v["endQ"]), int(v["begT"]), int(v["endT"]), int(v["score"]), int(v["rc"]), int(v["kind"]))
        nbad += got[1:] != want[1:]
    print("x-drop sample mismatches", nbad, "OK" if nbad == 0 else "FAIL", flush=True)
    assert nbad == 0
print("ALL OK", flush=True)
