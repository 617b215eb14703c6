"""Text formats around the path (elba_amd/formats.py): the reference's SharedSeeds / Overlap printers, the dump its test.py reads and
test.py's seed check, driven by the oracle's B and overlaps (CPU only)."""
import os

import numpy as np

import util
from elba_amd import formats as fm
from oracle import pyoracle as po

G = util.GOLDEN


def _oracle(name="small_err", k=17, lo=2, up=8):
    seqs = util.read_fasta(os.path.join(G, name + ".fa"))
    buf, off, lens = po.pack_reads(seqs)
    o = po.Oracle(k, lo, up); o.count_and_build(buf, off, lens); o.spgemm(1)
    return seqs, buf, off, lens, o


def test_shared_seeds_printer_follows_operator_stream():
    v = np.zeros(3, dtype=po.SEED_DTYPE)
    v[0] = (11, 22, 33, 44, 7); v[1] = (5, 6, 0, 0, 1); v[2] = (0, 0, 0, 0, 0)
    assert fm.seed_str(v[0]) == "{(11,22),(33,44),7}"          # include/SharedSeeds.hpp:75-88
    assert fm.seed_str(v[1]) == "{(5,6),1}"                     # min(numshared, 2) seeds
    assert fm.seed_str(v[2]) == "{0}"
    assert fm.seed_brief_str(v[0]) == "2\t7" and fm.seed_brief_str(v[1]) == "1\t1"


def test_seed_dump_passes_the_reference_test_py_check(tmp_path):
    seqs, buf, off, lens, o = _oracle()
    B = o.B()
    p = str(tmp_path / "B.mtx")
    fm.write_testpy_dump(p, B)
    correct, incorrect = fm.check_seed_dump(p, [s.upper() for s in seqs], 17)
    offdiag = int((np.repeat(np.arange(B["M"]), np.diff(B["rowptr"])) != B["col"]).sum())
    assert incorrect == 0 and correct == 2 * offdiag
    # a corrupted seed is caught
    lines = open(p).read().split("\n")
    t = lines[3].split(); t[2] = str(int(t[2]) + 1); lines[3] = " ".join(t)
    open(p, "w").write("\n".join(lines))
    assert fm.check_seed_dump(p, [s.upper() for s in seqs], 17)[1] >= 1


def test_matrix_market_of_B_is_column_major_one_based(tmp_path):
    seqs, buf, off, lens, o = _oracle("small_clean")
    d = o.export_dcsc(0, o.stat("M"), 0, o.stat("M"))
    p = str(tmp_path / "B.mtx")
    fm.write_seed_matrix_mm(p, d, o.stat("M"))
    L = open(p).read().split("\n")
    assert L[0].startswith("%%MatrixMarket") and L[1] == "%d %d %d" % (o.stat("M"), o.stat("M"), o.stat("Y"))
    first = L[2].split("\t")
    assert int(first[1]) == int(d["jc"][0]) + 1 and int(first[0]) == int(d["ir"][0]) + 1 and first[2] == fm.seed_str(d["numx"][0])
    assert len([x for x in L[2:] if x]) == o.stat("Y")


def test_overlap_and_paf_lines(tmp_path):
    seqs, buf, off, lens, o = _oracle()
    rows, cols, ov, _ = o.align_upper(buf, off, lens)
    a = int(np.nonzero(ov["passed"])[0][0])
    i, j = int(rows[a]), int(cols[a])
    s = fm.overlap_str(ov[a], int(lens[i]), int(lens[j])).split("\t")
    assert len(s) == 10 and s[3] in "+-" and int(s[7]) == int(ov[a]["score"])          # include/Overlap.hpp:78-83
    names = ["r%d" % x for x in range(len(seqs))]
    p = str(tmp_path / "out.paf")
    fm.write_paf(p, dict(n=len(rows), rows=rows, cols=cols, vals=ov), names, lens)
    t = open(p).read().split("\n")[a].split("\t")
    assert t[0] == names[i] and t[5] == names[j] and int(t[9]) == int(ov[a]["score"]) and t[11] == "255" and t[12] == "1"
    assert int(t[10]) == max(int(ov[a]["endQ"]) - int(ov[a]["begQ"]), 0)                # the reference's maplen expression (src/main.cpp:536)


def test_string_paf_lists_the_reduced_graph_in_the_reference_order(tmp_path):
    """parallel_write_paf(*S, ...) (src/main.cpp:315): S's entries column by column; a line below the diagonal carries the transposed
    record (query = the row's read)."""
    seqs, buf, off, lens, o = _oracle("small_clean")
    rows, cols, ov, _ = o.align_upper(buf, off, lens)
    S, flags, st = po.string_graph(len(lens), rows, cols, ov, cutoff=0.0)
    assert S["n"] >= 10 and st["removed"] > 0
    names = ["r%d" % x for x in range(len(seqs))]
    p = str(tmp_path / "out.string.paf")
    fm.write_paf(p, S, names, lens)
    L = [x.split("\t") for x in open(p).read().split("\n") if x]
    assert len(L) == S["n"] == st["nnz"]
    keys = [(int(c), int(r)) for r, c in zip(S["rows"], S["cols"])]
    assert keys == sorted(keys)
    for a, t in enumerate(L):
        i, j = int(S["rows"][a]), int(S["cols"][a])
        assert t[0] == names[i] and int(t[1]) == int(lens[i]) and t[5] == names[j] and int(t[6]) == int(lens[j]) and t[12] == "1"
        assert int(t[2]) == int(S["vals"][a]["begQ"]) and int(t[7]) == int(S["vals"][a]["begT"])
