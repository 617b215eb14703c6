"""Text formats of the reference around the path — host-side, for interop with the reference's own tools:

* SharedSeeds as the reference prints it: operator<< `{(q0,t0),(q1,t1),n}` (include/SharedSeeds.hpp:75-88; min(numshared, 2) seeds) and
  IOHandlerBrief `numstored<TAB>numshared` (:66-73) — the value column of B.mtx (ELBALogger::log_seed_matrix, src/ELBALogger.cpp:22-35,
  CombBLAS ParallelWriteMM with one-based indices).
* the `row col q0 t0 q1 t1` dump that the reference's test.py reads (test.py:42-52: two header lines, zero-based read indices), and
  test.py's check itself (:53-65): every stored seed must be the same k-mer in both reads, forward or reverse complement.
* Overlap as the reference prints it (include/Overlap.hpp:78-83) and the PAF lines of parallel_write_paf (src/main.cpp:514-551),
  including its `maplen` expression as written there (`end_T - end_T`, i.e. max(endQ - begQ, 0)).

CombBLAS's ParallelWriteMM itself is not in the reference tree (un-vendored): its header line is restated from the MatrixMarket
convention (`%%MatrixMarket matrix coordinate real general`, then `nrows ncols nnz`), entries column by column as the DCSC stores them.
"""
import numpy as np

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def seed_str(v):
    """SharedSeeds operator<< (include/SharedSeeds.hpp:75-88)."""
    n = int(v["numshared"])
    stored = min(2, n)
    parts = []
    if stored >= 1:
        parts.append("(%d,%d)," % (int(v["q0"]), int(v["t0"])))
    if stored >= 2:
        parts.append("(%d,%d)," % (int(v["q1"]), int(v["t1"])))
    return "{" + "".join(parts) + "%d}" % n


def seed_brief_str(v):
    """SharedSeeds::IOHandlerBrief (include/SharedSeeds.hpp:66-73)."""
    n = int(v["numshared"])
    return "%d\t%d" % (min(2, n), n)


def write_seed_matrix_mm(path, dcsc, nrows, handler=seed_str):
    """B.mtx as ELBALogger::log_seed_matrix writes it (one-based indices; entries in DCSC order: column by column, rows ascending)."""
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (nrows, nrows, dcsc["nnz"]))
        for c in range(dcsc["nzc"]):
            col = int(dcsc["jc"][c]) + 1
            for e in range(int(dcsc["cp"][c]), int(dcsc["cp"][c + 1])):
                f.write("%d\t%d\t%s\n" % (int(dcsc["ir"][e]) + 1, col, handler(dcsc["numx"][e])))


def write_testpy_dump(path, csr):
    """The file test.py opens as B.mtx (test.py:42-52): two header lines, then `row col q0 t0 q1 t1`, zero-based."""
    rows = np.repeat(np.arange(csr["M"]), np.diff(csr["rowptr"]))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n%d %d %d\n" % (csr["M"], csr["M"], csr["Y"]))
        for e in range(csr["Y"]):
            v = csr["val"][e]
            f.write("%d %d %d %d %d %d\n" % (rows[e], int(csr["col"][e]), int(v["q0"]), int(v["t0"]), int(v["q1"]), int(v["t1"])))


def check_seed_dump(path, seqs, k):
    """test.py:53-67 — returns (correct, incorrect)."""
    correct = incorrect = 0
    with open(path) as f:
        next(f); next(f)
        for line in f:
            t = tuple(int(v) for v in line.split())
            iq, it = t[0], t[1]
            if iq == it:
                continue
            sq, st = seqs[iq], seqs[it]
            for bq, bt in ((t[2], t[3]), (t[4], t[5])):
                a, b = sq[bq:bq + k], st[bt:bt + k]
                if a != b and a != b.translate(_COMP)[::-1]:
                    incorrect += 1
                else:
                    correct += 1
    return correct, incorrect


def overlap_str(o, lenQ, lenT):
    """Overlap operator<< (include/Overlap.hpp:78-83)."""
    return "%d\t%d\t%d\t%s\t%d\t%d\t%d\t%d\t%d\t%d" % (lenQ, int(o["begQ"]), int(o["endQ"]), "-" if o["rc"] else "+", lenT, int(o["begT"]), int(o["endT"]),
                                                     int(o["score"]), int(o["direction"]), int(o["suffix"]))


def paf_line(o, nameQ, nameT, lenQ, lenT):
    """One line of parallel_write_paf (src/main.cpp:536-540)."""
    maplen = max(int(o["endQ"]) - int(o["begQ"]), int(o["endT"]) - int(o["endT"]))      # as written in the reference (:536)
    return "%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t255\t%d" % (nameQ, lenQ, int(o["begQ"]), int(o["endQ"]), "-" if o["rc"] else "+", nameT, lenT,
                                                                 int(o["begT"]), int(o["endT"]), int(o["score"]), maplen, int(o["passed"]))


def write_paf(path, overlaps, names, lens, dcsc_order=False):
    """dcsc_order: walk the triples as parallel_write_paf walks the local DCSC (columns ascending, rows ascending within a column)
    instead of in the order given."""
    idx = range(overlaps["n"])
    if dcsc_order:
        idx = np.lexsort((overlaps["rows"], overlaps["cols"])).tolist()
    with open(path, "w") as f:
        for a in idx:
            i, j = int(overlaps["rows"][a]), int(overlaps["cols"][a])
            f.write(paf_line(overlaps["vals"][a], names[i], names[j], int(lens[i]), int(lens[j])) + "\n")
