"""Helpers for the string-graph tests: hand-made Overlap records, random overlap graphs, and a pure-Python restatement of
src/main.cpp:305-312 + src/TransitiveReduction.cpp:3-90 with dense dict-of-dict matrices that follows the reference statement by
statement (loop included).  Independent of oracle/elba_oracle.c: it is what pins the C oracle on random inputs."""
import numpy as np

from oracle import pyoracle as po

INF = 2**31 - 1


def ov(direction, directionT, suffix, suffixT, passed=1, cq=0, ct=0, direction_none=False, **kw):
    o = np.zeros(1, dtype=po.OVERLAP_DTYPE)[0]
    o["direction"] = -1 if direction_none else direction
    o["directionT"] = -1 if direction_none else directionT
    o["suffix"] = suffix; o["suffixT"] = suffixT; o["passed"] = passed; o["containedQ"] = cq; o["containedT"] = ct
    for k, v in kw.items():
        o[k] = v
    return o


def random_overlaps(rng, M, density=0.3, p_fail=0.1, p_contained=0.02, p_nodir=0.03, suffix_range=3000):
    """Upper-triangular pairs in (row, col) order with random Overlap fields — directions, suffixes and flags are drawn independently
    (not from real alignments): every branch of the semiring and of the prunes gets exercised, which real reads rarely do."""
    rows, cols, vals = [], [], []
    for i in range(M):
        for j in range(i + 1, M):
            if rng.random() >= density:
                continue
            o = np.zeros(1, dtype=po.OVERLAP_DTYPE)[0]
            for f in ("begQ", "begT", "endQ", "endT"):
                o[f] = int(rng.integers(0, 20000))
            o["score"] = int(rng.integers(-1, 9000)); o["rc"] = int(rng.integers(0, 2)); o["kind"] = int(rng.integers(0, 5))
            if rng.random() < p_fail:
                o["direction"] = -1; o["directionT"] = -1
            else:
                o["passed"] = 1
                u = rng.random()
                if u < p_contained:
                    o["containedQ"] = 1; o["direction"] = -1; o["directionT"] = -1
                elif u < 2 * p_contained:
                    o["containedT"] = 1; o["direction"] = -1; o["directionT"] = -1
                elif u < 2 * p_contained + p_nodir:
                    o["direction"] = -1; o["directionT"] = int(rng.integers(0, 4))
                else:
                    o["direction"] = int(rng.integers(0, 4)); o["directionT"] = int(rng.integers(-1, 4))
                    o["suffix"] = int(rng.integers(-200, suffix_range)); o["suffixT"] = int(rng.integers(-200, suffix_range))
            rows.append(i); cols.append(j); vals.append(o)
    return np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals, dtype=po.OVERLAP_DTYPE) if vals else np.zeros(0, dtype=po.OVERLAP_DTYPE)


def transpose(o):
    """Overlap::Transpose, include/Overlap.hpp:43-69."""
    t = o.copy()
    t["begQ"], t["begT"] = o["begT"], o["begQ"]
    t["endQ"], t["endT"] = o["endT"], o["endQ"]
    t["suffix"], t["suffixT"] = o["suffixT"], o["suffix"]
    t["direction"], t["directionT"] = o["directionT"], o["direction"]
    t["containedQ"], t["containedT"] = o["containedT"], o["containedQ"]
    return t


def _multiply(e1, e2):
    """MinPlusSR::multiply (include/TransitiveReduction.hpp:88-104) on (direction, suffix, paths) triples."""
    out = [INF] * 4
    if e1[0] == -1 or e2[0] == -1:
        return out
    t1, h1, t2, h2 = (e1[0] >> 1) & 1, e1[0] & 1, (e2[0] >> 1) & 1, e2[0] & 1
    if t2 == h1:
        return out
    out[2 * t1 + h2] = e1[1] + e2[1]
    return out


def python_string_graph(M, rows, cols, vals, cutoff, fuzz):
    rows = [int(r) for r in rows]; cols = [int(c) for c in cols]
    # find_bad_reads (src/main.cpp:553-571)
    deg = [0] * M; pas = [0] * M
    for r, c, v in zip(rows, cols, vals):
        deg[r] += 1; deg[c] += 1
        if v["passed"]:
            pas[r] += 1; pas[c] += 1
    bad = [(pas[v] + 1) / (float(deg[v]) + 1) <= cutoff for v in range(M)]
    R = {(r, c): v for r, c, v in zip(rows, cols, vals) if v["passed"] and not bad[r] and not bad[c]}
    st = dict(bad_reads=sum(bad), edges_passed=len(R))
    # find_contained_reads (:573-583)
    cont = [False] * M
    for (r, c), v in R.items():
        if v["containedQ"]:
            cont[r] = True
        if v["containedT"]:
            cont[c] = True
    R = {(r, c): v for (r, c), v in R.items() if not cont[r] and not cont[c]}
    st.update(contained_reads=sum(cont), edges_kept=len(R))
    flags = [int(b) | (int(c) << 1) for b, c in zip(bad, cont)]
    # TransitiveReduction
    for (r, c), v in list(R.items()):
        R[(c, r)] = transpose(v)
    Rm = {key: (int(v["direction"]), int(v["suffix"])) for key, v in R.items()}
    rowsof = {}
    for (r, c) in Rm:
        rowsof.setdefault(r, []).append(c)
    P = dict(Rm)
    T = set()
    iters = 0
    first = None
    while True:
        prev = len(T)
        N = {}
        products = 0
        for (i, k), e1 in P.items():
            for j in rowsof.get(k, ()):
                prod = _multiply(e1, Rm[(k, j)])
                products += 1
                cur = N.get((i, j))
                N[(i, j)] = prod if cur is None else [min(a, b) for a, b in zip(cur, prod)]
        N = {key: p for key, p in N.items() if any(x < INF for x in p)}
        Ipat = set()
        for key, (d, s) in Rm.items():
            if key in N and d != -1 and s + fuzz >= N[key][d]:
                Ipat.add(key)
        if first is None:
            first = dict(products=products, nnzN=len(N), marked=len(Ipat))
        Ipat |= {(c, r) for (r, c) in Ipat}
        T |= Ipat
        P = {key: (-1, 0) for key in N}                 # entries of N come out of Overlap(): direction -1, suffix 0
        iters += 1
        if len(T) == prev:
            break
    S = [(r, c, v) for (r, c), v in R.items() if (r, c) not in T and v["direction"] != -1]
    S.sort(key=lambda t: (t[1], t[0]))
    st.update(first)
    st.update(removed=len(T), nnz=len(S), iterations=iters)
    return S, flags, st
