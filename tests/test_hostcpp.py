"""The C++ host mirror (elba_amd/hostcpp/elba_host.hpp): reference-named functions over the C ABI.  The self-test binary runs the front
half of the reference's main() and PairwiseAlignment's DCSC walk; on the GPU its output must match the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import util
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "elba_amd", "hostcpp", "test_host_mirror")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "elba_amd", "hostcpp")], stdout=subprocess.DEVNULL)


def test_mirror_builds_and_fails_loudly_without_gpu():
    _build()
    p = subprocess.run([BIN, os.path.join(util.GOLDEN, "small_err.fa"), "17", "2", "8"], capture_output=True, text=True)
    if p.returncode == 3:
        assert "no HIP device" in p.stderr          # no silent CPU path
    else:
        assert p.returncode == 0 and json.loads(p.stdout)["reads"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,idx", [("small_err", 0), ("small_clean", 1)])
def test_mirror_matches_oracle_on_gpu(name, idx):
    if not os.path.exists(BIN):
        _build()
    m = util.golden_meta()[name][idx]
    k, lo, up = m["k"], m["lower"], m["upper"]
    fa = os.path.join(util.GOLDEN, name + ".fa")
    p = subprocess.run([BIN, fa, str(k), str(lo), str(up)], capture_output=True, text=True, check=True)
    got = json.loads(p.stdout)
    packed, off, lens = po.pack_reads(util.read_fasta(fa))
    o = po.Oracle(k, lo, up)
    o.count_and_build(packed, off, lens)
    o.spgemm(1)
    B = o.B()
    rows = np.repeat(np.arange(B["M"], dtype=np.uint64), np.diff(B["rowptr"]))
    cols = B["col"].astype(np.uint64)
    up_mask = rows < cols
    v = B["val"][up_mask]
    chk = (v["q0"].astype(np.uint64) * np.uint64(1000003) + v["t0"].astype(np.uint64) + v["numshared"].astype(np.uint64) * np.uint64(7919)
           + rows[up_mask] * np.uint64(31) + cols[up_mask])
    with np.errstate(over="ignore"):
        checksum = int(chk.sum(dtype=np.uint64))
    ar, ac, ov, _ = o.align_upper(packed, off, lens)
    u = lambda a, dt=np.uint32: a.astype(dt).astype(np.uint64)
    with np.errstate(over="ignore"):
        achk = (u(ov["score"]) * np.uint64(1000003) + u(ov["begQ"]) * np.uint64(31) + u(ov["begT"]) * np.uint64(37) + u(ov["endQ"]) * np.uint64(41) + u(ov["endT"]) * np.uint64(43)
                + ov["rc"].astype(np.uint64) * np.uint64(7) + u(ov["direction"], np.uint8) * np.uint64(131) + u(ov["suffix"]) * np.uint64(8191)
                + ar.astype(np.uint64) * np.uint64(3) + ac.astype(np.uint64))
        achk = int(achk.sum(dtype=np.uint64))
    S, flags, sst = po.string_graph(len(lens), ar, ac, ov, cutoff=0.65, fuzz=1000)
    with np.errstate(over="ignore"):
        sv = S["vals"]
        schk = (S["rows"].astype(np.uint64) * np.uint64(1000003) + S["cols"].astype(np.uint64) * np.uint64(31) + u(sv["direction"], np.uint8) * np.uint64(131)
                + u(sv["suffix"]) * np.uint64(8191) + u(sv["begQ"]) * np.uint64(37) + u(sv["endT"]) * np.uint64(43) + lens[S["rows"]].astype(np.uint64) * np.uint64(3)
                + lens[S["cols"]].astype(np.uint64) + np.arange(S["n"], dtype=np.uint64) * np.uint64(7))
        schk = int(schk.sum(dtype=np.uint64))
    want = {"reads": m["M"], "nnzA": m["Z"], "kmers": m["N"], "nnzB": m["Y"], "candidates": o.stat("nupper"), "checksum": checksum,
            "alignments": len(ar), "passed": int(ov["passed"].sum()), "align_checksum": achk, "ingest_equal": -1,
            "bad_reads": sst["bad_reads"], "contained_reads": sst["contained_reads"], "string_nnz": S["n"], "string_checksum": schk}
    assert got == want
    # the same run fed through the C++ FastaIndex mirror and the GPU encoder (reads.fa.fai next to a copy of the FASTA)
    import shutil, tempfile
    from elba_amd import fasta as efa
    with tempfile.TemporaryDirectory() as td:
        fa2 = os.path.join(td, "reads.fa")
        shutil.copy(fa, fa2)
        efa.write_fai(fa2)
        p2 = subprocess.run([BIN, fa2, str(k), str(lo), str(up), "fai"], capture_output=True, text=True, check=True)
    want["ingest_equal"] = 1
    assert json.loads(p2.stdout) == want


@pytest.mark.gpu
def test_cpp_writers_equal_the_python_ones_fed_from_the_oracle(tmp_path):
    """SURVEY.md §8f-4 on the C++ side (elba_host.hpp: log_seed_matrix, parallel_write_paf over the GPU's B, R and S) against elba_amd/formats.py
    driven by the ORACLE's matrices: B.mtx in SharedSeeds notation, overlap.paf and string.paf in the order the reference walks its DCSC."""
    from elba_amd import formats as fm
    if not os.path.exists(BIN):
        _build()
    m = util.golden_meta()["small_err"][0]
    k, lo, up = m["k"], m["lower"], m["upper"]
    fa = os.path.join(util.GOLDEN, "small_err.fa")
    pfx = str(tmp_path) + os.sep
    subprocess.run([BIN, fa, str(k), str(lo), str(up)], capture_output=True, text=True, check=True, env=dict(os.environ, ELBA_OUT_PREFIX=pfx))
    packed, off, lens = po.pack_reads(util.read_fasta(fa))
    o = po.Oracle(k, lo, up); o.count_and_build(packed, off, lens); o.spgemm(1)
    M = m["M"]
    fm.write_seed_matrix_mm(pfx + "o_B.mtx", o.export_dcsc(0, M, 0, M), M)
    assert open(pfx + "B.mtx").read() == open(pfx + "o_B.mtx").read()
    names = ["read%d" % r for r in range(M)]
    ar, ac, ov, _ = o.align_upper(packed, off, lens)
    fm.write_paf(pfx + "o_overlap.paf", dict(n=len(ar), rows=ar, cols=ac, vals=ov), names, lens, dcsc_order=True)
    assert open(pfx + "overlap.paf").read() == open(pfx + "o_overlap.paf").read() and len(ar) > 0
    S, flags, sst = po.string_graph(len(lens), ar, ac, ov, cutoff=0.65, fuzz=1000)
    fm.write_paf(pfx + "o_string.paf", S, names, lens, dcsc_order=True)
    assert open(pfx + "string.paf").read() == open(pfx + "o_string.paf").read()


DIST_BIN = os.path.join(ROOT, "elba_amd", "hostcpp", "test_host_dist")


def test_rccl_host_builds_and_fails_loudly_without_gpu():
    _build()
    p = subprocess.run([DIST_BIN, os.path.join(util.GOLDEN, "small_err.fa"), "17", "2", "8"], capture_output=True, text=True)
    assert p.returncode in (0, 3) and (p.returncode == 0 or "no HIP device" in p.stderr)


@pytest.mark.gpu
def test_rccl_host_world_of_one_matches_oracle():
    """elba_host_dist.hpp (C++ over RCCL: ncclAllReduce / ncclAllGather / grouped ncclSend + ncclRecv) end to end with one rank: the rows of
    B it computes must be the oracle's (count, and a checksum over every entry's row, column, both seeds and numshared)."""
    if not os.path.exists(DIST_BIN):
        _build()
    fa = os.path.join(util.GOLDEN, "small_err.fa")
    p = subprocess.run([DIST_BIN, fa, "17", "2", "8"], capture_output=True, text=True, check=True)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    # the same through the step with one host synchronisation (elba_set_stream / elba_seed_matrix_send / _recv, equal-size ncclSend / ncclRecv)
    p2 = subprocess.run([DIST_BIN, fa, "17", "2", "8"], capture_output=True, text=True, check=True, env=dict(os.environ, ELBA_DIST_SLOTS="1"))
    got2 = json.loads(p2.stdout.strip().splitlines()[-1])
    assert (got2["nnzB"], got2["products"], got2["checksum"]) == (got["nnzB"], got["products"], got["checksum"])
    packed, off, lens = po.pack_reads(util.read_fasta(fa))
    o = po.Oracle(17, 2, 8); o.count_and_build(packed, off, lens); o.spgemm(1)
    B = o.B(); v = B["val"]
    rows = np.repeat(np.arange(B["M"], dtype=np.uint64), np.diff(B["rowptr"]))
    with np.errstate(over="ignore"):
        chk = (v["q0"].astype(np.uint64) * np.uint64(1000003) + v["t0"].astype(np.uint64) + v["q1"].astype(np.uint64) * np.uint64(7) + v["t1"].astype(np.uint64) * np.uint64(13)
               + v["numshared"].astype(np.uint64) * np.uint64(7919) + rows * np.uint64(31) + B["col"].astype(np.uint64))
        chk = int(chk.sum(dtype=np.uint64))
    assert (got["rows"], got["nnzB"], got["products"], got["owned_kmers"], got["owned_entries"], got["kmers_total"]) == (B["M"], B["Y"], o.stat("P"), o.stat("N"), o.stat("Z"), o.stat("N"))
    assert got["checksum"] == chk
