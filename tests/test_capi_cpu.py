"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/*.h declares, rejects bad configurations,
and fails loudly (no fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import elba_amd
from elba_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(elba_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = elba_amd.load_library()
    names = _declared("elba_amd.h") + _declared("elba_synth.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(capi.EXPORTED_SYMBOLS)
    assert L.elba_abi_version() == 3


def test_struct_layouts_match_header():
    assert C.sizeof(capi.Seed) == 20            # sizeof(SharedSeeds) == 20 in the reference
    assert capi.SEED_DTYPE.itemsize == 20
    assert C.sizeof(capi.Cfg) == 32
    assert C.sizeof(capi.Dcsc) == 64
    assert C.sizeof(capi.OverlapStats) == 11 * 8 + 2 * 4 + 4 * 4
    assert capi.OVERLAP_DTYPE.itemsize == 36      # elba_overlap_t: 7 x int32 + 8 x int8
    assert C.sizeof(capi.AlignStats) == 6 * 8 + 2 * 4
    assert C.sizeof(capi.StringStats) == 10 * 8 + 2 * 4 + 2 * 4


def test_bad_configurations_are_rejected():
    L = elba_amd.load_library()
    h = C.c_void_p()
    for (k, lo, up, want) in [(16, 2, 8, 1), (2, 2, 8, 1), (97, 2, 8, 1), (17, 0, 8, 1), (17, 9, 8, 1), (17, 2, 70000, 1), (96, 2, 8, 1), (17, 1, 8, 6)]:
        cfg = capi.Cfg(k, lo, up, 0, 0, 0, 0)
        assert L.elba_ctx_create(C.byref(h), C.byref(cfg)) == want, (k, lo, up)
        assert not h.value
    assert L.elba_ctx_create(None, None) == 1
    assert L.elba_strerror(2).decode().startswith("no HIP device")


def test_no_gpu_means_loud_failure_not_fallback():
    try:
        e = elba_amd.Engine(17, 2, 8)
    except elba_amd.ElbaError as err:
        assert err.status == 2          # ELBA_ERR_NO_DEVICE
    else:
        e.close()                       # a GPU is present: creating a context must simply work


def test_null_context_calls_fail_cleanly():
    L = elba_amd.load_library()
    assert L.elba_count_kmers(None, None) == 1
    assert L.elba_create_seed_matrix(None, None) == 1
    L.elba_ctx_destroy(None)
    L.elba_free_dcsc(None)


def test_synth_generator_is_deterministic_and_shardable():
    a = elba_amd.synth_reads(7, 50000, 8, 3000, 500, error_rate=0.1)
    b = elba_amd.synth_reads(7, 50000, 8, 3000, 500, error_rate=0.1)
    assert (a[0] == b[0]).all() and (a[2] == b[2]).all()
    n = a[3]["nreads"]
    h = n // 2
    s0 = elba_amd.synth_reads(7, 50000, 8, 3000, 500, error_rate=0.1, first_read=0, num_reads=h)
    s1 = elba_amd.synth_reads(7, 50000, 8, 3000, 500, error_rate=0.1, first_read=h, num_reads=n - h)
    assert (np.concatenate([s0[2], s1[2]]) == a[2]).all()
    assert s0[0][: int(s0[1][-1])].tobytes() == a[0][: int(s0[1][-1])].tobytes()
    assert abs(a[3]["total_bases"] / 50000 - 8) < 1.0     # depth as asked


def test_synth_reads_overlap_in_the_oracle():
    """End-to-end sanity of the generator with the oracle: error-free overlapping reads share k-mers; strands are mixed."""
    from oracle import pyoracle as po
    packed, off, lens, info = elba_amd.synth_reads(3, 20000, 10, 2000, 300, error_rate=0.0)
    o = po.Oracle(17, 2, 30)
    o.count_and_build(packed, off, lens)
    o.spgemm(2)
    assert o.stat("Y") > 3 * info["nreads"]
    assert 0 < info["strand"].sum() < info["nreads"]
    # ground truth: two reads overlapping by >= 200 bases on the genome must be in B
    B = o.B()
    pos, ln = info["genome_pos"], lens.astype(np.int64)
    rows = np.repeat(np.arange(B["M"]), np.diff(B["rowptr"]))
    have = set(zip(rows.tolist(), B["col"].tolist()))
    missing = 0
    for i in range(min(60, len(pos))):
        for j in range(len(pos)):
            if i != j and min(pos[i] + ln[i], pos[j] + ln[j]) - max(pos[i], pos[j]) >= 200 and (i, j) not in have:
                missing += 1
    assert missing == 0
