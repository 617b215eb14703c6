"""`bench.py --gpus 2` as the driver launches it (python -m torch.distributed.run, one process per rank), on ONE GPU: RCCL refuses two ranks on
one device, so the collectives travel over gloo and host copies (ELBA_DIST_BACKEND=gloo, elba_amd.distributed.HostStagedDist).  Everything else
is the N > 1 path of bench.py as it stands: shards of one read set, the distributed build of A (src/KmerOps.cpp:117-151,244-274,371-375), the
step with its mirror exchange, the barrier-bracketed timing, MAX over ranks, ONE JSON line from rank 0.  The matrix must be the one-GPU matrix."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_as_processes_on_one_gpu_equals_the_one_gpu_line():
    env = dict(os.environ, ELBA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--workload", "ecsample30x-like", "--no-cpu-baseline", "--steady-steps", "0", "--no-accounting"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    c1 = j1["counts"]
    want = "%d,%d,%d,%d" % (c1["kmer_instances"], c1["nnz_A"], c1["products"], c1["overlap_nnz"])
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29571",
              os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common
    two = subprocess.run(launch, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, ELBA_BENCH_EXPECT=want))
    assert two.returncode == 0, (two.stdout[-2000:], two.stderr[-4000:])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["steps"] == 2 and j2["scaling"] == "strong" and "REHEARSAL" in j2["transport"]["backend"] and j2["transport"]["world"] == 2
    for key in ("reads", "kmer_instances", "nnz_A", "products", "overlap_nnz"):
        assert j2["config"][key] == j1["config"][key], key
    # the run validated itself against the one-GPU build's counts (bench.py: EXPECTED_COUNTS / ELBA_BENCH_EXPECT)
    assert j2["counts_match_one_gpu"] is True and j2["counts"] == c1 and j2["counts_expected"] == [int(x) for x in want.split(",")]
    assert len(j2["rank_phases_ms"]) == 2 and sum(r["nnz"] for r in j2["rank_phases_ms"]) == j1["config"]["overlap_nnz"]
    assert j2["transport"]["per_rank_overlap_nnz"] == [r["nnz"] for r in j2["rank_phases_ms"]]
    assert j2["value"] > 0 and j2["ms_per_step"] > 0
    # ... and a run whose counts are NOT the expected ones ends loudly: exit code 3 on every rank, the line says so
    wrong = "%d,%d,%d,%d" % (c1["kmer_instances"], c1["nnz_A"], c1["products"], c1["overlap_nnz"] + 1)
    bad = subprocess.run(launch[:8] + ["--master-port", "29572"] + launch[10:], capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, ELBA_BENCH_EXPECT=wrong))
    assert bad.returncode != 0, "a count mismatch must fail the run"
    bl = [l for l in bad.stdout.strip().splitlines() if l.startswith("{")]
    assert len(bl) == 1 and json.loads(bl[0])["counts_match_one_gpu"] is False
    assert "differ from the one-GPU build" in bad.stderr
