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
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29571",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, (two.stdout[-2000:], two.stderr[-4000:])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["steps"] == 2 and j2["scaling"] == "strong" and "REHEARSAL" in j2["transport"]
    for key in ("reads", "kmer_instances", "nnz_A", "products", "overlap_nnz"):
        assert j2["config"][key] == j1["config"][key], key
    assert len(j2["rank_phases_ms"]) == 2 and sum(r["nnz"] for r in j2["rank_phases_ms"]) == j1["config"]["overlap_nnz"]
    assert j2["value"] > 0 and j2["ms_per_step"] > 0
