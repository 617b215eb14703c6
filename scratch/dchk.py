import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, elba_amd, dist_sim
from elba_amd.distributed import DistributedOverlap, HipBackend, partition_by_bases
for genome in (400_000, 1_500_000, 4_640_000):
    reads = elba_amd.synth_reads(1, genome, 30.0, 8240.0, 2000.0, error_rate=0.15, min_len=1000)
    packed, off, lens, info = reads
    e = elba_amd.Engine(17, 2, 8); e.set_reads(packed, off, lens); ks = e.count_kmers(); ms = e.create_kmer_matrix(); st = e.create_seed_matrix(); e.close()
    def body(rank, h):
        d = DistributedOverlap(17, 2, 8, device=0, rank=0, world=1, dist=h, backend=HipBackend(17, 2, 8, 0))
        d.set_reads(packed, off, lens, 0, np.array([0, len(lens)]))
        k2, m2 = d.build_kmer_matrix(); s2 = d.create_seed_matrix(); d.be.e.close(); return k2, m2, s2
    k2, m2, s2 = dist_sim.run_ranks(1, body)[0]
    print(genome, "single:", ks["instances"], ks["reliable"], ks["entries"], st["nnz"], "| dist:", k2["instances"], k2["reliable"], k2["entries"], m2["nnz"], s2["nnz"], flush=True)
