// test_host_dist.cpp — the multi-GPU overlap build driven from C++ over RCCL (elba_host_dist.hpp): this rank's shard of the reads ->
// value-range owners -> all-to-all #1 -> exact count -> k-mer ids by exclusive scan -> all-to-all #2 (column panels) -> this rank's rows of B.
// Usage: test_host_dist reads.fa K LOWER UPPER [RANK SIZE ID_FILE]     (one process per GPU; RANK's GPU = RANK unless ELBA_DEVICE is set)
// Prints one JSON line per rank: rows, nnz of its rows of B and a checksum over (row, col, seeds[0], numshared) of every entry.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include "elba_host_dist.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s reads.fa K LOWER UPPER [RANK SIZE ID_FILE]\n", argv[0]); return 2; }
    elba::Params prm;
    prm.kmer_size = std::atoi(argv[2]); prm.lower_kmer_freq = std::atoi(argv[3]); prm.upper_kmer_freq = std::atoi(argv[4]);
    const int rank = argc > 7 ? std::atoi(argv[5]) : 0, size = argc > 7 ? std::atoi(argv[6]) : 1;
    std::ifstream in(argv[1]);
    std::vector<std::string> seqs;
    std::string line, cur;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '>') { if (!cur.empty()) seqs.push_back(cur); cur.clear(); }
        else cur += line;
    }
    if (!cur.empty()) seqs.push_back(cur);
    // contiguous partition balanced by bases: the greedy rule of src/FastaIndex.cpp:47-94
    size_t totbases = 0;
    for (auto &s : seqs) totbases += s.size();
    const double avg = (double)totbases / size;
    std::vector<int64_t> displs((size_t)size + 1, 0);
    size_t at = 0;
    for (int i = 0; i < size - 1; ++i) {
        size_t sofar = 0;
        while (at < seqs.size() && sofar + seqs[at].size() < avg) sofar += seqs[at++].size();
        displs[(size_t)i + 1] = (int64_t)at;
    }
    displs[(size_t)size] = (int64_t)seqs.size();
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "no HIP device\n"); return 3; }
        const int device = std::getenv("ELBA_DEVICE") ? std::atoi(std::getenv("ELBA_DEVICE")) : rank % ndev;
        ELBA_DIST_HIP(hipSetDevice(device));
        ncclUniqueId id;
        if (size == 1) ELBA_DIST_NCCL(ncclGetUniqueId(&id));
        else id = elba::DistGrid::exchange_id_through_file(argv[7], rank);
        auto grid = std::make_shared<elba::DistGrid>(rank, size, device, id);
        std::vector<size_t> lens;
        for (int64_t r = displs[(size_t)rank]; r < displs[(size_t)rank + 1]; ++r) lens.push_back(seqs[(size_t)r].size());
        elba::DnaBuffer mydna(elba::DnaBuffer::computebufsize(lens));
        for (int64_t r = displs[(size_t)rank]; r < displs[(size_t)rank + 1]; ++r) mydna.push_back(seqs[(size_t)r].c_str(), seqs[(size_t)r].size());
        elba::DistributedOverlap d(grid, prm);
        d.set_reads(mydna, displs);
        elba_kmer_stats ks{}; elba_matrix_stats ms{};
        d.build_kmer_matrix(&ks, &ms);
        // ELBA_DIST_SLOTS=1: the step with one host synchronisation (fixed-size slots on the grid's stream) instead of begin / fill / end
        const elba_overlap_stats st = std::getenv("ELBA_DIST_SLOTS") ? d.create_seed_matrix_slots((uint64_t)ms.nnz) : d.create_seed_matrix();
        elba_csr_t B;
        d.engine()->check(elba_export_csr(d.engine()->ctx, d.row_lo(), d.row_hi(), &B));
        uint64_t checksum = 0;
        for (int64_t i = 0; i < B.nrows; ++i)
            for (int64_t e = B.rowptr[i]; e < B.rowptr[i + 1]; ++e)
                checksum += (uint64_t)B.val[e].q0 * 1000003ull + B.val[e].t0 + (uint64_t)B.val[e].q1 * 7ull + (uint64_t)B.val[e].t1 * 13ull + (uint64_t)B.val[e].numshared * 7919ull
                            + (uint64_t)(d.row_lo() + i) * 31ull + (uint64_t)B.col[e];
        std::printf("{\"rank\": %d, \"size\": %d, \"rows\": %lld, \"instances\": %lld, \"owned_kmers\": %lld, \"owned_entries\": %lld, \"kmers_total\": %lld, \"panel_entries\": %lld, "
                    "\"nnzB\": %lld, \"products\": %lld, \"checksum\": %llu, \"exchange_bytes\": %llu}\n", rank, size, (long long)B.nrows, (long long)ks.instances, (long long)ks.reliable,
                    (long long)ks.entries, (long long)ms.ncols, (long long)ms.nnz, (long long)B.nnz, (long long)st.products, (unsigned long long)checksum, (unsigned long long)d.exchange_bytes());
        elba_free_csr(&B);
    } catch (const elba::Error &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.status == ELBA_ERR_NO_DEVICE ? 3 : 1;
    }
    return 0;
}
