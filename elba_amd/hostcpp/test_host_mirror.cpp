// test_host_mirror.cpp — the front half of the reference's main() (src/main.cpp:191-285) written against elba_host.hpp, followed by
// the DCSC walk of PairwiseAlignment (src/PairwiseAlignment.cpp:28-56).  Prints one JSON line that tests/test_gpu_hostcpp.py
// compares with the oracle.  Usage: test_host_mirror reads.fa K LOWER UPPER
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include "elba_host.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s reads.fa K LOWER UPPER [fai]   (fai: reads.fa.fai exists; ingest through FastaIndex + the GPU encoder)\n", argv[0]); return 2; }
    const bool use_fai = argc > 5 && std::string(argv[5]) == "fai";
    elba::Params prm;
    prm.kmer_size = std::atoi(argv[2]); prm.lower_kmer_freq = std::atoi(argv[3]); prm.upper_kmer_freq = std::atoi(argv[4]);
    std::ifstream in(argv[1]);
    std::vector<std::string> seqs;
    std::string line, cur;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '>') { if (!cur.empty()) seqs.push_back(cur); cur.clear(); }
        else cur += line;
    }
    if (!cur.empty()) seqs.push_back(cur);
    std::vector<size_t> lens;
    for (auto &s : seqs) lens.push_back(s.size());
    elba::DnaBuffer mydna(elba::DnaBuffer::computebufsize(lens));
    for (auto &s : seqs) mydna.push_back(s.c_str(), s.size());
    auto commgrid = std::make_shared<elba::Grid>();
    try {
        int ingest_equal = -1;
        std::unique_ptr<elba::KmerCountMap> kmermap;
        if (use_fai) {
            // main.cpp:157-176: FastaIndex index(fasta_fname, commgrid); DnaBuffer mydna = index.getmydna(); — the encoding happens on the GPU here
            elba::FastaIndex index(argv[1], commgrid);
            kmermap = elba::get_kmer_count_map_keys(index, commgrid, prm);
            std::vector<uint8_t> packed(mydna.getbufsize() + 16);
            std::vector<uint64_t> offs(mydna.size());
            std::vector<uint32_t> lens2(mydna.size());
            kmermap->engine->check(elba_export_reads(kmermap->engine->ctx, packed.data(), (int64_t)mydna.getbufsize(), offs.data(), lens2.data(), (int64_t)mydna.size()));
            ingest_equal = index.gettotrecords() == mydna.size() && std::memcmp(packed.data(), mydna.data(), mydna.getbufsize()) == 0
                           && std::memcmp(offs.data(), mydna.offsets(), offs.size() * 8) == 0 && std::memcmp(lens2.data(), mydna.lengths(), lens2.size() * 4) == 0;
        } else
            kmermap = elba::get_kmer_count_map_keys(mydna, commgrid, prm);                 // main.cpp:192
        elba::get_kmer_count_map_values(mydna, *kmermap, commgrid);                         // main.cpp:225
        auto A = elba::create_kmer_matrix(mydna, *kmermap, commgrid);                       // main.cpp:259
        kmermap.reset();                                                                    // main.cpp:266
        auto AT = std::make_unique<elba::KmerMatrix>(*A);                                   // main.cpp:272
        AT->Transpose();                                                                    // main.cpp:273
        auto B = elba::create_seed_matrix(*A, *AT);                                         // main.cpp:281
        const int64_t nnzA = A->getnnz(), ncol = A->getncol();
        A.reset(); AT.reset();                                                              // main.cpp:284-285
        // PairwiseAlignment.cpp:16-56
        size_t localnnzs = (size_t)B->seqptr()->getnnz();
        auto dcsc = B->seqptr()->GetDCSC();
        int64_t nalign = 0;
        uint64_t checksum = 0;
        if (dcsc != nullptr)
            for (int64_t i = 0; i < dcsc->nzc; ++i)
                for (int64_t j = dcsc->cp[i]; j < dcsc->cp[i + 1]; ++j) {
                    int64_t localrow = dcsc->ir[j], localcol = dcsc->jc[i];
                    if ((localrow < localcol) || (localrow <= localcol && localrow < localcol)) {
                        const elba::SharedSeeds &s = dcsc->numx[j];
                        ++nalign;
                        checksum += (uint64_t)std::get<0>(s.getseeds()[0]) * 1000003ull + std::get<1>(s.getseeds()[0]) + (uint64_t)s.getnumshared() * 7919ull
                                    + (uint64_t)localrow * 31ull + (uint64_t)localcol;
                    }
                }
        // main.cpp:300: R = PairwiseAlignment(dfd, *B, mat, mis, gap, xdrop_cutoff) with the defaults of main.cpp:53-56
        auto R = elba::PairwiseAlignment(mydna, *B, 1, -1, -1, 15);
        uint64_t achk = 0; int64_t npassed = 0;
        for (size_t a = 0; a < R->vals.size(); ++a) {
            const elba::Overlap &o = R->vals[a];
            npassed += o.passed ? 1 : 0;
            achk += (uint64_t)(uint32_t)o.score * 1000003ull + std::get<0>(o.beg) * 31ull + std::get<1>(o.beg) * 37ull + std::get<0>(o.end) * 41ull + std::get<1>(o.end) * 43ull
                    + (uint64_t)(o.rc ? 7 : 0) + (uint64_t)(uint8_t)o.direction * 131ull + (uint64_t)(uint32_t)o.suffix * 8191ull + (uint64_t)R->rows[a] * 3ull + (uint64_t)R->cols[a];
        }
        // main.cpp:305-312 with bad_read_cutoff of main.cpp:61
        auto S = elba::TransitiveReduction(mydna, *R, 0.65);
        uint64_t schk = 0;
        for (size_t a = 0; a < S->vals.size(); ++a) {
            const elba::Overlap &o = S->vals[a];
            schk += (uint64_t)S->rows[a] * 1000003ull + (uint64_t)S->cols[a] * 31ull + (uint64_t)(uint8_t)o.direction * 131ull + (uint64_t)(uint32_t)o.suffix * 8191ull
                    + std::get<0>(o.beg) * 37ull + std::get<1>(o.end) * 43ull + std::get<0>(o.len) * 3ull + std::get<1>(o.len) + (uint64_t)a * 7ull;
        }
        // main.cpp:287 (elbalog.log_seed_matrix(*B)), :303 and :317 (parallel_write_paf of R and of S): written when ELBA_OUT_PREFIX is set
        if (const char *pfx = std::getenv("ELBA_OUT_PREFIX")) {
            std::vector<std::string> names;
            for (size_t r = 0; r < mydna.size(); ++r) names.push_back("read" + std::to_string(r));
            elba::log_seed_matrix(*B, std::string(pfx) + "B.mtx");
            elba::parallel_write_paf(*R, names, std::string(pfx) + "overlap.paf");
            elba::parallel_write_paf(*S, names, std::string(pfx) + "string.paf");
        }
        std::printf("{\"reads\": %zu, \"nnzA\": %lld, \"kmers\": %lld, \"nnzB\": %zu, \"candidates\": %lld, \"checksum\": %llu, \"alignments\": %lld, \"passed\": %lld, \"align_checksum\": %llu, \"ingest_equal\": %d, "
                    "\"bad_reads\": %zu, \"contained_reads\": %zu, \"string_nnz\": %lld, \"string_checksum\": %llu}\n", mydna.size(),
                    (long long)nnzA, (long long)ncol, localnnzs, (long long)nalign, (unsigned long long)checksum, (long long)R->getnnz(), (long long)npassed, (unsigned long long)achk, ingest_equal,
                    S->bad_reads.size(), S->contained_reads.size(), (long long)S->getnnz(), (unsigned long long)schk);
    } catch (const elba::Error &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.status == ELBA_ERR_NO_DEVICE ? 3 : 1;
    }
    return 0;
}
