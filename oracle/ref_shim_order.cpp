/*
 * ref_shim_order.cpp — TEST INFRASTRUCTURE (never shipped, never on the GPU box).
 *
 * the k-mer numbering a ONE-RANK run of the reference would produce.  The reference numbers k-mers by the iteration
 * order of its std::unordered_map<TKmer, KmerCountEntry> (create_kmer_matrix, src/KmerOps.cpp:380-394), which depends on: the hash
 * (std::hash<Kmer>, include/Kmer.hpp:75-82 -> Kmer::GetHash, murmur3 seed 313), the bucket count the map got from
 * reserve(ceil(HLL estimate / nprocs)) (src/KmerOps.cpp:60,72; the estimate comes from HyperLogLog::add over the ASCII k-mers,
 * include/KmerOps.hpp:58-69), the insert sequence of pass 1 (:158-187), the erases of pass 2 (count >= UPPER, :301-313) and the final
 * erase of count < LOWER (:335-340).  All of that is replayed here ON THE REFERENCE'S OWN compiled Kmer / HashFuncs / Bloom /
 * HyperLogLog code and this libstdc++; only the MPI exchange (a self-send on one rank: arrival order = read order) and the CombBLAS
 * matrix constructor are not there.  Output: the reliable k-mers in map iteration order (k-mer id = index), and the HLL estimate.
 *
 * Built by `make -C oracle ref_order` from the sources where they lie under /root/reference; HyperLogLog.hpp needs <mpi.h> (the
 * image's MPICH, /opt/conda/include) and two one-line definitions of include/common.h (MPI_Count_type, MPI_ALLREDUCE; MPI 3 branch,
 * :13,:35) that are passed on the command line because common.h itself includes the un-vendored CombBLAS.
 */
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <array>
#include <tuple>
#include <string>
#include <unordered_map>
#include "Kmer.hpp"
#include "DnaSeq.hpp"
#include "Bloom.hpp"
#include "HyperLogLog.hpp"

typedef Kmer<(KMER_SIZE + 31) / 32> TKmer;          /* include/Kmer.hpp:95-97 */
typedef uint32_t PosInRead;                          /* include/KmerOps.hpp:14-22 */
typedef int64_t ReadId;
typedef std::array<PosInRead, UPPER_KMER_FREQ> POSITIONS;
typedef std::array<ReadId, UPPER_KMER_FREQ> READIDS;
typedef std::tuple<READIDS, POSITIONS, int> KmerCountEntry;
typedef std::unordered_map<TKmer, KmerCountEntry> KmerCountMap;

extern "C" {

/* returns N (reliable k-mers) or -1 if cap is too small; out_kmers[id] = first word of the k-mer with id `id` (NLONGS == 1 only) */
int64_t ref_replay_order(const uint8_t *buf, const uint64_t *byte_off, const uint32_t *lens, int64_t nreads,
                         uint64_t *out_kmers, int64_t cap, double *hll_estimate, int64_t *bucket_count, int64_t *keys_after_pass1)
{
    if (TKmer::NBYTES != 8) return -2;
    auto reps_of = [&](int64_t r) { DnaSeq seq(lens[r], const_cast<uint8_t *>(buf + byte_off[r])); return TKmer::GetRepKmers(seq); };

    /* src/KmerOps.cpp:42-60: cardinality estimate over the ASCII strings of the canonical k-mers */
    HyperLogLog hll;
    for (int64_t r = 0; r < nreads; ++r) {
        if (lens[r] < (uint32_t)KMER_SIZE) continue;
        for (auto &mer : reps_of(r)) { auto s = mer.GetString(); hll.add(s.c_str()); }
    }
    const double cardinality = hll.estimate();
    const size_t avgcardinality = static_cast<size_t>(std::ceil(cardinality / 1));
    if (hll_estimate) *hll_estimate = cardinality;

    KmerCountMap kmermap;
    kmermap.reserve(avgcardinality);                                              /* :72 */
    Bloom bm(static_cast<int64_t>(std::ceil(cardinality)), 0.05);                  /* :73 */

    for (int64_t r = 0; r < nreads; ++r) {                                        /* pass 1, :158-187 */
        if (lens[r] < (uint32_t)KMER_SIZE) continue;
        for (auto &mer : reps_of(r)) {
            if (bm.Check(mer.GetBytes(), TKmer::NBYTES)) { if (kmermap.find(mer) == kmermap.end()) kmermap.insert({mer, KmerCountEntry({}, {}, 0)}); }
            else bm.Add(mer.GetBytes(), TKmer::NBYTES);
        }
    }
    if (keys_after_pass1) *keys_after_pass1 = (int64_t)kmermap.size();

    for (int64_t r = 0; r < nreads; ++r) {                                        /* pass 2, :283-318 */
        if (lens[r] < (uint32_t)KMER_SIZE) continue;
        PosInRead p = 0;
        for (auto &kmer : reps_of(r)) {
            const PosInRead pos = p++;
            if (!bm.Check(kmer.GetBytes(), TKmer::NBYTES)) continue;
            auto kmitr = kmermap.find(kmer);
            if (kmitr == kmermap.end()) continue;
            KmerCountEntry &entry = kmitr->second;
            int &count = std::get<2>(entry);
            if (count >= UPPER_KMER_FREQ) { kmermap.erase(kmer); continue; }
            std::get<0>(entry)[count] = r; std::get<1>(entry)[count] = pos; count++;
        }
    }
    auto itr = kmermap.begin();                                                    /* :335-340 */
    while (itr != kmermap.end()) { if (std::get<2>(itr->second) < LOWER_KMER_FREQ) itr = kmermap.erase(itr); else itr++; }

    if (bucket_count) *bucket_count = (int64_t)kmermap.bucket_count();
    if ((int64_t)kmermap.size() > cap) return -1;
    int64_t kmerid = 0;
    for (auto it = kmermap.cbegin(); it != kmermap.cend(); ++it) { uint64_t w; it->first.CopyDataInto(&w); out_kmers[kmerid++] = w; }      /* :380-394 */
    return kmerid;
}

}
