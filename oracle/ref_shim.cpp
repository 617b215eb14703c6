/*
 * oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * C entry points over the REFERENCE's own k-mer primitives, compiled from the
 * sources where they lie under /root/reference (see oracle/Makefile, target
 * `ref`).  Nothing of the reference is copied: this file only *calls*
 *   DnaSeq::DnaSeq(char const*, size_t, uint8_t*)   include/DnaSeq.hpp:44  (compress, src/DnaSeq.cpp:7-29)
 *   Kmer<N>::GetRepKmers / GetKmers / GetTwin / GetRep / GetHash   src/Kmer.cpp:149-242
 *   murmurhash3_64                                  src/HashFuncs.cpp:231-236
 *   Bloom                                           src/Bloom.cpp:6-73
 *   xdrop_aligner / classify_alignment              src/XDropAligner.cpp:7-44, :224-282  (the step after the path)
 *
 * Built with -DCOMMON_H_ : include/common.h's include guard is predefined so its
 * body (mpi.h + the absent, un-vendored CombBLAS) is skipped; none of the files
 * compiled here use anything it declares.  KmerOps.cpp / SharedSeeds.hpp DO need
 * CombBLAS types and are therefore unbuildable here.
 *
 * KMER_SIZE is a compile-time constant of the reference (include/compiletime.h),
 * so one shared object per k is produced: oracle/_ref/libelbaref_k<K>.so.
 */
#include "Kmer.hpp"
#include "HashFuncs.hpp"
#include "DnaSeq.hpp"
#include "Bloom.hpp"
#include "XDropAligner.hpp"
#include <cstdint>
#include <cstring>
#include <vector>
#include <unordered_map>
#include <algorithm>
#include <tuple>

extern "C" {

int ref_kmer_size(void) { return KMER_SIZE; }
int ref_kmer_nlongs(void) { return TKmer::NBYTES / 8; }

/* a1: ASCII -> 2-bit packed bytes, exactly what DnaBuffer::push_back stores. */
size_t ref_encode(const char *s, size_t len, uint8_t *mem)
{
    DnaSeq seq(s, len, mem);
    return seq.numbytes();
}

int ref_base_at(const uint8_t *mem, size_t len, size_t i)
{
    DnaSeq seq(len, const_cast<uint8_t*>(mem));
    return seq[i];
}

/* a2: all forward k-mers / canonical k-mers of a packed read; out has NLONGS u64 per k-mer. */
int64_t ref_kmers(const uint8_t *mem, size_t len, uint64_t *out, int canonical)
{
    DnaSeq seq(len, const_cast<uint8_t*>(mem));
    std::vector<TKmer> v = canonical ? TKmer::GetRepKmers(seq) : TKmer::GetKmers(seq);
    for (size_t i = 0; i < v.size(); ++i) v[i].CopyDataInto(out + i * (TKmer::NBYTES / 8));
    return (int64_t)v.size();
}

/* a2: k-mer from ASCII (set_kmer(char const*)), its twin and its representative. */
void ref_kmer_from_ascii(const char *s, uint64_t *fwd, uint64_t *twin, uint64_t *rep)
{
    TKmer k(s);
    k.CopyDataInto(fwd);
    k.GetTwin().CopyDataInto(twin);
    k.GetRep().CopyDataInto(rep);
}

/* a3: Kmer::GetHash of a packed k-mer. */
uint64_t ref_kmer_hash(const uint64_t *longs)
{
    TKmer k((const void*)longs);
    return k.GetHash();
}

void ref_murmur3_128(const void *key, uint32_t nbytes, uint64_t *out2)
{
    murmurhash3_128(key, nbytes, out2);
}

/* a6: Bloom filter as the reference builds it (entries, 0.05). Returns #"seen before" answers. */
int64_t ref_bloom_second_sightings(const uint64_t *keys, int64_t n, int64_t entries)
{
    Bloom bm(entries, 0.05);
    int64_t seen = 0;
    for (int64_t i = 0; i < n; ++i)
    {
        if (bm.Check(&keys[i], 8)) seen++;
        else bm.Add(&keys[i], 8);
    }
    return seen;
}

/*
 * a7+a8 replay on ONE rank: the control flow of src/KmerOps.cpp:158-187 (pass 1: Bloom
 * check -> insert key on "seen", else Bloom add), :283-318 (pass 2: Bloom re-check, find,
 * erase when count >= UPPER, else record (read,pos)) and :335-340 (erase count < LOWER),
 * driven by the reference's own Kmer::GetRepKmers and Bloom.  The map is a plain
 * std::unordered_map keyed by the packed k-mer (NLONGS == 1 only); lower/upper are run-time
 * here (the reference fixes them at compile time, include/compiletime.h:15-22).
 * bloom_entries plays the role of ceil(HLL estimate) (src/KmerOps.cpp:73).
 * Output: triples sorted by (kmer, read, pos); returns Z (or -1 if cap too small).
 */
int64_t ref_replay_count(const uint8_t *buf, const uint64_t *byte_off, const uint32_t *lens, int64_t nreads,
                         int lower, int upper, int64_t bloom_entries,
                         uint64_t *out_kmer, int64_t *out_read, uint32_t *out_pos, int64_t cap,
                         int64_t *keys_after_pass1)
{
    static_assert(TKmer::NBYTES == 8 || TKmer::NBYTES > 8, "");
    if (TKmer::NBYTES != 8) return -2;
    struct Entry { std::vector<int64_t> reads; std::vector<uint32_t> pos; int count = 0; };
    std::unordered_map<uint64_t, Entry> kmermap;
    Bloom bm(bloom_entries, 0.05);

    for (int64_t r = 0; r < nreads; ++r)
    {
        if (lens[r] < (uint32_t)KMER_SIZE) continue;
        DnaSeq seq(lens[r], const_cast<uint8_t*>(buf + byte_off[r]));
        std::vector<TKmer> reps = TKmer::GetRepKmers(seq);
        for (auto &mer : reps)
        {
            uint64_t w; mer.CopyDataInto(&w);
            if (bm.Check(mer.GetBytes(), TKmer::NBYTES)) { if (kmermap.find(w) == kmermap.end()) kmermap.insert({w, Entry()}); }
            else bm.Add(mer.GetBytes(), TKmer::NBYTES);
        }
    }
    if (keys_after_pass1) *keys_after_pass1 = (int64_t)kmermap.size();

    for (int64_t r = 0; r < nreads; ++r)
    {
        if (lens[r] < (uint32_t)KMER_SIZE) continue;
        DnaSeq seq(lens[r], const_cast<uint8_t*>(buf + byte_off[r]));
        std::vector<TKmer> reps = TKmer::GetRepKmers(seq);
        uint32_t p = 0;
        for (auto &mer : reps)
        {
            uint32_t pos = p++;
            uint64_t w; mer.CopyDataInto(&w);
            if (!bm.Check(mer.GetBytes(), TKmer::NBYTES)) continue;
            auto it = kmermap.find(w);
            if (it == kmermap.end()) continue;
            Entry &e = it->second;
            if (e.count >= upper) { kmermap.erase(it); continue; }
            e.reads.push_back(r); e.pos.push_back(pos); e.count++;
        }
    }
    for (auto it = kmermap.begin(); it != kmermap.end(); ) { if (it->second.count < lower) it = kmermap.erase(it); else ++it; }

    std::vector<std::tuple<uint64_t,int64_t,uint32_t>> tr;
    for (auto &kv : kmermap) for (int j = 0; j < kv.second.count; ++j) tr.emplace_back(kv.first, kv.second.reads[j], kv.second.pos[j]);
    std::sort(tr.begin(), tr.end());
    if ((int64_t)tr.size() > cap) return -1;
    for (size_t i = 0; i < tr.size(); ++i) { out_kmer[i] = std::get<0>(tr[i]); out_read[i] = std::get<1>(tr[i]); out_pos[i] = std::get<2>(tr[i]); }
    return (int64_t)tr.size();
}

/*
 * f1: the reference's x-drop seed-and-extend on one pair (src/XDropAligner.cpp:224-282) and its classification (:7-44).
 * out = {return value, begQ, endQ, begT, endT, score, rc, OverlapClass}
 */
void ref_xdrop(const uint8_t *qmem, size_t qlen, const uint8_t *tmem, size_t tlen, int begQ, int begT,
               int mat, int mis, int gap, int dropoff, int *out)
{
    DnaSeq q(qlen, const_cast<uint8_t*>(qmem)), t(tlen, const_cast<uint8_t*>(tmem));
    XSeed r;
    int ret = xdrop_aligner(q, t, begQ, begT, mat, mis, gap, dropoff, r);
    OverlapClass kind;
    classify_alignment(r, (int)qlen, (int)tlen, kind);
    out[0] = ret; out[1] = r.begQ; out[2] = r.endQ; out[3] = r.begT; out[4] = r.endT; out[5] = r.score; out[6] = r.rc ? 1 : 0; out[7] = (int)kind;
}

}
