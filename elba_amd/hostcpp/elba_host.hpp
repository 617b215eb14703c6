// elba_host.hpp — C++17 host-side mirror of the reference's operator surface for the overlap hot path, over the C ABI of
// include/elba_amd.h.  Same function names, argument meaning and ownership conventions as the reference so that
// src/main.cpp:191-300 reads the same with `elba::` types in place of the CombBLAS ones:
//
//   reference                                                        this header
//   ---------------------------------------------------------------  --------------------------------------------------
//   DnaBuffer            include/DnaBuffer.hpp:13-38                  elba::DnaBuffer   (same 2-bit layout, src/DnaSeq.cpp:7-29)
//   KmerCountMap         include/KmerOps.hpp:22                       elba::KmerCountMap (opaque: the counts live in HBM)
//   get_kmer_count_map_keys / _values   include/KmerOps.hpp:27-30     same names
//   create_kmer_matrix   include/KmerOps.hpp:24-25                    same name -> elba::KmerMatrix (CSR and CSC both resident)
//   AT = *A; AT->Transpose()            src/main.cpp:272-273          KmerMatrix copy + Transpose(): no work, both orientations exist
//   SharedSeeds          include/SharedSeeds.hpp:8-96                 elba::SharedSeeds (same members and accessors)
//   create_seed_matrix   include/SharedSeeds.hpp:98-99                same name -> elba::SeedMatrix
//   FastaIndex           include/FastaIndex.hpp, src/FastaIndex.cpp          same name: .fai records, base-balanced partition, chunk bounds;
//                                                                             the reads are 2-bit encoded on the GPU (elba_set_reads_fasta)
//   PairwiseAlignment    include/PairwiseAlignment.hpp, src/PairwiseAlignment.cpp:5-106     same name -> elba::OverlapMatrix (triples of elba::Overlap)
//   Bmat.seqptr()->getnnz() / GetDCSC()  src/PairwiseAlignment.cpp:16-19   SeedMatrix::seqptr()->getnnz() / GetDCSC()
//   find_bad_reads / find_contained_reads / TransitiveReduction   src/main.cpp:305-312, src/TransitiveReduction.cpp:3-90
//                                                                             elba::TransitiveReduction(R, cutoff): the prunes and the reduction in one
//                                                                             call on the GPU -> elba::StringGraph (S + the two read lists)
//
// Errors: the reference asserts/aborts; here every failing C-ABI status throws elba::Error (status + text).
// There is no CPU path: constructing an engine without a GPU throws ELBA_ERR_NO_DEVICE.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <ostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/elba_amd.h"

namespace elba {

using PosInRead = uint32_t;   // include/KmerOps.hpp:14
using ReadId = int64_t;       // include/KmerOps.hpp:15

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &t) : std::runtime_error("elba status " + std::to_string(s) + ": " + t), status(s) {}
};

// The process grid argument of the reference (std::shared_ptr<CommGrid>): one rank per GPU, 1D here.
struct Grid {
    int rank = 0, size = 1, device = 0;
    int GetRank() const { return rank; }
    int GetSize() const { return size; }
};

// 2-bit read buffer, 4 bases per byte, first base in bits 7-6, reads byte-aligned (src/DnaSeq.cpp:7-29, src/DnaBuffer.cpp:22-29).
class DnaBuffer {
public:
    static size_t bytesneeded(size_t n) { return (n + 3) / 4; }                       // include/DnaSeq.hpp:131
    static size_t computebufsize(const std::vector<size_t> &lens)
    {
        size_t s = 0;
        for (size_t l : lens) s += bytesneeded(l);
        return s;
    }
    explicit DnaBuffer(size_t bufsize = 0) { buf_.reserve(bufsize + 16); }
    void push_back(char const *s, size_t len)
    {
        const size_t nb = bytesneeded(len), at = buf_.size();
        buf_.resize(at + nb, 0);
        for (size_t i = 0; i < len; ++i) buf_[at + i / 4] |= (uint8_t)(code(s[i]) << (6 - 2 * (i % 4)));
        off_.push_back(at);
        len_.push_back((uint32_t)len);
    }
    size_t size() const { return len_.size(); }
    size_t getbufsize() const { return buf_.size(); }
    const uint8_t *data() const { return buf_.data(); }
    const uint64_t *offsets() const { return off_.data(); }
    const uint32_t *lengths() const { return len_.data(); }
    int base(size_t read, size_t i) const { return (buf_[off_[read] + i / 4] >> (6 - 2 * (i % 4))) & 3; }   // DnaSeq::operator[]

private:
    static uint8_t code(char c)                                                        // include/DnaSeq.hpp:136-154
    {
        switch (c) {
        case 'A': case 'a': case 'N': case 'n': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
        }
    }
    std::vector<uint8_t> buf_;
    std::vector<uint64_t> off_;
    std::vector<uint32_t> len_;
};

// include/FastaIndex.hpp — the host side of the ingest: the .fai next to the FASTA (`name len pos bases [width]`, src/FastaIndex.cpp:15-23), the
// greedy base-balanced contiguous partition (:47-94: a rank takes reads while the next one keeps it under total / nprocs, the last rank
// the rest) and the bytes of the file a rank needs (:222-224).  What getmydna does after reading that chunk — stripping line ends and
// DnaSeq::compress — runs on the GPU instead: get_kmer_count_map_keys(index, ...) below hands the raw chunk to elba_set_reads_fasta.
class FastaIndex {
public:
    typedef elba_fasta_record_t Record;                       // {len, pos, bases}: FastaIndex::Record, include/FastaIndex.hpp:10
    FastaIndex(const std::string &fasta_fname, std::shared_ptr<Grid> commgrid) : commgrid(commgrid), fasta_fname(fasta_fname)
    {
        std::ifstream in(get_faidx_fname());
        if (!in) throw Error(ELBA_ERR_INVALID_ARG, "cannot open " + get_faidx_fname());
        std::string line, name;
        while (std::getline(in, line)) {
            std::istringstream ls(line);
            Record r{};
            if (!(ls >> name >> r.len >> r.pos >> r.bases)) continue;      // get_faidx_record, src/FastaIndex.cpp:15-23
            rootrecords.push_back(r); rootnames.push_back(name);
        }
        const int nprocs = commgrid->GetSize();
        size_t totbases = 0;
        for (auto &r : rootrecords) totbases += r.len;
        const double avg = (double)totbases / nprocs;
        readdispls.assign((size_t)nprocs + 1, 0);
        size_t at = 0;
        for (int i = 0; i < nprocs - 1; ++i) {                               // getpartition, src/FastaIndex.cpp:47-94
            size_t sofar = 0;
            while (at < rootrecords.size() && sofar + rootrecords[at].len < avg) sofar += rootrecords[at++].len;
            readdispls[(size_t)i + 1] = at;
        }
        readdispls[(size_t)nprocs] = rootrecords.size();
        const size_t lo = readdispls[(size_t)commgrid->GetRank()], hi = readdispls[(size_t)commgrid->GetRank() + 1];
        myrecords.assign(rootrecords.begin() + (std::ptrdiff_t)lo, rootrecords.begin() + (std::ptrdiff_t)hi);
    }
    std::string get_fasta_fname() const { return fasta_fname; }
    std::string get_faidx_fname() const { return fasta_fname + ".fai"; }
    size_t gettotrecords() const { return rootrecords.size(); }
    size_t getmyreadcount() const { return myrecords.size(); }
    size_t getmyreaddispl() const { return readdispls[(size_t)commgrid->GetRank()]; }
    const std::vector<Record> &getmyrecords() const { return myrecords; }
    const std::vector<size_t> &getreaddispls() const { return readdispls; }
    const std::vector<std::string> &getnames() const { return rootnames; }
    std::shared_ptr<Grid> getcommgrid() const { return commgrid; }
    // the raw bytes of the file that hold this rank's records, and the file offset of the first one (src/FastaIndex.cpp:222-241)
    std::vector<char> readmychunk(uint64_t &startpos) const
    {
        std::vector<char> buf;
        startpos = 0;
        if (myrecords.empty()) return buf;
        std::ifstream in(fasta_fname, std::ios::binary | std::ios::ate);
        if (!in) throw Error(ELBA_ERR_INVALID_ARG, "cannot open " + fasta_fname);
        const uint64_t filesize = (uint64_t)in.tellg();
        startpos = myrecords.front().pos;
        uint64_t endpos = myrecords.back().pos + myrecords.back().len + myrecords.back().len / myrecords.back().bases;
        if (endpos > filesize) endpos = filesize;
        buf.resize((size_t)(endpos - startpos));
        in.seekg((std::streamoff)startpos);
        in.read(buf.data(), (std::streamsize)buf.size());
        return buf;
    }

private:
    std::shared_ptr<Grid> commgrid;
    std::vector<Record> myrecords, rootrecords;
    std::vector<size_t> readdispls;
    std::vector<std::string> rootnames;
    std::string fasta_fname;
};

// include/SharedSeeds.hpp:8-96 — same members, constructors and accessors (the Semiring runs on the GPU).
struct SharedSeeds {
    SharedSeeds() : numshared(0) {}
    SharedSeeds(PosInRead begQ, PosInRead begT) : numshared(1) { std::get<0>(seeds[0]) = begQ; std::get<1>(seeds[0]) = begT; }
    SharedSeeds(const std::tuple<PosInRead, PosInRead> &s1, const std::tuple<PosInRead, PosInRead> &s2, int n) : numshared(n) { seeds[0] = s1; seeds[1] = s2; }
    explicit SharedSeeds(const elba_seed_t &v) : numshared(v.numshared)               // field-wise: never memcpy
    {
        std::get<0>(seeds[0]) = v.q0; std::get<1>(seeds[0]) = v.t0;
        std::get<0>(seeds[1]) = v.q1; std::get<1>(seeds[1]) = v.t1;
    }
    int getnumstored() const { return numshared < 2 ? numshared : 2; }
    int getnumshared() const { return numshared; }
    const std::tuple<PosInRead, PosInRead> *getseeds() const { return &seeds[0]; }
    std::tuple<PosInRead, PosInRead> seeds[2];
    int numshared;
};

namespace detail {
struct Engine {
    elba_ctx *ctx = nullptr;
    explicit Engine(const elba_cfg &cfg)
    {
        int rc = elba_ctx_create(&ctx, &cfg);
        if (rc != ELBA_OK) throw Error(rc, elba_strerror(rc));
    }
    ~Engine() { elba_ctx_destroy(ctx); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    void check(int rc) const { if (rc != ELBA_OK) throw Error(rc, std::string(elba_strerror(rc)) + ": " + elba_last_error(ctx)); }
};
}  // namespace detail

// Compile-time constants of the reference (Makefile:1-6, include/compiletime.h) are run-time parameters here.
struct Params {
    int kmer_size = 31, lower_kmer_freq = 15, upper_kmer_freq = 35;
};

class KmerCountMap {
public:
    std::shared_ptr<detail::Engine> engine;
    elba_kmer_stats stats{};
    size_t size() const { return (size_t)stats.reliable; }       // reliable 'column' k-mers (src/KmerOps.cpp:343-346)
};

// What the reference's CT<SharedSeeds>::PSpDCCols exposes to PairwiseAlignment (src/PairwiseAlignment.cpp:16-32).
struct Dcsc {
    int64_t nzc = 0;
    std::vector<int64_t> jc, cp, ir;
    std::vector<SharedSeeds> numx;
};

class SeqSeedMatrix {
public:
    int64_t getnnz() const { return (int64_t)dcsc_.numx.size(); }
    const Dcsc *GetDCSC() const { return getnnz() ? &dcsc_ : nullptr; }
    Dcsc dcsc_;
};

class SeedMatrix {
public:
    std::shared_ptr<detail::Engine> engine;
    elba_overlap_stats stats{};
    int64_t nrows = 0;
    int64_t getnnz() const { return stats.nnz; }
    int64_t getnrow() const { return nrows; }
    int64_t getncol() const { return nrows; }
    // local block of this rank's grid cell; one rank: the whole matrix
    const SeqSeedMatrix *seqptr()
    {
        if (!seq_) {
            elba_dcsc_t d;
            engine->check(elba_export_dcsc(engine->ctx, 0, nrows, 0, nrows, &d));
            seq_ = std::make_unique<SeqSeedMatrix>();
            Dcsc &o = seq_->dcsc_;
            o.nzc = d.nzc;
            o.jc.assign(d.jc, d.jc + d.nzc);
            o.cp.assign(d.cp, d.cp + d.nzc + 1);
            o.ir.assign(d.ir, d.ir + d.nnz);
            o.numx.reserve((size_t)d.nnz);
            for (int64_t i = 0; i < d.nnz; ++i) o.numx.emplace_back(d.numx[i]);
            elba_free_dcsc(&d);
        }
        return seq_.get();
    }

private:
    std::unique_ptr<SeqSeedMatrix> seq_;
};

class KmerMatrix {
public:
    std::shared_ptr<detail::Engine> engine;
    elba_matrix_stats stats{};
    int64_t getnrow() const { return stats.nrows; }
    int64_t getncol() const { return stats.ncols; }
    int64_t getnnz() const { return stats.nnz; }
    void Transpose() {}        // CSR and CSC of A are both resident on the device: AT is a view, not a copy (src/main.cpp:272-273)
};

inline std::unique_ptr<KmerCountMap> get_kmer_count_map_keys(const DnaBuffer &myreads, std::shared_ptr<Grid> grid, const Params &prm = Params())
{
    if (grid->GetSize() != 1) throw Error(ELBA_ERR_UNSUPPORTED, "the C++ mirror drives one GPU; multi-GPU runs go through the elba_dist_* entry points");
    elba_cfg cfg{};
    cfg.k = prm.kmer_size; cfg.lower = prm.lower_kmer_freq; cfg.upper = prm.upper_kmer_freq; cfg.device = grid->device;
    auto map = std::make_unique<KmerCountMap>();
    map->engine = std::make_shared<detail::Engine>(cfg);
    map->engine->check(elba_set_reads(map->engine->ctx, myreads.data(), myreads.offsets(), myreads.lengths(), (int64_t)myreads.size(), 0));
    // both passes of the reference (keys: src/KmerOps.cpp:18-204, values: :206-350) are one exact count on the GPU
    map->engine->check(elba_count_kmers(map->engine->ctx, &map->stats));
    return map;
}

// The same entry point for a caller that has not parsed its reads yet: the rank's chunk of the FASTA goes to the GPU as it is in the file.
// index.getmydna()'s result can still be had from the context (elba_export_reads) — e.g. for the lengths PairwiseAlignment needs: `reads` is
// filled with them here (the mirror's DnaBuffer keeps lengths and offsets; the packed bytes are fetched only on demand).
inline std::unique_ptr<KmerCountMap> get_kmer_count_map_keys(const FastaIndex &index, std::shared_ptr<Grid> grid, const Params &prm, elba_ingest_stats *ingest = nullptr)
{
    if (grid->GetSize() != 1) throw Error(ELBA_ERR_UNSUPPORTED, "the C++ mirror drives one GPU; multi-GPU runs go through the elba_dist_* entry points");
    elba_cfg cfg{};
    cfg.k = prm.kmer_size; cfg.lower = prm.lower_kmer_freq; cfg.upper = prm.upper_kmer_freq; cfg.device = grid->device;
    auto map = std::make_unique<KmerCountMap>();
    map->engine = std::make_shared<detail::Engine>(cfg);
    uint64_t startpos = 0;
    const std::vector<char> chunk = index.readmychunk(startpos);
    elba_ingest_stats is{};
    map->engine->check(elba_set_reads_fasta(map->engine->ctx, chunk.data(), (int64_t)chunk.size(), startpos, index.getmyrecords().data(),
                                            (int64_t)index.getmyreadcount(), (int64_t)index.getmyreaddispl(), &is));
    if (ingest) *ingest = is;
    map->engine->check(elba_count_kmers(map->engine->ctx, &map->stats));
    return map;
}

inline void get_kmer_count_map_values(const DnaBuffer &, KmerCountMap &kmermap, std::shared_ptr<Grid>)
{
    if (!kmermap.engine) throw Error(ELBA_ERR_STATE, "get_kmer_count_map_values: empty k-mer map");
}

inline std::unique_ptr<KmerMatrix> create_kmer_matrix(const DnaBuffer &, const KmerCountMap &kmermap, std::shared_ptr<Grid>)
{
    auto A = std::make_unique<KmerMatrix>();
    A->engine = kmermap.engine;
    A->engine->check(elba_create_kmer_matrix(A->engine->ctx, &A->stats));
    return A;
}

inline std::unique_ptr<SeedMatrix> create_seed_matrix(KmerMatrix &A, KmerMatrix &AT)
{
    if (A.engine != AT.engine) throw Error(ELBA_ERR_INVALID_ARG, "create_seed_matrix: A and AT must come from the same create_kmer_matrix");
    auto B = std::make_unique<SeedMatrix>();
    B->engine = A.engine;
    B->nrows = A.stats.nrows;
    B->engine->check(elba_create_seed_matrix(B->engine->ctx, &B->stats));   // Mult_AnXBn_DoubleBuff + Prune(numshared <= 1), src/SharedSeeds.cpp:7-8
    return B;
}

// The fields of the reference's Overlap that extend_overlap fills (include/Overlap.hpp:22-28; src/Overlap.cpp:24-73), same names.
struct Overlap {
    std::tuple<PosInRead, PosInRead> beg, end, len, seed;
    int score = 0, suffix = 0, suffixT = 0;
    int8_t direction = -1, directionT = -1;
    bool rc = false, passed = false, containedQ = false, containedT = false;
};

// What PairwiseAlignment hands to SpParMat<Overlap>(numreads, numreads, drows, dcols, dvals, false): the local triples (src/PairwiseAlignment.cpp:97-103)
class OverlapMatrix {
public:
    int64_t numreads = 0;
    std::vector<int64_t> rows, cols;
    std::vector<Overlap> vals;
    elba_align_stats stats{};
    std::shared_ptr<detail::Engine> engine;  // the alignments stay on the device for the string-graph stage
    int64_t getnnz() const { return (int64_t)vals.size(); }
};

// PairwiseAlignment(dfd, Bmat, mat, mis, gap, dropoff) — src/PairwiseAlignment.cpp:5-106 on one rank: every stored B(i,j), i < j, is
// aligned from seeds[0] on the GPU (the reads are the ones the k-mer stage was given; DistributedFastaData's row/column buffers are
// the same buffer on one rank).
inline std::unique_ptr<OverlapMatrix> PairwiseAlignment(const DnaBuffer &myreads, SeedMatrix &Bmat, int mat, int mis, int gap, int dropoff)
{
    auto R = std::make_unique<OverlapMatrix>();
    R->numreads = Bmat.getnrow();
    R->engine = Bmat.engine;
    Bmat.engine->check(elba_align_seeds(Bmat.engine->ctx, mat, mis, gap, dropoff, &R->stats));
    elba_overlaps_t o;
    Bmat.engine->check(elba_export_overlaps(Bmat.engine->ctx, &o));
    R->rows.assign(o.rows, o.rows + o.n); R->cols.assign(o.cols, o.cols + o.n);
    R->vals.resize((size_t)o.n);
    // (Overlap::seed, a copy of seeds[0] of B(i,j), is not carried back: the caller still owns B)
    for (int64_t a = 0; a < o.n; ++a) {
        const elba_overlap_t &v = o.vals[a];
        Overlap &w = R->vals[(size_t)a];
        w.beg = std::make_tuple((PosInRead)v.begQ, (PosInRead)v.begT); w.end = std::make_tuple((PosInRead)v.endQ, (PosInRead)v.endT);
        w.len = std::make_tuple((PosInRead)myreads.lengths()[o.rows[a]], (PosInRead)myreads.lengths()[o.cols[a]]);
        w.score = v.score; w.suffix = v.suffix; w.suffixT = v.suffixT; w.direction = v.direction; w.directionT = v.directionT;
        w.rc = v.rc; w.passed = v.passed; w.containedQ = v.containedQ; w.containedT = v.containedT;
    }
    elba_free_overlaps(&o);
    return R;
}

// The string graph S of src/main.cpp:312 with what main() derives on the way: the reads find_bad_reads (:305) and find_contained_reads
// (:310) return.  Entries in the order parallel_write_paf walks S (:527-541): columns ascending, rows ascending within a column.
class StringGraph {
public:
    int64_t numreads = 0;
    std::vector<int64_t> rows, cols;
    std::vector<Overlap> vals;
    std::vector<int64_t> bad_reads, contained_reads;
    elba_string_stats stats{};
    int64_t getnnz() const { return (int64_t)vals.size(); }
};

// src/main.cpp:305-312 — bad_reads = find_bad_reads(*R, cutoff); R->Prune(!passed); R->PruneFull(bad_reads, bad_reads);
// contained = find_contained_reads(*R); R->PruneFull(contained, contained); S = TransitiveReduction(*R) — as one call on the
// alignments PairwiseAlignment left on the device.  R itself is not modified (the reference prunes it in place and then drops it).
inline std::unique_ptr<StringGraph> TransitiveReduction(const DnaBuffer &myreads, OverlapMatrix &R, double bad_read_cutoff, int fuzz = 1000)
{
    auto S = std::make_unique<StringGraph>();
    S->numreads = R.numreads;
    R.engine->check(elba_transitive_reduction(R.engine->ctx, bad_read_cutoff, fuzz, &S->stats));
    elba_overlaps_t o;
    R.engine->check(elba_export_string_graph(R.engine->ctx, &o));
    S->rows.assign(o.rows, o.rows + o.n); S->cols.assign(o.cols, o.cols + o.n);
    S->vals.resize((size_t)o.n);
    for (int64_t a = 0; a < o.n; ++a) {
        const elba_overlap_t &v = o.vals[a];
        Overlap &w = S->vals[(size_t)a];
        w.beg = std::make_tuple((PosInRead)v.begQ, (PosInRead)v.begT); w.end = std::make_tuple((PosInRead)v.endQ, (PosInRead)v.endT);
        w.len = std::make_tuple((PosInRead)myreads.lengths()[o.rows[a]], (PosInRead)myreads.lengths()[o.cols[a]]);
        w.score = v.score; w.suffix = v.suffix; w.suffixT = v.suffixT; w.direction = v.direction; w.directionT = v.directionT;
        w.rc = v.rc; w.passed = v.passed; w.containedQ = v.containedQ; w.containedT = v.containedT;
    }
    elba_free_overlaps(&o);
    std::vector<uint8_t> flags((size_t)R.numreads);
    R.engine->check(elba_export_read_flags(R.engine->ctx, flags.data(), R.numreads));
    for (int64_t v = 0; v < R.numreads; ++v) {
        if (flags[(size_t)v] & 1) S->bad_reads.push_back(v);
        if (flags[(size_t)v] & 2) S->contained_reads.push_back(v);
    }
    return S;
}

// ---- the reference's text outputs -------------------------------------------------------------------------------
// SharedSeeds as operator<< prints it (include/SharedSeeds.hpp:75-88): `{(q0,t0),(q1,t1),n}` with min(numshared, 2) seeds.
inline std::ostream &operator<<(std::ostream &os, const SharedSeeds &o)
{
    const int seedstoprint = o.getnumstored() < 2 ? o.getnumstored() : 2;
    os << "{";
    for (int i = 0; i < seedstoprint; ++i) os << "(" << std::get<0>(o.seeds[i]) << "," << std::get<1>(o.seeds[i]) << "),";
    os << o.getnumshared() << "}";
    return os;
}

// Overlap as operator<< prints it (include/Overlap.hpp:78-83)
inline std::ostream &operator<<(std::ostream &os, const Overlap &o)
{
    os << std::get<0>(o.len) << "\t" << std::get<0>(o.beg) << "\t" << std::get<0>(o.end) << "\t" << (o.rc ? '-' : '+') << "\t" << std::get<1>(o.len) << "\t" << std::get<1>(o.beg) << "\t"
       << std::get<1>(o.end) << "\t" << o.score << "\t" << static_cast<int>(o.direction) << "\t" << static_cast<int>(o.suffix);
    return os;
}

// ELBALogger::log_seed_matrix (src/ELBALogger.cpp:22-35): B.ParallelWriteMM("B.mtx", true, SharedSeeds::IOHandler()) — MatrixMarket
// coordinate file, one-based indices, entries as the local DCSC stores them (column by column, rows ascending), values through
// IOHandler::save = operator<<.  (CombBLAS's writer itself is un-vendored: the header line follows the MatrixMarket convention.)
inline void log_seed_matrix(SeedMatrix &B, const std::string &fname)
{
    std::ofstream os(fname);
    if (!os) throw Error(ELBA_ERR_INVALID_ARG, "cannot open " + fname);
    const SeqSeedMatrix *seq = B.seqptr();
    const Dcsc *dcsc = seq->GetDCSC();
    os << "%%MatrixMarket matrix coordinate real general\n" << B.getnrow() << " " << B.getncol() << " " << seq->getnnz() << "\n";
    if (dcsc)
        for (int64_t i = 0; i < dcsc->nzc; ++i)
            for (int64_t j = dcsc->cp[(size_t)i]; j < dcsc->cp[(size_t)i + 1]; ++j) os << dcsc->ir[(size_t)j] + 1 << "\t" << dcsc->jc[(size_t)i] + 1 << "\t" << dcsc->numx[(size_t)j] << "\n";
}

// parallel_write_paf (src/main.cpp:514-551) for the triples of R or S: the reference walks the local DCSC (columns ascending, rows ascending
// within a column) and prints `maplen` as written there (:536: max(endQ - begQ, endT - endT)).
namespace detail {
template <class M>
inline void write_paf(const M &R, const std::vector<std::string> &names, const std::string &pafname)
{
    std::vector<size_t> order(R.vals.size());
    for (size_t a = 0; a < order.size(); ++a) order[a] = a;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return R.cols[x] != R.cols[y] ? R.cols[x] < R.cols[y] : R.rows[x] < R.rows[y]; });
    std::ofstream ss(pafname);
    if (!ss) throw Error(ELBA_ERR_INVALID_ARG, "cannot open " + pafname);
    for (size_t a : order) {
        const Overlap &o = R.vals[a];
        const long long begQ = std::get<0>(o.beg), endQ = std::get<0>(o.end), endT = std::get<1>(o.end);
        const long long maplen = std::max(endQ - begQ, endT - endT);
        ss << names[(size_t)R.rows[a]] << "\t" << std::get<0>(o.len) << "\t" << std::get<0>(o.beg) << "\t" << std::get<0>(o.end) << "\t" << "+-"[o.rc ? 1 : 0] << "\t"
           << names[(size_t)R.cols[a]] << "\t" << std::get<1>(o.len) << "\t" << std::get<1>(o.beg) << "\t" << std::get<1>(o.end) << "\t" << o.score << "\t" << maplen << "\t255\t"
           << static_cast<int>(o.passed) << "\n";
    }
}
}  // namespace detail
inline void parallel_write_paf(const OverlapMatrix &R, const std::vector<std::string> &names, const std::string &pafname) { detail::write_paf(R, names, pafname); }
inline void parallel_write_paf(const StringGraph &S, const std::vector<std::string> &names, const std::string &pafname) { detail::write_paf(S, names, pafname); }

}  // namespace elba
