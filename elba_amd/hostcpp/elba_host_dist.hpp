// elba_host_dist.hpp — the multi-GPU overlap build in C++17 over RCCL: one process per GPU of a node, 1D read-row shards x value-range-
// owned k-mer columns.  The heavy lifting is in libelba_amd.so (elba_dist_* of include/elba_amd.h); this header sequences
// the stages and issues the collectives — exactly what elba_amd/distributed.py does through torch.distributed, here directly on RCCL:
//
//   reference (MPI, src/KmerOps.cpp)                                   here (RCCL over xGMI)
//   ----------------------------------------------------------------   ---------------------------------------------------------------------
//   GetKmerOwner = hash -> rank                       :352-359          owner = value range; ncclAllReduce of a 4096-bin histogram picks the ranges
//   MPI_ALLTOALLV of the k-mers (pass 1) and of
//   (k-mer, read, pos) seeds (pass 2)                 :117-151,:244-274  ONE all-to-all of (k-mer, read << 32 | pos) records: grouped ncclSend / ncclRecv
//   MPI_Exscan of the local map sizes -> k-mer ids    :371-375          ncclAllGather of the owners' counts, exclusive scan on the host
//   SpParMat ctor / Transpose() redistribute A, AT    :396-400, main.cpp:272-273   ONE all-to-all of column panels: each column, whole, to every rank owning one of its reads
//   SUMMA broadcasts inside create_seed_matrix        src/SharedSeeds.cpp:7       ONE all-to-all of mirrored entries: a cross-rank pair is accumulated by one rank only
//
// All-to-all rounds are capped at 512 MiB per peer (the reference batches too, include/KmerOps.hpp:33-56; RCCL 2.26 was seen to deliver
// only the first GiB of a >= 2 GiB message, profiles/r01_notes.md).  Needs <rccl/rccl.h> and the HIP runtime API (device buffers, one stream).
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <chrono>
#include <thread>
#include "elba_host.hpp"

namespace elba {

#define ELBA_DIST_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) throw ::elba::Error(ELBA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)
#define ELBA_DIST_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) throw ::elba::Error(ELBA_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r__)); } while (0)

// The reference's CommGrid for this path: rank, size, the GPU, and the communicator (the reference: MPI_COMM_WORLD on a sqrt(p) x sqrt(p) grid).
struct DistGrid : Grid {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DistGrid(int rank_, int size_, int device_, const ncclUniqueId &id)
    {
        rank = rank_; size = size_; device = device_;
        ELBA_DIST_HIP(hipSetDevice(device));
        ELBA_DIST_HIP(hipStreamCreate(&stream));
        ELBA_DIST_NCCL(ncclCommInitRank(&comm, size, id, rank));
    }
    ~DistGrid() { if (comm) (void)ncclCommDestroy(comm); if (stream) (void)hipStreamDestroy(stream); }
    DistGrid(const DistGrid &) = delete;
    DistGrid &operator=(const DistGrid &) = delete;
    // single-node bootstrap without MPI: rank 0 publishes the RCCL id in a file, the others wait for it
    static ncclUniqueId exchange_id_through_file(const std::string &path, int rank)
    {
        ncclUniqueId id;
        if (rank == 0) {
            ELBA_DIST_NCCL(ncclGetUniqueId(&id));
            std::ofstream tmp(path + ".tmp", std::ios::binary);
            tmp.write(reinterpret_cast<const char *>(&id), sizeof(id));
            tmp.close();
            std::rename((path + ".tmp").c_str(), path.c_str());
        } else {
            for (int tries = 0;; ++tries) {
                std::ifstream in(path, std::ios::binary);
                if (in && in.read(reinterpret_cast<char *>(&id), sizeof(id))) break;
                if (tries > 6000) throw Error(ELBA_ERR_INTERNAL, "no RCCL id in " + path);
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        return id;
    }
};

namespace detail {
struct DevMem {                      // a device buffer that only grows
    void *p = nullptr; size_t cap = 0;
    ~DevMem() { if (p) (void)hipFree(p); }
    void reserve(size_t n) { if (n <= cap) return; if (p) (void)hipFree(p); p = nullptr; cap = 0; ELBA_DIST_HIP(hipMalloc(&p, n ? n : 8)); cap = n ? n : 8; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
}  // namespace detail

class DistributedOverlap {
public:
    DistributedOverlap(std::shared_ptr<DistGrid> grid, const Params &prm) : grid_(grid)
    {
        elba_cfg cfg{};
        cfg.k = prm.kmer_size; cfg.lower = prm.lower_kmer_freq; cfg.upper = prm.upper_kmer_freq; cfg.device = grid->device;
        engine_ = std::make_shared<detail::Engine>(cfg);
        kw_ = prm.kmer_size > 64 ? 3 : (prm.kmer_size > 32 ? 2 : 1);
    }
    std::shared_ptr<detail::Engine> engine() const { return engine_; }
    int64_t row_lo() const { return bounds_[(size_t)grid_->rank]; }
    int64_t row_hi() const { return bounds_[(size_t)grid_->rank + 1]; }
    int64_t nreads_total() const { return bounds_.back(); }

    // this rank's shard of the read set and the global partition readdispls[size + 1] (FastaIndex::getreaddispls)
    void set_reads(const DnaBuffer &mine, const std::vector<int64_t> &readdispls)
    {
        if ((int)readdispls.size() != grid_->size + 1) throw Error(ELBA_ERR_INVALID_ARG, "set_reads: partition needs size + 1 bounds");
        bounds_ = readdispls;
        my_lens_.assign(mine.lengths(), mine.lengths() + mine.size());
        engine_->check(elba_set_reads(engine_->ctx, mine.data(), mine.offsets(), mine.lengths(), (int64_t)mine.size(), row_lo()));
    }
    void packed_exchange(bool on) { packed_exchange_ = on; }      // exchange #1 in 8-byte records where an instance fits them (default on; elba_dist_packed_format)
    bool exchange_was_packed() const { return was_packed_; }

    // get_kmer_count_map_keys + _values + create_kmer_matrix + Transpose, distributed (src/main.cpp:192-273)
    void build_kmer_matrix(elba_kmer_stats *kstats = nullptr, elba_matrix_stats *mstats = nullptr)
    {
        const int W = grid_->size;
        exchange_bytes_ = 0;      // (per build: the panel reload of create_seed_matrix(false) adds to THIS build's figure)
        // owners by value range, balanced on the all-reduced histogram of the instances
        if (W > 1) {
            std::vector<uint64_t> hist(ELBA_OWNER_BINS);
            engine_->check(elba_dist_value_histogram(engine_->ctx, hist.data(), ELBA_OWNER_BINS));
            scratch_.reserve(ELBA_OWNER_BINS * 8);
            ELBA_DIST_HIP(hipMemcpyAsync(scratch_.p, hist.data(), ELBA_OWNER_BINS * 8, hipMemcpyHostToDevice, grid_->stream));
            ELBA_DIST_NCCL(ncclAllReduce(scratch_.p, scratch_.p, ELBA_OWNER_BINS, ncclUint64, ncclSum, grid_->comm, grid_->stream));
            ELBA_DIST_HIP(hipMemcpyAsync(hist.data(), scratch_.p, ELBA_OWNER_BINS * 8, hipMemcpyDeviceToHost, grid_->stream));
            ELBA_DIST_HIP(hipStreamSynchronize(grid_->stream));
            std::vector<uint32_t> upper((size_t)W);
            uint64_t total = 0;
            for (uint64_t h : hist) total += h;
            uint64_t cum = 0; size_t bin = 0;
            for (int r = 0; r < W; ++r) {                         // rank r ends at the first bin where the running count reaches (r + 1) / W of the total
                const uint64_t target = (total * (uint64_t)(r + 1) + (uint64_t)W - 1) / (uint64_t)W;
                while (bin < ELBA_OWNER_BINS && cum < target) cum += hist[bin++];
                upper[(size_t)r] = total ? (uint32_t)(bin ? bin : 1) : (uint32_t)((size_t)ELBA_OWNER_BINS * (size_t)(r + 1) / (size_t)W);
                if (r && upper[(size_t)r] < upper[(size_t)r - 1]) upper[(size_t)r] = upper[(size_t)r - 1];
            }
            upper[(size_t)W - 1] = ELBA_OWNER_BINS;
            engine_->check(elba_dist_set_owner_ranges(engine_->ctx, W, upper.data()));
        }
        // exchange #1: every instance to the owner of its k-mer
        std::vector<uint64_t> sc((size_t)W), rc;
        engine_->check(elba_dist_count_owners(engine_->ctx, W, sc.data()));
        rc = exchange_counts(sc);
        // one-word k-mers travel as 8-byte records where (value inside the owner's range, instance index in the sender's reads) fit 64 bits: every rank then
        // needs every rank's read lengths — one all-gather of 4 bytes per read (include/elba_amd.h: elba_dist_packed_format)
        int fits = 0;
        if (kw_ == 1 && packed_exchange_) {
            const std::vector<uint32_t> all = all_gather_lengths();
            int vb = 0, ib = 0;
            engine_->check(elba_dist_packed_format(engine_->ctx, W, bounds_.data(), all.data(), &fits, &vb, &ib));
        }
        was_packed_ = fits != 0;
        const size_t rw = fits ? 1 : (size_t)kw_ + 1;             // words per record
        send_.reserve(sum(sc) * rw * 8);
        const std::vector<uint64_t> soff = offsets(sc);
        if (fits) engine_->check(elba_dist_fill_send_packed(engine_->ctx, W, send_.p, soff.data()));
        else engine_->check(elba_dist_fill_send(engine_->ctx, W, send_.p, soff.data()));
        recv_.reserve(sum(rc) * rw * 8);
        all_to_all(send_, sc, recv_, rc, rw * 8);
        elba_kmer_stats ks{};
        if (fits) {      // (the owner's side: 16-byte records again, in the send buffer — it is free — or a buffer of their own when that is too small)
            detail::DevMem &exp = send_.cap >= sum(rc) * 16 ? send_ : unpacked_;
            exp.reserve(sum(rc) * 16);
            engine_->check(elba_dist_unpack_records(engine_->ctx, W, grid_->rank, recv_.p, rc.data(), exp.p));
            engine_->check(elba_dist_count_records(engine_->ctx, exp.p, (int64_t)sum(rc), &ks));
        } else
        engine_->check(elba_dist_count_records(engine_->ctx, recv_.p, (int64_t)sum(rc), &ks));
        // global k-mer ids: exclusive scan of the owners' counts (src/KmerOps.cpp:371-375)
        std::vector<uint64_t> ns = all_gather_u64((uint64_t)ks.reliable);
        uint64_t base = 0, nall = 0;
        for (int r = 0; r < W; ++r) { if (r < grid_->rank) base += ns[(size_t)r]; nall += ns[(size_t)r]; }
        engine_->check(elba_dist_set_kmer_id_base(engine_->ctx, (int64_t)base, (int64_t)nall));
        nall_ = nall;
        elba_matrix_stats ms = load_panel();
        ks.instances = (int64_t)sum(sc);
        if (kstats) *kstats = ks;
        if (mstats) *mstats = ms;
        exchange_bytes_ += sum(sc) * rw * 8;
    }

    // exchange #2: every column, whole, to each rank that owns one of its reads; the panel becomes the context's A.  Collective.  Inline partners
    // in the panel's rows follow the pair-ownership rule of the mirror exchange: written when that is how the panel will be multiplied
    // (panel_inline_exchange, default) — create_seed_matrix(false) on such a panel loads it again without them.
    elba_matrix_stats load_panel()
    {
        const int W = grid_->size;
        std::vector<uint64_t> ub(bounds_.begin(), bounds_.end()), pc((size_t)W), prc;
        engine_->check(elba_dist_panel_counts(engine_->ctx, W, ub.data(), pc.data()));
        prc = exchange_counts(pc);
        send_.reserve(sum(pc) * 16);
        const std::vector<uint64_t> poff = offsets(pc);
        engine_->check(elba_dist_panel_fill(engine_->ctx, W, ub.data(), send_.p, poff.data()));
        recv_.reserve(sum(prc) * 16);
        all_to_all(send_, pc, recv_, prc, 16);
        elba_matrix_stats ms{};
        engine_->check(elba_set_option(engine_->ctx, "panel_inline", panel_inline_ ? 1 : 0));
        engine_->check(elba_dist_set_panel(engine_->ctx, recv_.p, (int64_t)sum(prc), nreads_total(), (int64_t)nall_, row_lo(), row_hi(), &ms));
        panel_has_inline_ = panel_inline_;
        exchange_bytes_ += sum(pc) * 16;
        slot_ = 0;
        return ms;
    }

    // create_seed_matrix (include/SharedSeeds.hpp:98-99) for this rank's rows of B.  With more than one rank every pair of rows that live on
    // two ranks is accumulated by ONE of them (elba_seed_matrix_begin) and its mirror image travels to the other in one all-to-all of
    // 32-byte records (the only exchange inside the call; the reference's SUMMA stages have no other counterpart), then elba_seed_matrix_end
    // completes the rows.  exchange = false: no communication, both ranks accumulate the pair.
    void panel_inline_exchange(bool on) { panel_inline_ = on; }      // before build_kmer_matrix: false saves the reload if the panel will be multiplied with exchange = false
    elba_overlap_stats create_seed_matrix(bool exchange = true)
    {
        elba_overlap_stats st{};
        const int W = grid_->size;
        if (!exchange) {      // (every rank takes this branch together: the reload is a collective)
            if (panel_has_inline_) { panel_inline_ = false; (void)load_panel(); }
            engine_->check(elba_create_seed_matrix(engine_->ctx, &st));
            return st;
        }
        std::vector<uint64_t> ub(bounds_.begin(), bounds_.end()), sc((size_t)W), rc;
        engine_->check(elba_seed_matrix_begin(engine_->ctx, W, ub.data(), sc.data()));
        rc = exchange_counts(sc);
        send_.reserve(sum(sc) * 32);
        const std::vector<uint64_t> soff = offsets(sc);
        engine_->check(elba_seed_matrix_fill(engine_->ctx, send_.p, soff.data()));
        recv_.reserve(sum(rc) * 32);
        all_to_all(send_, sc, recv_, rc, 32);
        engine_->check(elba_seed_matrix_end(engine_->ctx, recv_.p, (int64_t)sum(rc), &st));
        mirror_bytes_ = sum(sc) * 32;
        return st;
    }
    // The same step with ONE host synchronisation (elba_set_stream / elba_seed_matrix_send / _recv): the library launches on the grid's
    // stream, the mirror images travel in fixed-size slots — grouped ncclSend / ncclRecv of equal size, nothing about them crosses the host —
    // and "did everything fit?" is answered once, at the end, by every rank alike (a flag in every slot's header): then the step is repeated
    // with the slot size that was needed.  `panel_records`: entries of this rank's panel (first slot guess, all-reduced).
    elba_overlap_stats create_seed_matrix_slots(uint64_t panel_records)
    {
        const int W = grid_->size;
        if (!shares_stream_) { engine_->check(elba_set_stream(engine_->ctx, (void *)grid_->stream)); shares_stream_ = true; }
        if (slot_ == 0) {
            uint64_t g = W > 1 ? std::max<uint64_t>(panel_records / (8 * (uint64_t)(W - 1)), 1u << 12) : (1u << 12);
            for (uint64_t x : all_gather_u64(g)) g = std::max(g, x);
            slot_ = g + 1;
        }
        std::vector<uint64_t> ub(bounds_.begin(), bounds_.end());
        for (int attempt = 0; attempt < 6; ++attempt) {
            const uint64_t slot = slot_;
            send_.reserve((size_t)W * slot * 32); recv_.reserve((size_t)W * slot * 32);
            engine_->check(elba_seed_matrix_send(engine_->ctx, W, ub.data(), send_.p, (int64_t)slot));
            ELBA_DIST_NCCL(ncclGroupStart());
            for (int p = 0; p < W; ++p) {
                ELBA_DIST_NCCL(ncclSend(send_.as<char>() + (size_t)p * slot * 32, slot * 32, ncclChar, p, grid_->comm, grid_->stream));
                ELBA_DIST_NCCL(ncclRecv(recv_.as<char>() + (size_t)p * slot * 32, slot * 32, ncclChar, p, grid_->comm, grid_->stream));
            }
            ELBA_DIST_NCCL(ncclGroupEnd());
            elba_overlap_stats st{};
            int64_t need = 0;
            const int rc = elba_seed_matrix_recv(engine_->ctx, recv_.p, (int64_t)slot, &st, &need);
            if (rc == ELBA_OK) { slot_ = std::max<uint64_t>((uint64_t)need, 2); mirror_bytes_ = (uint64_t)W * slot * 32; return st; }      // (what the step needed + 1/8, the same on every rank)
            slot_ = std::max<uint64_t>(slot_, (uint64_t)need);
            if (rc != ELBA_ERR_RETRY) engine_->check(rc);
        }
        throw std::runtime_error("create_seed_matrix_slots: the mirror exchange did not settle");
    }
    uint64_t mirror_bytes() const { return mirror_bytes_; }
    uint64_t exchange_bytes() const { return exchange_bytes_; }

private:
    static uint64_t sum(const std::vector<uint64_t> &v) { uint64_t s = 0; for (uint64_t x : v) s += x; return s; }
    static std::vector<uint64_t> offsets(const std::vector<uint64_t> &cnt)
    {
        std::vector<uint64_t> off(cnt.size());
        uint64_t at = 0;
        for (size_t r = 0; r < cnt.size(); ++r) { off[r] = at; at += cnt[r]; }
        return off;
    }
    std::vector<uint64_t> all_gather_u64(uint64_t mine)
    {
        const int W = grid_->size;
        scratch_.reserve((size_t)(W + 1) * 8);
        uint64_t *d = scratch_.as<uint64_t>();
        ELBA_DIST_HIP(hipMemcpyAsync(d + W, &mine, 8, hipMemcpyHostToDevice, grid_->stream));
        ELBA_DIST_NCCL(ncclAllGather(d + W, d, 1, ncclUint64, grid_->comm, grid_->stream));
        std::vector<uint64_t> out((size_t)W);
        ELBA_DIST_HIP(hipMemcpyAsync(out.data(), d, (size_t)W * 8, hipMemcpyDeviceToHost, grid_->stream));
        ELBA_DIST_HIP(hipStreamSynchronize(grid_->stream));
        return out;
    }
    // the lengths of ALL reads in global order: an all-gather of the shards' lengths, padded to the longest shard
    std::vector<uint32_t> all_gather_lengths()
    {
        const int W = grid_->size;
        size_t mx = 1;
        for (int r = 0; r < W; ++r) mx = std::max<size_t>(mx, (size_t)(bounds_[(size_t)r + 1] - bounds_[(size_t)r]));
        scratch_.reserve((size_t)(W + 1) * mx * 4);
        uint32_t *d = scratch_.as<uint32_t>();
        std::vector<uint32_t> pad(mx, 0u);
        std::copy(my_lens_.begin(), my_lens_.end(), pad.begin());
        ELBA_DIST_HIP(hipMemcpyAsync(d + (size_t)W * mx, pad.data(), mx * 4, hipMemcpyHostToDevice, grid_->stream));
        ELBA_DIST_NCCL(ncclAllGather(d + (size_t)W * mx, d, mx, ncclUint32, grid_->comm, grid_->stream));
        std::vector<uint32_t> got((size_t)W * mx), out;
        ELBA_DIST_HIP(hipMemcpyAsync(got.data(), d, (size_t)W * mx * 4, hipMemcpyDeviceToHost, grid_->stream));
        ELBA_DIST_HIP(hipStreamSynchronize(grid_->stream));
        out.reserve((size_t)bounds_.back());
        for (int r = 0; r < W; ++r) out.insert(out.end(), got.begin() + (size_t)r * mx, got.begin() + (size_t)r * mx + (size_t)(bounds_[(size_t)r + 1] - bounds_[(size_t)r]));
        return out;
    }
    // recv_counts[p] = what rank p sends to this rank: an all-gather of every rank's send counts (W x W words), this rank's column of it
    std::vector<uint64_t> exchange_counts(const std::vector<uint64_t> &sendcnt)
    {
        const int W = grid_->size;
        scratch_.reserve((size_t)(W * W + W) * 8);
        uint64_t *d = scratch_.as<uint64_t>();
        ELBA_DIST_HIP(hipMemcpyAsync(d + (size_t)W * W, sendcnt.data(), (size_t)W * 8, hipMemcpyHostToDevice, grid_->stream));
        ELBA_DIST_NCCL(ncclAllGather(d + (size_t)W * W, d, (size_t)W, ncclUint64, grid_->comm, grid_->stream));
        std::vector<uint64_t> all((size_t)W * W), out((size_t)W);
        ELBA_DIST_HIP(hipMemcpyAsync(all.data(), d, (size_t)W * W * 8, hipMemcpyDeviceToHost, grid_->stream));
        ELBA_DIST_HIP(hipStreamSynchronize(grid_->stream));
        for (int p = 0; p < W; ++p) out[(size_t)p] = all[(size_t)p * W + (size_t)grid_->rank];
        return out;
    }
    // all-to-all of records of `recbytes` bytes: grouped ncclSend / ncclRecv, at most 512 MiB per peer and round
    void all_to_all(const detail::DevMem &send, const std::vector<uint64_t> &sc, detail::DevMem &recv, const std::vector<uint64_t> &rc, size_t recbytes)
    {
        const int W = grid_->size;
        const std::vector<uint64_t> soff = offsets(sc), roff = offsets(rc);
        const uint64_t CH = ((uint64_t)512 << 20) / recbytes;
        uint64_t most = 0;
        for (int p = 0; p < W; ++p) { most = std::max(most, sc[(size_t)p]); most = std::max(most, rc[(size_t)p]); }
        uint64_t rounds = (most + CH - 1) / CH;
        {   // every rank must run the same number of rounds
            std::vector<uint64_t> all = all_gather_u64(rounds);
            for (uint64_t r : all) rounds = std::max(rounds, r);
        }
        for (uint64_t r = 0; r < rounds; ++r) {
            ELBA_DIST_NCCL(ncclGroupStart());
            for (int p = 0; p < W; ++p) {
                const uint64_t s0 = std::min(r * CH, sc[(size_t)p]), s1 = std::min((r + 1) * CH, sc[(size_t)p]);
                const uint64_t r0 = std::min(r * CH, rc[(size_t)p]), r1 = std::min((r + 1) * CH, rc[(size_t)p]);
                if (s1 > s0) ELBA_DIST_NCCL(ncclSend(send.as<char>() + (soff[(size_t)p] + s0) * recbytes, (s1 - s0) * recbytes, ncclChar, p, grid_->comm, grid_->stream));
                if (r1 > r0) ELBA_DIST_NCCL(ncclRecv(recv.as<char>() + (roff[(size_t)p] + r0) * recbytes, (r1 - r0) * recbytes, ncclChar, p, grid_->comm, grid_->stream));
            }
            ELBA_DIST_NCCL(ncclGroupEnd());
        }
        ELBA_DIST_HIP(hipStreamSynchronize(grid_->stream));      // the library works on its own stream: hand over explicitly
    }

    std::shared_ptr<DistGrid> grid_;
    std::shared_ptr<detail::Engine> engine_;
    std::vector<int64_t> bounds_;
    detail::DevMem send_, recv_, scratch_, unpacked_;
    std::vector<uint32_t> my_lens_;
    bool packed_exchange_ = true, was_packed_ = false;
    int kw_ = 1;
    uint64_t exchange_bytes_ = 0, mirror_bytes_ = 0, slot_ = 0;
    bool shares_stream_ = false;
    bool panel_inline_ = true, panel_has_inline_ = false;
    uint64_t nall_ = 0;
};

}  // namespace elba
