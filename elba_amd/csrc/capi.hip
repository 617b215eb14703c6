// capi.hip — the extern "C" surface declared in include/elba_amd.h.  Every entry point converts elba::Error into a status
// code + elba_last_error text; nothing here computes on the CPU except index bookkeeping of exported copies.
#include "common.hpp"
#include <new>

using namespace elba;

struct elba_ctx {
    Ctx c;
};

namespace {

template <class F>
int guarded(elba_ctx *ctx, F &&f)
{
    if (!ctx) return ELBA_ERR_INVALID_ARG;
    try {
        ELBA_HIP(hipSetDevice(ctx->c.device));
        f(ctx->c);
        return ELBA_OK;
    } catch (const Error &e) {
        ctx->c.last_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc &) {
        ctx->c.last_error = "host allocation failed";
        return ELBA_ERR_OUT_OF_MEMORY;
    } catch (...) {
        ctx->c.last_error = "unknown exception";
        return ELBA_ERR_INTERNAL;
    }
}

template <class T>
T *host_alloc(size_t n)
{
    T *p = static_cast<T *>(malloc((n ? n : 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    return p;
}

template <class T>
std::vector<T> download(Ctx &c, const void *d, size_t n)
{
    std::vector<T> h(n);
    if (n) ELBA_HIP(hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, c.stream));
    ELBA_HIP(hipStreamSynchronize(c.stream));
    return h;
}

}  // namespace

extern "C" {

int elba_abi_version(void) { return ELBA_ABI_VERSION; }

const char *elba_strerror(int status)
{
    switch (status) {
    case ELBA_OK: return "ok";
    case ELBA_ERR_INVALID_ARG: return "invalid argument";
    case ELBA_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
    case ELBA_ERR_HIP: return "HIP runtime error";
    case ELBA_ERR_OUT_OF_MEMORY: return "out of memory";
    case ELBA_ERR_STATE: return "stage called out of order";
    case ELBA_ERR_UNSUPPORTED: return "unsupported configuration";
    case ELBA_ERR_INTERNAL: return "internal error";
    case ELBA_ERR_RETRY: return "some rank ran out of room: repeat the step";
    default: return "unknown status";
    }
}

const char *elba_last_error(const elba_ctx *ctx) { return ctx ? ctx->c.last_error.c_str() : "null context"; }

int elba_ctx_create(elba_ctx **out, const elba_cfg *cfg)
{
    if (!out || !cfg) return ELBA_ERR_INVALID_ARG;
    *out = nullptr;
    // include/compiletime.h:10,21: k odd, 2 < k < 96, 0 < L <= U <= 65535.  Here: one 64-bit word per k-mer (k <= 31).
    if (cfg->k < 3 || !(cfg->k & 1) || cfg->k >= 96) return ELBA_ERR_INVALID_ARG;
    if (cfg->lower < 1 || cfg->lower > cfg->upper || cfg->upper > 65535) return ELBA_ERR_INVALID_ARG;
    /* 3 <= k <= 95, odd: the reference's range (include/compiletime.h:10); one, two or three words per k-mer */
    if (cfg->lower < 2) return ELBA_ERR_UNSUPPORTED;   // LOWER == 1 is nondeterministic in the reference
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ELBA_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return ELBA_ERR_INVALID_ARG;
    elba_ctx *ctx = new (std::nothrow) elba_ctx();
    if (!ctx) return ELBA_ERR_OUT_OF_MEMORY;
    ctx->c.cfg = *cfg;
    ctx->c.device = cfg->device;
    int rc = guarded(ctx, [&](Ctx &c) {
        ELBA_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        hipDeviceProp_t prop;
        ELBA_HIP(hipGetDeviceProperties(&prop, c.device));
        c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    });
    if (rc != ELBA_OK) { delete ctx; return rc; }
    *out = ctx;
    return ELBA_OK;
}

void elba_ctx_destroy(elba_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->c.device);
    if (ctx->c.stream) { (void)hipStreamSynchronize(ctx->c.stream); if (ctx->c.own_stream) (void)hipStreamDestroy(ctx->c.stream); }
    delete ctx;
}

static void check_reads_host(const uint64_t *byte_off, const uint32_t *len, int64_t nreads, int64_t *packed_bytes)
{
    int64_t end = 0;
    for (int64_t r = 0; r < nreads; ++r) {
        int64_t e = (int64_t)byte_off[r] + ((int64_t)len[r] + 3) / 4;
        if (e > end) end = e;
    }
    *packed_bytes = end;
}

int elba_set_reads(elba_ctx *ctx, const uint8_t *packed, const uint64_t *byte_off, const uint32_t *len, int64_t nreads, int64_t first_global_id)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(nreads >= 0 && (nreads == 0 || (packed && byte_off && len)), ELBA_ERR_INVALID_ARG, "set_reads: null array");
        ELBA_REQUIRE(nreads < 0xFFFFFFFFll, ELBA_ERR_UNSUPPORTED, "set_reads: more than 2^32-1 reads on one GPU");
        int64_t pb = 0;
        check_reads_host(byte_off, len, nreads, &pb);
        c.own_packed.reserve((size_t)pb + 16);       // +16: the enumerate kernel reads whole 8-byte windows
        c.own_byte_off.reserve((size_t)(nreads + 1) * 8);
        c.own_len.reserve((size_t)(nreads + 1) * 4);
        ELBA_HIP(hipMemsetAsync(c.own_packed.p, 0, (size_t)pb + 16, c.stream));
        if (pb) ELBA_HIP(hipMemcpyAsync(c.own_packed.p, packed, (size_t)pb, hipMemcpyHostToDevice, c.stream));
        if (nreads) {
            ELBA_HIP(hipMemcpyAsync(c.own_byte_off.p, byte_off, (size_t)nreads * 8, hipMemcpyHostToDevice, c.stream));
            ELBA_HIP(hipMemcpyAsync(c.own_len.p, len, (size_t)nreads * 4, hipMemcpyHostToDevice, c.stream));
        }
        ELBA_HIP(hipStreamSynchronize(c.stream));
        c.d_packed = c.own_packed.as<uint8_t>(); c.d_byte_off = c.own_byte_off.as<uint64_t>(); c.d_len = c.own_len.as<uint32_t>();
        c.h_len.assign(len, len + nreads);
        c.h_byte_off.assign(byte_off, byte_off + nreads);
        c.nreads = nreads; c.first_global_id = first_global_id; c.packed_bytes = pb;
        // a new read set invalidates everything derived from the old one (as stage_set_reads_fasta does)
        c.have_reads = true; c.have_counts = false; c.have_aln = false; c.have_edges = false; c.have_S = false;
        if (c.A_has_kmers) { c.have_A = false; c.have_B = false; }      // (an A handed over as triples / a panel does not come from these reads)
    });
}

int elba_set_reads_fasta(elba_ctx *ctx, const char *chunk, int64_t chunk_bytes, uint64_t chunk_file_offset, const elba_fasta_record_t *recs, int64_t nreads,
                         int64_t first_global_id, elba_ingest_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) { stage_set_reads_fasta(c, chunk, chunk_bytes, chunk_file_offset, recs, nreads, first_global_id, stats); });
}

int elba_export_reads(elba_ctx *ctx, uint8_t *packed, int64_t packed_capacity, uint64_t *byte_off, uint32_t *len, int64_t nreads_capacity)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(c.have_reads, ELBA_ERR_STATE, "export_reads: no reads");
        if (packed) {
            ELBA_REQUIRE(packed_capacity >= c.packed_bytes, ELBA_ERR_INVALID_ARG, "export_reads: packed buffer too small");
            if (c.packed_bytes) ELBA_HIP(hipMemcpyAsync(packed, c.d_packed, (size_t)c.packed_bytes, hipMemcpyDeviceToHost, c.stream));
        }
        if (byte_off || len) ELBA_REQUIRE(nreads_capacity >= c.nreads, ELBA_ERR_INVALID_ARG, "export_reads: offset/length arrays too small");
        if (byte_off && c.nreads) ELBA_HIP(hipMemcpyAsync(byte_off, c.d_byte_off, (size_t)c.nreads * 8, hipMemcpyDeviceToHost, c.stream));
        if (len && c.nreads) ELBA_HIP(hipMemcpyAsync(len, c.d_len, (size_t)c.nreads * 4, hipMemcpyDeviceToHost, c.stream));
        ELBA_HIP(hipStreamSynchronize(c.stream));
    });
}

int elba_set_reads_device(elba_ctx *ctx, const void *d_packed, int64_t packed_bytes, const void *d_byte_off, const void *d_len, int64_t nreads, int64_t first_global_id)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(nreads >= 0 && packed_bytes >= 0 && (nreads == 0 || (d_packed && d_byte_off && d_len)), ELBA_ERR_INVALID_ARG, "set_reads_device: null array");
        ELBA_REQUIRE(nreads < 0xFFFFFFFFll, ELBA_ERR_UNSUPPORTED, "set_reads_device: more than 2^32-1 reads on one GPU");
        // the packed buffer is copied once device-to-device so that the 16 guard bytes behind it exist; offsets/lengths are borrowed
        c.own_packed.reserve((size_t)packed_bytes + 16);
        ELBA_HIP(hipMemsetAsync(c.own_packed.p, 0, (size_t)packed_bytes + 16, c.stream));
        if (packed_bytes) ELBA_HIP(hipMemcpyAsync(c.own_packed.p, d_packed, (size_t)packed_bytes, hipMemcpyDeviceToDevice, c.stream));
        c.d_packed = c.own_packed.as<uint8_t>();
        c.d_byte_off = static_cast<const uint64_t *>(d_byte_off);
        c.d_len = static_cast<const uint32_t *>(d_len);
        c.h_len = download<uint32_t>(c, d_len, (size_t)nreads);
        c.h_byte_off = download<uint64_t>(c, d_byte_off, (size_t)nreads);
        for (int64_t r = 0; r < nreads; ++r)
            ELBA_REQUIRE((int64_t)c.h_byte_off[r] + ((int64_t)c.h_len[r] + 3) / 4 <= packed_bytes, ELBA_ERR_INVALID_ARG, "set_reads_device: read exceeds the packed buffer");
        c.nreads = nreads; c.first_global_id = first_global_id; c.packed_bytes = packed_bytes;
        // a new read set invalidates everything derived from the old one (as stage_set_reads_fasta does)
        c.have_reads = true; c.have_counts = false; c.have_aln = false; c.have_edges = false; c.have_S = false;
        if (c.A_has_kmers) { c.have_A = false; c.have_B = false; }      // (an A handed over as triples / a panel does not come from these reads)
    });
}

int elba_count_kmers(elba_ctx *ctx, elba_kmer_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        stage_count_kmers(c);
        if (stats) *stats = c.kstats;
    });
}

int elba_create_kmer_matrix(elba_ctx *ctx, elba_matrix_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        EventTimer t;
        t.start(c.stream);
        stage_create_kmer_matrix(c);
        t.stop(c.stream);
        if (stats) { stats->nrows = c.M; stats->ncols = c.N; stats->nnz = c.Z; stats->max_row_nnz = c.max_row_nnz; stats->ms_total = t.ms(); }
    });
}

int elba_set_kmer_matrix(elba_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t *rows, const int64_t *cols, const uint32_t *vals, elba_matrix_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        EventTimer t;
        t.start(c.stream);
        stage_set_kmer_matrix(c, nrows, ncols, nnz, rows, cols, vals);
        t.stop(c.stream);
        if (stats) { stats->nrows = c.M; stats->ncols = c.N; stats->nnz = c.Z; stats->max_row_nnz = c.max_row_nnz; stats->ms_total = t.ms(); }
    });
}

int elba_set_kmer_matrix_device(elba_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz, const void *d_rows, const void *d_cols, const void *d_vals, elba_matrix_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        EventTimer t;
        t.start(c.stream);
        stage_set_kmer_matrix_device(c, nrows, ncols, nnz, static_cast<const int64_t *>(d_rows), static_cast<const int64_t *>(d_cols), static_cast<const uint32_t *>(d_vals));
        t.stop(c.stream);
        if (stats) { stats->nrows = c.M; stats->ncols = c.N; stats->nnz = c.Z; stats->max_row_nnz = c.max_row_nnz; stats->ms_total = t.ms(); }
    });
}

int elba_export_triples_device(elba_ctx *ctx, void *d_rows, void *d_cols, void *d_vals)
{
    return guarded(ctx, [&](Ctx &c) { stage_export_triples_device(c, static_cast<int64_t *>(d_rows), static_cast<int64_t *>(d_cols), static_cast<uint32_t *>(d_vals)); });
}

int elba_create_seed_matrix(elba_ctx *ctx, elba_overlap_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        stage_create_seed_matrix(c);
        if (stats) *stats = c.ostats;
    });
}

int elba_seed_matrix_begin(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, uint64_t *send_counts)
{
    return guarded(ctx, [&](Ctx &c) { stage_seed_matrix_begin(c, nranks, read_bounds, send_counts); });
}

int elba_set_stream(elba_ctx *ctx, void *hip_stream)
{
    return guarded(ctx, [&](Ctx &c) {
        if (c.stream) ELBA_HIP(hipStreamSynchronize(c.stream));
        if (c.own_stream && c.stream) (void)hipStreamDestroy(c.stream);
        c.stream = static_cast<hipStream_t>(hip_stream);
        c.own_stream = false;
    });
}

int elba_seed_matrix_send(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, void *d_send, int64_t slot_records)
{
    return guarded(ctx, [&](Ctx &c) { stage_seed_matrix_send(c, nranks, read_bounds, d_send, slot_records); });
}

int elba_seed_matrix_recv(elba_ctx *ctx, void *d_recv, int64_t slot_records, elba_overlap_stats *stats, int64_t *slot_records_needed)
{
    bool done = false;
    const int rc = guarded(ctx, [&](Ctx &c) {
        done = stage_seed_matrix_recv(c, d_recv, slot_records, slot_records_needed);
        if (done && stats) *stats = c.ostats;
    });
    return rc != ELBA_OK ? rc : (done ? ELBA_OK : ELBA_ERR_RETRY);
}

int elba_seed_matrix_fill(elba_ctx *ctx, void *d_send, const uint64_t *offsets)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(offsets, ELBA_ERR_INVALID_ARG, "seed_matrix_fill: null offsets");
        stage_seed_matrix_fill(c, d_send, offsets);
    });
}

int elba_seed_matrix_end(elba_ctx *ctx, const void *d_recv, int64_t nrecords, elba_overlap_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        stage_seed_matrix_end(c, d_recv, nrecords);
        if (stats) *stats = c.ostats;
    });
}

int elba_align_seeds(elba_ctx *ctx, int mat, int mis, int gap, int dropoff, elba_align_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        stage_align_seeds(c, mat, mis, gap, dropoff);
        if (stats) *stats = c.astats;
    });
}

int elba_dist_set_all_reads(elba_ctx *ctx, const void *d_packed, int64_t packed_bytes, const void *d_byte_off, const void *d_len, int64_t nreads_total)
{
    return guarded(ctx, [&](Ctx &c) { stage_dist_set_all_reads(c, d_packed, packed_bytes, d_byte_off, d_len, nreads_total); });
}

int elba_export_overlaps(elba_ctx *ctx, elba_overlaps_t *out)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(out, ELBA_ERR_INVALID_ARG, "export_overlaps: null output");
        memset(out, 0, sizeof(*out));
        ELBA_REQUIRE(c.have_aln, ELBA_ERR_STATE, "export_overlaps: no alignments (call elba_align_seeds)");
        const int64_t n = c.naln;
        out->n = n;
        out->rows = host_alloc<int64_t>((size_t)n); out->cols = host_alloc<int64_t>((size_t)n); out->vals = host_alloc<elba_overlap_t>((size_t)n);
        if (n) {
            ELBA_HIP(hipMemcpyAsync(out->rows, c.aln_rows.p, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
            ELBA_HIP(hipMemcpyAsync(out->cols, c.aln_cols.p, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
            ELBA_HIP(hipMemcpyAsync(out->vals, c.aln_out.p, (size_t)n * sizeof(elba_overlap_t), hipMemcpyDeviceToHost, c.stream));
        }
        ELBA_HIP(hipStreamSynchronize(c.stream));
        for (int64_t a = 0; a < n; ++a) { out->rows[a] += c.first_global_id_rows(); out->cols[a] += c.first_global_id_rows(); }
    });
}

void elba_free_overlaps(elba_overlaps_t *o)
{
    if (!o) return;
    free(o->rows); free(o->cols); free(o->vals);
    memset(o, 0, sizeof(*o));
}

int elba_set_overlaps(elba_ctx *ctx, int64_t nreads, const int64_t *rows, const int64_t *cols, const elba_overlap_t *vals, int64_t n)
{
    return guarded(ctx, [&](Ctx &c) { stage_set_overlaps(c, nreads, rows, cols, vals, n); });
}

int elba_transitive_reduction(elba_ctx *ctx, double bad_read_cutoff, int fuzz, elba_string_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        stage_transitive_reduction(c, bad_read_cutoff, fuzz);
        if (stats) *stats = c.sstats;
    });
}

int elba_export_string_graph(elba_ctx *ctx, elba_overlaps_t *out)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(out, ELBA_ERR_INVALID_ARG, "export_string_graph: null output");
        memset(out, 0, sizeof(*out));
        ELBA_REQUIRE(c.have_S, ELBA_ERR_STATE, "export_string_graph: no string graph (call elba_transitive_reduction)");
        const int64_t n = c.tr_nnz;
        out->n = n;
        out->rows = host_alloc<int64_t>((size_t)n); out->cols = host_alloc<int64_t>((size_t)n); out->vals = host_alloc<elba_overlap_t>((size_t)n);
        if (n) {
            ELBA_HIP(hipMemcpyAsync(out->rows, c.tr_out_rows.p, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
            ELBA_HIP(hipMemcpyAsync(out->cols, c.tr_out_cols.p, (size_t)n * 8, hipMemcpyDeviceToHost, c.stream));
            ELBA_HIP(hipMemcpyAsync(out->vals, c.tr_out_vals.p, (size_t)n * sizeof(elba_overlap_t), hipMemcpyDeviceToHost, c.stream));
        }
        ELBA_HIP(hipStreamSynchronize(c.stream));
        for (int64_t a = 0; a < n; ++a) { out->rows[a] += c.tr_id_base; out->cols[a] += c.tr_id_base; }
    });
}

int elba_export_read_flags(elba_ctx *ctx, uint8_t *flags, int64_t nreads)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(c.have_S, ELBA_ERR_STATE, "export_read_flags: no string graph (call elba_transitive_reduction)");
        ELBA_REQUIRE(nreads == c.tr_M && (flags || nreads == 0), ELBA_ERR_INVALID_ARG, "export_read_flags: need one byte per read of the graph");
        if (nreads) ELBA_HIP(hipMemcpyAsync(flags, c.tr_flags.p, (size_t)nreads, hipMemcpyDeviceToHost, c.stream));
        ELBA_HIP(hipStreamSynchronize(c.stream));
    });
}

int elba_export_csr(elba_ctx *ctx, int64_t row_lo, int64_t row_hi, elba_csr_t *out)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(out, ELBA_ERR_INVALID_ARG, "export_csr: null output");
        memset(out, 0, sizeof(*out));
        ELBA_REQUIRE(c.have_B, ELBA_ERR_STATE, "export_csr: no seed matrix");
        ELBA_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= c.M, ELBA_ERR_INVALID_ARG, "export_csr: bad row range");
        auto rp = download<int64_t>(c, c.b_rowptr.as<int64_t>() + row_lo, (size_t)(row_hi - row_lo + 1));
        const int64_t e0 = rp.front(), e1 = rp.back(), n = e1 - e0;
        auto col = download<uint32_t>(c, c.b_col.as<uint32_t>() + e0, (size_t)n);
        out->nrows = row_hi - row_lo; out->ncols = c.M; out->nnz = n;
        out->rowptr = host_alloc<int64_t>((size_t)(row_hi - row_lo + 1));
        out->col = host_alloc<int64_t>((size_t)n);
        out->val = host_alloc<elba_seed_t>((size_t)n);
        for (size_t i = 0; i < rp.size(); ++i) out->rowptr[i] = rp[i] - e0;
        for (int64_t i = 0; i < n; ++i) out->col[i] = col[(size_t)i];
        if (n) ELBA_HIP(hipMemcpyAsync(out->val, c.b_val.as<elba_seed_t>() + e0, (size_t)n * sizeof(elba_seed_t), hipMemcpyDeviceToHost, c.stream));
        ELBA_HIP(hipStreamSynchronize(c.stream));
    });
}

void elba_free_csr(elba_csr_t *c)
{
    if (!c) return;
    free(c->rowptr); free(c->col); free(c->val);
    memset(c, 0, sizeof(*c));
}

int elba_export_dcsc(elba_ctx *ctx, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi, elba_dcsc_t *out)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(out, ELBA_ERR_INVALID_ARG, "export_dcsc: null output");
        memset(out, 0, sizeof(*out));
        ELBA_REQUIRE(c.have_B, ELBA_ERR_STATE, "export_dcsc: no seed matrix");
        ELBA_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= c.M && col_lo >= 0 && col_lo <= col_hi && col_hi <= c.M, ELBA_ERR_INVALID_ARG, "export_dcsc: bad block");
        // rows [row_lo,row_hi) of the device CSR -> host; column-major regrouping is index bookkeeping on the exported copy
        auto rp = download<int64_t>(c, c.b_rowptr.as<int64_t>() + row_lo, (size_t)(row_hi - row_lo + 1));
        const int64_t e0 = rp.front(), n = rp.back() - e0;
        auto col = download<uint32_t>(c, c.b_col.as<uint32_t>() + e0, (size_t)n);
        auto val = download<elba_seed_t>(c, c.b_val.as<elba_seed_t>() + e0, (size_t)n);
        const int64_t ncols = col_hi - col_lo;
        std::vector<int64_t> cnt((size_t)ncols + 1, 0);
        int64_t total = 0;
        for (int64_t e = 0; e < n; ++e) { int64_t j = col[(size_t)e]; if (j >= col_lo && j < col_hi) { cnt[(size_t)(j - col_lo)]++; ++total; } }
        int64_t nzc = 0;
        for (int64_t j = 0; j < ncols; ++j) if (cnt[(size_t)j]) ++nzc;
        out->nrows = row_hi - row_lo; out->ncols = ncols; out->nnz = total; out->nzc = nzc;
        out->jc = host_alloc<int64_t>((size_t)nzc);
        out->cp = host_alloc<int64_t>((size_t)nzc + 1);
        out->ir = host_alloc<int64_t>((size_t)total);
        out->numx = host_alloc<elba_seed_t>((size_t)total);
        std::vector<int64_t> start((size_t)ncols + 1, 0);
        int64_t run = 0, ci = 0;
        for (int64_t j = 0; j < ncols; ++j) {
            start[(size_t)j] = run;
            if (cnt[(size_t)j]) { out->jc[ci] = j; out->cp[ci] = run; ++ci; }
            run += cnt[(size_t)j];
        }
        out->cp[nzc] = run;
        for (int64_t r = 0; r < row_hi - row_lo; ++r)
            for (int64_t e = rp[(size_t)r] - e0; e < rp[(size_t)r + 1] - e0; ++e) {
                int64_t j = col[(size_t)e];
                if (j >= col_lo && j < col_hi) { int64_t d = start[(size_t)(j - col_lo)]++; out->ir[d] = r; out->numx[d] = val[(size_t)e]; }
            }
    });
}

void elba_free_dcsc(elba_dcsc_t *d)
{
    if (!d) return;
    free(d->jc); free(d->cp); free(d->ir); free(d->numx);
    memset(d, 0, sizeof(*d));
}

int elba_export_kmer_matrix(elba_ctx *ctx, elba_kmer_matrix_t *out)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(out, ELBA_ERR_INVALID_ARG, "export_kmer_matrix: null output");
        memset(out, 0, sizeof(*out));
        ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "export_kmer_matrix: no k-mer matrix");
        const size_t M = (size_t)c.M, N = (size_t)c.N, Z = (size_t)c.Z;
        auto rp = download<uint32_t>(c, c.a_rowptr.p, M + 1);
        auto cp = download<uint32_t>(c, c.a_colptr.p, N + 1);
        auto csr = download<uint64_t>(c, c.a_csr.p, Z);
        auto csc = download<uint64_t>(c, c.a_csc.p, Z);
        out->nrows = c.M; out->ncols = c.N; out->nnz = c.Z;
        out->colptr = host_alloc<int64_t>(N + 1); out->rowptr = host_alloc<int64_t>(M + 1);
        out->csc_row = host_alloc<int64_t>(Z); out->csc_val = host_alloc<uint32_t>(Z);
        out->csr_col = host_alloc<int64_t>(Z); out->csr_val = host_alloc<uint32_t>(Z);
        for (size_t i = 0; i <= N; ++i) out->colptr[i] = cp[i];
        for (size_t i = 0; i <= M; ++i) out->rowptr[i] = rp[i];
        for (size_t z = 0; z < Z; ++z) {
            out->csc_row[z] = (int64_t)(csc[z] >> 32) + c.first_global_id_rows(); out->csc_val[z] = (uint32_t)csc[z];
            out->csr_col[z] = (int64_t)(csr[z] >> 32); out->csr_val[z] = (uint32_t)csr[z] & (c.csr_suffix ? 0xFFFFu : (c.csr_hints ? 0x3FFFFFFFu : 0xFFFFFFFFu));      // (above the position: SpGEMM hints, or column length and place of a dense matrix)
        }
        if (c.csr_inline) {
            // rows with inline partners (Ctx::csr_inline) do not name the k-mer of such an entry: the rows are rebuilt from the columns — walked in
            // k-mer order, every column's entries in (read, pos) order, a row fills up in (k-mer id, pos) order
            std::vector<size_t> fill(rp.begin(), rp.end() - 1);
            for (size_t k = 0; k < N; ++k)
                for (size_t z = cp[k]; z < cp[k + 1]; ++z) { const size_t at = fill[(size_t)(csc[z] >> 32)]++; out->csr_col[at] = (int64_t)k; out->csr_val[at] = (uint32_t)csc[z]; }
        }
        if (c.A_has_kmers) {
            out->kmers = host_alloc<uint64_t>(N);
            if (N) ELBA_HIP(hipMemcpyAsync(out->kmers, c.rel_kmers.p, N * 8, hipMemcpyDeviceToHost, c.stream));
            if (c.cfg.k > 32) {
                out->kmers_lo = host_alloc<uint64_t>(N);
                if (N) ELBA_HIP(hipMemcpyAsync(out->kmers_lo, c.rel_kmers_lo.p, N * 8, hipMemcpyDeviceToHost, c.stream));
            }
            if (c.cfg.k > 64) {
                out->kmers_lo2 = host_alloc<uint64_t>(N);
                if (N) ELBA_HIP(hipMemcpyAsync(out->kmers_lo2, c.rel_kmers_lo2.p, N * 8, hipMemcpyDeviceToHost, c.stream));
            }
            ELBA_HIP(hipStreamSynchronize(c.stream));
        }
    });
}

void elba_free_kmer_matrix(elba_kmer_matrix_t *m)
{
    if (!m) return;
    free(m->kmers); free(m->kmers_lo); free(m->kmers_lo2); free(m->colptr); free(m->csc_row); free(m->csc_val); free(m->rowptr); free(m->csr_col); free(m->csr_val);
    memset(m, 0, sizeof(*m));
}

int elba_kmer_histogram(elba_ctx *ctx, int64_t *hist, int64_t len)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(hist && len > 0, ELBA_ERR_INVALID_ARG, "kmer_histogram: null output");
        ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "kmer_histogram: no k-mer matrix");
        auto cp = download<uint32_t>(c, c.a_colptr.p, (size_t)c.N + 1);
        for (int64_t i = 0; i < len; ++i) hist[i] = 0;
        for (int64_t k = 0; k < c.N; ++k) { int64_t n = (int64_t)cp[(size_t)k + 1] - cp[(size_t)k]; if (n < len) hist[n]++; }
    });
}

int elba_get_device_view(elba_ctx *ctx, elba_device_view *v)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(v, ELBA_ERR_INVALID_ARG, "get_device_view: null output");
        memset(v, 0, sizeof(*v));
        v->stream = (void *)c.stream;
        if (c.have_A) { v->M = c.M; v->N = c.N; v->Z = c.Z; v->a_rowptr = c.a_rowptr.p; v->a_csr = c.a_csr.p; v->a_colptr = c.a_colptr.p; v->a_csc = c.a_csc.p;
                        v->a_csr_format = c.csr_inline ? ELBA_CSR_INLINE : (c.csr_suffix ? ELBA_CSR_DENSE : (c.csr_hints ? ELBA_CSR_HINTS : ELBA_CSR_PLAIN));
                        v->a_csr_pos_mask = c.csr_suffix ? 0xFFFFu : (c.csr_hints ? 0x3FFFFFFFu : 0xFFFFFFFFu);
                        v->a_kmers = c.A_has_kmers ? c.rel_kmers.p : nullptr;
                        v->a_gather_slots = c.use_ell && c.ell_compact ? c.ell_nslots : 0;
                        v->a_slot_kid = c.use_ell && c.ell_compact ? c.ell_slot_kid.p : nullptr; }
        if (c.have_B) { v->Y = c.Y; v->b_rowptr = c.b_rowptr.p; v->b_col = c.b_col.p; v->b_val = c.b_val.p; }
    });
}

int elba_set_option(elba_ctx *ctx, const char *name, int64_t value)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(name, ELBA_ERR_INVALID_ARG, "set_option: null name");
        struct { const char *n; bool *b; } flags[] = {
            {"overlap_cold_calls", &c.cold_calls}, {"no_symmetry", &c.opt.no_symmetry}, {"no_ell", &c.opt.no_ell}, {"no_pay", &c.opt.no_pay}, {"mir32", &c.opt.mir32},
            {"no_hints", &c.opt.no_hints}, {"no_sample", &c.opt.no_sample}, {"no_slab", &c.opt.no_slab}, {"no_ell_compact", &c.opt.no_ell_compact}, {"msd_no_emit8", &c.opt.msd_no_emit8}, {"msd_no_rank", &c.opt.msd_no_rank}, {"msd_rank", &c.opt.msd_rank}, {"csr_pairs_late", &c.opt.csr_pairs_late}, {"no_suffix", &c.opt.no_suffix}, {"no_row_order", &c.opt.no_row_order}, {"no_inline", &c.opt.no_inline}, {"panel_inline", &c.opt.panel_inline}, {"kmer_pairs", &c.opt.kmer_pairs},
            {"kmer_unfused", &c.opt.kmer_unfused}, {"kmer_no_msd", &c.opt.kmer_no_msd}, {"kmer_msd", &c.opt.kmer_msd}, {"csr_pairs", &c.opt.csr_pairs}, {"emit_plain", &c.opt.emit_plain}, {"trace", &c.opt.trace}, {"measure_prep", &c.opt.measure_prep}};
        for (auto &f : flags) if (!strcmp(name, f.n)) { *f.b = value != 0; return; }
        if (!strcmp(name, "kmer_drop")) { ELBA_REQUIRE(value >= 0 && value <= 3, ELBA_ERR_INVALID_ARG, "set_option: kmer_drop is 0..3"); c.opt.kmer_drop = (int)value; }
        else if (!strcmp(name, "dense_up")) { ELBA_REQUIRE(value >= 0 && value <= 3, ELBA_ERR_INVALID_ARG, "set_option: dense_up is 0..3"); c.opt.dense_up = (int)value; }
        else if (!strcmp(name, "dense_wgs")) { ELBA_REQUIRE(value >= 1 && value <= 16, ELBA_ERR_INVALID_ARG, "set_option: dense_wgs is 1..16"); c.opt.dense_wgs = (int)value; }
        else if (!strncmp(name, "tune", 4) && name[4] >= '0' && name[4] <= '7' && name[5] == 0) c.opt.tune[name[4] - '0'] = value;
        else if (!strcmp(name, "msd_small_cap")) c.opt.msd_small_cap = (int)value;
        else if (!strcmp(name, "kmer_batch_instances")) { ELBA_REQUIRE(value >= 0, ELBA_ERR_INVALID_ARG, "set_option: kmer_batch_instances is >= 0"); c.opt.kmer_batch_instances = value; }
        else if (!strcmp(name, "msd_wide_bits")) c.opt.msd_wide_bits = (int)value;
        else if (!strcmp(name, "ell_slot_cap")) c.opt.ell_slot_cap = (int)value;
        else if (!strcmp(name, "slab_pct")) { ELBA_REQUIRE(value >= 1 && value <= 1000, ELBA_ERR_INVALID_ARG, "set_option: slab_pct is 1..1000"); c.opt.slab_pct = (int)value; }
        else if (!strcmp(name, "slab_q16")) { ELBA_REQUIRE(value >= 0 && value < (1ll << 31), ELBA_ERR_INVALID_ARG, "set_option: slab_q16 is 0..2^31"); c.opt.slab_q16 = (int)value; }
        else if (!strcmp(name, "dk")) { ELBA_REQUIRE(value == -1 || value == 0 || value == 1 || value == 2 || value == 4, ELBA_ERR_INVALID_ARG, "set_option: dk is -1 (chosen per matrix), 0, 1, 2 or 4"); c.opt.dk = (int)value; }
        else if (!strcmp(name, "aln_tiers")) { ELBA_REQUIRE(value >= 0, ELBA_ERR_INVALID_ARG, "set_option: aln_tiers is a string of the digits 1, 2, 4, 8"); c.opt.aln_tiers = value; }
        else if (!strcmp(name, "aln_wide_hint")) c.opt.aln_wide_hint = (int)value;
        else if (!strcmp(name, "aln_long_hint")) c.opt.aln_long_hint = (int)value;
        else throw Error{ELBA_ERR_INVALID_ARG, std::string("set_option: unknown option ") + name};
    });
}

int elba_release_workspace(elba_ctx *ctx)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(!c.dist_owner && c.ov_phase == 0, ELBA_ERR_STATE, "release_workspace: the context holds exchanged records or is inside a sharded multiplication");
        ELBA_HIP(hipStreamSynchronize(c.stream));
        c.ws_a.release(); c.ws_b.release(); c.ws_c.release(); c.ws_d.release(); c.ws_e.release(); c.ws_f.release(); c.ws_g.release(); c.ws_h.release(); c.ws_sort.release();
        c.csr_words.release(); c.kid_of_entry.release();
        if (c.have_counts) { c.pre_ready = false; c.pre_consumed = true; }    // (the CSR sort keys / column ids of the entries are gone: create_kmer_matrix rebuilds them from the column pointers)
    });
}

int elba_get_stat(elba_ctx *ctx, const char *name, int64_t *value)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(name && value, ELBA_ERR_INVALID_ARG, "get_stat: null name or value");
        if (!strcmp(name, "overlap_mirror_placed")) *value = c.ov_mir_placed;
        else if (!strcmp(name, "overlap_slab_q16")) *value = (int64_t)c.ov_slab_q16_used;
        else if (!strcmp(name, "kmer_path")) *value = c.kmer_path;
        else if (!strcmp(name, "kmer_passes")) *value = c.kmer_passes;      // value-range passes the last elba_count_kmers took (1: the whole input at once)
        else if (!strcmp(name, "spgemm_prep_us")) *value = c.prep_us;      // (option "measure_prep"; -1: not measured — the option was off, or the path taken has no emit kernels of its own)
        else if (!strcmp(name, "emit_us")) *value = c.emit_us;
        else if (!strcmp(name, "triples_path")) *value = c.triples_path;
        else if (!strcmp(name, "padded_columns")) *value = c.have_A && c.use_ell ? 1 : 0;
        else if (!strcmp(name, "gather_slots")) *value = c.have_A && c.use_ell ? c.ell_nslots : 0;
        else if (!strcmp(name, "resident_bytes_A")) {
            int64_t b = 0;
            if (c.have_A) {
                b = (int64_t)(c.M + 1) * 4 + (int64_t)(c.N + 1) * 4 + 16 * c.Z;
                if (c.use_ell) b += 8 * c.ell_nslots * (int64_t)c.s_stride + (c.ell_compact ? 4 * c.ell_nslots : 0);
                if (c.csr_suffix) b += (4ll * c.N) << c.j_shift;
            }
            *value = b;
        }
        else throw Error{ELBA_ERR_INVALID_ARG, std::string("get_stat: unknown counter ") + name};
    });
}

int elba_kmer_hash_owner(elba_ctx *ctx, const uint64_t *kmers, int64_t n, int nprocs, uint64_t *hash, int32_t *owner)
{
    return guarded(ctx, [&](Ctx &c) { stage_ref_hash_owner(c, kmers, n, nprocs, hash, owner); });
}

int elba_dist_value_histogram(elba_ctx *ctx, uint64_t *hist, int64_t nbins)
{
    return guarded(ctx, [&](Ctx &c) { stage_dist_value_histogram(c, hist, nbins); });
}

int elba_dist_set_owner_ranges(elba_ctx *ctx, int nranks, const uint32_t *upper_bins)
{
    return guarded(ctx, [&](Ctx &c) { stage_dist_set_owner_ranges(c, nranks, upper_bins); });
}

int elba_dist_set_kmer_id_base(elba_ctx *ctx, int64_t base, int64_t nall)
{
    return guarded(ctx, [&](Ctx &c) { stage_dist_set_kmer_id_base(c, base, nall); });
}

int elba_dist_count_owners(elba_ctx *ctx, int nranks, uint64_t *counts)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(counts, ELBA_ERR_INVALID_ARG, "dist_count_owners: null output");
        stage_dist_count_owners(c, nranks, counts);
    });
}

int elba_dist_fill_send(elba_ctx *ctx, int nranks, void *d_send, const uint64_t *offsets)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(offsets && (d_send || c.I == 0), ELBA_ERR_INVALID_ARG, "dist_fill_send: null argument");
        stage_dist_fill_send(c, nranks, d_send, offsets);
    });
}

int elba_dist_packed_format(elba_ctx *ctx, int nranks, const int64_t *read_bounds, const uint32_t *all_lens, int *fits, int *value_bits, int *index_bits)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(fits, ELBA_ERR_INVALID_ARG, "dist_packed_format: null output");
        *fits = stage_dist_packed_format(c, nranks, read_bounds, all_lens, value_bits, index_bits) ? 1 : 0;
    });
}

int elba_dist_fill_send_packed(elba_ctx *ctx, int nranks, void *d_send, const uint64_t *offsets)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(offsets && (d_send || c.I == 0), ELBA_ERR_INVALID_ARG, "dist_fill_send_packed: null argument");
        stage_dist_fill_send_packed(c, nranks, d_send, offsets);
    });
}

int elba_dist_unpack_records(elba_ctx *ctx, int nranks, int rank, const void *d_packed, const uint64_t *recv_counts, void *d_records)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(recv_counts, ELBA_ERR_INVALID_ARG, "dist_unpack_records: null counts");
        stage_dist_unpack_records(c, nranks, rank, d_packed, recv_counts, d_records);
    });
}

int elba_dist_count_records(elba_ctx *ctx, const void *d_records, int64_t nrecords, elba_kmer_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        stage_dist_count_records(c, d_records, nrecords);
        if (stats) *stats = c.kstats;
    });
}

int elba_dist_get_reliable_kmers(elba_ctx *ctx, const void **d_kmers, int64_t *n)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(d_kmers && n, ELBA_ERR_INVALID_ARG, "dist_get_reliable_kmers: null output");
        ELBA_REQUIRE(c.have_counts && c.dist_owner, ELBA_ERR_STATE, "dist_get_reliable_kmers: call dist_count_records first");
        ELBA_REQUIRE(c.cfg.k <= 31, ELBA_ERR_UNSUPPORTED, "dist_get_reliable_kmers: multi-word k-mers are handed out interleaved by elba_dist_copy_reliable_kmers");
        *d_kmers = c.rel_kmers.p; *n = c.own_N;
    });
}

int elba_dist_copy_reliable_kmers(elba_ctx *ctx, void *d_dst, int64_t capacity)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(c.have_counts && c.dist_owner, ELBA_ERR_STATE, "dist_copy_reliable_kmers: call dist_count_records first");
        ELBA_REQUIRE(capacity >= c.own_N && (d_dst || c.own_N == 0), ELBA_ERR_INVALID_ARG, "dist_copy_reliable_kmers: buffer too small");
        stage_dist_copy_reliable_kmers(c, d_dst);
    });
}

int elba_dist_set_global_kmers(elba_ctx *ctx, const void *d_all_kmers, int64_t nall)
{
    return guarded(ctx, [&](Ctx &c) { stage_dist_set_global_kmers(c, d_all_kmers, nall); });
}

int elba_dist_panel_counts(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, uint64_t *counts)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(read_bounds && counts, ELBA_ERR_INVALID_ARG, "dist_panel_counts: null argument");
        stage_dist_panel(c, nranks, read_bounds, nullptr, nullptr, false, nullptr, counts);
    });
}

int elba_dist_panel_fill(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, void *d_send, const uint64_t *offsets)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(read_bounds && offsets, ELBA_ERR_INVALID_ARG, "dist_panel_fill: null argument");
        stage_dist_panel(c, nranks, read_bounds, nullptr, nullptr, true, d_send, const_cast<uint64_t *>(offsets));
    });
}

int elba_dist_panel_counts_win(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, const uint64_t *win_lo, const uint64_t *win_hi, uint64_t *counts)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(read_bounds && win_lo && win_hi && counts, ELBA_ERR_INVALID_ARG, "dist_panel_counts_win: null argument");
        stage_dist_panel(c, nranks, read_bounds, win_lo, win_hi, false, nullptr, counts);
    });
}

int elba_dist_panel_fill_win(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, const uint64_t *win_lo, const uint64_t *win_hi, void *d_send, const uint64_t *offsets)
{
    return guarded(ctx, [&](Ctx &c) {
        ELBA_REQUIRE(read_bounds && win_lo && win_hi && offsets, ELBA_ERR_INVALID_ARG, "dist_panel_fill_win: null argument");
        stage_dist_panel(c, nranks, read_bounds, win_lo, win_hi, true, d_send, const_cast<uint64_t *>(offsets));
    });
}

int elba_dist_set_panel(elba_ctx *ctx, const void *d_records, int64_t nrecords, int64_t nreads_total, int64_t nkmers_total, int64_t row_lo, int64_t row_hi, elba_matrix_stats *stats)
{
    return guarded(ctx, [&](Ctx &c) {
        EventTimer t;
        t.start(c.stream);
        stage_dist_set_panel(c, d_records, nrecords, nreads_total, nkmers_total, row_lo, row_hi);
        t.stop(c.stream);
        if (stats) { stats->nrows = c.M; stats->ncols = c.N_global >= 0 ? c.N_global : c.N; stats->nnz = c.Z; stats->max_row_nnz = c.max_row_nnz; stats->ms_total = t.ms(); }
    });
}

}  // extern "C"
