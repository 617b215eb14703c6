// Internal declarations shared by the HIP translation units of libelba_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/elba_amd.h"

namespace elba {

struct Error {
    int code;
    std::string msg;
};

#define ELBA_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            throw ::elba::Error{e__ == hipErrorOutOfMemory ? ELBA_ERR_OUT_OF_MEMORY : ELBA_ERR_HIP,      \
                                std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + __FILE__ +  \
                                    ":" + std::to_string(__LINE__) + ")"};                               \
        }                                                                                                \
    } while (0)

#define ELBA_REQUIRE(cond, code, text)                       \
    do {                                                     \
        if (!(cond)) throw ::elba::Error{(code), (text)};    \
    } while (0)

// Work that must run once per DEVICE and process (hipFuncSetAttribute applies to the current device: a host process that drives two GPUs, or
// ranks that run as threads, must each set it on their own device) — serialised, so that threads never race on the flags.
struct DeviceOnce {
    std::mutex m;
    bool done[64] = {};
    template <class F> void run(int dev, F &&f)
    {
        std::lock_guard<std::mutex> g(m);
        if (dev >= 0 && dev < 64 && done[dev]) return;
        f();
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
};

// Growable device buffer (never shrinks; capacity is reused across calls so the steady state allocates nothing).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        release();
        size_t want = bytes + (bytes >> 3) + 256;   // slack so that small growth does not reallocate
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { want = bytes; e = hipMalloc(&p, want); }
        if (e != hipSuccess) { p = nullptr; throw Error{ELBA_ERR_OUT_OF_MEMORY, "hipMalloc of " + std::to_string(bytes) + " bytes failed"}; }
        cap = want;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
    void swap(DevBuf &o) { void *tp = p; p = o.p; o.p = tp; size_t tc = cap; cap = o.cap; o.cap = tc; }
};

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    void init() { if (!a) { ELBA_HIP(hipEventCreate(&a)); ELBA_HIP(hipEventCreate(&b)); } }
    void start(hipStream_t s) { init(); ELBA_HIP(hipEventRecord(a, s)); }
    void stop(hipStream_t s) { ELBA_HIP(hipEventRecord(b, s)); }
    float ms() { float t = 0; ELBA_HIP(hipEventSynchronize(b)); ELBA_HIP(hipEventElapsedTime(&t, a, b)); return t; }
    ~EventTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

// Phase marks of one call on one stream: N events, one record per phase boundary (every record costs stream time — the overlap
// SpGEMM's whole call is a few hundred microseconds, so boundaries are shared instead of bracketing every phase with its own pair)
template <int N>
struct PhaseMarks {
    hipEvent_t e[N] = {};
    void mark(int k, hipStream_t s) { if (!e[k]) ELBA_HIP(hipEventCreate(&e[k])); ELBA_HIP(hipEventRecord(e[k], s)); }
    float ms(int from, int to) { float t = 0; ELBA_HIP(hipEventSynchronize(e[to])); ELBA_HIP(hipEventElapsedTime(&t, e[from], e[to])); return t; }
    ~PhaseMarks() { for (int k = 0; k < N; ++k) if (e[k]) (void)hipEventDestroy(e[k]); }
};

// ---- primitives (prims.hip) -------------------------------------------------------------------------------------
// Exclusive prefix sums; in and out may alias.  tmp is grown as needed.
void exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, int64_t n, DevBuf &tmp);
void exclusive_scan_u32_to_i64(hipStream_t s, const uint32_t *in, int64_t *out, int64_t n, DevBuf &tmp);
// Stable LSD radix sort of (key u64, value u64) pairs on key bits [bit_lo, bit_hi).  Ping-pongs between the two
// buffer pairs; returns 0 if the result is in (k0,v0), 1 if in (k1,v1).
// hint bits (see Ctx::csr_hints) of the entry of read i in a column of L <= 64 entries (read << 32 | pos)
__device__ __forceinline__ uint32_t column_hint(const uint64_t *col, uint32_t L, uint32_t i, uint32_t win_lo, uint32_t win_hi)
{
    bool own_g = false, own_w = false;
    uint32_t mult = 0;
    for (uint32_t t = 0; t < L; ++t) {
        const uint32_t j = (uint32_t)(col[t] >> 32);
        if (j == i) { ++mult; continue; }
        const bool par = ((i ^ j) & 1u) ? j < i : j > i;
        own_g |= par; own_w |= par || j < win_lo || j >= win_hi;
    }
    if (mult >= 2) return 0u;
    return (own_g ? 0u : 1u) | (own_w ? 0u : 2u);
}
constexpr uint32_t HINT_MAX_COL = 12;     // longer columns are not examined: a row owns no pair of an L-read column with probability 2^-(L-1), and the test costs L loads

int radix_sort_pairs(hipStream_t s, uint64_t *k0, uint64_t *v0, uint64_t *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp);
// The same for bare 64-bit words (whatever rides in the bits outside [bit_lo, bit_hi) moves with them): returns 0 if the result is in k0, 1 if in k1.
int radix_sort_keys(hipStream_t s, uint64_t *k0, uint64_t *k1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp, bool first_hist_done = false);
int radix_sort_where(int64_t n, int bit_lo, int bit_hi);
// The CSR build's sort of one-word keys (read << rs | id << (pb + 2) | hint << pb | pos, or an inline partner: bit 63, matrix.hip) on the read bits
// [rs, rs + mb), with the LAST pass writing the rows of A themselves: csr[z] = the unpacked entry (k_unpack_csr_words' format) and rowptr[0 .. M]
// (the first entry of every read, empty rows included) — no pass over the sorted keys behind the sort.  The keys in k0 are consumed.
struct CsrFin { int idbits, pb, rs, mb, pbi; uint64_t *csr; uint32_t *rowptr; int64_t M; };
void radix_sort_keys_to_csr(hipStream_t s, uint64_t *k0, uint64_t *k1, int64_t n, const CsrFin &f, DevBuf &tmp);
int radix_sort_pairs_k32(hipStream_t s, uint32_t *k0, uint64_t *v0, uint32_t *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp);
void group_offsets_k32(hipStream_t s, const uint32_t *sorted_keys, int64_t n, uint32_t *ptr, int64_t nkeys);
void radix_column_scan(hipStream_t s, uint32_t *rows, int64_t nrows, uint32_t nbins, DevBuf &tmp);      // (prims.hip)
uint32_t *radix_first_histogram(int64_t n, int bit_lo, int bit_hi, DevBuf &tmp, int *shift, int *bits, int *tile);      // (prims.hip)
void fill_u32(hipStream_t s, uint32_t *p, uint32_t v, int64_t n);
void fill_u64(hipStream_t s, uint64_t *p, uint64_t v, int64_t n);
// ptr[k] = first index z with keys[z] >= k, for k in [0, nkeys]; keys ascending (sorted group ids -> CSR/CSC pointers)
void group_offsets_u32(hipStream_t s, const uint64_t *sorted_keys, int key_shift, int64_t n, uint32_t *ptr, int64_t nkeys);
uint64_t reduce_max_u64(hipStream_t s, const uint64_t *p, int64_t n, DevBuf &tmp);   // synchronises

// ---- tuning and A/B switches (elba_set_option) --------------------------------------------------------------------------
// Every default is the production path and none of them changes a result: the alternatives are kept for measurements and for the parity
// tests that walk them.  They used to be environment variables read inside the stages; a library does not read its caller's environment.
struct Options {
    bool no_symmetry = false;   // SpGEMM: accumulate both triangles instead of one + mirror
    bool no_ell = false;        // plain CSC columns instead of the padded column store
    bool no_pay = false;        // 32-bit accumulators + seed look-ups instead of the position-carrying 64-bit ones
    bool mir32 = false;         // 32-byte staging / mirror records instead of 16-byte words
    bool no_hints = false;      // no ownership bits in the rows of A
    bool no_sample = false;     // a cold call does not compute a sample of rows first
    bool no_ell_compact = false;// k-mer stage: the padded column store holds every column, row entries name k-mer ids (no gather slots)
    bool no_slab = false;       // SpGEMM: mirrored entries wait in the staging area for k_mirror instead of going straight to their row's slab (spgemm.hip: "mirror slabs")
    bool csr_pairs_late = false; // dense matrices from the two-level partition: the CSR build's sort pairs written by the CSR build, not by the bucket kernels (A/B)
    bool msd_rank = false;      // k-mer stage: column ranks whatever UPPER is (tests)
    bool msd_no_rank = false;   // k-mer stage: entries without their column's rank — the emit kernels sort by ranges of the 16 value bits (A/B; what runs when the payload leaves no room)
    bool msd_no_emit8 = false;  // k-mer stage: buckets of up to 2048 entries through the 512-lane emit kernel too (A/B)
    int ell_slot_cap = 0;       // test hook: the padded column store pretends to hold this many gather slots only (0: its real size)
    int slab_pct = 175;         // SpGEMM: a row's slab holds this many percent of the mirrored entries the measured ratio predicts for it (+ SLAB_PAD)
    int slab_q16 = 0;           // test hook: slab entries per row entry of A in 1/65536 units, instead of the measured ratio (small matrices take no sample)
    bool panel_inline = false;  // inline partners in the rows of a windowed matrix too (a shard's panel): such a matrix is multiplied with the mirror exchange only
    bool no_inline = false;     // no inline partners in the rows of A (the owner's entry of a two-read column carries the other read: no column fetch)
    bool no_suffix = false;     // dense matrices stay on the general kernel
    bool no_row_order = false;  // dense matrices: partners are named by their row, not by a label that brings reads of one locus together
    bool kmer_pairs = false;    // (value, payload) pairs through the k-mer sort instead of one packed word
    bool kmer_unfused = false;  // per-head column emission (k_runs<true> + k_instance_entries) instead of k_runs_emit
    bool kmer_msd = false;      // force the two-level partition path on inputs below its size threshold (tests)
    bool kmer_no_msd = false;   // k <= 17: keep the LSD sort of the whole value instead of the two-level partition + LDS count (kmer_msd.hip)
    bool csr_pairs = false;     // (read, entry) pairs through the CSR sort instead of one word
    bool emit_plain = false;    // k-mer emit without the fused first histogram
    bool trace = false;         // progress lines on stderr
    bool measure_prep = false;  // diagnostic (bench.py, "spgemm_prep"): elba_count_kmers runs its emit kernels a second time WITHOUT what they write for the SpGEMM's sake alone (hint bits,
                                // inline partners, gather slots + padded columns), both runs bracketed by events: elba_get_stat("spgemm_prep_us") = the difference (kmer_msd.hip)
    int64_t tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // A/B switches of the round in progress ("tune0" .. "tune7"): what each means is said where it is read
    int msd_wide_bits = 0;      // tests: value bits the partition of the 19 <= k <= 31 path takes (0: chosen from the number of instances)
    int64_t kmer_batch_instances = 0;      // k-mer stage (k <= 17, reads): more instances than this are counted in passes over value ranges (0: 0xE0000000 — what a 32-bit place holds); tests force passes on small sets
    int msd_small_cap = 0;      // tests: buckets with more entries than this go to the crowded-bucket kernel (0 = its real capacity)
    int kmer_drop = 0;          // test hook: force that many dropped index bits on a small input (1..3)
    int dense_up = 1;           // SpGEMM, dense path: the tier its rows start on at least (1: eight wavefronts share a 1024-slot table — 32 per CU as with four on 512 slots, half the load)
    int dense_wgs = 8;          // SpGEMM, dense path: workgroups of the 512-slot tier per CU
    int dk = -1;                // SpGEMM: gather trips per iteration of the padded-column loop: 0 = one, 1 = two, 2 = four, 4 = eight; -1 = chosen per matrix (spgemm.hip)
    int64_t aln_tiers = 0;      // x-drop register tiers as decimal digits (1248 = all), 0 = default
    int aln_wide_hint = 6, aln_long_hint = 6000;
};

// ---- context ----------------------------------------------------------------------------------------------------
struct Ctx {
    elba_cfg cfg{};
    Options opt;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    int num_cus = 256;

    // reads (DnaBuffer layout)
    int64_t nreads = 0, first_global_id = 0, packed_bytes = 0;
    const uint8_t *d_packed = nullptr;     // either owned (own_*) or borrowed from the caller
    const uint64_t *d_byte_off = nullptr;
    const uint32_t *d_len = nullptr;
    DevBuf own_packed, own_byte_off, own_len;
    bool have_reads = false;
    uint32_t max_read_len = 0;             // longest read (set by stage_count_kmers)
    std::vector<uint32_t> h_len;           // host copy of lengths (instance offsets are a host-side prefix sum)
    std::vector<uint64_t> h_byte_off;

    // k-mer stage results (device)
    bool have_counts = false;
    int triples_path = 0; // diagnostic: how the last elba_set_kmer_matrix_device built the matrix — 0 radix sorts of the whole matrix (matrix.hip), 1 the k-mer stage's bucket kernels (kmer_msd.hip)
    int kmer_passes = 1;  // diagnostic: value-range passes of the last elba_count_kmers (kmer_msd.hip)
    int kmer_path = 0;    // diagnostic: how the last elba_count_kmers counted — 0 the sort of kmer.hip, 1 two-level partition + LDS count tables (k <= 17), 2 the same on 16-byte records + LDS sort (19 <= k <= 31)
    int64_t I = 0, ndistinct = 0;
    DevBuf inst_off;      // u64[M+1] instance offset of each read
    DevBuf rel_kmers;     // u64[N] reliable k-mers ascending (right-aligned value order == packed order)
    DevBuf rel_kmers_lo;  // u64[N] their second word when k > 32
    DevBuf rel_kmers_lo2; // u64[N] their third word when k > 64
    DevBuf rel_counts;    // u32[N]
    DevBuf csr_words;     // u64[Z] read << (pre_nb + pre_pb) | k-mer id << pre_pb | pos of every entry of a_csc, when pre_words (k_runs_emit -> the CSR build's sort)
    bool pre_pairs = false;      // dense matrix from kmer_msd.hip: the (read, kid | L | place | pos) pairs of the CSR sort are written (keys in ws_b; values where a sort of Z pairs on the read bits that ENDS in a_csr starts: a_csr or ws_d)
    bool pre_ready = false, pre_consumed = false, pre_words = false, pre_hints = false, pre_hints_done = false, pre_ell_done = false, pre_inline = false, pre_inline_pending = false /* the sort keys leave room for inline partners: k_add_hints writes them */; int pre_rs = 0, pre_pbi = 0; int pre_nb = 0, pre_pb = 0; uint64_t pre_maxpos = 0;
    // Ownership hints of the SpGEMM, two bits in every a_csr entry (kid << 32 | hint << 30 | pos; positions below 2^30): bit 30 = under the
    // parity rule of owns_pair (spgemm_direct.hpp) this row accumulates NO pair of the entry's column and appears in it once — the column
    // need not be fetched at all, the entry only counts one diagonal product; bit 31 = the same with every partner outside the row window
    // counted as owned (calls without the mirror exchange between ranks).  They are a property of A, written when A is built (k_runs_emit /
    // k_csc_to_csr_words: every entry sees its whole column there anyway).  csr_hints false: both bits are zero / positions use all 32 bits.
    bool csr_hints = false, ov_hints_used = false, ov_rec16 = false;
    // Inline partners (whole-matrix windows, positions below 2^16, read ids and positions narrow enough for one sort word): four in five columns of
    // 15 %-error reads hold TWO reads, and under the parity rule exactly one of the two rows accumulates the pair (of longer columns: an entry whose
    // row accumulates exactly ONE pair of the column).  That row's entry then carries the pair itself — a_csr entry = 1 << 63 | (partner >> 1) << 32 | posQ | posT << 16, the partner's low bit follows from the rule — and the SpGEMM
    // fetches no column for it (58 % of its gathers on BASELINE config 3).  Such an entry has no k-mer id: exports rebuild CSR from the columns.
    bool csr_inline = false;
    bool csr_inline_window = false;      // ... written for a matrix with a row window (option "panel_inline"): valid under the parity rule over all rows only, i.e. for calls with the mirror exchange
    bool csr_suffix = false;  // dense matrices: a_csr entries are kid << 32 | column length << 23 | own place in the column << 16 | pos, pairs owned by the smaller row (matrix.hip)
    DevBuf ov_sample;         // u32[256]: the rows a cold SpGEMM call computes first (spgemm.hip)
    int64_t A_products = 0;   // sum over the window's row entries of their column's length (what the SpGEMM reports as `products`)
    DevBuf prod_ctr;
    DevBuf kid_of_entry;  // u64[Z] k-mer id of every entry of a_csc (written with the columns; what the CSR build sorts by read)
    elba_kmer_stats kstats{};

    // A (device)
    bool have_A = false;
    bool A_has_kmers = false;
    int64_t M = 0, N = 0, Z = 0, max_row_nnz = 0, max_col_nnz = 0;
    DevBuf a_rowptr, a_csr, a_colptr, a_csc;   // u32[M+1], u64[Z], u32[N+1], u64[Z]
    DevBuf a_ell;                              // u64[N * s_stride]: the columns padded to a common stride (entries, then all ones) — the column store the
                                               // SpGEMM gathers from when no column is longer than 64 entries: column kid starts at kid * s_stride, no pointer
    bool use_ell = false;
    // gather slots (kmer_msd.hip, BucketOut): with inline partners the padded store holds only the columns some row entry still fetches; such an entry
    // names its column's slot in the id field of its a_csr word (the others keep the k-mer id), ell_slot_kid[slot] = k-mer id.  Not compact: slot == k-mer id.
    bool ell_compact = false;
    int64_t ell_nslots = 0, ell_cap_cols = 0;  // slots in use (compact: an upper bound, chunks are drawn whole) / columns the store was sized for
    DevBuf ell_slot_kid;                       // u32[ell_nslots]
    DevBuf a_ellj;                             // u32[N << j_shift]: the partner reads of every column, right-aligned in an aligned block of 32 or 64 slots (dense matrices, Ctx::csr_suffix)
    DevBuf row_order, row_label, row_keys; bool have_row_order = false;      // u32[M] each, dense matrices: the rows sorted by their smallest k-mer id (label -> row) and its inverse (row -> label); a_ellj then names partners by label
    uint32_t j_shift = 5;
    DevBuf col_w0;                             // u8[N]: rotation of every padded column of a dense matrix with a row window (matrix.hip: k_fill_ell)
    uint32_t s_stride = 4, lpc_log2 = 1;       // padded column stride in entries (4, or a multiple of 8: a whole number of 64-byte lines); lanes of the SpGEMM per row entry 2^lpc_log2 >= s_stride / 2
    bool cold_calls = false;                   // every elba_create_seed_matrix call forgets what earlier calls learned (prior, tier usage): elba_set_option
    bool pos16 = false;                        // every position among the entries is < 65536 (mirrored entries of B travel as 16-byte records then)
    uint32_t fbits = 1;                        // bits of the column-position field of a product sequence number
    int64_t row_lo = 0, row_hi = -1;           // rows of B computed by this context (-1: all)
    // rows of an A built from reads are local read indices; exported triples carry global ids (src/KmerOps.cpp:215-219)
    int64_t first_global_id_rows() const { return A_has_kmers ? first_global_id : 0; }

    // distributed owner state (kmer.hip, second half)
    const uint64_t *d_records = nullptr; int64_t nrecords = 0;
    bool dist_owner = false;
    DevBuf dist_gid;          // u32[N_local] global k-mer id of each local column
    int64_t dist_nall = -1;
    DevBuf dist_all_off; std::vector<int64_t> dist_bounds; int dist_pack_vb = 0, dist_pack_ib = 0;      // exchange #1 with 8-byte records (kmer.hip, stage_dist_packed_format): every rank's instance offsets (rank r's at bounds[r] + r), the read bounds, the format
    std::vector<uint32_t> owner_upper;   // value-range owners: rank r owns the value bins [owner_upper[r-1], owner_upper[r]) (kmer.hip)
    int64_t N_global = -1;    // a panel context: k-mers of the whole run (its own columns are renumbered locally)
    DevBuf own_colptr, own_csc;   // the columns this rank OWNS (u32[own_N + 1], u64[own_Z]): kept apart from the context's A, which the panels overwrite —
    int64_t own_N = 0, own_Z = 0; // a rank serves one panel per row block of every rank (elba_dist_panel_*_win)

    // B (device)
    bool have_B = false;
    int64_t Y = 0;
    DevBuf b_rowptr, b_col, b_val;             // i64[M+1], u32[Y], elba_seed_t[Y]
    elba_overlap_stats ostats{};

    // alignments (align.hip)
    bool have_aln = false;
    int64_t naln = 0;
    DevBuf aln_tasks, aln_ext, aln_cnt, aln_ptr, aln_ctr, aln_ofl, aln_scratch, aln_rows, aln_cols, aln_out;
    elba_align_stats astats{};
    DevBuf aln_all_packed, aln_all_off, aln_all_len;     // every read of the run, replicated for a row shard's alignments (elba_dist_set_all_reads)
    int64_t aln_all_n = -1; uint32_t aln_all_maxlen = 0;

    // string graph (tr.hip)
    bool have_edges = false;                   // an edge list loaded with elba_set_overlaps (otherwise this context's alignments are the input)
    int64_t tr_in_M = 0, tr_in_n = 0;
    DevBuf tr_in_rows, tr_in_cols, tr_in_vals;
    bool have_S = false;
    int64_t tr_M = 0, tr_nnz = 0, tr_id_base = 0;
    DevBuf tr_deg, tr_pas, tr_flags, tr_k0, tr_v0, tr_k1, tr_v1, tr_ptr, tr_sym, tr_src, tr_mark, tr_ctr, tr_sel, tr_out_rows, tr_out_cols, tr_out_vals;
    elba_string_stats sstats{};

    // workspaces
    DevBuf ws_scan, ws_sort, ws_a, ws_b, ws_c, ws_d, ws_e, ws_f;
    DevBuf ws_cursor;       // the gather-slot cursor of the k-mer stage's emit kernels (kmer_msd.hip)
    DevBuf ws_g, ws_h;      // crowded buckets of the wide k-mer partition (kmer_msd.hip: k31_gather_crowded ...): their records / the pseudo-buckets' arrays
    DevBuf ov_totcnt, ov_mir, ov_tmp, ov_sum_tmp;  // u32[M+1] mirrored entries per row (ticket counters); mirrored entries laid out like B (32-byte records); staging area (32-byte records)
    bool ov_low_clean = false;                 // the ticket counters are all zero (handed back clean by the previous call)
    DevBuf ov_rowub, ov_rowcnt, ov_rowoff, ov_lists, ov_counters, ov_gtable, ov_sortkeys;
    int64_t ov_tmp_cap = 0;
    bool ov_sort_used[2] = {false, false};      // wide-row sorts used by the previous call
    bool ov_tiers_known = false, ov_tier_used[8] = {false, false, false, false, false, false, false, false};   // tiers that got rows in the previous call
    int64_t b_cap_entries = 0;      // capacity of b_col/b_val the next overlap call may assume (0 = unknown: size it after the numeric pass)
    uint64_t ov_calls = 0;          // steady-state overlap calls so far (phase events are recorded on every cfg.timing_stride-th)
    uint32_t ov_prior_q16 = 0;      // distinct-partner / product ratio measured by the previous overlap call (x 65536), 0 = unknown
    uint32_t ov_slab_q16 = 0;       // mirrored entries per row entry of A measured by the previous call on the whole matrix (x 65536, without the margin), 0 = unknown
    DevBuf ov_slab;                 // uint4[ov_slab_cap]: the rows' mirror slabs (spgemm.hip)
    int64_t ov_slab_cap = 0;
    DevBuf ov_tickrows;             // u32[M / 32 + 1]: rows that staged an entry whose image took a ticket (spgemm.hip: OvParams::tick_rows)
    DevBuf ov_slabpos, ov_slabn;    // u64[M] slab end << 32 | next free entry; u32[M] entries in every row's slab (spgemm.hip)
    bool ov_slab_on = false;        // the running call has slabs (ov_launch_finalize reads them)
    uint32_t ov_slab_q16_used = 0;  // diagnostic: the ratio the last call's slabs were sized by (margin included), 0 = none
    int64_t ov_mir_placed = 0;      // diagnostic: mirrored entries of the last call that did NOT go to a slab (placed by k_mirror)
    // sharded call with mirror exchange (spgemm.hip: stage_seed_matrix_begin / _fill / _end)
    int ov_phase = 0;               // 1: begin has run (numeric done, staged records waiting), end not yet
    int ov_pend_passes = 1; bool ov_pend_timed = false; float ov_pend_ms[3] = {0, 0, 0};
    std::vector<uint64_t> ov_remote_bounds;
    DevBuf ov_remote;               // mirror images received from other ranks (32-byte records)
    int64_t ov_send_slot = 0;       // slot size of the step in flight (stage_seed_matrix_send): recv must be given the same
    DevBuf ov_cursors;              // per-destination cursors of the fixed-slot exchange + the receive side's check words (stage_seed_matrix_send / _recv)
    bool own_stream = true;         // c.stream was created by the context (elba_set_stream: the caller's)

    EventTimer t_total, t_a, t_b, t_c;
    EventTimer t_emit, t_emit_plain; int64_t prep_us = -1, emit_us = -1;      // Options::measure_prep
    struct PinnedHost { void *p = nullptr; size_t cap = 0; void reserve(size_t n) { if (n <= cap) return; if (p) (void)hipHostFree(p); p = nullptr; cap = 0; ELBA_HIP(hipHostMalloc(&p, n, hipHostMallocDefault)); cap = n; } ~PinnedHost() { if (p) (void)hipHostFree(p); } };
    PinnedHost ov_host;            // pinned landing area of the per-call counter read-back (a pageable target makes the copy a staged, blocking one)
    PhaseMarks<5> ov_marks;        // overlap SpGEMM: 0 call start, 1 numeric start, 2 numeric end, 3 call end, 4 finalize start when the host synchronised before it
};

// ---- stages -----------------------------------------------------------------------------------------------------
void stage_set_reads_fasta(Ctx &c, const char *chunk, int64_t chunk_bytes, uint64_t chunk_file_offset, const elba_fasta_record_t *recs, int64_t nreads,
                           int64_t first_global_id, elba_ingest_stats *stats);      // ingest.hip
void stage_count_kmers(Ctx &c);                                   // kmer.hip
void stage_create_kmer_matrix(Ctx &c);                            // kmer.hip
bool msd_matrix_from_triples(Ctx &c, int64_t M, int64_t N, int64_t Z, const int64_t *d_rows, const int64_t *d_cols, const uint32_t *d_vals);   // kmer_msd.hip: false = matrix.hip sorts
bool msd_count_kmers(Ctx &c, uint64_t I, elba_kmer_stats &st);   // kmer_msd.hip: false = not applicable (the caller sorts)
void choose_column_store(Ctx &c, int64_t N, int64_t max_col);     // matrix.hip: padded column store or plain CSC, strides, sequence-number bits
void stage_set_kmer_matrix(Ctx &c, int64_t M, int64_t N, int64_t Z, const int64_t *rows, const int64_t *cols, const uint32_t *vals);  // matrix.hip
void stage_set_kmer_matrix_device(Ctx &c, int64_t M, int64_t N, int64_t Z, const int64_t *d_rows, const int64_t *d_cols, const uint32_t *d_vals);
void stage_export_triples_device(Ctx &c, int64_t *d_rows, int64_t *d_cols, uint32_t *d_vals);
void stage_create_seed_matrix(Ctx &c);                            // spgemm.hip
void stage_seed_matrix_begin(Ctx &c, int nranks, const uint64_t *bounds_host, uint64_t *send_counts_host);
void stage_seed_matrix_fill(Ctx &c, void *d_send, const uint64_t *offsets_host);
void stage_seed_matrix_end(Ctx &c, const void *d_recv, int64_t nrecv);
void stage_seed_matrix_send(Ctx &c, int nranks, const uint64_t *bounds_host, void *d_send, int64_t slot);
bool stage_seed_matrix_recv(Ctx &c, void *d_recv, int64_t slot, int64_t *slot_needed);
void stage_align_seeds(Ctx &c, int mat, int mis, int gap, int dropoff);   // align.hip
void stage_dist_set_all_reads(Ctx &c, const void *d_packed, int64_t packed_bytes, const void *d_byte_off, const void *d_len, int64_t nreads_total);   // align.hip
void stage_set_overlaps(Ctx &c, int64_t nreads, const int64_t *rows, const int64_t *cols, const elba_overlap_t *vals, int64_t n);   // tr.hip
void stage_transitive_reduction(Ctx &c, double bad_read_cutoff, int fuzz);   // tr.hip
void stage_dist_count_owners(Ctx &c, int nranks, uint64_t *counts_host);                                   // kmer.hip
void stage_dist_value_histogram(Ctx &c, uint64_t *hist_host, int64_t nbins);
void stage_ref_hash_owner(Ctx &c, const uint64_t *kmers_host, int64_t n, int nprocs, uint64_t *hash_host, int32_t *owner_host);
void stage_dist_set_owner_ranges(Ctx &c, int nranks, const uint32_t *upper_bins);
void stage_dist_set_kmer_id_base(Ctx &c, int64_t base, int64_t nall);
void stage_dist_fill_send(Ctx &c, int nranks, void *d_send, const uint64_t *offsets_host);
bool stage_dist_packed_format(Ctx &c, int nranks, const int64_t *bounds, const uint32_t *all_lens, int *value_bits, int *index_bits);
void stage_dist_fill_send_packed(Ctx &c, int nranks, void *d_send, const uint64_t *offsets_host);
void stage_dist_unpack_records(Ctx &c, int nranks, int rank, const void *d_packed, const uint64_t *recv_counts_host, void *d_out);
void stage_dist_count_records(Ctx &c, const void *d_rec, int64_t nrec);
void stage_dist_copy_reliable_kmers(Ctx &c, void *d_dst);       // N k-mers of 1 + (k > 32) + (k > 64) words each, interleaved
void stage_dist_set_global_kmers(Ctx &c, const void *d_all, int64_t nall);
void stage_dist_panel(Ctx &c, int nranks, const uint64_t *bounds_host, const uint64_t *win_lo_host, const uint64_t *win_hi_host, bool fill, void *d_send, uint64_t *counts_or_offsets_host);
void stage_dist_set_panel(Ctx &c, const void *d_rec, int64_t nrec, int64_t M_total, int64_t N_total, int64_t row_lo, int64_t row_hi);

}  // namespace elba
