/*
 * elba_amd.h — C ABI of the MI355X-native overlap-detection engine (libelba_amd.so).
 *
 * This is the drop-in boundary for ELBA's hot path.  Each entry point names the
 * reference interface it replaces; all file:line citations are into PASSIONLab/ELBA @ v2.
 *
 *   reference (C++/MPI/CombBLAS, one call chain in src/main.cpp:191-285)      this ABI
 *   -----------------------------------------------------------------------   --------------------------------
 *   DnaBuffer (include/DnaBuffer.hpp:13-38), index.getmydna() main.cpp:126      elba_set_reads / elba_set_reads_device
 *   get_kmer_count_map_keys   (include/KmerOps.hpp:27-28, main.cpp:192)   \
 *   get_kmer_count_map_values (include/KmerOps.hpp:30,    main.cpp:225)   /    elba_count_kmers
 *   create_kmer_matrix        (include/KmerOps.hpp:24-25, main.cpp:259)   \
 *   AT = *A; AT->Transpose()  (src/main.cpp:272-273)                      /    elba_create_kmer_matrix
 *   (A handed over by a caller that already owns it: the SpParMat triples
 *    of src/KmerOps.cpp:380-400)                                               elba_set_kmer_matrix
 *   create_seed_matrix        (include/SharedSeeds.hpp:98-99, main.cpp:281)    elba_create_seed_matrix
 *   Bmat.seqptr()->GetDCSC()  (src/PairwiseAlignment.cpp:16-19)                elba_export_dcsc
 *   SharedSeeds               (include/SharedSeeds.hpp:8-96)                   elba_seed_t
 *
 * Conventions: plain pointers and sizes only; every function returns an int status (ELBA_OK == 0);
 * host arrays handed in are borrowed for the duration of the call; arrays handed out are owned by
 * the library and released with the matching elba_free_* call.  One context per process and GPU,
 * calls on a context are serialised by the caller (the reference is single-threaded per rank on this
 * path).  The library fails loudly (ELBA_ERR_NO_DEVICE) when no HIP device is present:
 * there is no CPU fallback.
 *
 * Canonical result order: k-mer ids are the ranks of the packed canonical k-mer
 * values (ascending); entries of a column of A are ordered by (read, pos), of a row by (k-mer id, pos);
 * B(i,j).seeds[0] / seeds[1] are the products with minimal / maximal (k-mer id, posQ, posT) — what an
 * ascending-k left fold of SharedSeeds::Semiring::add (include/SharedSeeds.hpp:41-46) yields.
 */
#ifndef ELBA_AMD_H_
#define ELBA_AMD_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELBA_ABI_VERSION 3

enum {
    ELBA_OK = 0,
    ELBA_ERR_INVALID_ARG   = 1,  /* bad pointer / size / (k, lower, upper) outside include/compiletime.h:10,21 */
    ELBA_ERR_NO_DEVICE     = 2,  /* no HIP device: the product path has no CPU fallback */
    ELBA_ERR_HIP           = 3,  /* a HIP runtime call failed (elba_last_error has the text) */
    ELBA_ERR_OUT_OF_MEMORY = 4,
    ELBA_ERR_STATE         = 5,  /* stage called before its inputs exist */
    ELBA_ERR_UNSUPPORTED   = 6,  /* k outside 3..95, index ranges beyond 32 bit, LOWER == 1 */
    ELBA_ERR_INTERNAL      = 7,
    ELBA_ERR_RETRY         = 8   /* elba_seed_matrix_recv: some rank ran out of room; every rank got this answer: repeat the step */
};

typedef struct elba_ctx elba_ctx;

/* SharedSeeds with an explicit field order (the reference's std::tuple pair is laid out reversed in
 * memory, SURVEY.md a11 — never memcpy into it, use std::get<>). */
typedef struct { uint32_t q0, t0, q1, t1; int32_t numshared; } elba_seed_t;

/* The fields of the reference's Overlap that Overlap::extend_overlap fills (include/Overlap.hpp:22-28, src/Overlap.cpp:24-73),
 * explicit order (beg/end are std::tuples in the reference: same caveat as above), plus the OverlapClass it was derived from. */
typedef struct {
    int32_t begQ, begT, endQ, endT;     /* beg = (begQ, begT), end = (endQ, endT) */
    int32_t score, suffix, suffixT;
    int8_t  direction, directionT;      /* -1 = none */
    uint8_t rc, passed, containedQ, containedT;
    uint8_t kind;                       /* OverlapClass: 0 BAD_ALIGNMENT, 1 FIRST_CONTAINED, 2 SECOND_CONTAINED, 3 FIRST_TO_SECOND_OVERLAP, 4 SECOND_TO_FIRST_OVERLAP */
    uint8_t reserved;
} elba_overlap_t;

/* One line of the samtools-style .fai next to the FASTA: FastaIndex::Record (include/FastaIndex.hpp:10; name and line width with
 * newline are not kept by the reference either: src/FastaIndex.cpp:15-23 reads `name len pos bases`). */
typedef struct { uint64_t len, pos, bases; } elba_fasta_record_t;

typedef struct {
    int64_t nreads, bases, packed_bytes, chunk_bytes;
    float   ms_total;       /* H2D copy of the chunk + encode */
    float   ms_encode;      /* the encode kernel alone */
} elba_ingest_stats;

typedef struct {
    int64_t nalignments;    /* candidate pairs aligned: stored B(i,j) with i < j (src/PairwiseAlignment.cpp:52 on one rank) */
    int64_t seeds_rejected; /* xdrop_aligner returned -1 (src/XDropAligner.cpp:231-245) */
    int64_t passed;         /* Overlap::passed */
    int64_t contained;      /* containedQ or containedT */
    int64_t extensions_strided; /* extensions whose band outgrew one wavefront (redone by the strided kernel) */
    int64_t cells;          /* DP cells computed (both extensions of every pair) — the unit of the stage's throughput */
    float   ms_total;       /* device time of the stage */
    float   ms_extend;      /* the extension kernels (dominant) */
} elba_align_stats;

/* rows[a], cols[a] (global read ids) and vals[a] of the a-th aligned pair, in CSR order of B's strict upper triangle:
 * the triples PairwiseAlignment hands to SpParMat<Overlap> (src/PairwiseAlignment.cpp:97-103). */
typedef struct {
    int64_t n;
    int64_t *rows, *cols;
    elba_overlap_t *vals;
} elba_overlaps_t;

/* From the aligned pairs to the string graph (src/main.cpp:305-312). */
typedef struct {
    int64_t nreads, nedges;     /* reads, aligned pairs handed in (entries of the upper-triangular R of PairwiseAlignment) */
    int64_t bad_reads;          /* find_bad_reads (src/main.cpp:553-571) */
    int64_t edges_passed;       /* entries left by Prune(!passed) + PruneFull(bad reads) */
    int64_t contained_reads;    /* find_contained_reads (src/main.cpp:573-583) */
    int64_t edges_kept;         /* entries left by PruneFull(contained reads): the R handed to TransitiveReduction (upper triangle) */
    int64_t products;           /* semiring products of R (x) R on the symmetrised R = sum over reads of degree^2: lookups done */
    int64_t marked;             /* nnz(I) before it is symmetrised: entries of R with suffix + FUZZ >= the two-edge path of their direction */
    int64_t removed;            /* nnz(T) */
    int64_t nnz;                /* nnz(S): both triangles */
    int32_t iterations;         /* passes of the reference's do-while this stands for (2 when something was removed, else 1; see tr.hip) */
    int32_t reserved;
    float   ms_total;           /* device time of the stage */
    float   ms_minplus;         /* the masked min-plus kernel alone */
} elba_string_stats;

typedef struct {
    int32_t k;          /* KMER_SIZE: odd, 3..95 as in the reference (one to three 64-bit words per k-mer; reference: compile-time, Makefile:1) */
    int32_t lower;      /* LOWER_KMER_FREQ >= 2 (SURVEY.md App. A.4 precondition)            */
    int32_t upper;      /* UPPER_KMER_FREQ <= 65535                                          */
    int32_t device;     /* HIP device ordinal                                                */
    int64_t workspace_hint_bytes; /* 0 = decide from the input; otherwise pre-size the overlap workspace */
    int32_t flags;      /* reserved, 0 */
    int32_t timing_stride; /* elba_create_seed_matrix records its phase events (ms_* of elba_overlap_stats) on every timing_stride-th
                              steady-state call only; 0 or 1 = every call.  An event record costs ~5 us of stream time on a ~0.25 ms call:
                              a caller that does not read the phase times every call should not pay for them every call. */
} elba_cfg;

typedef struct {
    int64_t nreads;        /* M                                                       */
    int64_t instances;     /* I  = sum_reads max(0, len-k+1)                          */
    int64_t distinct;      /* distinct canonical k-mers seen                          */
    int64_t reliable;      /* N  = k-mers with lower <= count <= upper                */
    int64_t entries;       /* Z  = nnz(A)                                             */
    float   ms_total;      /* device time of the stage (HIP events on the library's stream) */
    float   ms_count;      /* "collecting distinct k-mers": enumerate + value partition (k <= 31), or enumerate + sort */
    float   ms_lookup;     /* unused (0): there is no second enumeration — an instance carries its read and position */
    float   ms_sort;       /* "counting recording k-mer seeds": per-bucket exact count, [lower, upper] filter, columns */
} elba_kmer_stats;

typedef struct {
    int64_t nrows, ncols, nnz;   /* M, N, Z */
    int64_t max_row_nnz;
    float   ms_total;
} elba_matrix_stats;

typedef struct {
    int64_t nrows;          /* M */
    int64_t products;       /* P = sum_k c_k^2 (semiring multiplies)                      */
    int64_t nnz_before_prune; /* Y_raw = nnz of the raw product                            */
    int64_t nnz;            /* Y = nnz(B) after Prune(numshared <= 1) — the metric's unit */
    int64_t nnz_diag;       /* diagonal entries kept                                      */
    int64_t nnz_upper;      /* strict upper triangle = #alignment candidates on one rank  */
    int64_t max_numshared;
    int64_t rows_lds;       /* rows accumulated in LDS tables                             */
    int64_t rows_global;    /* rows spilled to the HBM table path                         */
    int64_t rows_escalated; /* optimistic-table overflows re-queued on a larger tier      */
    int64_t algorithmic_bytes; /* 16Z + 8(2M+N+3) + 24Y                    */
    int32_t passes;         /* 1, or 2 when the output workspace had to grow              */
    int32_t timed;          /* 1: ms_* below were measured in this call; 0: this call recorded no events (cfg.timing_stride), ms_* are 0 */
    float   ms_total;       /* whole timed region: resident A -> resident pruned CSR B    */
    float   ms_symbolic;    /* row upper bounds + binning                                 */
    float   ms_numeric;     /* hash-accumulate kernels (the dominant kernels)             */
    float   ms_finalize;    /* row-pointer scan + per-row column sort + copy              */
} elba_overlap_stats;

/* CombBLAS-shaped DCSC of a block of B with LOCAL indices: exactly the arrays walked by
 * src/PairwiseAlignment.cpp:28-32 (jc[nzc], cp[nzc+1], ir[nnz], numx[nnz]). */
typedef struct {
    int64_t nrows, ncols, nnz, nzc;
    int64_t *jc, *cp, *ir;
    elba_seed_t *numx;
} elba_dcsc_t;

/* B (or a row range of it) as CSR, columns ascending within a row. */
typedef struct {
    int64_t nrows, ncols, nnz;
    int64_t *rowptr;       /* [nrows+1] */
    int64_t *col;          /* [nnz]     */
    elba_seed_t *val;      /* [nnz]     */
} elba_csr_t;

/* A in both orientations, for parity checks and for callers that want CombBLAS triples back. */
typedef struct {
    int64_t nrows, ncols, nnz;
    uint64_t *kmers;       /* [ncols] packed canonical k-mer of each column (NULL if A came from elba_set_kmer_matrix); first word */
    uint64_t *kmers_lo;    /* [ncols] second word (bases 32..63, left-aligned) when k > 32, else NULL */
    uint64_t *kmers_lo2;   /* [ncols] third word (bases 64..k-1) when k > 64, else NULL */
    int64_t *colptr;       /* [ncols+1] */
    int64_t *csc_row;      /* [nnz] read id, within a column ordered by (read, pos) */
    uint32_t *csc_val;     /* [nnz] position */
    int64_t *rowptr;       /* [nrows+1] */
    int64_t *csr_col;      /* [nnz] k-mer id, within a row ordered by (kid, pos) */
    uint32_t *csr_val;     /* [nnz] position */
} elba_kmer_matrix_t;

/* Device-resident views (HIP pointers, valid until the next stage call / destroy): for consumers that stay on the GPU
 * (next rows f1/f2 of SURVEY.md §8f) and for the benchmark harness. */
typedef struct {
    int64_t M, N, Z, Y;
    const void *a_rowptr;  /* u32[M+1] */
    const void *a_csr;     /* u64[Z]: kid<<32 | hint<<30 | pos: positions below 2^30 (else hint = 0 and pos takes 32 bits); hint = two ownership bits
                              the SpGEMM reads (an entry whose row accumulates no pair of its column skips the column).  Dense matrices (a column longer than 16
                              entries, positions below 2^16): kid<<32 | column length<<23 | the entry's place in its column<<16 | pos.
                              elba_export_kmer_matrix returns plain positions */
    const void *a_colptr;  /* u32[N+1] */
    const void *a_csc;     /* u64[Z]: read<<32 | pos */
    const void *b_rowptr;  /* i64[M+1] */
    const void *b_col;     /* u32[Y] */
    const void *b_val;     /* elba_seed_t[Y] */
    void *stream;          /* hipStream_t the library launches on */
    uint32_t a_csr_format; /* which of the four encodings of a_csr is active (ELBA_CSR_*) */
    uint32_t a_csr_pos_mask; /* position of an a_csr entry = low word & a_csr_pos_mask, whatever the format */
    const void *a_kmers;   /* u64[N]: packed canonical k-mer (first word) of every column, ascending; NULL when A came from triples */
    int64_t a_gather_slots;/* ELBA_CSR_INLINE matrices built by elba_count_kmers: > 0 = an a_csr entry WITHOUT bit 63 and with hint == 0 (an entry that
                              fetches its column) names its column's gather slot in the id field, not the k-mer id; a_slot_kid maps back.  0: ids are k-mer ids */
    const void *a_slot_kid;/* u32[a_gather_slots]: k-mer id of every gather slot (slots are drawn in chunks: unused ones hold garbage) */
} elba_device_view;
enum { ELBA_CSR_PLAIN = 0 /* kid<<32 | pos */, ELBA_CSR_HINTS = 1 /* kid<<32 | hint<<30 | pos */, ELBA_CSR_DENSE = 2 /* kid<<32 | L<<23 | idx<<16 | pos */,
       ELBA_CSR_INLINE = 3 /* as HINTS, but an entry with bit 63 set carries the one partner of its two-read column instead of the k-mer id:
                              1<<63 | (partner>>1)<<32 | pos | partner's pos<<16 (positions below 2^16; the partner's low bit follows from the
                              ownership rule: DESIGN.md §4.1).  elba_export_kmer_matrix and the column side (a_colptr / a_csc / a_kmers) stay canonical */ };

int  elba_abi_version(void);
const char *elba_strerror(int status);
const char *elba_last_error(const elba_ctx *ctx);           /* detail text of the last failure on this context */

int  elba_ctx_create(elba_ctx **out, const elba_cfg *cfg);
void elba_ctx_destroy(elba_ctx *ctx);

/* DnaBuffer layout (src/DnaSeq.cpp:7-29, src/DnaBuffer.cpp:22-29): read r occupies bytes
 * [byte_off[r], byte_off[r] + (len[r]+3)/4) of `packed`, 4 bases per byte, first base in bits 7-6.
 * first_global_id: global id of local read 0 (the reference's MPI_Exscan offset, src/KmerOps.cpp:215). */
int  elba_set_reads(elba_ctx *ctx, const uint8_t *packed, const uint64_t *byte_off, const uint32_t *len,
                    int64_t nreads, int64_t first_global_id);
/* Same, with all three arrays already resident in HBM on cfg.device (no copy; caller keeps them alive). */
int  elba_set_reads_device(elba_ctx *ctx, const void *d_packed, int64_t packed_bytes, const void *d_byte_off,
                           const void *d_len, int64_t nreads, int64_t first_global_id);

/* FastaIndex::getmydna (src/FastaIndex.cpp:191-290) without the host-side parse: `chunk` holds the raw bytes of the FASTA file from
 * file offset `chunk_file_offset` on (the reference's MPI_File_read_at_all buffer, :236-241), `recs` the .fai records of this rank's
 * reads in order; the reads are 2-bit encoded on the device straight into the DnaBuffer layout (DnaSeq::compress, src/DnaSeq.cpp:7-29)
 * and become the context's reads, as after elba_set_reads. */
int  elba_set_reads_fasta(elba_ctx *ctx, const char *chunk, int64_t chunk_bytes, uint64_t chunk_file_offset,
                          const elba_fasta_record_t *recs, int64_t nreads, int64_t first_global_id, elba_ingest_stats *stats);
/* The resident reads back on the host in DnaBuffer layout: packed[packed_bytes], byte_off[nreads], len[nreads] (caller-allocated;
 * any pointer may be NULL to skip it).  Sizes: elba_ingest_stats, or nreads and sum((len+3)/4). */
int  elba_export_reads(elba_ctx *ctx, uint8_t *packed, int64_t packed_capacity, uint64_t *byte_off, uint32_t *len, int64_t nreads_capacity);
int  elba_count_kmers(elba_ctx *ctx, elba_kmer_stats *stats);
int  elba_create_kmer_matrix(elba_ctx *ctx, elba_matrix_stats *stats);

/* A as COO triples in any order (duplicates kept: SumDuplicates=false, src/KmerOps.cpp:400). */
int  elba_set_kmer_matrix(elba_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz,
                          const int64_t *rows, const int64_t *cols, const uint32_t *vals, elba_matrix_stats *stats);

/* The same with the three arrays resident in HBM (int64 rows, int64 cols, uint32 vals): what a GPU-resident caller of create_seed_matrix(A, AT)
 * (include/SharedSeeds.hpp:98-99) hands over.  elba_export_triples_device writes the resident A back as such triples into
 * caller-allocated device arrays of nnz elements each; their ORDER is unspecified (row-major for most matrices, column-major when the rows
 * carry inline partners, a_csr_format == ELBA_CSR_INLINE: such rows are rebuilt from the columns) — elba_set_kmer_matrix_device takes any order. */
int  elba_set_kmer_matrix_device(elba_ctx *ctx, int64_t nrows, int64_t ncols, int64_t nnz,
                                 const void *d_rows, const void *d_cols, const void *d_vals, elba_matrix_stats *stats);
int  elba_export_triples_device(elba_ctx *ctx, void *d_rows, void *d_cols, void *d_vals);

int  elba_create_seed_matrix(elba_ctx *ctx, elba_overlap_stats *stats);

/* PairwiseAlignment (src/PairwiseAlignment.cpp:5-106) on one rank, on the GPU: x-drop seed-and-extend from seeds[0] of every stored
 * B(i,j), i < j (xdrop_aligner src/XDropAligner.cpp:224-282 with the reference's defaults mat 1, mis -1, gap -1, dropoff 15,
 * src/main.cpp:53-56), classification and Overlap fields.  Needs the reads and B resident on this context (all reads local). */
int  elba_align_seeds(elba_ctx *ctx, int mat, int mis, int gap, int dropoff, elba_align_stats *stats);
/* Multi-GPU: on a row shard of B (after elba_dist_set_panel + elba_create_seed_matrix) elba_align_seeds aligns this rank's share of the
 * candidate pairs — pair {i < j} belongs to the rank of row i when i + j is even, to the rank of row j when it is odd; always as
 * (query i, target j) — once every read of the run is resident: device arrays in DnaBuffer layout with GLOBAL read ids, replicated by
 * the driver with one all-gather (the reads are 2 bits per base). */
int  elba_dist_set_all_reads(elba_ctx *ctx, const void *d_packed, int64_t packed_bytes, const void *d_byte_off, const void *d_len, int64_t nreads_total);
int  elba_export_overlaps(elba_ctx *ctx, elba_overlaps_t *out);
void elba_free_overlaps(elba_overlaps_t *o);

/* The string graph: what src/main.cpp:305-312 does to the R of PairwiseAlignment —
 *   find_bad_reads (src/main.cpp:553-571; reads whose (passed + 1) / (aligned + 1) <= bad_read_cutoff), Prune(!passed), PruneFull(bad),
 *   find_contained_reads (:573-583), PruneFull(contained), TransitiveReduction (src/TransitiveReduction.cpp:3-90; MinPlusSR,
 *   include/TransitiveReduction.hpp:78-110; fuzz = FUZZ, :16, 1000 in the reference).
 * Works on the alignments elba_align_seeds left on this context (one context holding the whole of B), or on an edge list loaded with
 * elba_set_overlaps: host arrays, rows[a] < cols[a] < nreads, strictly ascending in (row, col) — the order elba_export_overlaps gives;
 * a multi-GPU driver concatenates its ranks' shares (merged into that order) and hands them to every rank or to one.
 * elba_export_string_graph returns the entries of S, both triangles, in the order parallel_write_paf walks the reference's S
 * (src/main.cpp:527-541: columns ascending, rows ascending within a column); an entry below the diagonal carries Overlap::Transpose
 * (include/Overlap.hpp:43-69) of its mirror image.  elba_export_read_flags: flags[v] bit 0 = bad read, bit 1 = contained read. */
int  elba_set_overlaps(elba_ctx *ctx, int64_t nreads, const int64_t *rows, const int64_t *cols, const elba_overlap_t *vals, int64_t n);
int  elba_transitive_reduction(elba_ctx *ctx, double bad_read_cutoff, int fuzz, elba_string_stats *stats);
int  elba_export_string_graph(elba_ctx *ctx, elba_overlaps_t *out);
int  elba_export_read_flags(elba_ctx *ctx, uint8_t *flags, int64_t nreads);

int  elba_export_dcsc(elba_ctx *ctx, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi, elba_dcsc_t *out);
void elba_free_dcsc(elba_dcsc_t *d);
int  elba_export_csr(elba_ctx *ctx, int64_t row_lo, int64_t row_hi, elba_csr_t *out);
void elba_free_csr(elba_csr_t *c);
int  elba_export_kmer_matrix(elba_ctx *ctx, elba_kmer_matrix_t *out);
void elba_free_kmer_matrix(elba_kmer_matrix_t *m);
/* histogram of column counts: hist[c] = #reliable k-mers occurring c times, c in [0, upper] (src/main.cpp:449-485) */
int  elba_kmer_histogram(elba_ctx *ctx, int64_t *hist, int64_t len);

int  elba_get_device_view(elba_ctx *ctx, elba_device_view *view);

/* Run-time options by name (unknown name: ELBA_ERR_INVALID_ARG).  Besides the one below: tuning and A/B switches that never change a
 * result (struct Options in elba_amd/csrc/common.hpp lists them: "no_symmetry", "no_ell", "no_pay", "mir32", "no_hints", "no_sample",
 * "no_suffix", "no_row_order", "dense_up", "dense_wgs", "panel_inline", "kmer_pairs", "kmer_unfused", "kmer_no_msd", "csr_pairs", "emit_plain",
 * "trace", "kmer_drop", "dk", "aln_tiers", ...).
 * Options that shape A must be set before elba_count_kmers / elba_set_kmer_matrix.
 *   "overlap_cold_calls" (0 | 1): 1 = every elba_create_seed_matrix call forgets what earlier calls on the same matrix learned (the
 *       distinct-partner ratio that picks the starting table tiers, which tiers and column sorts received rows): what a caller that
 *       multiplies every matrix once pays — the reference's create_seed_matrix is called once per A (src/main.cpp:281).  Buffers stay
 *       allocated.  Default 0.
 *   "kmer_batch_instances" (>= 0): LIMITS.  The reference batches its k-mer exchange so that the input's size is no limit
 *       (include/KmerOps.hpp:10-12,33-56).  Here a context counts any number of k-mer instances for 9 <= k <= 17 — more than this option allows at
 *       once (0, the default: 0xE0000000, what a 32-bit place holds) are counted in PASSES over ranges of the k-mer value (consecutive first
 *       digits of the value partition: the passes yield consecutive k-mer ids and consecutive stretches of the columns; every pass enumerates the
 *       reads again and the whole input is counted twice — once for the sizes A's layout depends on, once to write it) — bounded by device memory
 *       (16 bytes per instance of the largest pass) and by nnz(A) < 2^32 per context (32-bit row / column pointers).  Tests set a small value to
 *       force passes; the result does not depend on it.  Every other k-mer path (k > 17, the sort) holds 32-bit places: < 2^32 instances.
 *   "measure_prep" (0 | 1): diagnostic — elba_count_kmers runs its emit kernels a second time without what they write for the SpGEMM's sake alone and
 *       brackets both runs with events (elba_get_stat "spgemm_prep_us").
 *   "tune0" .. "tune7": A/B switches of the round in progress (what each means is said where the library reads it); never a result-changing switch. */
int  elba_set_option(elba_ctx *ctx, const char *name, int64_t value);
/* Diagnostic counters of the last stage call by name (unknown name: ELBA_ERR_INVALID_ARG); none of them is part of a result.
 *   "overlap_mirror_placed"  mirrored entries of the last elba_create_seed_matrix call that waited in the staging area for the placement pass
 *                            instead of going straight to their row's slab ("mirror slabs")
 *   "overlap_slab_q16"       slab entries reserved per row entry of A in that call, x 65536 (0: the call ran without slabs)
 *   "resident_bytes_A"       device bytes the resident k-mer matrix occupies (CSR, columns, padded column store in use, pointers)
 *   "kmer_path"              how the last elba_count_kmers counted: 0 = sort, 1 = two-level value partition + LDS count tables (k <= 17),
 *                            2 = two-level partition of 16-byte records + LDS sort per bucket (19 <= k <= 31)
 *   "triples_path"           how the last elba_set_kmer_matrix_device built the matrix: 1 = two-level partition by column + the k-mer stage's
 *                            bucket kernels (matrices of some size whose columns all hold entries), 0 = radix sorts of the whole matrix
 *   "padded_columns"         1: the resident matrix has its padded column store (what the fast SpGEMM paths gather from); 0: columns longer than 64
 *                            entries, or the store did not fit a third of the free device memory when the matrix was built
 *                            (elba_release_workspace on other contexts of the device, then build again)
 *   "gather_slots"           columns of the padded column store in use (with inline partners: only the columns some row entry still fetches)
 *   "kmer_passes"            value-range passes the last elba_count_kmers took (1: the whole input at once; option "kmer_batch_instances")
 *   "spgemm_prep_us"         option "measure_prep": device microseconds the emit kernels of the last elba_count_kmers spent on hint bits, inline partners,
 *                            gather slots and padded columns (the kernels as built minus the same kernels without them; -1: not measured)
 *   "emit_us"                ... and the emit kernels as built */
int  elba_get_stat(elba_ctx *ctx, const char *name, int64_t *value);

/* Gives the stage calls' scratch memory back to the device (the sort / partition buffers of elba_count_kmers, elba_create_kmer_matrix and
 * elba_set_kmer_matrix_device: ~32 bytes per k-mer instance, 64 GB on BASELINE config 3), keeping the resident matrices, the reads and the
 * buffers of elba_create_seed_matrix.  A caller that builds A once and multiplies many times (or runs a second context beside this one)
 * calls it after elba_create_kmer_matrix; the next stage call allocates again.  The sort keys elba_count_kmers left for
 * elba_create_kmer_matrix go with the scratch: a later elba_create_kmer_matrix rebuilds them from the columns.
 * ELBA_ERR_STATE on a context that counted exchanged records (elba_dist_count_records) or inside a sharded multiplication. */
int  elba_release_workspace(elba_ctx *ctx);

/* ---- distributed building blocks (one context per rank/GPU; the collectives are issued by the host driver) ----------------
 * They replace, for a 1D read-row partition over the GPUs of one node, what the reference does with MPI inside the same four
 * functions: the two k-mer all-to-alls (src/KmerOps.cpp:117-151, :244-274), the k-mer id Exscan (:371-375), and the
 * redistribution of A / AT to their consumers (SpParMat ctor :396-400, Transpose src/main.cpp:272-273).
 * Exchange #1 records: (packed canonical k-mer, global read id << 32 | pos) — W + 1 uint64 words, W = words per k-mer (1 for k <= 31,
 * 2 up to 63, 3 up to 95; NLONGS of include/Kmer.hpp:95-97), most significant word first.
 * Exchange #2 records: two words, (global k-mer id, global read id << 32 | pos), a column's entries contiguous and ordered by (read,pos). */
#define ELBA_MAX_RANKS 64
/* The reference's own k-mer hash and owner on the device: Kmer::GetHash (src/Kmer.cpp:207-213: h1 of murmurhash3_x64_128, seed 313, over
 * the 8 * NLONGS key bytes) and GetKmerOwner (src/KmerOps.cpp:352-359: (size_t)(double(h) * nprocs / double(UINT64_MAX))).  kmers: n
 * packed k-mers of W = NLONGS adjacent words each (first word first); hash / owner: host outputs, either may be NULL. */
int  elba_kmer_hash_owner(elba_ctx *ctx, const uint64_t *kmers, int64_t n, int nprocs, uint64_t *hash, int32_t *owner);

/* Owners by VALUE RANGE.  The value space of the packed canonical k-mers is cut into ELBA_OWNER_BINS equal bins (leading 12 bits of the
 * first word); elba_dist_value_histogram counts this rank's instances per bin; the driver all-reduces the histograms, picks boundaries
 * that balance the instances and announces them with elba_dist_set_owner_ranges: rank r owns the bins [upper_bins[r-1], upper_bins[r]),
 * upper_bins[nranks-1] == ELBA_OWNER_BINS.  (The reference hashes, GetKmerOwner src/KmerOps.cpp:352-359; with ranges the owners'
 * reliable k-mers are disjoint ascending runs, so the global k-mer id — rank of the value — is the owner's local index plus the
 * exclusive scan of the owners' counts, the reference's MPI_Exscan of src/KmerOps.cpp:371-375: elba_dist_set_kmer_id_base.) */
#define ELBA_OWNER_BINS 4096
int  elba_dist_value_histogram(elba_ctx *ctx, uint64_t *hist, int64_t nbins);
int  elba_dist_set_owner_ranges(elba_ctx *ctx, int nranks, const uint32_t *upper_bins);
/* after elba_dist_count_records: global id of this owner's i-th reliable k-mer = base + i; nall = reliable k-mers of all owners */
int  elba_dist_set_kmer_id_base(elba_ctx *ctx, int64_t base, int64_t nall);
/* instances of this rank's reads per owner rank */
int  elba_dist_count_owners(elba_ctx *ctx, int nranks, uint64_t *counts);
/* write this rank's records into d_send (device, 8 (W + 1) bytes per record) grouped by owner; offsets[r] = first record index of owner r */
int  elba_dist_fill_send(elba_ctx *ctx, int nranks, void *d_send, const uint64_t *offsets);
/* Exchange #1 with 8-byte records (one-word k-mers, k <= 31; round 5).  The reference ships (k-mer, read, pos) per instance
 * (src/KmerOps.cpp:105-151: a k-mer of NBYTES + the 8-byte seed, batched); here an instance travels as
 * (value - first value of the owner's range) << index_bits | instance index in the SENDER's reads — the source rank is the segment of the
 * receive buffer a record arrives in — where value_bits + index_bits <= 64 (k = 17 at two or more ranks of BASELINE config 3; never k = 31).
 * elba_dist_packed_format: read_bounds[nranks + 1] = first global read of every rank, all_lens = the lengths of ALL reads (one all-gather of
 * 4 bytes per read); decides the format — the same on every rank — and stores every rank's instance offsets on the device; *fits = 0: keep
 * elba_dist_fill_send's 16-byte records.  Call after elba_dist_set_owner_ranges.  elba_dist_fill_send_packed: as elba_dist_fill_send, 8 bytes
 * per record.  elba_dist_unpack_records: the owner side — recv_counts[p] packed records of rank p, segment after segment in d_packed, become
 * 16-byte records (k-mer, global read << 32 | pos) in d_records, what elba_dist_count_records takes; `rank` = this owner. */
int  elba_dist_packed_format(elba_ctx *ctx, int nranks, const int64_t *read_bounds, const uint32_t *all_lens, int *fits, int *value_bits, int *index_bits);
int  elba_dist_fill_send_packed(elba_ctx *ctx, int nranks, void *d_send, const uint64_t *offsets);
int  elba_dist_unpack_records(elba_ctx *ctx, int nranks, int rank, const void *d_packed, const uint64_t *recv_counts, void *d_records);
/* owner side: exact count + [lower, upper] filter of the received records; builds the owner's columns (device; records are
 * borrowed until elba_dist_set_global_kmers returns) */
int  elba_dist_count_records(elba_ctx *ctx, const void *d_records, int64_t nrecords, elba_kmer_stats *stats);
/* device pointer to this owner's reliable k-mers, ascending (input of the all-gather); one-word k-mers only */
int  elba_dist_get_reliable_kmers(elba_ctx *ctx, const void **d_kmers, int64_t *n);
/* copy them into a caller-owned device buffer of at least W * capacity words, the W words of a k-mer adjacent (e.g. a torch tensor that
 * takes part in the all-gather); capacity counts k-mers */
int  elba_dist_copy_reliable_kmers(elba_ctx *ctx, void *d_dst, int64_t capacity);
/* alternative to elba_dist_set_kmer_id_base for owners that are not value ranges: all owners' reliable k-mers concatenated (any order;
 * nall k-mers of W adjacent words): global k-mer id = rank of the packed value */
int  elba_dist_set_global_kmers(elba_ctx *ctx, const void *d_all_kmers, int64_t nall);
/* column panels: read_bounds[r] = first global read id of rank r (read_bounds[nranks] = total reads).  counts[r] = records this
 * owner sends to rank r: every column, whole, for every rank that owns at least one of its reads */
int  elba_dist_panel_counts(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, uint64_t *counts);
int  elba_dist_panel_fill(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, void *d_send, const uint64_t *offsets);
/* Row-block batching: the same with one ROW BLOCK per rank, [win_lo[r], win_hi[r]) inside rank r's rows — a column goes to rank r only
 * if it has a read in that block.  A rank whose whole panel would not fit (dense columns: nearly every column reaches every rank, cf.
 * the reference's batched exchange, include/KmerOps.hpp:33-56) walks its rows block by block: panel_*_win -> all-to-all ->
 * elba_dist_set_panel(block) -> elba_create_seed_matrix -> rows of the block -> next block.  The owner's columns stay resident. */
int  elba_dist_panel_counts_win(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, const uint64_t *win_lo, const uint64_t *win_hi, uint64_t *counts);
int  elba_dist_panel_fill_win(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, const uint64_t *win_lo, const uint64_t *win_hi, void *d_send, const uint64_t *offsets);
/* create_seed_matrix on a row shard WITH mirror exchange.  elba_create_seed_matrix on a shard accumulates every pair that has a row
 * outside the shard on both ranks concerned (no communication, about twice the accumulator work and table sizes of a one-GPU run).  The
 * three calls below accumulate each pair {i, j} on ONE rank — the rank of the smaller row when i + j is even, of the larger when odd —
 * and hand the mirrored entry to the other: begin (classify + numeric; send_counts[r] = 32-byte records for rank r), fill (writes them
 * grouped by rank at offsets[r]), the driver's all-to-all, end (merges what arrived, row pointers, per-row column sort: this rank's rows of
 * B are then complete, exactly as after elba_create_seed_matrix).  read_bounds as in elba_dist_panel_counts.  The only exchange inside
 * the reference's create_seed_matrix this corresponds to are the SUMMA stages of Mult_AnXBn_DoubleBuff (src/SharedSeeds.cpp:7). */
int  elba_seed_matrix_begin(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, uint64_t *send_counts);
int  elba_seed_matrix_fill(elba_ctx *ctx, void *d_send, const uint64_t *offsets);
int  elba_seed_matrix_end(elba_ctx *ctx, const void *d_recv, int64_t nrecords, elba_overlap_stats *stats);
/* The same step with ONE host synchronisation — nothing about the exchange has to be known on the host:
 *   elba_set_stream         the library launches on the caller's HIP stream from now on (the stream its collectives are ordered against)
 *   elba_seed_matrix_send   queues classify + numeric + the grouping of the cross-rank mirror images into d_send = nranks slots of slot_records
 *                           32-byte records: record 0 of slot r is a header written on the device (count, "repeat" flag, slot size needed), the
 *                           images for rank r follow.  Does not wait.
 *   (the driver's all-to-all with EQUAL splits: slot r of d_send -> slot <this rank> of rank r's d_recv)
 *   elba_seed_matrix_recv   queues the merge of d_recv's slots (d_recv is scratch: tickets are written into it), row pointers, column sort;
 *                           synchronises once.  ELBA_OK: this rank's rows of B are complete.  ELBA_ERR_RETRY: this rank's staging area or some
 *                           rank's slot was too small — the flag travels in every header, so every rank gets this answer in the same step —
 *                           capacities have grown: repeat send / all-to-all / recv with *slot_records_needed (the same value on every rank).
 *                           With ELBA_OK *slot_records_needed is the slot size this step would have got by with (+ 1/8; <= slot_records, again
 *                           the same on every rank): a caller whose first guess was generous uses it for its next step — whole slots travel. */
int  elba_set_stream(elba_ctx *ctx, void *hip_stream);
int  elba_seed_matrix_send(elba_ctx *ctx, int nranks, const uint64_t *read_bounds, void *d_send, int64_t slot_records);
int  elba_seed_matrix_recv(elba_ctx *ctx, void *d_recv, int64_t slot_records, elba_overlap_stats *stats, int64_t *slot_records_needed);
/* receiver side: the panel of every column touching rows [row_lo,row_hi) -> columns (renumbered by rank among the columns present) + CSR; elba_create_seed_matrix
 * then computes exactly those rows of B (global column ids); elba_export_csr(row_lo,row_hi) / elba_export_dcsc read them */
int  elba_dist_set_panel(elba_ctx *ctx, const void *d_records, int64_t nrecords, int64_t nreads_total, int64_t nkmers_total,
                         int64_t row_lo, int64_t row_hi, elba_matrix_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* ELBA_AMD_H_ */
