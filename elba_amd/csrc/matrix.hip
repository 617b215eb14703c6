// matrix.hip — device construction of A in both orientations.
//   CSC(A)  = the reference's AT (src/main.cpp:272-273): column k-mer, entries (read, pos) ordered by (read, pos)
//   CSR(A)  = the reference's A  (src/KmerOps.cpp:361-401): row read, entries (kid, pos) ordered by (kid, pos)
// Entries are single u64 words (hi = read or kid, lo = pos) so that one 8-byte load fetches an entry and a row /
// column segment sorted as u64 is in canonical order.  Duplicate (read,kid) entries are kept (SumDuplicates=false,
// src/KmerOps.cpp:400).  A CSR entry also carries two ownership bits above its position (Ctx::csr_hints, common.hpp), and the columns
// exist a second time padded to a common stride (a_ell): the store the SpGEMM fetches from.
#include "common.hpp"
#include "matrix.hpp"
#include <algorithm>

namespace elba {

namespace {

// hint bits of entry z (read i) from its column, and the column's length added to the window's product count (one atomic per wavefront)
__device__ __forceinline__ uint32_t entry_hint(const uint32_t *colptr, const uint64_t *csc, uint64_t kid, uint32_t i, bool hints, uint32_t win_lo, uint32_t win_hi, unsigned long long *prod_ctr)
{
    const uint32_t c0 = colptr[kid], L = colptr[kid + 1] - c0;
    unsigned long long mine = i >= win_lo && i < win_hi ? L : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&prod_ctr[(blockIdx.x & 63u) * 16u], mine);      // (64 counters on lines of their own)
    return hints && L <= HINT_MAX_COL ? column_hint(csc + c0, L, i, win_lo, win_hi) : 0u;
}

__global__ void k_csc_to_csr_keys(const uint64_t *kid_keys, int kid_shift, uint64_t kid_mask, const uint32_t *colptr, const uint64_t *csc, int64_t Z, uint32_t *row_keys, uint64_t *csr_vals,
                                  bool hints, bool suffix, uint32_t win_lo, uint32_t win_hi, unsigned long long *prod_ctr, const uint8_t *colw0)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = z < Z;
    if (!in) z = Z - 1;                                      // (whole wavefronts reach the reduction inside entry_hint)
    uint64_t kid = (kid_keys[z] >> kid_shift) & kid_mask;
    uint64_t e = csc[z];
    const uint32_t h = entry_hint(colptr, csc, kid, in ? (uint32_t)(e >> 32) : 0xFFFFFFFFu, hints && in, in ? win_lo : 0u, in ? win_hi : 0u, prod_ctr);
    if (!in) return;
    row_keys[z] = (uint32_t)(e >> 32);                       // read
    if (suffix) {      // dense matrices (Ctx::csr_suffix): the entry knows its column's length and its own place in it — kid | L << 23 | idx << 16 | pos (16 bits)
        const uint32_t c0 = colptr[kid], L = colptr[kid + 1] - c0;
        uint32_t idx = (uint32_t)z - c0;
        if (colw0) { const uint32_t w0 = colw0[kid]; idx = idx >= w0 ? idx - w0 : idx + L - w0; }      // (a row window: the padded column is stored rotated, see k_fill_ell)
        csr_vals[z] = (kid << 32) | ((uint64_t)L << 23) | ((uint64_t)idx << 16) | (e & 0xFFFFull);
    } else csr_vals[z] = (kid << 32) | ((uint64_t)h << 30) | (e & 0xFFFFFFFFull);         // kid | hint | pos
}

// CSR build with ONE word per entry when read, k-mer id, the two hint bits and the position fit 64 bits together:
// read << (nb + pb + 2) | kid << (pb + 2) | hint << pb | pos, sorted on the read bits only (stable: rows come out in (kid, pos) order)
__global__ void k_csc_to_csr_words(const uint64_t *kid_keys, int kid_shift, const uint32_t *colptr, const uint64_t *csc, int64_t Z, int nb, int pb, int rs, int pbi, uint64_t *words,
                                   bool hints, uint32_t win_lo, uint32_t win_hi, unsigned long long *prod_ctr)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = z < Z;
    if (!in) z = Z - 1;
    const uint64_t kid = kid_keys[z] >> kid_shift, e = csc[z];
    const uint32_t h = entry_hint(colptr, csc, kid, in ? (uint32_t)(e >> 32) : 0xFFFFFFFFu, hints && in, in ? win_lo : 0u, in ? win_hi : 0u, prod_ctr);
    if (!in) return;
    uint64_t word = ((e >> 32) << rs) | (kid << (pb + 2)) | ((uint64_t)h << pb) | (e & 0xFFFFFFFFull);
    if (pbi) {
        // Ctx::csr_inline, as kmer_msd.hip writes it: a row that accumulates exactly ONE pair of this column under the parity rule over ALL rows
        // carries that pair in its own entry — flag | read << rs | (partner >> 1) << 2 pbi | posQ << pbi | posT — when both positions fit pbi bits
        const uint32_t c0 = colptr[kid], L = colptr[kid + 1] - c0, i = (uint32_t)(e >> 32);
        if (L <= HINT_MAX_COL) {
            uint32_t nown = 0, mult = 0;
            uint64_t other = 0;
            for (uint32_t t = 0; t < L; ++t) {
                const uint64_t y = csc[c0 + t];
                const uint32_t j = (uint32_t)(y >> 32);
                if (j == i) { ++mult; continue; }
                if (((i ^ j) & 1u) ? j < i : j > i) { ++nown; other = y; }
            }
            const uint64_t pos = e & 0xFFFFFFFFull, opos = other & 0xFFFFFFFFull;
            if (mult == 1u && nown == 1u && ((pos | opos) >> pbi) == 0)
                word = (1ull << 63) | ((e >> 32) << rs) | (((other >> 32) >> 1) << (2 * pbi)) | (pos << pbi) | opos;
        }
    }
    words[z] = word;
}
// hint bits for sort keys that came without them (k_runs_emit, kmer.hip): the entry's k-mer id is in the word, its column in the CSC
// (one lane per entry; one lane per COLUMN, its entries in registers, was measured: 7.0 ms against 5.2)
__global__ void k_add_hints(const uint32_t *colptr, const uint64_t *csc, int64_t Z, int nb, int pb, int rs, int pbi, uint64_t *words)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    const uint64_t w = words[z];
    const uint64_t kid = (w >> (pb + 2)) & ((1ull << nb) - 1);
    const uint32_t c0 = colptr[kid], L = colptr[kid + 1] - c0;
    if (L > HINT_MAX_COL) return;
    const uint64_t e = csc[z];
    const uint32_t i = (uint32_t)(e >> 32);
    if (pbi) {
        // Ctx::csr_inline (the whole matrix, parity rule over all rows): exactly one pair of this column accumulated by this row -> the entry carries it
        uint32_t nown = 0, mult = 0;
        uint64_t other = 0;
        for (uint32_t t = 0; t < L; ++t) {
            const uint64_t y = csc[c0 + t];
            const uint32_t j = (uint32_t)(y >> 32);
            if (j == i) { ++mult; continue; }
            if (((i ^ j) & 1u) ? j < i : j > i) { ++nown; other = y; }
        }
        const uint64_t pos = e & 0xFFFFFFFFull, opos = other & 0xFFFFFFFFull;
        if (mult == 1u && nown == 1u && ((pos | opos) >> pbi) == 0) { words[z] = (1ull << 63) | ((e >> 32) << rs) | (((other >> 32) >> 1) << (2 * pbi)) | (pos << pbi) | opos; return; }
        if (mult < 2u && nown == 0u) words[z] = w | (3ull << pb);      // (the whole matrix: both hint bits agree)
        return;
    }
    words[z] = w | ((uint64_t)column_hint(csc + c0, L, i, 0u, 0xFFFFFFFFu) << pb);      // (the window is the whole matrix)
}
// sorted keys -> CSR entries, and the row pointers with them: entry z opens the rows (read of z - 1, read of z]; launched with Z + 1 lanes,
// the last of which closes the rows behind the last entry
__global__ void k_unpack_csr_words(const uint64_t *words, int64_t Z, int nb, int pb, int rs, int mb, int pbi, uint64_t *csr, uint32_t *rowptr, int64_t M)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z > Z) return;
    const uint64_t rmask = (1ull << mb) - 1;
    const int64_t prev = z == 0 ? -1 : (int64_t)((words[z - 1] >> rs) & rmask);
    const uint64_t w = z < Z ? words[z] : 0;
    const int64_t cur = z < Z ? (int64_t)((w >> rs) & rmask) : M;
    for (int64_t k = prev + 1; k <= cur; ++k) rowptr[k] = (uint32_t)z;
    if (z < Z) {
        if (w >> 63) {      // inline partner (Ctx::csr_inline): flag | partner >> 1 | posQ | posT << 16
            const uint64_t pm = (1ull << pbi) - 1;
            csr[z] = (1ull << 63) | (((w >> (2 * pbi)) & ((1ull << (mb - 1)) - 1)) << 32) | ((w >> pbi) & pm) | ((w & pm) << 16);
        } else csr[z] = (((w >> (pb + 2)) & ((1ull << nb) - 1)) << 32) | (((w >> pb) & 3ull) << 30) | (w & ((1ull << pb) - 1));
    }
}

__global__ void k_colrow_to_csc(const uint64_t *colrow, const uint64_t *pos, int64_t Z, uint64_t *csc)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    csc[z] = ((colrow[z] & 0xFFFFFFFFull) << 32) | (pos[z] & 0xFFFFFFFFull);
}

__global__ void k_max_seg_len(const uint32_t *ptr, int64_t nseg, unsigned long long *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long m = 0;
    for (; i < nseg; i += stride) {
        unsigned long long l = ptr[i + 1] - ptr[i];
        m = l > m ? l : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// largest position (low word) among the column entries
__global__ void k_max_low32(const uint64_t *v, int64_t n, unsigned long long *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long m = 0;
    for (; i < n; i += stride) { const unsigned long long l = (uint32_t)v[i]; m = l > m ? l : m; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// padded column store: slot j of column kid (at kid * stride + j) holds the column's j-th entry, or all ones behind its end; one lane per
// slot, so the stores of a wavefront are one contiguous 512 bytes (no fill pass before, no column-id array)
// The dense path's column store (a_ellj): the partner READS of column kid, right-aligned in the aligned block of Sj = 1 << jsh four-byte slots
// from kid * Sj — its L entries are the block's last L slots, all ones in front of them.  What a row entry owns (the column behind its own
// place) is then the tail of the block: one aligned 128-byte segment for every entry but the first ones of columns longer than 33.  Copied
// from the padded 8-byte store (whatever rotation it carries, k_fill_ell), one lane per 16-byte piece of the output.
// (ell == null: no rotation — the whole matrix —: the columns themselves, csc, are read instead of the padded store — 8 bytes per entry, not per slot)
__global__ void k_ell_partners(const uint64_t *ell, const uint64_t *csc, const uint32_t *colptr, uint64_t N, uint32_t S, uint32_t jsh, uint32_t *ellj, const uint32_t *label)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, nq = N << (jsh - 2u);
    const uint32_t Sj = 1u << jsh;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nq; t += stride) {
        const uint64_t kid = t >> (jsh - 2u);
        const uint32_t c0 = colptr[kid], s0 = ((uint32_t)t & ((Sj >> 2) - 1u)) << 2, L = colptr[kid + 1] - c0;
        const uint64_t *col = ell ? ell + kid * S : csc + c0;
        uint4 v;
        v.x = s0 + L >= Sj ? (uint32_t)(col[s0 + L - Sj] >> 32) : 0xFFFFFFFFu;
        v.y = s0 + 1u + L >= Sj ? (uint32_t)(col[s0 + 1u + L - Sj] >> 32) : 0xFFFFFFFFu;
        v.z = s0 + 2u + L >= Sj ? (uint32_t)(col[s0 + 2u + L - Sj] >> 32) : 0xFFFFFFFFu;
        v.w = s0 + 3u + L >= Sj ? (uint32_t)(col[s0 + 3u + L - Sj] >> 32) : 0xFFFFFFFFu;
        if (label) {      // (partners by their LABEL — Ctx::row_label: reads of one locus get neighbouring labels)
            if (v.x != 0xFFFFFFFFu) v.x = label[v.x];
            if (v.y != 0xFFFFFFFFu) v.y = label[v.y];
            if (v.z != 0xFFFFFFFFu) v.z = label[v.z];
            if (v.w != 0xFFFFFFFFu) v.w = label[v.w];
        }
        reinterpret_cast<uint4 *>(ellj)[t] = v;
    }
}
// colw0 (dense matrices with a row window): column kid is stored ROTATED by colw0[kid] = its entries of reads below the window — the window's
// entries first, then the reads above it, then the reads below it.  With pairs owned by the smaller row INSIDE the window and every partner
// outside it kept (owns_pair), the candidates a window row's entry owns are then exactly the slots behind its own, as without a window.
// (Relative order inside each of the three groups is kept, so a product's sequence number still orders the products of one pair.)
__global__ void k_col_w0(const uint32_t *colptr, const uint64_t *csc, uint64_t N, uint32_t win_lo, uint8_t *colw0)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const uint32_t c0 = colptr[k], L = colptr[k + 1] - c0;
    uint32_t w0 = 0;
    while (w0 < L && (uint32_t)(csc[c0 + w0] >> 32) < win_lo) ++w0;
    colw0[k] = (uint8_t)(w0 == L ? 0u : w0);      // (a column wholly below the window — it holds no window row — keeps its order)
}
__global__ void k_fill_ell(const uint32_t *colptr, const uint64_t *csc, uint64_t nslots, uint32_t cs, uint64_t *ell, const uint8_t *colw0)
{
    // two slots (16 bytes) per lane: the stride is even, so a pair never straddles two columns.  (Adding the hint bits of the entries here, by the lane
    // that copies them, was measured: 15.8 ms against 4.8 + 5.2 for this kernel and k_add_hints — most lanes hold padding.)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, npairs = nslots >> 1;
    for (uint64_t t2 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t2 < npairs; t2 += stride) {
        const uint64_t t = t2 << 1, kid = t / cs;
        const uint32_t j = (uint32_t)(t - kid * cs), c0 = colptr[kid], L = colptr[kid + 1] - c0, w0 = colw0 ? colw0[kid] : 0u;
        ulonglong2 v;
        v.x = j < L ? csc[c0 + (j + w0 < L ? j + w0 : j + w0 - L)] : ~0ull;
        v.y = j + 1u < L ? csc[c0 + (j + 1u + w0 < L ? j + 1u + w0 : j + 1u + w0 - L)] : ~0ull;
        reinterpret_cast<ulonglong2 *>(ell)[t2] = v;
    }
}

// dense matrices: smallest k-mer id of the row << 32 | row (an empty row: behind all others); the sorted keys give label -> row and row -> label
__global__ void k_row_minimizer(const uint32_t *rowptr, const uint64_t *csr, uint32_t M, uint64_t N, uint64_t *keys)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const uint32_t rs = rowptr[r], re = rowptr[r + 1];
    const uint64_t kid = rs < re ? csr[rs] >> 32 : N;      // (dense row entries: k-mer id in the upper word)
    keys[r] = kid << 32 | r;
}
__global__ void k_order_and_labels(const uint64_t *keys, uint32_t M, uint32_t *order, uint32_t *label)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) { const uint32_t r = (uint32_t)keys[t]; order[t] = r; label[r] = t; }
}

int bits_for(uint64_t maxval)
{
    int b = 1;
    while (b < 64 && (maxval >> b)) ++b;
    return b;
}

}  // namespace

int64_t max_segment_len(Ctx &c, const uint32_t *ptr, int64_t nseg)
{
    if (nseg <= 0) return 0;
    c.ws_scan.reserve(64);
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 8, c.stream));
    int64_t nb = (nseg + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_max_seg_len, dim3((unsigned)nb), dim3(256), 0, c.stream, ptr, nseg, c.ws_scan.as<unsigned long long>());
    uint64_t h = 0;
    ELBA_HIP(hipMemcpyAsync(&h, c.ws_scan.p, 8, hipMemcpyDeviceToHost, c.stream));
    ELBA_HIP(hipStreamSynchronize(c.stream));
    return (int64_t)h;
}

// The column store the SpGEMM gathers from.  No column longer than 64 entries (UPPER <= 64: every configuration the reference
// documents): columns padded to a common stride S (4 entries, or whole 64-byte lines), column kid at kid * S — its address is arithmetic, and a
// group of S/2 lanes reads it as one aligned segment.  Longer columns (or no room for the padding): the plain CSC, reached through a_colptr.
// Sets use_ell, s_stride, lpc_log2, fbits and reserves a_ell (guard words written); the caller fills it.
void choose_column_store(Ctx &c, int64_t N, int64_t max_col)
{
    hipStream_t s = c.stream;
    const uint32_t mc = (uint32_t)(max_col > 0 ? max_col : 1);
    // stride: 4 entries, else the longest column rounded up to whole 64-byte lines (8 entries) — a power of two is not needed: the kernel
    // multiplies (U = 35: 40 entries = 5 lines per gather where 64 entries would fetch 8)
    const uint32_t stride = mc <= 4 ? 4u : (mc + 7u) & ~7u;
    size_t free_b = 0, total_b = 0;
    ELBA_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t ell_bytes = (size_t)N * stride * 8;
    c.use_ell = mc <= 64 && N > 0 && !c.opt.no_ell && ell_bytes <= (free_b + c.a_ell.cap) / 3 && (uint64_t)N * stride < (1ull << 40);
    if (!c.use_ell && mc <= 64 && N > 0 && !c.opt.no_ell && c.opt.trace)
        fprintf(stderr, "[elba] the padded column store (%zu bytes) does not fit a third of the free device memory (%zu bytes): columns are gathered from CSC\n", ell_bytes, free_b + c.a_ell.cap);
    c.ell_compact = false; c.ell_nslots = c.use_ell ? N : 0; c.ell_cap_cols = c.use_ell ? N : 0;      // (kmer_msd.hip may compact the store: gather slots)
    if (c.use_ell) {
        uint32_t lb = 1, fb = 2;
        while ((2u << lb) < stride) ++lb;              // lanes per row entry: 16 bytes (two entries) each
        while ((1u << fb) < stride) ++fb;
        c.s_stride = stride; c.lpc_log2 = lb; c.fbits = fb;
        c.a_ell.reserve(ell_bytes + 64);
        ELBA_HIP(hipMemsetAsync(c.a_ell.as<char>() + ell_bytes, 0xFF, 64, s));      // (guard words behind the last column)
    } else {
        // lanes per row entry: half the longest column, between 2 and 64 (a column is walked in chunks of 2 * lanes entries)
        uint32_t lb = 1;
        while (lb < 6 && (2u << lb) < mc) ++lb;
        c.lpc_log2 = lb; c.s_stride = 0;
        int fb = 1;
        while (fb < 31 && ((uint64_t)(mc > 1 ? mc - 1 : 1) >> fb)) ++fb;
        c.fbits = (uint32_t)fb;
    }
}

// Input: Z entries sorted by (kid, read, pos): kid_keys[z] >> kid_shift = kid, csc[z] = read<<32|pos (device, in c.a_csc or elsewhere).
// Produces c.a_colptr, c.a_csc (copy if csc is not already c.a_csc), c.a_rowptr, c.a_csr, max_row_nnz, max_col_nnz.
// [win_lo, win_hi) = the rows of B this context computes (win_hi < 0: all); the product schedule is laid out for those rows only.
void finish_matrix_from_sorted_csc(Ctx &c, int64_t M, int64_t N, int64_t Z, const uint64_t *kid_keys, int kid_shift, const uint64_t *csc, int64_t win_lo, int64_t win_hi, bool pre)
{
    // pre: the k-mer stage left the column pointers, a bound on the positions and (pre_words) the CSR sort keys behind (k_runs_emit, kmer.hip)
    hipStream_t s = c.stream;
    c.M = M; c.N = N; c.Z = Z;
    c.a_colptr.reserve((size_t)(N + 1) * 4);
    c.a_rowptr.reserve((size_t)(M + 1) * 4);
    c.a_csc.reserve((size_t)(Z + 8) * 8);   // + guard entries: the SpGEMM gathers up to four consecutive entries without a bounds check
    c.a_csr.reserve((size_t)(Z + 1) * 8);
    if (!pre) group_offsets_u32(s, kid_keys, kid_shift, Z, c.a_colptr.as<uint32_t>(), N);
    if (csc != c.a_csc.as<uint64_t>() && Z > 0)
        ELBA_HIP(hipMemcpyAsync(c.a_csc.p, csc, (size_t)Z * 8, hipMemcpyDeviceToDevice, s));
    // largest position among the entries: below 2^16 the SpGEMM's 64-bit accumulators carry both positions of a seed (spgemm_direct.hpp);
    // it also decides whether an entry fits one word for the CSR sort
    uint64_t maxpos = pre ? c.pre_maxpos : 0;
    if (Z > 0 && !pre) {
        int64_t nbz = (Z + 255) / 256;
        if (nbz > 2048) nbz = 2048;
        c.ws_scan.reserve(64);
        ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 8, s));
        hipLaunchKernelGGL(k_max_low32, dim3((unsigned)nbz), dim3(256), 0, s, c.a_csc.as<uint64_t>(), Z, c.ws_scan.as<unsigned long long>());
        ELBA_HIP(hipMemcpyAsync(&maxpos, c.ws_scan.p, 8, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
    }
    c.pos16 = maxpos < 65536;
    ELBA_REQUIRE(win_lo >= 0 && (win_hi < 0 || (win_lo <= win_hi && win_hi <= M)), ELBA_ERR_INVALID_ARG, "bad row window");
    // stable sort by read: rows come out ordered by (kid, pos)
    c.ws_a.reserve((size_t)(Z + 1) * 8); c.ws_c.reserve((size_t)(Z + 1) * 8);
    const int mb = bits_for((uint64_t)(M > 0 ? M - 1 : 0)), nb = bits_for((uint64_t)(N > 0 ? N - 1 : 0)), pb = bits_for(maxpos);
    bool hints = pre ? c.pre_hints : (pb <= 30 && !c.opt.no_hints);
    const uint32_t wlo = (uint32_t)win_lo, whi = (uint32_t)(win_hi < 0 ? M : win_hi);
    c.prod_ctr.reserve(64 * 128);
    unsigned long long *prod_ctr = c.prod_ctr.as<unsigned long long>();
    if (!pre) ELBA_HIP(hipMemsetAsync(prod_ctr, 0, 64 * 128, s));      // (pre: k_runs_emit has counted)
    // Dense matrices (columns of more than 16 reads, e.g. UPPER = 35 on 40x low-error reads: every row owns pairs of every column, hundreds of
    // products per surviving pair): pairs are owned by their SMALLER row and an entry's owned candidates are the column's entries behind its
    // own — the row entries then carry the column's length and their own place in it, and the SpGEMM hands out exactly those candidates to
    // its lanes (spgemm_direct.hpp, "suffix" path).  Positions below 2^16, the padded column store.  With a row window (a shard, a row block
    // of a shard: elba_dist_set_panel) the padded columns are stored rotated so that this stays true (k_fill_ell).
    const bool windowed = !(win_lo == 0 && (win_hi < 0 || win_hi == M));
    c.csr_inline = false; c.csr_inline_window = false;
    const uint8_t *colw0 = nullptr;
    if (!(pre && c.pre_ell_done)) {      // (the two-level partition of kmer_msd.hip writes the padded columns with the columns themselves)
        c.max_col_nnz = max_segment_len(c, c.a_colptr.as<uint32_t>(), N);
        choose_column_store(c, N, c.max_col_nnz);
    }
    c.csr_suffix = c.use_ell && c.pos16 && c.max_col_nnz > 16 && !c.opt.no_pay && !c.opt.no_suffix;
    if (c.csr_suffix && windowed && N > 0) {
        c.col_w0.reserve((size_t)N + 16);
        hipLaunchKernelGGL(k_col_w0, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), (uint64_t)N, wlo, c.col_w0.as<uint8_t>());
        colw0 = c.col_w0.as<uint8_t>();
    }
    if (!(pre && c.pre_ell_done) && c.use_ell) {
        const uint64_t nslots = (uint64_t)N * c.s_stride;
        hipLaunchKernelGGL(k_fill_ell, dim3((unsigned)std::min<uint64_t>((nslots / 2 + 255) / 256, 1ull << 30)), dim3(256), 0, s, c.a_colptr.as<uint32_t>(), c.a_csc.as<uint64_t>(), nslots, c.s_stride, c.a_ell.as<uint64_t>(), colw0);
    }
    if (c.csr_suffix) {
        hints = false;
        c.j_shift = c.max_col_nnz <= 32 ? 5u : 6u;      // (use_ell: no column longer than 64)
    }
    if (!c.csr_suffix && mb + nb + pb + 2 <= 64 && !c.opt.csr_pairs) {
        const bool have_words = pre && c.pre_words && c.pre_nb == nb && c.pre_pb == pb;
        uint64_t *w0 = have_words ? c.csr_words.as<uint64_t>() : c.ws_a.as<uint64_t>(), *w1 = c.ws_c.as<uint64_t>();
        ELBA_REQUIRE(!pre || have_words, ELBA_ERR_INTERNAL, "create_kmer_matrix: the sort keys of the k-mer stage do not match the matrix");
        int rs = have_words ? c.pre_rs : nb + pb + 2, pbi = have_words ? c.pre_pbi : 0;
        // inline partners for a matrix that did not come from the k-mer stage of this context (triples, a panel of the sharded build): the keys are
        // written here.  The rule they follow is the parity rule over ALL rows — what a whole matrix uses, and a shard when the mirror exchange
        // assigns every pair of the job to one of its two rows (elba_seed_matrix_send / _begin); a windowed matrix gets them only on request
        // (option "panel_inline") and can then not be multiplied without the exchange.
        bool inl_here = false;
        if (!have_words && hints && c.use_ell && c.pos16 && !c.opt.no_pay && !c.opt.no_inline && !c.opt.no_symmetry && (!windowed || c.opt.panel_inline) && N < (1ll << 31) && mb >= 2) {
            const int rs2 = 63 - mb, pbi2 = std::min(pb, (rs2 - (mb - 1)) / 2);
            if (rs2 >= nb + pb + 2 && pbi2 >= 10) { rs = rs2; pbi = pbi2; inl_here = true; }
        }
        if (Z > 0 && !have_words)
            hipLaunchKernelGGL(k_csc_to_csr_words, dim3((unsigned)((Z + 255) / 256)), dim3(256), 0, s, kid_keys, kid_shift, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), Z, nb, pb, rs, inl_here ? pbi : 0, w0,
                               hints, wlo, whi, prod_ctr);
        // (sort keys of the k-mer stage's sort path: they left room for inline partners — whether the matrix qualifies is known here)
        const bool inl_late = have_words && hints && !c.pre_hints_done && c.pre_inline_pending && c.use_ell && !c.csr_suffix && c.pos16 && !windowed;
        if (Z > 0 && have_words && hints && !c.pre_hints_done)
            hipLaunchKernelGGL(k_add_hints, dim3((unsigned)((Z + 255) / 256)), dim3(256), 0, s, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), Z, nb, pb, rs, inl_late ? pbi : 0, w0);
        // (gather slots, kmer_msd.hip: the id field of a key may hold a slot beyond the last k-mer id — slots are drawn in chunks — and is as wide as
        //  the key leaves room below the read)
        const int idbits = have_words && c.ell_compact ? std::min(32, rs - pb - 2) : nb;
        const int pbi_used = (have_words && c.pre_inline) || inl_here || inl_late ? pbi : 0;
        if (Z > 0 && M > 0 && c.opt.tune[1] != 1) {
            // the sort's last pass writes the rows themselves and their pointers (prims.hip: k_rs_scatter<FIN>): no pass over the sorted keys behind it
            // (round 4: k_unpack_csr_words read and wrote all of them once more — 1.9 ms of the 10 ms CSR build of BASELINE config 3)
            const CsrFin fin{idbits, pb, rs, mb, pbi_used, c.a_csr.as<uint64_t>(), c.a_rowptr.as<uint32_t>(), M};
            radix_sort_keys_to_csr(s, w0, w1, Z, fin, c.ws_sort);
        } else {
        const int where = radix_sort_keys(s, w0, w1, Z, rs, rs + mb, c.ws_sort);
        const uint64_t *sorted = where ? w1 : w0;
        hipLaunchKernelGGL(k_unpack_csr_words, dim3((unsigned)((Z + 1 + 255) / 256)), dim3(256), 0, s, sorted, Z, idbits, pb, rs, mb, pbi_used, c.a_csr.as<uint64_t>(), c.a_rowptr.as<uint32_t>(), M);
        }
        c.csr_inline = (have_words && c.pre_inline) || inl_here || inl_late;
        c.csr_inline_window = inl_here && windowed;
    } else {
        const bool pre_pairs = pre && c.pre_pairs;           // (kmer_msd.hip has written the pairs of a dense matrix: keys in ws_b, values where the sort must start to end in a_csr)
        ELBA_REQUIRE(!pre || !c.pre_words || c.csr_suffix || pre_pairs, ELBA_ERR_INTERNAL, "create_kmer_matrix: the sort keys of the k-mer stage do not match the matrix");
        ELBA_REQUIRE(!pre_pairs || c.csr_suffix, ELBA_ERR_INTERNAL, "create_kmer_matrix: the k-mer stage wrote the pairs of a dense matrix, the matrix is not one");
        const bool kid_in_words = pre && c.pre_words && !pre_pairs;        // (k_runs_emit left sort keys, not column ids: the k-mer id is a field of the word)
        // the sorted values are the rows of A: the sort's buffers are handed over so that it ENDS in a_csr (no copy behind it)
        c.ws_b.reserve((size_t)(Z + 1) * 8); c.ws_d.reserve((size_t)(Z + 1) * 8);
        const bool ends_in_second = radix_sort_where(Z, 0, mb) != 0;
        // (row ids as 32-bit keys: 12 bytes per pair and pass)
        uint32_t *k0 = pre_pairs ? c.ws_b.as<uint32_t>() : c.ws_a.as<uint32_t>(), *k1 = pre_pairs ? c.ws_a.as<uint32_t>() : c.ws_c.as<uint32_t>();
        uint64_t *spare = pre_pairs ? c.ws_d.as<uint64_t>() : c.ws_b.as<uint64_t>();
        uint64_t *v0 = ends_in_second ? spare : c.a_csr.as<uint64_t>(), *v1 = ends_in_second ? c.a_csr.as<uint64_t>() : spare;
        const uint64_t *kk = pre_pairs ? v0 : kid_in_words ? c.csr_words.as<uint64_t>() : kid_keys;      // (pairs that must be written again — a row window rotates the columns —: the k-mer id is the value's upper half)
        const int ks = pre_pairs ? 32 : kid_in_words ? c.pre_pb + 2 : kid_shift;
        const uint64_t km = pre_pairs ? 0xFFFFFFFFull : kid_in_words ? (1ull << c.pre_nb) - 1 : ~0ull;
        if (Z > 0 && !(pre_pairs && !windowed)) {
            int64_t nbk = (Z + 255) / 256;
            if (pre) ELBA_HIP(hipMemsetAsync(prod_ctr, 0, 64 * 128, s));
            hipLaunchKernelGGL(k_csc_to_csr_keys, dim3((unsigned)nbk), dim3(256), 0, s, kk, ks, km, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), Z, k0, v0,
                               hints, c.csr_suffix, wlo, whi, prod_ctr, colw0);
        }
        const int where = radix_sort_pairs_k32(s, k0, v0, k1, v1, Z, 0, mb, c.ws_sort);
        ELBA_REQUIRE((where != 0) == ends_in_second, ELBA_ERR_INTERNAL, "create_kmer_matrix: the CSR sort ended in the other buffer");
        group_offsets_k32(s, where ? k1 : k0, Z, c.a_rowptr.as<uint32_t>(), M);
    }
    c.csr_hints = hints;
    {
        unsigned long long hp[64 * 16];
        ELBA_HIP(hipMemcpyAsync(hp, prod_ctr, sizeof(hp), hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        c.A_products = 0;
        for (int q = 0; q < 64; ++q) c.A_products += (int64_t)hp[q * 16];
    }
    c.row_lo = win_lo; c.row_hi = win_hi;
    // a new matrix: the tier queues and the tier / sort usage of the previous one are forgotten.  The OUTPUT capacity is kept as a guess (the
    // buffers exist): the first SpGEMM call on this matrix then runs without a host round trip in its middle and checks afterwards that
    // everything fitted (spgemm.hip repeats the call on the synchronising path otherwise)
    c.ov_tiers_known = false; c.ov_sort_used[0] = c.ov_sort_used[1] = true;
    c.ov_prior_q16 = 0;            // a new matrix: forget the partner/product ratio measured on the previous one
    c.ov_slab_q16 = 0;
    c.ov_phase = 0;                // ... and a sharded call that was begun on the previous one
    c.max_row_nnz = max_segment_len(c, c.a_rowptr.as<uint32_t>(), M);
    {
        ELBA_REQUIRE(c.fbits < 31 && (uint64_t)c.max_row_nnz < (1ull << (32 - c.fbits)), ELBA_ERR_UNSUPPORTED,
                     "row nnz x column nnz exceeds the 32-bit product sequence number");
    }
    // Dense matrices: a row meets the same ~100 partners tens of thousands of times, and what its accumulator pays for is every look-up that
    // does not find its partner in the slot its hash names (it queues for the general insert: profiles/r03_notes.md).  Partner ids are random
    // — reads that overlap stand anywhere in the input — so the keys of a table collide as random keys do.  Reads of one locus share their
    // MINIMIZER, the smallest k-mer id of the row (its first entry: rows are in column order): ~20 reads of a 40x set share one.  The rows
    // sorted by minimizer give every read a LABEL (its rank); a row's partners are then a handful of runs of consecutive labels, which the
    // multiplicative hash keeps apart (config 5 / 25: 10.1 -> 9.7 ms).  The partner store of the dense path (a_ellj) holds labels, the
    // table is keyed by them, and the kernel turns the few surviving partners back into rows when it writes them.
    c.have_row_order = false;
    if (c.csr_suffix && !c.opt.no_row_order && M > 1 && Z > 0) {
        c.row_order.reserve((size_t)M * 4 + 64); c.row_label.reserve((size_t)M * 4 + 64);
        c.row_keys.reserve((size_t)(M + 1) * 16);
        uint64_t *k0 = c.row_keys.as<uint64_t>(), *k1 = k0 + (M + 1);
        hipLaunchKernelGGL(k_row_minimizer, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, (const uint32_t *)c.a_rowptr.as<uint32_t>(), (const uint64_t *)c.a_csr.as<uint64_t>(), (uint32_t)M, (uint64_t)N, k0);
        const int where = radix_sort_keys(s, k0, k1, M, 32, 32 + bits_for((uint64_t)(N > 0 ? N : 1)), c.ws_sort);
        hipLaunchKernelGGL(k_order_and_labels, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, (const uint64_t *)(where ? k1 : k0), (uint32_t)M, c.row_order.as<uint32_t>(), c.row_label.as<uint32_t>());
        c.have_row_order = true;
    }
    if (c.csr_suffix) {
        const uint64_t nslots = (uint64_t)N << c.j_shift;
        c.a_ellj.reserve((size_t)nslots * 4 + 64);
        hipLaunchKernelGGL(k_ell_partners, dim3((unsigned)std::min<uint64_t>((nslots / 4 + 255) / 256, 1ull << 20)), dim3(256), 0, s, windowed ? (const uint64_t *)c.a_ell.as<uint64_t>() : (const uint64_t *)nullptr, (const uint64_t *)c.a_csc.as<uint64_t>(), (const uint32_t *)c.a_colptr.as<uint32_t>(), (uint64_t)N, c.s_stride,
                           c.j_shift, c.a_ellj.as<uint32_t>(), c.have_row_order ? (const uint32_t *)c.row_label.as<uint32_t>() : (const uint32_t *)nullptr);
    }
    c.have_A = true;
    c.have_B = false;
}

namespace {
// device triples -> (key = pos, value = col << 32 | row) pairs of the three-pass sort; out-of-range indices are counted, the largest position kept
__global__ void k_pack_triples(const int64_t *rows, const int64_t *cols, const uint32_t *vals, int64_t Z, int64_t M, int64_t N, uint64_t *k0, uint64_t *v0, unsigned long long *chk)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0, mx = 0;
    if (z < Z) {
        const int64_t r = rows[z], cc = cols[z];
        const uint32_t v = vals[z];
        if (r < 0 || r >= M || cc < 0 || cc >= N) bad = 1;
        else { k0[z] = v; v0[z] = ((uint64_t)cc << 32) | (uint64_t)r; mx = v; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { bad += __shfl_xor(bad, d, 64); const unsigned long long o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
    // (one hot word: a wavefront only touches it when it would raise it — 8 M wavefronts each doing an atomic on it took 95 ms)
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&chk[0], bad); if (mx > __hip_atomic_load(&chk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&chk[1], mx); }
}
// the same check alone (out-of-range indices counted, largest position kept): what the one-word sort needs to know before it packs
__global__ void k_check_triples(const int64_t *rows, const int64_t *cols, const uint32_t *vals, int64_t Z, int64_t M, int64_t N, unsigned long long *chk)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0, mx = 0;
    if (z < Z) {
        const int64_t r = rows[z], cc = cols[z];
        if (r < 0 || r >= M || cc < 0 || cc >= N) bad = 1; else mx = vals[z];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { bad += __shfl_xor(bad, d, 64); const unsigned long long o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&chk[0], bad); if (mx > __hip_atomic_load(&chk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&chk[1], mx); }
}
// device triples -> ONE sort word each, col << (mb + pb) | row << pb | pos: sorted as 64-bit keys they are in (col, row, pos) order
__global__ void k_pack_triple_words(const int64_t *rows, const int64_t *cols, const uint32_t *vals, int64_t Z, int mb, int pb, uint64_t *w)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z < Z) w[z] = ((uint64_t)cols[z] << (mb + pb)) | ((uint64_t)rows[z] << pb) | (uint64_t)vals[z];
}
// sorted words -> what finish_matrix_from_sorted_csc takes: col << 32 | row and row << 32 | pos
__global__ void k_unpack_triple_words(const uint64_t *w, int64_t Z, int mb, int pb, uint64_t *keys, uint64_t *csc)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    const uint64_t x = w[z], row = (x >> pb) & ((1ull << mb) - 1), pos = x & ((1ull << pb) - 1);
    keys[z] = ((x >> (mb + pb)) << 32) | row;
    csc[z] = (row << 32) | pos;
}
// A as triples, from the resident matrix (rows: the context's row ids, cols: k-mer ids, vals: positions), in CSR order
__global__ void k_export_triples(const uint32_t *rowptr, const uint64_t *csr, uint32_t M, uint32_t pos_mask, int64_t *rows, int64_t *cols, uint32_t *vals)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < M; i += nwaves)
        for (uint32_t e = rowptr[i] + lane, re = rowptr[i + 1]; e < re; e += 64) { const uint64_t x = csr[e]; rows[e] = i; cols[e] = (int64_t)(x >> 32); vals[e] = (uint32_t)x & pos_mask; }
}
__global__ void k_export_triples_csc(const uint32_t *colptr, const uint64_t *csc, uint64_t N, int64_t *rows, int64_t *cols, uint32_t *vals)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    for (uint32_t z = colptr[k], z1 = colptr[k + 1]; z < z1; ++z) { rows[z] = (int64_t)(csc[z] >> 32); cols[z] = (int64_t)k; vals[z] = (uint32_t)csc[z]; }
}
}  // namespace

// k0 / v0 (= c.ws_a / c.ws_b) hold (pos, col << 32 | row) of every triple
static void set_kmer_matrix_from_pairs(Ctx &c, int64_t M, int64_t N, int64_t Z, uint64_t maxpos)
{
    hipStream_t s = c.stream;
    uint64_t *k0 = c.ws_a.as<uint64_t>(), *v0 = c.ws_b.as<uint64_t>(), *k1 = c.ws_c.as<uint64_t>(), *v1 = c.ws_d.as<uint64_t>();
    // (1) by pos, (2) by row, (3) by col — LSD, stable: final order (col, row, pos)
    int w = radix_sort_pairs(s, k0, v0, k1, v1, Z, 0, bits_for(maxpos), c.ws_sort);
    // swap roles: key := (col<<32|row), value := pos
    uint64_t *ck = w ? v1 : v0, *cv = w ? k1 : k0, *ok = w ? v0 : v1, *ov = w ? k0 : k1;
    int w2 = radix_sort_pairs(s, ck, cv, ok, ov, Z, 0, bits_for((uint64_t)(M > 0 ? M - 1 : 0)), c.ws_sort);
    if (w2) { uint64_t *t; t = ck; ck = ok; ok = t; t = cv; cv = ov; ov = t; }
    int w3 = radix_sort_pairs(s, ck, cv, ok, ov, Z, 32, 32 + bits_for((uint64_t)(N > 0 ? N - 1 : 0)), c.ws_sort);
    if (w3) { uint64_t *t; t = ck; ck = ok; ok = t; t = cv; cv = ov; ov = t; }
    // ck = (col<<32|row) sorted, cv = pos.  Move the keys out of the ws_a..d pool (finish_* reuses it).
    uint64_t *keys = c.ws_e.as<uint64_t>();
    uint64_t *csc = c.ws_f.as<uint64_t>();
    if (Z > 0) {
        ELBA_HIP(hipMemcpyAsync(keys, ck, (size_t)Z * 8, hipMemcpyDeviceToDevice, s));
        int64_t nb = (Z + 255) / 256;
        hipLaunchKernelGGL(k_colrow_to_csc, dim3((unsigned)nb), dim3(256), 0, s, ck, cv, Z, csc);
    }
    c.A_has_kmers = false;
    finish_matrix_from_sorted_csc(c, M, N, Z, keys, 32, csc);
}

void stage_set_kmer_matrix_device(Ctx &c, int64_t M, int64_t N, int64_t Z, const int64_t *d_rows, const int64_t *d_cols, const uint32_t *d_vals)
{
    ELBA_REQUIRE(M >= 0 && N >= 0 && Z >= 0, ELBA_ERR_INVALID_ARG, "negative matrix dimension");
    ELBA_REQUIRE(M < 0xFFFFFFFFll && N < 0xFFFFFFFFll && Z < 0xFFFFFFFFll, ELBA_ERR_UNSUPPORTED, "matrix dimension beyond 32-bit device indices");
    ELBA_REQUIRE(Z == 0 || (d_rows && d_cols && d_vals), ELBA_ERR_INVALID_ARG, "null triple array");
    hipStream_t s = c.stream;
    c.ws_scan.reserve(64);
    const unsigned nbz = (unsigned)((Z + 255) / 256);
    // Matrices of some size whose columns all hold entries (the reference's A: a reliable k-mer occurs LOWER times at least) go through the k-mer stage's
    // bucket kernels — two-level partition by column, every bucket of ~256 columns sorted in LDS, columns, hints, inline partners, gather slots and the
    // CSR build's sort keys written from there (kmer_msd.hip; it checks the indices as it packs them) — instead of seven radix passes over the whole
    // matrix and a pass per by-product
    c.triples_path = 0;
    if (msd_matrix_from_triples(c, M, N, Z, d_rows, d_cols, d_vals)) {
        c.triples_path = 1;
        c.A_has_kmers = false;
        finish_matrix_from_sorted_csc(c, M, N, Z, nullptr, 0, c.a_csc.as<uint64_t>(), 0, -1, true);
        c.pre_ready = false; c.pre_consumed = true;
        return;
    }
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 16, s));
    if (Z > 0) hipLaunchKernelGGL(k_check_triples, dim3(nbz), dim3(256), 0, s, d_rows, d_cols, d_vals, Z, M, N, c.ws_scan.as<unsigned long long>());
    unsigned long long chk[2] = {0, 0};
    ELBA_HIP(hipMemcpyAsync(chk, c.ws_scan.p, 16, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    ELBA_REQUIRE(chk[0] == 0, ELBA_ERR_INVALID_ARG, "triple index out of range");
    c.ws_a.reserve((size_t)(Z + 1) * 8); c.ws_b.reserve((size_t)(Z + 1) * 8);
    c.ws_c.reserve((size_t)(Z + 1) * 8); c.ws_d.reserve((size_t)(Z + 1) * 8);
    c.ws_e.reserve((size_t)(Z + 1) * 8); c.ws_f.reserve((size_t)(Z + 1) * 8);
    // k-mer id, read and position in ONE 64-bit word when they fit (28 + 18 + 14 bits on the 200 k-read set): one sort of 8-byte keys over
    // all their bits instead of three stable sorts of 16-byte (key, value) pairs — 7 passes of ~3 ms instead of 8 of ~7 ms there
    const int mb = bits_for((uint64_t)(M > 0 ? M - 1 : 0)), nb = bits_for((uint64_t)(N > 0 ? N - 1 : 0)), pb = bits_for(chk[1]);
    if (mb + nb + pb <= 64 && Z > 0 && !c.opt.csr_pairs) {
        uint64_t *w0 = c.ws_a.as<uint64_t>(), *w1 = c.ws_b.as<uint64_t>();
        hipLaunchKernelGGL(k_pack_triple_words, dim3(nbz), dim3(256), 0, s, d_rows, d_cols, d_vals, Z, mb, pb, w0);
        const int where = radix_sort_keys(s, w0, w1, Z, 0, mb + nb + pb, c.ws_sort);
        hipLaunchKernelGGL(k_unpack_triple_words, dim3(nbz), dim3(256), 0, s, (const uint64_t *)(where ? w1 : w0), Z, mb, pb, c.ws_e.as<uint64_t>(), c.ws_f.as<uint64_t>());
        c.A_has_kmers = false;
        finish_matrix_from_sorted_csc(c, M, N, Z, c.ws_e.as<uint64_t>(), 32, c.ws_f.as<uint64_t>());
        return;
    }
    if (Z > 0)
        hipLaunchKernelGGL(k_pack_triples, dim3(nbz), dim3(256), 0, s, d_rows, d_cols, d_vals, Z, M, N, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>(), c.ws_scan.as<unsigned long long>());
    ELBA_HIP(hipStreamSynchronize(s));
    set_kmer_matrix_from_pairs(c, M, N, Z, chk[1]);
}

void stage_export_triples_device(Ctx &c, int64_t *d_rows, int64_t *d_cols, uint32_t *d_vals)
{
    ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "export_triples_device: no k-mer matrix");
    ELBA_REQUIRE(c.Z == 0 || (d_rows && d_cols && d_vals), ELBA_ERR_INVALID_ARG, "null triple array");
    if (c.csr_inline && c.N > 0 && c.Z > 0) {      // (rows with inline partners do not name every entry's k-mer: the triples come from the columns)
        hipLaunchKernelGGL(k_export_triples_csc, dim3((unsigned)((c.N + 255) / 256)), dim3(256), 0, c.stream, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), (uint64_t)c.N, d_rows, d_cols, d_vals);
    } else if (c.M > 0 && c.Z > 0) {
        int nb = (int)std::min<int64_t>((c.M + 3) / 4, (int64_t)c.num_cus * 32);
        hipLaunchKernelGGL(k_export_triples, dim3(nb), dim3(256), 0, c.stream, (const uint32_t *)c.a_rowptr.as<uint32_t>(), (const uint64_t *)c.a_csr.as<uint64_t>(), (uint32_t)c.M,
                           c.csr_suffix ? 0xFFFFu : (c.csr_hints ? 0x3FFFFFFFu : 0xFFFFFFFFu), d_rows, d_cols, d_vals);
    }
    ELBA_HIP(hipStreamSynchronize(c.stream));
}

void stage_set_kmer_matrix(Ctx &c, int64_t M, int64_t N, int64_t Z, const int64_t *rows, const int64_t *cols, const uint32_t *vals)
{
    ELBA_REQUIRE(M >= 0 && N >= 0 && Z >= 0, ELBA_ERR_INVALID_ARG, "negative matrix dimension");
    ELBA_REQUIRE(M < 0xFFFFFFFFll && N < 0xFFFFFFFFll && Z < 0xFFFFFFFFll, ELBA_ERR_UNSUPPORTED, "matrix dimension beyond 32-bit device indices");
    ELBA_REQUIRE(Z == 0 || (rows && cols && vals), ELBA_ERR_INVALID_ARG, "null triple array");
    hipStream_t s = c.stream;
    std::vector<uint64_t> hk((size_t)Z + 1), hv((size_t)Z + 1);
    uint32_t maxpos = 0;
    for (int64_t z = 0; z < Z; ++z) {
        ELBA_REQUIRE(rows[z] >= 0 && rows[z] < M && cols[z] >= 0 && cols[z] < N, ELBA_ERR_INVALID_ARG, "triple index out of range");
        hk[(size_t)z] = vals[z];
        hv[(size_t)z] = ((uint64_t)cols[z] << 32) | (uint64_t)rows[z];
        if (vals[z] > maxpos) maxpos = vals[z];
    }
    c.ws_a.reserve((size_t)(Z + 1) * 8); c.ws_b.reserve((size_t)(Z + 1) * 8);
    c.ws_c.reserve((size_t)(Z + 1) * 8); c.ws_d.reserve((size_t)(Z + 1) * 8);
    c.ws_e.reserve((size_t)(Z + 1) * 8); c.ws_f.reserve((size_t)(Z + 1) * 8);
    if (Z > 0) {
        ELBA_HIP(hipMemcpyAsync(c.ws_a.p, hk.data(), (size_t)Z * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.ws_b.p, hv.data(), (size_t)Z * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipStreamSynchronize(s));      // (the host vectors go out of scope below)
    }
    set_kmer_matrix_from_pairs(c, M, N, Z, maxpos);
}

}  // namespace elba
