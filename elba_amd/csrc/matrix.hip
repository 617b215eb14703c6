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

__global__ void k_csc_to_csr_keys(const uint64_t *kid_keys, int kid_shift, uint64_t kid_mask, const uint32_t *colptr, const uint64_t *csc, int64_t Z, uint64_t *row_keys, uint64_t *csr_vals,
                                  bool hints, bool suffix, uint32_t win_lo, uint32_t win_hi, unsigned long long *prod_ctr)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = z < Z;
    if (!in) z = Z - 1;                                      // (whole wavefronts reach the reduction inside entry_hint)
    uint64_t kid = (kid_keys[z] >> kid_shift) & kid_mask;
    uint64_t e = csc[z];
    const uint32_t h = entry_hint(colptr, csc, kid, in ? (uint32_t)(e >> 32) : 0xFFFFFFFFu, hints && in, in ? win_lo : 0u, in ? win_hi : 0u, prod_ctr);
    if (!in) return;
    row_keys[z] = e >> 32;                                   // read
    if (suffix) {      // dense matrices (Ctx::csr_suffix): the entry knows its column's length and its own place in it — kid | L << 23 | idx << 16 | pos (16 bits)
        const uint32_t c0 = colptr[kid], L = colptr[kid + 1] - c0, idx = (uint32_t)z - c0;
        csr_vals[z] = (kid << 32) | ((uint64_t)L << 23) | ((uint64_t)idx << 16) | (e & 0xFFFFull);
    } else csr_vals[z] = (kid << 32) | ((uint64_t)h << 30) | (e & 0xFFFFFFFFull);         // kid | hint | pos
}

// CSR build with ONE word per entry when read, k-mer id, the two hint bits and the position fit 64 bits together:
// read << (nb + pb + 2) | kid << (pb + 2) | hint << pb | pos, sorted on the read bits only (stable: rows come out in (kid, pos) order)
__global__ void k_csc_to_csr_words(const uint64_t *kid_keys, int kid_shift, const uint32_t *colptr, const uint64_t *csc, int64_t Z, int nb, int pb, uint64_t *words,
                                   bool hints, uint32_t win_lo, uint32_t win_hi, unsigned long long *prod_ctr)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = z < Z;
    if (!in) z = Z - 1;
    const uint64_t kid = kid_keys[z] >> kid_shift, e = csc[z];
    const uint32_t h = entry_hint(colptr, csc, kid, in ? (uint32_t)(e >> 32) : 0xFFFFFFFFu, hints && in, in ? win_lo : 0u, in ? win_hi : 0u, prod_ctr);
    if (!in) return;
    words[z] = ((e >> 32) << (nb + pb + 2)) | (kid << (pb + 2)) | ((uint64_t)h << pb) | (e & 0xFFFFFFFFull);
}
// hint bits for sort keys that came without them (k_runs_emit, kmer.hip): the entry's k-mer id is in the word, its column in the CSC
// (one lane per entry; one lane per COLUMN, its entries in registers, was measured: 7.0 ms against 5.2)
__global__ void k_add_hints(const uint32_t *colptr, const uint64_t *csc, int64_t Z, int nb, int pb, uint64_t *words)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    const uint64_t w = words[z];
    const uint64_t kid = (w >> (pb + 2)) & ((1ull << nb) - 1);
    const uint32_t c0 = colptr[kid], L = colptr[kid + 1] - c0;
    if (L <= HINT_MAX_COL) words[z] = w | ((uint64_t)column_hint(csc + c0, L, (uint32_t)(csc[z] >> 32), 0u, 0xFFFFFFFFu) << pb);      // (the window is the whole matrix)
}
// sorted keys -> CSR entries, and the row pointers with them: entry z opens the rows (read of z - 1, read of z]; launched with Z + 1 lanes,
// the last of which closes the rows behind the last entry
__global__ void k_unpack_csr_words(const uint64_t *words, int64_t Z, int nb, int pb, uint64_t *csr, uint32_t *rowptr, int64_t M)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z > Z) return;
    const int rs = nb + pb + 2;
    const int64_t prev = z == 0 ? -1 : (int64_t)(words[z - 1] >> rs);
    const uint64_t w = z < Z ? words[z] : 0;
    const int64_t cur = z < Z ? (int64_t)(w >> rs) : M;
    for (int64_t k = prev + 1; k <= cur; ++k) rowptr[k] = (uint32_t)z;
    if (z < Z) csr[z] = (((w >> (pb + 2)) & ((1ull << nb) - 1)) << 32) | (((w >> pb) & 3ull) << 30) | (w & ((1ull << pb) - 1));
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
__global__ void k_fill_ell(const uint32_t *colptr, const uint64_t *csc, uint64_t nslots, uint32_t cs, uint64_t *ell)
{
    // two slots (16 bytes) per lane: the stride is even, so a pair never straddles two columns.  (Adding the hint bits of the entries here, by the lane
    // that copies them, was measured: 15.8 ms against 4.8 + 5.2 for this kernel and k_add_hints — most lanes hold padding.)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, npairs = nslots >> 1;
    for (uint64_t t2 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t2 < npairs; t2 += stride) {
        const uint64_t t = t2 << 1, kid = t / cs;
        const uint32_t j = (uint32_t)(t - kid * cs), c0 = colptr[kid], L = colptr[kid + 1] - c0;
        ulonglong2 v;
        v.x = j < L ? csc[c0 + j] : ~0ull;
        v.y = j + 1u < L ? csc[c0 + j + 1u] : ~0ull;
        reinterpret_cast<ulonglong2 *>(ell)[t2] = v;
    }
}

// 16-byte descriptors that carry the position of their row entry in the upper half of w (the packed form has no room for it)
__global__ __launch_bounds__(256) void k_fold_desc_pos(const RowHot *hdr, uint32_t M, HotDesc *hot, const uint64_t *dec, uint32_t fbits)
{
    for (uint32_t i = blockIdx.x; i < M; i += gridDim.x) {
        const RowHot h = hdr[i];
        for (uint32_t t = threadIdx.x; t < h.nd; t += blockDim.x) {
            const HotDesc d = hot[h.hs + t];
            hot[h.hs + t].w = (d.w & 0xFFFFu) | (uint32_t)dec[h.rs + (d.y >> fbits)] << 16;
        }
    }
}

// entries whose position is >= thr
__global__ void k_count_pos_ge(const uint64_t *v, int64_t n, uint32_t thr, unsigned long long *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long m = 0;
    for (; i < n; i += stride) m += (uint32_t)v[i] >= thr ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m += __shfl_xor(m, d, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, m);
}

// products per row: ub_i = sum over the row's entries of the length of the entry's column (one wavefront per row); seed-decoding array
// for the canonical column layout (replaced by k_dec_permuted when the columns are permuted)
__global__ __launch_bounds__(256) void k_row_products(const uint32_t *rowptr, const uint64_t *csr, const uint32_t *colptr, uint32_t M, uint32_t *rowprod, uint64_t *dec)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < M; i += nwaves) {
        const uint32_t rs = rowptr[i], re = rowptr[i + 1];
        unsigned long long ub = 0;
        for (uint32_t e = rs + lane; e < re; e += 64) {
            const uint32_t kid = (uint32_t)(csr[e] >> 32);
            const uint32_t c0 = colptr[kid], len = colptr[kid + 1] - c0;
            dec[e] = ((uint64_t)c0 << 32) | (uint32_t)csr[e];
            ub += len;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ub += __shfl_xor(ub, d, 64);
        if (lane == 0) rowprod[i] = ub > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ub;
    }
}

// ---- hot format of the SpGEMM: columns stored in FIRST-OCCURRENCE order --------------------------------------------------------
// k-mer ids are ranks of the k-mer VALUE, i.e. random with respect to the genome: a read's columns are scattered over the whole
// CSC and every 8-byte column gather costs a 64-byte sector (4.3x the algorithmic bytes measured, profiles/r01_notes.md).  Stored
// in the order of their first entry (read, pos), the columns a read shares with the reads before it — and all the columns it
// introduces itself — lie next to each other: 2.2x fewer sectors per row on 15 %-error reads, far fewer on accurate ones.
// Canonical order is untouched: a_csc / a_colptr / a_csr stay as they are (exports, seed decoding); only the arrays the hot
// loop walks are permuted: a_cscp (columns) and the row descriptors that point into it.
__global__ void k_first_entry_keys(const uint32_t *colptr, const uint64_t *csc, uint64_t N, uint64_t *keys, uint64_t *vals)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const uint32_t c0 = colptr[k], c1 = colptr[k + 1];
    keys[k] = c1 > c0 ? csc[c0] : ~0ull;            // empty columns (panels index by GLOBAL k-mer id) go last
    vals[k] = k;
}

__global__ void k_perm_counts(const uint64_t *sorted_cols, const uint32_t *colptr, uint64_t N, uint32_t *cnt)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const uint64_t k = sorted_cols[r];
    cnt[r] = colptr[k + 1] - colptr[k];
}

__global__ void k_perm_copy(const uint64_t *sorted_cols, const uint32_t *newstart_sorted, const uint32_t *colptr, const uint64_t *csc, uint64_t N,
                            uint32_t *newstart, uint64_t *cscp)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const uint64_t k = sorted_cols[r];
    const uint32_t c0 = colptr[k], c1 = colptr[k + 1], d0 = newstart_sorted[r];
    newstart[k] = d0;
    for (uint32_t a = c0; a < c1; ++a) cscp[d0 + (a - c0)] = csc[a];
}

// seed decoding by canonical rank: address of the entry's column in a_cscp << 32 | position in the read
__global__ void k_dec_permuted(const uint64_t *csr, const uint32_t *newstart, uint64_t Z, uint64_t *dec)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Z) return;
    dec[e] = ((uint64_t)newstart[(uint32_t)(csr[e] >> 32)] << 32) | (uint32_t)csr[e];
}

__global__ void k_roworder_keys(const RowHot *hdr, uint64_t M, uint64_t *keys, uint64_t *vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    keys[i] = (uint64_t)(0xFFFFFFFFu - hdr[i].work);      // ascending sort of the complement = descending products
    vals[i] = i;
}

// 16-byte descriptors -> 8-byte words (the dominant stream of the SpGEMM's numeric loop: half the bytes, half the cache lines)
__global__ void k_pack_desc(const HotDesc *in, uint64_t n, uint32_t xb, uint32_t yb, uint32_t zb, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const HotDesc d = in[i];
    out[i] = (uint64_t)d.x | (uint64_t)d.y << xb | (uint64_t)d.z << (xb + yb) | (uint64_t)d.w << (xb + yb + zb);
}

// partner read of every column entry (+ guard entries), the numeric loop's gather target; pb != 0: partner read << pb | position in it
__global__ void k_high_u32(const uint64_t *in, uint64_t n, uint64_t nguard, uint32_t pb, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t esc = (1u << pb) - 1u, pos = (uint32_t)in[i];          // a position that does not fit is written as all ones: "look it up" (spgemm_rows.hpp)
        out[i] = pb ? ((uint32_t)(in[i] >> 32) << pb) | (pos < esc ? pos : esc) : (uint32_t)(in[i] >> 32);
    }
    else if (i < n + nguard) out[i] = 0xFFFFFFFFu;
}

// packed descriptors that also carry the position of their row entry in its read (above w): one workgroup per row, which knows where
// the row's entries start in a_dec (rank of the entry = y >> fbits)
__global__ __launch_bounds__(256) void k_pack_desc_pos(const RowHot *hdr, uint32_t M, const HotDesc *in, const uint64_t *dec, uint32_t fbits, uint32_t xb, uint32_t yb, uint32_t zb, uint64_t *out)
{
    for (uint32_t i = blockIdx.x; i < M; i += gridDim.x) {
        const RowHot h = hdr[i];
        for (uint32_t t = threadIdx.x; t < h.nd; t += blockDim.x) {
            const HotDesc d = in[h.hs + t];
            const uint64_t qpos = (uint32_t)dec[h.rs + (d.y >> fbits)];
            out[h.hs + t] = (uint64_t)d.x | (uint64_t)d.y << xb | (uint64_t)d.z << (xb + yb) | (uint64_t)d.w << (xb + yb + zb) | qpos << (xb + yb + 2 * zb);
        }
    }
}

__global__ void k_narrow_u32(const uint64_t *in, uint64_t n, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}

// ---- row descriptors ------------------------------------------------------------------------------------------------------------
// What the SpGEMM walks is a per-ENTRY format of A, O(nnz(A)) like CSR itself: for the row entry (i, k, pos) the contiguous ranges of
// column k that hold its partners.  Columns are ordered by (read, pos), so with d0..d1 the run of read i itself in column k:
//     [0, d0)   partners j < i        [d0, d1)   read i itself (the diagonal of B)        [d1, len)   partners j > i.
// B is symmetric up to swapping each seed's two positions: the canonical seeds of (i,j) are the lexicographic min/max of
// (kid, pos in i, pos in j) over a CROSS product of positions per shared k-mer, so min/max of (kid, pos in j, pos in i) is the same
// pair of products with its positions exchanged — exactly, not approximately.  With `half` a pair of rows of this context's window
// [lo, hi) is therefore described on its smaller row only (range [d1, len)), partners below the window in full ([0, w0), w0 = entries
// with read < lo: their own rows live on another rank); the SpGEMM mirrors the surviving in-window pairs into the partner's row
// (spgemm.hip: k_mirror).  Nothing here is a product or a value of B: every partner entry is gathered, every pair accumulated and the
// diagonal counted inside the SpGEMM call.  Only the order of a row's descriptors is chosen for the kernel: by descending number of
// partners (the lanes of a wavefront then walk equally long ranges), then by address (neighbouring lanes gather neighbouring sectors).
template <bool FILL>
__global__ __launch_bounds__(256) void k_entry_ranges(const uint32_t *rowptr, const uint64_t *csr, const uint32_t *colptr, const uint32_t *newstart, const uint64_t *cscp,
                                                      uint32_t fbits, uint32_t lo, uint32_t hi, bool half, uint32_t cbits, uint32_t cmax,
                                                      uint32_t *cnt, RowHot *hdr, const uint32_t *dptr, HotDesc *desc, uint64_t *key_addr, uint64_t *key_row, uint64_t *val)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = lo + wave; i < hi; i += nwaves) {             // rows outside the window keep an empty header
        const uint32_t rs = rowptr[i], re = rowptr[i + 1];
        unsigned long long work = 0;
        for (uint32_t e = rs + lane; e < re; e += 64) {
            const uint32_t kid = (uint32_t)(csr[e] >> 32), pos = (uint32_t)csr[e];
            const uint32_t st = newstart[kid], len = colptr[kid + 1] - colptr[kid];
            uint32_t d0 = len, d1 = 0, w0 = 0, own = 0;
            for (uint32_t f = 0; f < len; ++f) {
                const uint64_t v = cscp[st + f];
                const uint32_t j = (uint32_t)(v >> 32);
                if (j < lo) ++w0;
                if (j == i) { d0 = f < d0 ? f : d0; d1 = f + 1; if ((uint32_t)v == pos) own = f; }
            }
            const uint32_t run = d1 - d0;                                  // >= 1: the entry itself
            const uint32_t cA = half ? w0 : d0, cB = len - d1;
            uint32_t nde = (cA ? 1u : 0u) + (cB ? 1u : 0u);
            if (nde == 0 && run >= 2) nde = 1;                              // no partners, but the diagonal count needs the run
            if (!FILL) {
                cnt[e] = nde;
                work += (unsigned long long)cA + cB;
                if (e == rs) hdr[i].own0 = own;
                if (e == re - 1) hdr[i].ownl = own;
            } else {
                uint32_t at = dptr[e], w = run - 1;
                const uint32_t rank = e - rs;
                auto put = [&](uint32_t f0, uint32_t c) {
                    desc[at] = HotDesc{st + f0, (rank << fbits) | f0, c, w};
                    key_addr[at] = (uint64_t)(st + f0);
                    key_row[at] = ((uint64_t)i << cbits) | (uint64_t)(cmax - (c < cmax ? c : cmax));
                    val[at] = at;
                    ++at; w = 0;
                };
                if (cA) put(0u, cA);
                if (cB) put(d1, cB);
                if (!cA && !cB && run >= 2) put(d0, 0u);
            }
        }
        if (!FILL) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) work += __shfl_xor(work, d, 64);
            if (lane == 0) { hdr[i].rs = rs; hdr[i].nnz = re - rs; hdr[i].work = work > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)work; }
        } else if (lane == 0) {
            hdr[i].hs = dptr[rs]; hdr[i].nd = dptr[re] - dptr[rs];
        }
    }
}

__global__ void k_gather_keys(const uint64_t *val, const uint64_t *key_by_index, uint64_t n, uint64_t *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = key_by_index[val[t]];
}

__global__ void k_permute_desc(const uint64_t *val, const HotDesc *in, uint64_t n, HotDesc *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[val[t]];
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

// Permuted columns (see the comment above k_first_entry_keys).  ELBA_NO_PERMUTE keeps the canonical layout (a_cscp == a_csc).
static void build_hot_format(Ctx &c)
{
    hipStream_t s = c.stream;
    const int64_t N = c.N, Z = c.Z;
    c.a_newstart.reserve((size_t)(N + 2) * 4);
    c.a_cscp_is_csc = Z == 0 || getenv("ELBA_NO_PERMUTE");
    if (c.a_cscp_is_csc) {          // k_row_products already wrote the canonical seed-decoding array
        if (N > 0) ELBA_HIP(hipMemcpyAsync(c.a_newstart.p, c.a_colptr.p, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
        return;
    }
    c.a_cscp.reserve((size_t)(Z + 8) * 8);       // + guard entries: the SpGEMM gathers up to four consecutive entries without a bounds check
    c.ws_a.reserve((size_t)(std::max(N, Z) + 1) * 8); c.ws_b.reserve((size_t)(std::max(N, Z) + 1) * 8);
    c.ws_c.reserve((size_t)(std::max(N, Z) + 1) * 8); c.ws_d.reserve((size_t)(std::max(N, Z) + 1) * 8);
    c.ws_e.reserve((size_t)(N + 2) * 4);
    uint64_t *k0 = c.ws_a.as<uint64_t>(), *v0 = c.ws_b.as<uint64_t>(), *k1 = c.ws_c.as<uint64_t>(), *v1 = c.ws_d.as<uint64_t>();
    const unsigned nbN = (unsigned)((N + 255) / 256);
    // columns by first entry (read << 32 | pos): LSD over the pos bits that can be set, then the read bits (an empty column's key is all
    // ones: it sorts with the largest values, and where it lands among them does not matter — it has no entries)
    hipLaunchKernelGGL(k_first_entry_keys, dim3(nbN), dim3(256), 0, s, c.a_colptr.as<uint32_t>(), c.a_csc.as<uint64_t>(), (uint64_t)N, k0, v0);
    uint32_t maxlen = 1;
    for (uint32_t l : c.h_len) maxlen = l > maxlen ? l : maxlen;
    const int posbits = c.have_reads && c.A_has_kmers && !c.h_len.empty() ? bits_for(maxlen) : 32;
    int w = radix_sort_pairs(s, k0, v0, k1, v1, N, 0, posbits, c.ws_sort);
    uint64_t *ck = w ? k1 : k0, *cv = w ? v1 : v0, *ok = w ? k0 : k1, *ov = w ? v0 : v1;
    int w2 = radix_sort_pairs(s, ck, cv, ok, ov, N, 32, 32 + bits_for((uint64_t)(c.M > 0 ? c.M : 1)), c.ws_sort);
    const uint64_t *sorted_cols = w2 ? ov : cv;
    uint32_t *cnt = c.ws_e.as<uint32_t>(), *newstart = c.a_newstart.as<uint32_t>();
    hipLaunchKernelGGL(k_perm_counts, dim3(nbN), dim3(256), 0, s, sorted_cols, c.a_colptr.as<uint32_t>(), (uint64_t)N, cnt);
    exclusive_scan_u32(s, cnt, cnt, N, c.ws_scan);
    hipLaunchKernelGGL(k_perm_copy, dim3(nbN), dim3(256), 0, s, sorted_cols, cnt, c.a_colptr.as<uint32_t>(), c.a_csc.as<uint64_t>(), (uint64_t)N, newstart, c.a_cscp.as<uint64_t>());
    ELBA_HIP(hipMemsetAsync(c.a_cscp.as<uint64_t>() + Z, 0xFF, 8 * 8, s));
    hipLaunchKernelGGL(k_dec_permuted, dim3((unsigned)((Z + 255) / 256)), dim3(256), 0, s, c.a_csr.as<uint64_t>(), newstart, (uint64_t)Z, c.a_dec.as<uint64_t>());
    ELBA_HIP(hipStreamSynchronize(s));
}

// Row headers + hot descriptors (see the comment above k_entry_ranges).  Needs a_newstart and the column copy (a_cscp or a_csc).
static void build_row_descriptors(Ctx &c)
{
    hipStream_t s = c.stream;
    const int64_t M = c.M, Z = c.Z;
    int fb = 1;
    while (fb < 31 && ((uint64_t)(c.max_col_nnz > 1 ? c.max_col_nnz - 1 : 1) >> fb)) ++fb;
    c.fbits = (uint32_t)fb;
    ELBA_REQUIRE(fb < 31 && (uint64_t)c.max_row_nnz < (1ull << (32 - fb)), ELBA_ERR_UNSUPPORTED,
                 "row nnz x column nnz exceeds the 32-bit product sequence number");
    c.a_hdr.reserve((size_t)(M + 1) * sizeof(RowHot));
    c.Pnd = 0; c.H = 0;
    c.half = !getenv("ELBA_NO_SYMMETRY");
    c.a_roworder.reserve((size_t)(M + 1) * 4);
    if (M == 0) return;
    const uint64_t *cols = c.a_cscp_is_csc ? c.a_csc.as<uint64_t>() : c.a_cscp.as<uint64_t>();
    const uint32_t cmax = (uint32_t)(c.max_col_nnz > 0 ? c.max_col_nnz : 1);
    const uint32_t cbits = (uint32_t)bits_for(cmax);
    // Field widths of the packed descriptor: x < Z + 8, y < max_row_nnz << fbits, z <= max_col_nnz, w < max_col_nnz.
    // Position-carrying formats (the SpGEMM's 64-bit accumulators, spgemm_rows.hpp): every position of every read of the matrix fits
    // pb <= 16 bits, a partner read and a position fit one 32-bit word (or nearly: see below), and the descriptor has pb bits to spare.  The largest position is
    // taken from the entries themselves (a multi-GPU shard does not know the lengths of the other ranks' reads, a matrix handed over as
    // triples comes without reads).
    const uint32_t hxb = (uint32_t)bits_for((uint64_t)Z + 8), hyb = (uint32_t)bits_for((uint64_t)(c.max_row_nnz > 0 ? c.max_row_nnz : 1)) + c.fbits, hzb = (uint32_t)bits_for((uint64_t)cmax + 1);
    const bool can_pack = hxb + hyb + 2 * hzb <= 64 && hxb <= 32 && hyb <= 32 && !getenv("ELBA_DESC16");
    c.pay_pb = 0;
    uint32_t pay_qbits = 16;                         // bits of the largest position (the descriptor's field)
    // (not for dense data — long columns, mean length >= 8: accurate reads with a high UPPER — where the 32-bit kernels combine runs of
    //  equal partners across lanes before the table, Table::insert_runs: measured 34.5 vs 51 ms on the dense-repeats set)
    uint64_t maxpos = 0;
    int64_t nbz = (Z + 255) / 256;
    if (nbz > 2048) nbz = 2048;
    if (Z > 0) {
        c.ws_scan.reserve(64);
        ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 8, s));
        hipLaunchKernelGGL(k_max_low32, dim3((unsigned)nbz), dim3(256), 0, s, cols, Z, c.ws_scan.as<unsigned long long>());
        ELBA_HIP(hipMemcpyAsync(&maxpos, c.ws_scan.p, 8, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
    }
    c.pos16 = maxpos < 65536;                        // every position fits 16 bits: the mirror pass writes 16-byte records (spgemm.hip)
    if (Z > 0 && Z < 8 * c.N && !getenv("ELBA_NO_PAY")) {
        const uint32_t pb = (uint32_t)bits_for(maxpos), mb = (uint32_t)bits_for((uint64_t)(M > 0 ? M - 1 : 0));
        pay_qbits = pb <= 16 ? pb : 16;
        if (pb <= 16) {
            if (mb + pb <= 32) c.pay_pb = pb;
            else if (mb < 32 && 32 - mb + 2 >= pb) {
                // One or two bits short (many reads, a few very long ones): the word keeps 32 - mb position bits and the entries beyond
                // them are marked "look it up" — worth it while they are rare (here: under 1 in 16: a marked position costs its seed the
                // two look-ups every seed used to pay)
                const uint32_t wb = 32 - mb;
                ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 8, s));
                hipLaunchKernelGGL(k_count_pos_ge, dim3((unsigned)nbz), dim3(256), 0, s, cols, Z, (1u << wb) - 1u, c.ws_scan.as<unsigned long long>());
                uint64_t nesc = 0;
                ELBA_HIP(hipMemcpyAsync(&nesc, c.ws_scan.p, 8, hipMemcpyDeviceToHost, s));
                ELBA_HIP(hipStreamSynchronize(s));
                if (nesc * 16 <= (uint64_t)Z) c.pay_pb = wb;
            }
        }
    }
    if (getenv("ELBA_TRACE")) fprintf(stderr, "[elba] hot format: packed descriptors %d (x %u y %u z %u bits), position-carrying words: %u position bits\n", (int)can_pack, hxb, hyb, hzb, c.pay_pb);
    c.a_cscj.reserve((size_t)(Z + 8) * 4);
    hipLaunchKernelGGL(k_high_u32, dim3((unsigned)((Z + 8 + 255) / 256)), dim3(256), 0, s, cols, (uint64_t)Z, (uint64_t)8, c.pay_pb, c.a_cscj.as<uint32_t>());
    const uint32_t lo = (uint32_t)c.row_lo, hi = (uint32_t)(c.row_hi < 0 ? M : c.row_hi);
    int nb = (int)((hi - lo + 3) / 4);
    if (nb > c.num_cus * 8) nb = c.num_cus * 8;
    if (nb < 1) nb = 1;
    DevBuf cntbuf, dptrbuf;         // u32[Z+1] descriptors per entry and their exclusive scan (released when the format is built)
    cntbuf.reserve((size_t)(Z + 2) * 4);
    uint32_t *cnt = cntbuf.as<uint32_t>();
    ELBA_HIP(hipMemsetAsync(cnt, 0, (size_t)(Z + 2) * 4, s));
    ELBA_HIP(hipMemsetAsync(c.a_hdr.p, 0, (size_t)(M + 1) * sizeof(RowHot), s));
    hipLaunchKernelGGL((k_entry_ranges<false>), dim3(nb), dim3(256), 0, s, c.a_rowptr.as<uint32_t>(), c.a_csr.as<uint64_t>(), c.a_colptr.as<uint32_t>(), c.a_newstart.as<uint32_t>(), cols,
                       c.fbits, lo, hi, c.half, cbits, cmax, cnt, c.a_hdr.as<RowHot>(), (const uint32_t *)nullptr, (HotDesc *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr);
    // descriptors per entry -> offsets (Z + 1 of them); the total must fit 32-bit addressing
    c.ws_e.reserve((size_t)(Z + 2) * 8);
    int64_t *dptr64 = c.ws_e.as<int64_t>();
    exclusive_scan_u32_to_i64(s, cnt, dptr64, Z + 1, c.ws_scan);
    int64_t H = 0;
    ELBA_HIP(hipMemcpyAsync(&H, dptr64 + Z, 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    ELBA_REQUIRE(H < 0xFFFFFFF0ll, ELBA_ERR_UNSUPPORTED, "row descriptors beyond 32-bit device indices");
    c.H = H;
    dptrbuf.reserve((size_t)(Z + 2) * 4);
    uint32_t *dptr = dptrbuf.as<uint32_t>();
    hipLaunchKernelGGL(k_narrow_u32, dim3((unsigned)((Z + 1 + 255) / 256)), dim3(256), 0, s, (const uint64_t *)dptr64, (uint64_t)(Z + 1), dptr);
    c.a_hot.reserve((size_t)(H + 1) * sizeof(HotDesc));
    DevBuf raw;                     // descriptors in entry order, before the per-row sort
    raw.reserve((size_t)(H + 1) * sizeof(HotDesc));
    c.ws_a.reserve((size_t)(std::max(H, M) + 1) * 8); c.ws_b.reserve((size_t)(std::max(H, M) + 1) * 8); c.ws_c.reserve((size_t)(std::max(H, M) + 1) * 8); c.ws_d.reserve((size_t)(std::max(H, M) + 1) * 8);
    c.ws_f.reserve((size_t)(H + 1) * 8);
    uint64_t *k0 = c.ws_a.as<uint64_t>(), *v0 = c.ws_b.as<uint64_t>(), *k1 = c.ws_c.as<uint64_t>(), *v1 = c.ws_d.as<uint64_t>();
    uint64_t *key_row = c.ws_f.as<uint64_t>();
    hipLaunchKernelGGL((k_entry_ranges<true>), dim3(nb), dim3(256), 0, s, c.a_rowptr.as<uint32_t>(), c.a_csr.as<uint64_t>(), c.a_colptr.as<uint32_t>(), c.a_newstart.as<uint32_t>(), cols,
                       c.fbits, lo, hi, c.half, cbits, cmax, (uint32_t *)nullptr, c.a_hdr.as<RowHot>(), dptr, raw.as<HotDesc>(), k0, key_row, v0);
    if (H > 0) {
        // stable LSD: by address, then by (row, descending partner count)
        int w = radix_sort_pairs(s, k0, v0, k1, v1, H, 0, bits_for((uint64_t)Z + 8), c.ws_sort);
        uint64_t *cv = w ? v1 : v0, *ck = w ? k1 : k0, *ok = w ? k0 : k1, *ov = w ? v0 : v1;
        const unsigned nbH = (unsigned)((H + 255) / 256);
        hipLaunchKernelGGL(k_gather_keys, dim3(nbH), dim3(256), 0, s, cv, key_row, (uint64_t)H, ck);
        int w2 = radix_sort_pairs(s, ck, cv, ok, ov, H, 0, (int)cbits + bits_for((uint64_t)(M > 0 ? M - 1 : 0)), c.ws_sort);
        hipLaunchKernelGGL(k_permute_desc, dim3(nbH), dim3(256), 0, s, w2 ? ov : cv, raw.as<HotDesc>(), (uint64_t)H, c.a_hot.as<HotDesc>());
    }
    // rows by descending work: the SpGEMM queues them in this order so that a workgroup's static share of a tier
    // mixes heavy rows first and light rows last (longest-processing-time order: short tail)
    const unsigned nbM = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(k_roworder_keys, dim3(nbM), dim3(256), 0, s, c.a_hdr.as<RowHot>(), (uint64_t)M, k0, v0);
    int wr = radix_sort_pairs(s, k0, v0, k1, v1, M, 0, 32, c.ws_sort);
    hipLaunchKernelGGL(k_narrow_u32, dim3(nbM), dim3(256), 0, s, wr ? v1 : v0, (uint64_t)M, c.a_roworder.as<uint32_t>());
    c.hot_xb = 0;
    if (c.pay_pb && !(can_pack && hxb + hyb + 2 * hzb + pay_qbits <= 64)) {
        // position-carrying formats whose descriptors cannot spare the bits in 8 bytes: the 16-byte form, position folded into w
        if (H > 0) hipLaunchKernelGGL(k_fold_desc_pos, dim3((unsigned)std::min<int64_t>(M, (int64_t)c.num_cus * 16)), dim3(256), 0, s, c.a_hdr.as<RowHot>(), (uint32_t)M, c.a_hot.as<HotDesc>(),
                                      c.a_dec.as<uint64_t>(), c.fbits);
    } else if (can_pack) {
        c.a_hot8.reserve((size_t)(H + 1) * 8);
        if (H > 0 && c.pay_pb)
            hipLaunchKernelGGL(k_pack_desc_pos, dim3((unsigned)std::min<int64_t>(M, (int64_t)c.num_cus * 16)), dim3(256), 0, s, c.a_hdr.as<RowHot>(), (uint32_t)M, c.a_hot.as<HotDesc>(),
                               c.a_dec.as<uint64_t>(), c.fbits, hxb, hyb, hzb, c.a_hot8.as<uint64_t>());
        else if (H > 0) hipLaunchKernelGGL(k_pack_desc, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, s, c.a_hot.as<HotDesc>(), (uint64_t)H, hxb, hyb, hzb, c.a_hot8.as<uint64_t>());
        c.hot_xb = hxb; c.hot_yb = hyb; c.hot_zb = hzb;
    }
    // products the descriptors stand for (statistics only)
    std::vector<RowHot> hh((size_t)M);
    ELBA_HIP(hipMemcpyAsync(hh.data(), c.a_hdr.p, (size_t)M * sizeof(RowHot), hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    for (int64_t i = 0; i < M; ++i) c.Pnd += hh[(size_t)i].work;
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
    c.plan = getenv("ELBA_PLAN") != nullptr;
    // stable sort by read: rows come out ordered by (kid, pos)
    c.ws_a.reserve((size_t)(Z + 1) * 8); c.ws_c.reserve((size_t)(Z + 1) * 8);
    const int mb = bits_for((uint64_t)(M > 0 ? M - 1 : 0)), nb = bits_for((uint64_t)(N > 0 ? N - 1 : 0)), pb = bits_for(maxpos);
    bool hints = pre ? c.pre_hints : (pb <= 30 && !c.plan && !getenv("ELBA_NO_HINTS"));
    const uint32_t wlo = (uint32_t)win_lo, whi = (uint32_t)(win_hi < 0 ? M : win_hi);
    c.prod_ctr.reserve(64 * 128);
    unsigned long long *prod_ctr = c.prod_ctr.as<unsigned long long>();
    if (!pre) ELBA_HIP(hipMemsetAsync(prod_ctr, 0, 64 * 128, s));      // (pre: k_runs_emit has counted)
    // The column store the SpGEMM gathers from.  No column longer than 64 entries (UPPER <= 64: every configuration the reference
    // documents): columns padded to a common stride S (4 entries, or whole 64-byte lines), column kid at kid * S — its address is arithmetic, and a group of S/2
    // lanes reads it as one aligned segment.  Longer columns (or no room for the padding): the plain CSC, reached through a_colptr.
    c.max_col_nnz = max_segment_len(c, c.a_colptr.as<uint32_t>(), N);
    {
        const uint32_t mc = (uint32_t)(c.max_col_nnz > 0 ? c.max_col_nnz : 1);
        // stride: 4 entries, else the longest column rounded up to whole 64-byte lines (8 entries) — a power of two is not needed: the kernel
        // multiplies (U = 35: 40 entries = 5 lines per gather where 64 entries would fetch 8)
        const uint32_t stride = mc <= 4 ? 4u : (mc + 7u) & ~7u;
        size_t free_b = 0, total_b = 0;
        ELBA_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t ell_bytes = (size_t)N * stride * 8;
        c.use_ell = mc <= 64 && N > 0 && !getenv("ELBA_NO_ELL") && ell_bytes <= (free_b + c.a_ell.cap) / 3 && (uint64_t)N * stride < (1ull << 40);
        if (c.use_ell) {
            uint32_t lb = 1, fb = 2;
            while ((2u << lb) < stride) ++lb;              // lanes per row entry: 16 bytes (two entries) each
            while ((1u << fb) < stride) ++fb;
            c.s_stride = stride; c.lpc_log2 = lb; c.fbits = fb;
            c.a_ell.reserve(ell_bytes + 64);
            ELBA_HIP(hipMemsetAsync(c.a_ell.as<char>() + ell_bytes, 0xFF, 64, s));      // (guard words behind the last column)
            const uint64_t nslots = (uint64_t)N * stride;
            hipLaunchKernelGGL(k_fill_ell, dim3((unsigned)std::min<uint64_t>((nslots / 2 + 255) / 256, 1ull << 30)), dim3(256), 0, s, c.a_colptr.as<uint32_t>(), c.a_csc.as<uint64_t>(), nslots, stride, c.a_ell.as<uint64_t>());
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
    // Dense matrices (columns of more than 16 reads, e.g. UPPER = 35 on 40x low-error reads: every row owns pairs of every column, hundreds of
    // products per surviving pair): pairs are owned by their SMALLER row and an entry's owned candidates are the column's entries behind its
    // own — the row entries then carry the column's length and their own place in it, and the SpGEMM hands out exactly those candidates to
    // its lanes (spgemm_direct.hpp, "suffix" path).  One GPU / whole-matrix window, positions below 2^16, the padded column store.
    c.csr_suffix = c.use_ell && c.pos16 && c.max_col_nnz > 16 && !c.plan && win_lo == 0 && (win_hi < 0 || win_hi == M) && !getenv("ELBA_NO_PAY") && !getenv("ELBA_NO_SUFFIX");
    if (c.csr_suffix) hints = false;
    if (!c.csr_suffix && mb + nb + pb + 2 <= 64 && !getenv("ELBA_CSR_PAIRS")) {
        const bool have_words = pre && c.pre_words && c.pre_nb == nb && c.pre_pb == pb;
        uint64_t *w0 = have_words ? c.csr_words.as<uint64_t>() : c.ws_a.as<uint64_t>(), *w1 = c.ws_c.as<uint64_t>();
        ELBA_REQUIRE(!pre || have_words, ELBA_ERR_INTERNAL, "create_kmer_matrix: the sort keys of the k-mer stage do not match the matrix");
        if (Z > 0 && !have_words)
            hipLaunchKernelGGL(k_csc_to_csr_words, dim3((unsigned)((Z + 255) / 256)), dim3(256), 0, s, kid_keys, kid_shift, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), Z, nb, pb, w0,
                               hints, wlo, whi, prod_ctr);
        if (Z > 0 && have_words && hints)
            hipLaunchKernelGGL(k_add_hints, dim3((unsigned)((Z + 255) / 256)), dim3(256), 0, s, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), Z, nb, pb, w0);
        const int where = radix_sort_keys(s, w0, w1, Z, nb + pb + 2, nb + pb + 2 + mb, c.ws_sort);
        const uint64_t *sorted = where ? w1 : w0;
        hipLaunchKernelGGL(k_unpack_csr_words, dim3((unsigned)((Z + 1 + 255) / 256)), dim3(256), 0, s, sorted, Z, nb, pb, c.a_csr.as<uint64_t>(), c.a_rowptr.as<uint32_t>(), M);
    } else {
        ELBA_REQUIRE(!pre || !c.pre_words || c.csr_suffix, ELBA_ERR_INTERNAL, "create_kmer_matrix: the sort keys of the k-mer stage do not match the matrix");
        const bool kid_in_words = pre && c.pre_words;        // (k_runs_emit left sort keys, not column ids: the k-mer id is a field of the word)
        const uint64_t *kk = kid_in_words ? c.csr_words.as<uint64_t>() : kid_keys;
        const int ks = kid_in_words ? c.pre_pb + 2 : kid_shift;
        const uint64_t km = kid_in_words ? (1ull << c.pre_nb) - 1 : ~0ull;
        c.ws_b.reserve((size_t)(Z + 1) * 8); c.ws_d.reserve((size_t)(Z + 1) * 8);
        uint64_t *k0 = c.ws_a.as<uint64_t>(), *v0 = c.ws_b.as<uint64_t>(), *k1 = c.ws_c.as<uint64_t>(), *v1 = c.ws_d.as<uint64_t>();
        if (Z > 0) {
            int64_t nbk = (Z + 255) / 256;
            if (pre) ELBA_HIP(hipMemsetAsync(prod_ctr, 0, 64 * 128, s));
            hipLaunchKernelGGL(k_csc_to_csr_keys, dim3((unsigned)nbk), dim3(256), 0, s, kk, ks, km, (const uint32_t *)c.a_colptr.as<uint32_t>(), (const uint64_t *)c.a_csc.as<uint64_t>(), Z, k0, v0,
                               hints, c.csr_suffix, wlo, whi, prod_ctr);
        }
        int where = radix_sort_pairs(s, k0, v0, k1, v1, Z, 0, mb, c.ws_sort);
        const uint64_t *rk = where ? k1 : k0, *rv = where ? v1 : v0;
        group_offsets_u32(s, rk, 0, Z, c.a_rowptr.as<uint32_t>(), M);
        if (Z > 0) ELBA_HIP(hipMemcpyAsync(c.a_csr.p, rv, (size_t)Z * 8, hipMemcpyDeviceToDevice, s));
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
    if (c.plan) {
        c.a_rowprod.reserve((size_t)(M + 1) * 4);
        c.a_dec.reserve((size_t)(Z + 1) * 8);
        if (M > 0) {
            int nb = (int)((M + 3) / 4);
            if (nb > c.num_cus * 8) nb = c.num_cus * 8;
            hipLaunchKernelGGL(k_row_products, dim3(nb), dim3(256), 0, s, c.a_rowptr.as<uint32_t>(), c.a_csr.as<uint64_t>(), c.a_colptr.as<uint32_t>(), (uint32_t)M, c.a_rowprod.as<uint32_t>(), c.a_dec.as<uint64_t>());
        }
    }
    // a new matrix: the tier queues and the tier / sort usage of the previous one are forgotten.  The OUTPUT capacity is kept as a guess (the
    // buffers exist): the first SpGEMM call on this matrix then runs without a host round trip in its middle and checks afterwards that
    // everything fitted (spgemm.hip repeats the call on the synchronising path otherwise)
    c.ov_tiers_known = false; c.ov_class_valid = false; c.ov_sort_used[0] = c.ov_sort_used[1] = true;
    c.ov_prior_q16 = 0;            // a new matrix: forget the partner/product ratio measured on the previous one
    c.max_row_nnz = max_segment_len(c, c.a_rowptr.as<uint32_t>(), M);
    {
        ELBA_REQUIRE(c.fbits < 31 && (uint64_t)c.max_row_nnz < (1ull << (32 - c.fbits)), ELBA_ERR_UNSUPPORTED,
                     "row nnz x column nnz exceeds the 32-bit product sequence number");
    }
    if (c.plan) {
        build_hot_format(c);
        build_row_descriptors(c);
    }
    c.have_A = true;
    c.have_B = false;
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
    uint64_t *k0 = c.ws_a.as<uint64_t>(), *v0 = c.ws_b.as<uint64_t>(), *k1 = c.ws_c.as<uint64_t>(), *v1 = c.ws_d.as<uint64_t>();
    if (Z > 0) {
        ELBA_HIP(hipMemcpyAsync(k0, hk.data(), (size_t)Z * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(v0, hv.data(), (size_t)Z * 8, hipMemcpyHostToDevice, s));
    }
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

}  // namespace elba
