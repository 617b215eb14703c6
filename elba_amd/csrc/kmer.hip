// kmer.hip — k-mer front end on the GPU: packed 2-bit reads -> reliable canonical k-mers -> A (CSC + CSR).
//
// Replaces get_kmer_count_map_keys / get_kmer_count_map_values / create_kmer_matrix (src/KmerOps.cpp:18-401) and the
// explicit transpose (src/main.cpp:272-273).  The reference's Bloom filter + HyperLogLog + two all-to-all passes have
// one net effect when LOWER >= 2: a canonical k-mer is kept iff its total instance count c obeys
// LOWER <= c <= UPPER, and every instance of a kept k-mer becomes one entry (read, pos).  That is computed here exactly, by SORTING:
// every pass is a coalesced stream over HBM (no random atomics):
//   k_kmer_emit_packed_hist   every k-mer instance -> ONE word, canonical value << pb | (instance index >> drop), written in instance order =
//                             (read, pos) order, a whole tile of the sort per workgroup together with the tile's first-digit counts
//                             ((value, read << 32 | pos) pairs — k_kmer_emit — when more than 3 index bits would have to be dropped)
//   radix sort (prims.hip)    stable LSD sort on the 2k value bits: equal k-mers become one run, its entries still in (read, pos) order
//   k_runs<false>             run lengths = the exact counts; per block: reliable runs (LOWER <= count <= UPPER), their entries, all runs
//   k_runs_emit               reliable runs numbered in value order (k-mer id = rank of the value); every item of a reliable
//                             run writes its entry (read, pos) — that IS the CSC of A, columns already sorted — and the entry's one-word
//                             sort key for the CSR build.  (UPPER > 62 or pairs: k_runs<true> + k_instance_entries.)
//   Multi-word k-mers (k > 31): k_kmer_emit2/3, an index permutation sorted last word first, runs compare every word.
//
// Packed k-mer layout (src/Kmer.cpp:67-87): base i at bits 2*(31-i), low 64-2k bits zero; canonical = min(kmer, twin)
// (src/Kmer.cpp:200-205); position = forward start index (include/KmerOps.hpp:91-103).
#include "common.hpp"
#include "matrix.hpp"

namespace elba {

namespace {

#include "kmer_enum.hpp"

__device__ __forceinline__ uint64_t mix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

// one lane per column: sort its (<= UPPER) entries ascending as u64 == by (read, pos); also emits the column id of every entry
// (the distributed owner: its records arrive in no particular order)
__global__ void k_sort_columns(const uint32_t *colptr, uint64_t *csc, uint64_t *kid_keys, uint64_t N)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const uint32_t c0 = colptr[k], c1 = colptr[k + 1];
    for (uint32_t a = c0 + 1; a < c1; ++a) {
        const uint64_t v = csc[a];
        uint32_t b = a;
        while (b > c0 && csc[b - 1] > v) { csc[b] = csc[b - 1]; --b; }
        csc[b] = v;
    }
    for (uint32_t a = c0; a < c1; ++a) kid_keys[a] = k;
}

// ---- sort-based counting ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EN_THREADS) void k_kmer_emit(EnumParams e, uint64_t *keys, uint64_t *vals)
{
    for_each_instance(e, [&](uint64_t g, uint32_t r, uint32_t p, uint64_t km) { keys[g] = km; vals[g] = ((uint64_t)r << 32) | p; });
}

// One word per instance: canonical value (right-aligned) << ib | instance index.  The index is the payload: (read, pos) follow from it
// through the instance offsets, and they are needed for the few instances that survive the count filter only — the sort moves
// 8 bytes per instance instead of 16.
__global__ __launch_bounds__(EN_THREADS) void k_kmer_emit_packed(EnumParams e, int pb, int drop, uint64_t *words)
{
    const int k2 = 2 * e.k;
    for_each_instance(e, [&](uint64_t g, uint32_t, uint32_t, uint64_t km) { words[g] = ((km >> (64 - k2)) << pb) | (g >> drop); });
}

// The same, and the first radix pass's histogram with it: a workgroup writes one tile of the sort (SUB * EN_PER_BLOCK words) and the row of
// digit counts the sort expects for it (radix_first_histogram, prims.hip) — the sort then starts with its scatter.
template <int SUB>
__global__ __launch_bounds__(EN_THREADS) void k_kmer_emit_packed_hist(EnumParams e, const BlockInfo *block_read, int pb, int drop, uint64_t *words, int shift, int bits, uint32_t *hist)
{
    static_assert(SUB * EN_PER_WAVE == (1 << IB_SHIFT), "a wavefront's share of the tile is one block of the instance -> read table");
    __shared__ uint32_t h[512];
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    for (uint32_t i = threadIdx.x; i < nbins; i += EN_THREADS) h[i] = 0;
    __syncthreads();
    const int k2 = 2 * e.k;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t base = ((uint64_t)blockIdx.x * (EN_THREADS / 64) + (threadIdx.x >> 6)) * (uint64_t)(SUB * EN_PER_WAVE);
    if (base < e.I) {
        // the read of the wavefront's first instance: ONE table load (a binary search over the reads is 18 dependent loads, and was repeated for
        // every 512 instances); every lane then walks on by itself, the read's bounds in registers
        const ReadCursor rc = cursor_at(e, block_read, base);
        uint32_t r = rc.lo;
        uint64_t off_lo = rc.off_lo, off_hi = rc.off_hi, boff = rc.boff;
#pragma unroll 8
        for (int it = 0; it < SUB * EN_ITEMS; ++it) {
            const uint64_t g = base + (uint64_t)it * 64 + lane;
            if (g >= e.I) break;
            while (g >= off_hi) { ++r; off_lo = off_hi; off_hi = e.inst_off[r + 1]; boff = e.byte_off[r]; }
            const uint64_t km = canonical_at_off(e, boff, (uint32_t)(g - off_lo));
            const uint64_t wd = ((km >> (64 - k2)) << pb) | (g >> drop);
            words[g] = wd;
            atomicAdd(&h[(uint32_t)(wd >> shift) & dmask], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += EN_THREADS) hist[(size_t)blockIdx.x * nbins + i] = h[i];
}

__global__ __launch_bounds__(EN_THREADS) void k_kmer_emit2(EnumParams e, uint64_t *khi, uint64_t *klo, uint64_t *vals, uint64_t *idx)
{
    for_each_position(e, [&](uint64_t g, uint32_t r, uint32_t p) {
        uint64_t hi, lo;
        canonical2_at(e, r, p, hi, lo);
        khi[g] = hi; klo[g] = lo; vals[g] = ((uint64_t)r << 32) | p; idx[g] = g;
    });
}

__global__ __launch_bounds__(EN_THREADS) void k_kmer_emit3(EnumParams e, uint64_t *k0, uint64_t *k1, uint64_t *k2, uint64_t *vals, uint64_t *idx)
{
    for_each_position(e, [&](uint64_t g, uint32_t r, uint32_t p) {
        uint64_t a, b, c;
        canonical3_at(e, r, p, a, b, c);
        k0[g] = a; k1[g] = b; k2[g] = c; vals[g] = ((uint64_t)r << 32) | p; idx[g] = g;
    });
}

__global__ void k_gather_u64(const uint64_t *idx, const uint64_t *in, uint64_t n, uint64_t *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[idx[t]];
}

// flag[g] = 1 where a run of equal k-mers starts (flag[I] = 1 closes the last run); keys_lo: second word of two-word k-mers, or null
// (ib > 0: packed words, the k-mer value sits above the ib index bits)
__global__ void k_run_flags(const uint64_t *keys, const uint64_t *keys_lo, const uint64_t *keys_lo2, uint64_t I, uint32_t *flag, int ib)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > I) return;
    flag[g] = (g == 0 || g == I || (keys[g] >> ib) != (keys[g - 1] >> ib) || (keys_lo && keys_lo[g] != keys_lo[g - 1]) || (keys_lo2 && keys_lo2[g] != keys_lo2[g - 1])) ? 1u : 0u;
}

// ---- runs -> reliable columns, fused -------------------------------------------------------------------------------------------------
// After the sort equal k-mers are adjacent.  A lane owns RUN_ITEMS consecutive items; an item whose left neighbour holds another k-mer
// heads a run, and the head measures its run by walking right (at most UPPER + 1 steps: longer runs are unreliable whatever their
// length) — no flag array, no run ids, no head positions, no scans over the instance stream.  Two passes with the same walk:
//   k_runs<false>  per block: reliable runs, their entries, all runs (= distinct k-mers)            -> three u32 per 2048 items
//   (exclusive scans of those per-block triples: a few MB)
//   k_runs<true>   block-local scan on top of the block's offsets: every reliable head knows its k-mer id (rank of the value) and its
//                  column pointer and writes the k-mer, the count, the pointer, and for each entry its payload and its k-mer id.
constexpr int RUN_THREADS = 256, RUN_ITEMS = 8, RUN_TILE = RUN_THREADS * RUN_ITEMS;
struct RunParams {
    const uint64_t *keys, *lo, *lo2, *vals;      // sorted words (value above `ib` payload bits) or (k-mer words, payload) arrays
    uint64_t I;
    int ib, k2;
    uint32_t lower, upper;
};
__device__ __forceinline__ bool same_kmer(const RunParams &p, uint64_t a, uint64_t b)
{
    return (p.keys[a] >> p.ib) == (p.keys[b] >> p.ib) && (!p.lo || p.lo[a] == p.lo[b]) && (!p.lo2 || p.lo2[a] == p.lo2[b]);
}
template <bool EMIT>
__global__ __launch_bounds__(RUN_THREADS) void k_runs(RunParams p, uint32_t *blk_rel, uint32_t *blk_ent, uint32_t *blk_heads, const uint32_t *off_rel, const uint32_t *off_ent,
                                                      uint64_t *rel_kmers, uint64_t *rel_kmers_lo, uint64_t *rel_kmers_lo2, uint32_t *rel_counts, uint32_t *colptr,
                                                      uint64_t *payload, uint64_t *kid_of_entry)
{
    // Item q of the tile (q = i * 256 + tid: consecutive lanes, consecutive items — coalesced) is compared with its left neighbour once;
    // the answers live in LDS as one bit per item (+ 64 items of halo behind the tile), and a head reads its run length off the bits:
    // the number of consecutive "same as my left neighbour" bits that follow it.  Only a run that outgrows the halo walks global memory.
    constexpr int NW = RUN_THREADS / 64, SLICES = RUN_ITEMS * NW;
    __shared__ uint64_t eqmask[SLICES + 2];
    __shared__ uint32_t srel[SLICES + 1], sent[SLICES + 1], shead[NW];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t lt = (1ull << lane) - 1;
    const uint64_t T0 = (uint64_t)blockIdx.x * RUN_TILE;
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint64_t g = T0 + (uint64_t)i * RUN_THREADS + tid;
        const uint64_t bal = __ballot(g < p.I && g > 0 && same_kmer(p, g, g - 1));
        if (lane == 0) eqmask[i * NW + w] = bal;
    }
    if (w == 0) {
        const uint64_t g = T0 + RUN_TILE + lane;
        const uint64_t bal = __ballot(g < p.I && same_kmer(p, g, g - 1));
        if (lane == 0) { eqmask[SLICES] = bal; eqmask[SLICES + 1] = 0; }
    }
    __syncthreads();
    uint32_t len[RUN_ITEMS];                      // 0: not a head; else run length, capped at upper + 1
    uint32_t nheads = 0;
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint32_t q = (uint32_t)i * RUN_THREADS + tid;
        const uint64_t g = T0 + q;
        len[i] = 0;
        const bool head = g < p.I && !((eqmask[q >> 6] >> (q & 63u)) & 1ull);
        uint32_t l = 0;
        if (head) {
            const uint32_t q1 = q + 1u, s1 = q1 & 63u;
            const uint64_t w0 = eqmask[q1 >> 6], w1 = eqmask[(q1 >> 6) + 1];
            const uint64_t win = s1 ? (w0 >> s1) | (w1 << (64u - s1)) : w0;      // the 64 bits that follow the head
            const uint32_t ones = win == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~win);
            l = 1u + ones;
            // the window ends at the halo's end: a run that fills it (or the part of it the bits cover) goes on in global memory
            const uint32_t covered = (uint32_t)RUN_TILE + 64u - q1;               // bits behind the head that the masks hold
            if (ones >= (covered < 64u ? covered : 64u)) {
                l = 1u + (covered < 64u ? covered : 64u);
                while (l <= p.upper && g + l < p.I && same_kmer(p, g, g + l)) ++l;
            }
            if (l > p.upper + 1u) l = p.upper + 1u;
            len[i] = l;
        }
        const bool rel = head && l >= p.lower && l <= p.upper;
        const uint64_t bh = __ballot(head), br = __ballot(rel);
        uint32_t e = rel ? l : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e += __shfl_xor(e, d, 64);
        nheads += (uint32_t)__popcll(bh);
        if (lane == 0) { srel[i * NW + w] = (uint32_t)__popcll(br); sent[i * NW + w] = e; }
    }
    if (lane == 0) shead[w] = nheads;
    __syncthreads();
    if (!EMIT) {
        if (tid == 0) {
            uint32_t x = 0, y = 0, z = 0;
            for (int k2 = 0; k2 < SLICES; ++k2) { x += srel[k2]; y += sent[k2]; }
            for (int k2 = 0; k2 < NW; ++k2) z += shead[k2];
            blk_rel[blockIdx.x] = x; blk_ent[blockIdx.x] = y; blk_heads[blockIdx.x] = z;
        }
        return;
    }
    if (tid == 0) {                               // exclusive prefixes over the slices, in item order (slice = i * NW + wave)
        uint32_t x = off_rel[blockIdx.x], y = off_ent[blockIdx.x];
        for (int k2 = 0; k2 < SLICES; ++k2) { const uint32_t a = srel[k2], b2 = sent[k2]; srel[k2] = x; sent[k2] = y; x += a; y += b2; }
    }
    __syncthreads();
    const uint64_t pmask = p.ib ? (1ull << p.ib) - 1 : 0;
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint32_t l = len[i];
        const bool rel = l != 0 && l >= p.lower && l <= p.upper;
        const uint64_t br = __ballot(rel);
        // entries of the reliable heads before this lane in its slice: wave-wide exclusive scan
        uint32_t inc = rel ? l : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        if (rel) {
            const uint64_t g = T0 + (uint64_t)i * RUN_THREADS + tid;
            const uint32_t kid = srel[i * NW + w] + (uint32_t)__popcll(br & lt), at = sent[i * NW + w] + inc - l;
            rel_kmers[kid] = p.ib ? (p.keys[g] >> p.ib) << (64 - p.k2) : p.keys[g];
            if (p.lo) rel_kmers_lo[kid] = p.lo[g];
            if (p.lo2) rel_kmers_lo2[kid] = p.lo2[g];
            rel_counts[kid] = l; colptr[kid] = at;
            for (uint32_t t = 0; t < l; ++t) { payload[at + t] = p.ib ? (p.keys[g + t] & pmask) : p.vals[g + t]; kid_of_entry[at + t] = kid; }
        }
    }
}

// Payload -> read << 32 | pos, one lane per entry.  The payload is the instance index with its `drop` lowest bits cut off (they did not
// fit beside the value in one 64-bit word, see stage_count_kmers): the entry is the dup-th instance among the 2^drop candidates whose
// canonical k-mer is the column's, dup = entries of the same column with the same payload before this one (equal payloads are adjacent:
// the sort is stable).  The instance's read is found by binary search in the reads' instance offsets (they stay in L2).
// Second pass over the sorted words, fused with the entries' (read, pos): EVERY item of a reliable run writes its own entry — consecutive
// lanes hold consecutive items and the entries of consecutive reliable runs are consecutive in the output, so the stores coalesce (the
// per-head loops of k_runs<true> wrote one 8-byte word per lane and step, and k_instance_entries read all of it back).  An item finds its
// head by counting the "same as my left neighbour" bits that end at it (LDS bit masks, as above), the head leaves its k-mer id and column
// pointer in LDS; runs that reach across the tile's end are finished by the tile of their head (64 items of halo), so UPPER <= 62 here.
// The entry is the dup-th instance among the 2^drop candidates behind the payload whose canonical k-mer is the run's (see
// k_instance_entries); its read comes from the block table.  With `csr_words` the entry is also written as the one-word sort key of the CSR
// build (read << (nb + pb + 2) | k-mer id << (pb + 2) | pos, matrix.hip: its two hint bits are added there): no column-id array, no conversion pass.
struct EmitOut {
    uint64_t *rel_kmers; uint32_t *rel_counts, *colptr;
    uint64_t *csc, *csr_words, *kid_of_entry;
    unsigned long long *prod_ctr;     // += sum over reliable runs of length^2 (the SpGEMM's product count, Ctx::A_products)
    int nb, pb, rs;                   // CSR sort key: read << rs | kid << (pb + 2) | pos  (rs >= nb + pb + 2)
};
__global__ __launch_bounds__(RUN_THREADS) __attribute__((amdgpu_waves_per_eu(6))) void k_runs_emit(RunParams p, EnumParams e, const BlockInfo *block_read, int drop, const uint32_t *off_rel, const uint32_t *off_ent, EmitOut o)
{
    constexpr int NW = RUN_THREADS / 64, SLICES = RUN_ITEMS * NW;
    __shared__ uint64_t eqmask[SLICES + 2];
    __shared__ uint32_t srel[SLICES + 1], sent[SLICES + 1];
    __shared__ uint32_t hkid[RUN_TILE], hat[RUN_TILE];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t lt = (1ull << lane) - 1;
    const uint64_t T0 = (uint64_t)blockIdx.x * RUN_TILE;
    const int ib = p.ib;
    uint64_t word[RUN_ITEMS];
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint64_t g = T0 + (uint64_t)i * RUN_THREADS + tid;
        word[i] = g < p.I ? p.keys[g] : 0;
        const uint64_t left = g > 0 && g < p.I ? p.keys[g - 1] : 0;
        const uint64_t bal = __ballot(g < p.I && g > 0 && (word[i] >> ib) == (left >> ib));
        if (lane == 0) eqmask[i * NW + w] = bal;
    }
    uint64_t hword = 0;
    if (w == 0) {
        const uint64_t g = T0 + RUN_TILE + lane;
        hword = g < p.I ? p.keys[g] : 0;
        const uint64_t left = g < p.I ? p.keys[g - 1] : 0;
        const uint64_t bal = __ballot(g < p.I && (hword >> ib) == (left >> ib));
        if (lane == 0) { eqmask[SLICES] = bal; eqmask[SLICES + 1] = 0; }
    }
    __syncthreads();
    uint32_t len[RUN_ITEMS];                      // 0: not a head; else run length (anything beyond 63 reads as 64: unreliable, UPPER <= 62)
    unsigned long long sq = 0;
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint32_t q = (uint32_t)i * RUN_THREADS + tid;
        const uint64_t g = T0 + q;
        const bool head = g < p.I && !((eqmask[q >> 6] >> (q & 63u)) & 1ull);
        uint32_t l = 0;
        if (head) {
            const uint32_t q1 = q + 1u, s1 = q1 & 63u;
            const uint64_t w0 = eqmask[q1 >> 6], w1 = eqmask[(q1 >> 6) + 1];
            const uint64_t win = s1 ? (w0 >> s1) | (w1 << (64u - s1)) : w0;      // the 64 bits that follow the head (tile + halo always hold them)
            l = 1u + (win == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~win));
        }
        len[i] = l;
        const bool rel = head && l >= p.lower && l <= p.upper;
        const uint64_t br = __ballot(rel);
        uint32_t en = rel ? l : 0u;
        sq += rel ? (unsigned long long)l * l : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) en += __shfl_xor(en, d, 64);
        if (lane == 0) { srel[i * NW + w] = (uint32_t)__popcll(br); sent[i * NW + w] = en; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
    if (lane == 0 && sq) atomicAdd(&o.prod_ctr[(blockIdx.x & 63u) * 16u], sq);      // (64 counters on lines of their own: one hot word would serialise 4 M wavefronts)
    __syncthreads();
    if (tid == 0) {                               // exclusive prefixes over the slices, in item order (slice = i * NW + wave)
        uint32_t x = off_rel[blockIdx.x], y = off_ent[blockIdx.x];
        for (int k2 = 0; k2 < SLICES; ++k2) { const uint32_t a = srel[k2], b2 = sent[k2]; srel[k2] = x; sent[k2] = y; x += a; y += b2; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint32_t l = len[i];
        const bool rel = l != 0 && l >= p.lower && l <= p.upper;
        const uint64_t br = __ballot(rel);
        uint32_t inc = rel ? l : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o2 = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o2; }
        const uint32_t q = (uint32_t)i * RUN_THREADS + tid;
        if (rel) {
            const uint32_t kid = srel[i * NW + w] + (uint32_t)__popcll(br & lt), at = sent[i * NW + w] + inc - l;
            o.rel_kmers[kid] = (word[i] >> ib) << (64 - p.k2);
            o.rel_counts[kid] = l; o.colptr[kid] = at;
            hkid[q] = kid; hat[q] = at;
        } else if (l != 0) hat[q] = 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint64_t pmask = (1ull << ib) - 1;
    auto emit = [&](uint32_t q, uint64_t wd) {        // q: place in tile + halo; wd: its word
        const uint32_t a = q >> 6, s = q & 63u;
        const uint64_t cur = eqmask[a], prv = a ? eqmask[a - 1] : ~0ull;
        const uint64_t win = (cur << (63u - s)) | (s < 63u ? prv >> (s + 1u) : 0ull);       // bit 63 = this item's bit, bit 62 = its left neighbour's, ...
        const uint32_t dist = win == ~0ull ? 64u : (uint32_t)__builtin_clzll(~win);
        if (dist > q || dist >= 63u) return;          // the head sits in the previous tile (its tile writes this entry), or the run is too long to be reliable
        const uint32_t h = q - dist;
        if (h >= (uint32_t)RUN_TILE) return;          // (halo items only) the head is in the halo: the next tile's
        const uint32_t at = hat[h];
        if (at == 0xFFFFFFFFu) return;
        const uint32_t kid = hkid[h];
        const uint64_t gq = T0 + q;
        const uint64_t hp = wd & pmask;
        uint64_t g = hp << drop;
        const ReadCursor rc = cursor_at(e, block_read, g);
        uint32_t lo = rc.lo;
        uint64_t off_lo = rc.off_lo, off_hi = rc.off_hi, boff = rc.boff;
        auto advance = [&](uint64_t gg) { while (gg >= off_hi) { ++lo; off_lo = off_hi; off_hi = e.inst_off[lo + 1]; boff = e.byte_off[lo]; } };
        advance(g);
        if (drop) {
            const uint64_t want = (wd >> ib) << (64 - p.k2);
            uint32_t dup = 0;
            for (uint32_t t = 1; t <= dist && t < (1u << drop) && p.keys[gq - t] == wd; ++t) ++dup;
            const uint64_t gend = g + (1ull << drop) < e.I ? g + (1ull << drop) : e.I;
            for (; g < gend; ++g) {
                advance(g);
                if (canonical_at_off(e, boff, (uint32_t)(g - off_lo)) == want) { if (dup == 0) break; --dup; }
            }
        }
        const uint32_t pos = (uint32_t)(g - off_lo);
        const uint32_t z = at + dist;
        o.csc[z] = ((uint64_t)lo << 32) | pos;
        if (o.csr_words) o.csr_words[z] = ((uint64_t)lo << o.rs) | ((uint64_t)kid << (o.pb + 2)) | pos;      // (hint bits: k_add_hints, matrix.hip)
        else o.kid_of_entry[z] = kid;
    };
#pragma unroll
    for (int i = 0; i < RUN_ITEMS; ++i) {
        const uint32_t q = (uint32_t)i * RUN_THREADS + tid;
        if (T0 + q < p.I) emit(q, word[i]);
    }
    if (w == 0 && T0 + RUN_TILE + lane < p.I) emit((uint32_t)RUN_TILE + lane, hword);
}

__global__ void k_instance_entries(const uint64_t *payload, const uint64_t *kid_of_entry, const uint64_t *rel_kmers, uint64_t *csc, uint64_t Z, EnumParams e, int drop,
                                   const BlockInfo *block_read)
{
    const uint64_t z = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    const uint64_t h = payload[z];
    uint64_t g = h << drop;
    const ReadCursor rc = cursor_at(e, block_read, g);
    uint32_t lo = rc.lo;                                              // last read with inst_off[read] <= g
    uint64_t off_lo = rc.off_lo, off_hi = rc.off_hi, boff = rc.boff;
    auto advance = [&](uint64_t gg) {                                 // (a block that holds a read boundary: walk on through the global arrays)
        while (gg >= off_hi) { ++lo; off_lo = off_hi; off_hi = e.inst_off[lo + 1]; boff = e.byte_off[lo]; }
    };
    advance(g);
    if (drop) {
        const uint64_t kid = kid_of_entry[z], want = rel_kmers[kid];
        uint32_t dup = 0;
        for (uint64_t y = z; y > 0 && kid_of_entry[y - 1] == kid && payload[y - 1] == h; --y) ++dup;
        const uint64_t gend = g + (1ull << drop) < e.I ? g + (1ull << drop) : e.I;
        for (; g < gend; ++g) {
            advance(g);
            if (canonical_at_off(e, boff, (uint32_t)(g - off_lo)) == want) { if (dup == 0) break; --dup; }
        }
    }
    csc[z] = ((uint64_t)lo << 32) | (uint32_t)(g - off_lo);
}

int bits_needed(uint64_t maxval)
{
    int b = 1;
    while (b < 64 && (maxval >> b)) ++b;
    return b;
}

int next_pow2_bits(uint64_t v)
{
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

EnumParams make_enum(Ctx &c)
{
    EnumParams e{};
    e.packed = c.d_packed; e.byte_off = c.d_byte_off; e.len = c.d_len; e.inst_off = c.inst_off.as<uint64_t>();
    e.nreads = (uint32_t)c.nreads; e.I = (uint64_t)c.I; e.k = c.cfg.k;
    return e;
}

}  // namespace

// Sorted words / (k-mer, value) pairs -> runs -> reliable columns: rel_kmers / rel_counts / a_colptr / a_csc / kid_of_entry of the context
// (see the comment above k_runs).  ib: payload bits below the value in `skeys` (0: the payload is svals); drop: low bits of the instance
// index that were cut off the payload.  scratch: at least (Z + 8) u64 (entry payloads), allocated by the caller's pool.
static void runs_to_columns(Ctx &c, const uint64_t *skeys, const uint64_t *svals, uint64_t I, uint64_t &nruns_out, uint64_t &N_out, uint64_t &Z_out,
                            const uint64_t *skeys_lo = nullptr, const uint64_t *skeys_lo2 = nullptr, int ib = 0, int drop = 0)
{
    hipStream_t s = c.stream;
    const uint32_t nblocks = (uint32_t)((I + RUN_TILE - 1) / RUN_TILE);
    c.ws_e.reserve((size_t)(nblocks + 2) * 4 * 5);
    uint32_t *blk_rel = c.ws_e.as<uint32_t>(), *blk_ent = blk_rel + (nblocks + 2), *blk_heads = blk_ent + (nblocks + 2), *off_rel = blk_heads + (nblocks + 2), *off_ent = off_rel + (nblocks + 2);
    RunParams p{};
    p.keys = skeys; p.lo = skeys_lo; p.lo2 = skeys_lo2; p.vals = svals; p.I = I; p.ib = ib; p.k2 = 2 * c.cfg.k;
    p.lower = (uint32_t)c.cfg.lower; p.upper = (uint32_t)c.cfg.upper;
    uint64_t nruns = 0, N = 0, Z = 0;
    if (I > 0) {
        ELBA_HIP(hipMemsetAsync(blk_rel, 0, (size_t)(nblocks + 2) * 4 * 3, s));
        hipLaunchKernelGGL((k_runs<false>), dim3(nblocks), dim3(RUN_THREADS), 0, s, p, blk_rel, blk_ent, blk_heads, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                           (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr);
        exclusive_scan_u32(s, blk_rel, off_rel, (int64_t)nblocks + 1, c.ws_scan);
        exclusive_scan_u32(s, blk_ent, off_ent, (int64_t)nblocks + 1, c.ws_scan);      // Z <= I < 2^32: the 32-bit scan cannot wrap
        exclusive_scan_u32(s, blk_heads, blk_heads, (int64_t)nblocks + 1, c.ws_scan);
        uint32_t h3[3] = {0, 0, 0};
        ELBA_HIP(hipMemcpyAsync(&h3[0], off_rel + nblocks, 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipMemcpyAsync(&h3[1], off_ent + nblocks, 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipMemcpyAsync(&h3[2], blk_heads + nblocks, 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        N = h3[0]; Z = h3[1]; nruns = h3[2];
    }
    ELBA_REQUIRE(Z < 0xFFFFFFF0ull, ELBA_ERR_UNSUPPORTED, "count_kmers: nnz(A) beyond 32-bit device offsets");
    c.rel_kmers.reserve((size_t)(N + 1) * 8);
    if (skeys_lo) c.rel_kmers_lo.reserve((size_t)(N + 1) * 8);
    if (skeys_lo2) c.rel_kmers_lo2.reserve((size_t)(N + 1) * 8);
    c.rel_counts.reserve((size_t)(N + 2) * 4);
    c.a_colptr.reserve((size_t)(N + 2) * 4);
    c.a_csc.reserve((size_t)(Z + 8) * 8);   // + guard entries (matrix.hip)
    c.kid_of_entry.reserve((size_t)(Z + 8) * 8);
    // Packed words with UPPER <= 62: one fused pass writes the columns, the entries' (read, pos) and — when read, k-mer id and position fit
    // one word — the sort keys of the CSR build (k_runs_emit).  Otherwise: heads write payloads and column ids, a second kernel converts.
    c.pre_ready = false; c.pre_consumed = false; c.pre_hints_done = false; c.pre_ell_done = false; c.pre_inline_pending = false; c.pre_pairs = false;
    const bool fused = ib && p.upper <= 62 && !c.opt.kmer_unfused;
    if (fused && Z > 0) {
        EnumParams e = make_enum(c);
        const uint64_t nib = (I >> IB_SHIFT) + 1;
        c.ws_b.reserve((size_t)(nib + 1) * sizeof(BlockInfo));
        hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((nib + 255) / 256)), dim3(256), 0, s, e.inst_off, e.byte_off, e.nreads, nib, c.ws_b.as<BlockInfo>());
        uint32_t maxlen = 0;
        for (int64_t r = 0; r < c.nreads; ++r) maxlen = c.h_len[(size_t)r] > maxlen ? c.h_len[(size_t)r] : maxlen;
        const uint64_t maxpos = maxlen >= (uint32_t)c.cfg.k ? maxlen - (uint32_t)c.cfg.k : 0;      // (a bound: the largest position any entry can have)
        const int mb = bits_needed((uint64_t)(c.nreads > 0 ? c.nreads - 1 : 0)), nb = bits_needed((uint64_t)(N > 0 ? N - 1 : 0)), pb = bits_needed(maxpos);
        const bool words = mb + nb + pb + 2 <= 64 && !c.opt.csr_pairs;
        const bool hints = pb <= 30 && !c.opt.no_hints;
        c.prod_ctr.reserve(64 * 128);
        ELBA_HIP(hipMemsetAsync(c.prod_ctr.p, 0, 64 * 128, s));
        EmitOut o{};
        o.prod_ctr = c.prod_ctr.as<unsigned long long>();
        o.rel_kmers = c.rel_kmers.as<uint64_t>(); o.rel_counts = c.rel_counts.as<uint32_t>(); o.colptr = c.a_colptr.as<uint32_t>();
        o.csc = c.a_csc.as<uint64_t>(); o.kid_of_entry = c.kid_of_entry.as<uint64_t>(); o.nb = nb; o.pb = pb;
        // inline partners (Ctx::csr_inline), as kmer_msd.hip: the keys leave room for them — read as high as it goes, pbi position bits — and
        // the hint pass that sees every entry's column anyway (k_add_hints, matrix.hip) writes them, if the matrix turns out to qualify
        // (padded columns, not dense: UPPER <= 16 guarantees the latter here, where no column has been seen yet)
        int rs = nb + pb + 2, pbi = 0;
        bool inl = words && hints && maxpos < 65536 && p.upper <= 16 && !c.opt.no_pay && !c.opt.no_inline && !c.opt.no_symmetry && !c.opt.no_ell && N < (1ll << 31) && mb >= 2;
        if (inl) {
            const int rs2 = 63 - mb;
            pbi = std::min(pb, (rs2 - (mb - 1)) / 2);
            if (rs2 >= rs && pbi >= 10) rs = rs2; else inl = false;
        }
        o.rs = rs;
        if (words) { c.csr_words.reserve((size_t)(Z + 8) * 8); o.csr_words = c.csr_words.as<uint64_t>(); }
        hipLaunchKernelGGL(k_runs_emit, dim3(nblocks), dim3(RUN_THREADS), 0, s, p, e, (const BlockInfo *)c.ws_b.as<BlockInfo>(), drop, (const uint32_t *)off_rel, (const uint32_t *)off_ent, o);
        c.pre_ready = true; c.pre_words = words; c.pre_hints = hints; c.pre_nb = nb; c.pre_pb = pb; c.pre_maxpos = maxpos; c.pre_rs = rs; c.pre_inline = false; c.pre_inline_pending = inl; c.pre_pbi = inl ? pbi : 0;
    } else {
    // entry payloads: written straight into a_csc when they are final (pairs: read << 32 | pos), else into scratch and converted
    DevBuf &scratch = c.ws_f;
    if (ib) scratch.reserve((size_t)(Z + 8) * 8);
    uint64_t *pay = ib ? scratch.as<uint64_t>() : c.a_csc.as<uint64_t>();
    if (I > 0)
        hipLaunchKernelGGL((k_runs<true>), dim3(nblocks), dim3(RUN_THREADS), 0, s, p, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, off_rel, off_ent,
                           c.rel_kmers.as<uint64_t>(), skeys_lo ? c.rel_kmers_lo.as<uint64_t>() : (uint64_t *)nullptr, skeys_lo2 ? c.rel_kmers_lo2.as<uint64_t>() : (uint64_t *)nullptr,
                           c.rel_counts.as<uint32_t>(), c.a_colptr.as<uint32_t>(), pay, c.kid_of_entry.as<uint64_t>());
    if (ib && Z > 0) {
        EnumParams e = make_enum(c);
        const uint64_t nib = (I >> IB_SHIFT) + 1;
        c.ws_b.reserve((size_t)(nib + 1) * sizeof(BlockInfo));
        hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((nib + 255) / 256)), dim3(256), 0, s, e.inst_off, e.byte_off, e.nreads, nib, c.ws_b.as<BlockInfo>());
        hipLaunchKernelGGL(k_instance_entries, dim3((unsigned)((Z + 255) / 256)), dim3(256), 0, s, pay, c.kid_of_entry.as<uint64_t>(), c.rel_kmers.as<uint64_t>(), c.a_csc.as<uint64_t>(), Z, e, drop,
                           c.ws_b.as<BlockInfo>());
    }
    }
    const uint32_t Zz = (uint32_t)Z;
    ELBA_HIP(hipMemcpyAsync(c.a_colptr.as<uint32_t>() + N, &Zz, 4, hipMemcpyHostToDevice, s));
    ELBA_HIP(hipStreamSynchronize(s));
    nruns_out = nruns; N_out = N; Z_out = Z;
}

void stage_count_kmers(Ctx &c)
{
    ELBA_REQUIRE(c.have_reads, ELBA_ERR_STATE, "count_kmers: no reads (call elba_set_reads)");
    hipStream_t s = c.stream;
    const int k = c.cfg.k;
    const int64_t M = c.nreads;
    c.have_counts = false; c.have_A = false; c.have_B = false; c.dist_owner = false;
    c.kmer_path = 0;
    elba_kmer_stats st{};
    st.nreads = M;

    // instance offsets: a host prefix sum over the read lengths (include/KmerOps.hpp:118-119: reads shorter than k contribute nothing)
    std::vector<uint64_t> off((size_t)M + 1);
    uint64_t I = 0;
    uint32_t maxlen = 0;
    for (int64_t r = 0; r < M; ++r) { const uint32_t l = c.h_len[(size_t)r]; off[(size_t)r] = I; if ((int64_t)l >= k) I += (uint64_t)l - k + 1; maxlen = l > maxlen ? l : maxlen; }
    off[(size_t)M] = I;
    c.max_read_len = maxlen;      // (the one host walk over the read lengths of this stage: kmer_msd.hip sizes its position field by it)
    // (more than 2^32 instances: the two-level partition of kmer_msd.hip counts them in passes over value ranges — k <= 17; every other path holds 32-bit places)
    ELBA_REQUIRE(I < 0xFFFFFFF0ull || (k <= 17 && k >= 9 && !c.opt.kmer_no_msd), ELBA_ERR_UNSUPPORTED, "count_kmers: more than 2^32 k-mer instances on one GPU (k <= 17 only)");
    c.I = (int64_t)I; c.kmer_passes = 1;
    c.inst_off.reserve((size_t)(M + 1) * 8);
    ELBA_HIP(hipMemcpyAsync(c.inst_off.p, off.data(), (size_t)(M + 1) * 8, hipMemcpyHostToDevice, s));

    if (k > 31) {
        // ---- two- and three-word k-mers: sort an index permutation, last word first (stable LSD over all words), then gather ----
        const int words = k > 64 ? 3 : 2;
        DevBuf khi, klo, klo2, val, i0, i1, t0, t1, s2buf;
        for (DevBuf *b : {&khi, &klo, &val, &i0, &i1, &t0, &t1}) b->reserve((size_t)(I + 2) * 8);
        if (words == 3) { klo2.reserve((size_t)(I + 2) * 8); s2buf.reserve((size_t)(I + 2) * 8); }
        c.ws_a.reserve((size_t)(I + 2) * 8); c.ws_b.reserve((size_t)(I + 2) * 8); c.ws_c.reserve((size_t)(I + 2) * 8);
        c.ws_e.reserve((size_t)(I + 2) * 4); c.ws_f.reserve((size_t)(I + 2) * 8);
        c.t_total.start(s);
        c.t_a.start(s);
        EnumParams e = make_enum(c);
        const uint64_t nblocks = (I + EN_PER_BLOCK - 1) / EN_PER_BLOCK;
        const unsigned nbI = (unsigned)((I + 255) / 256);
        uint64_t *shi = c.ws_a.as<uint64_t>(), *slo = c.ws_b.as<uint64_t>(), *sval = c.ws_c.as<uint64_t>(), *slo2 = words == 3 ? s2buf.as<uint64_t>() : nullptr;
        if (I > 0) {
            if (words == 3) hipLaunchKernelGGL(k_kmer_emit3, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, khi.as<uint64_t>(), klo.as<uint64_t>(), klo2.as<uint64_t>(), val.as<uint64_t>(), i0.as<uint64_t>());
            else hipLaunchKernelGGL(k_kmer_emit2, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, khi.as<uint64_t>(), klo.as<uint64_t>(), val.as<uint64_t>(), i0.as<uint64_t>());
            uint64_t *ia = i0.as<uint64_t>(), *ib = i1.as<uint64_t>(), *ka = t0.as<uint64_t>(), *kb = t1.as<uint64_t>();
            const uint64_t *wordsrc[3] = {khi.as<uint64_t>(), klo.as<uint64_t>(), words == 3 ? klo2.as<uint64_t>() : nullptr};
            for (int wd = words - 1; wd >= 0; --wd) {
                // keys of this pass = word wd in the current order
                if (wd == words - 1) ELBA_HIP(hipMemcpyAsync(ka, wordsrc[wd], (size_t)I * 8, hipMemcpyDeviceToDevice, s));
                else hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, ia, wordsrc[wd], I, ka);
                const int lo_bit = wd == words - 1 ? 64 - 2 * (k - 32 * (words - 1)) : 0;
                const int w = radix_sort_pairs(s, ka, ia, kb, ib, (int64_t)I, lo_bit, 64, c.ws_sort);
                if (w) { uint64_t *t; t = ia; ia = ib; ib = t; t = ka; ka = kb; kb = t; }
            }
            const uint64_t *fin = ia;
            hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, fin, khi.as<uint64_t>(), I, shi);
            hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, fin, klo.as<uint64_t>(), I, slo);
            if (words == 3) hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, fin, klo2.as<uint64_t>(), I, slo2);
            hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, fin, val.as<uint64_t>(), I, sval);
        }
        c.t_a.stop(s);
        c.t_b.start(s);
        uint64_t nruns = 0, N = 0, Z = 0;
        runs_to_columns(c, shi, sval, I, nruns, N, Z, slo, slo2);
        c.t_b.stop(s);
        c.t_total.stop(s);
        ELBA_HIP(hipStreamSynchronize(s));
        st.instances = (int64_t)I; st.distinct = (int64_t)nruns; st.reliable = (int64_t)N; st.entries = (int64_t)Z;
        st.ms_total = c.t_total.ms(); st.ms_count = c.t_a.ms(); st.ms_sort = c.t_b.ms(); st.ms_lookup = 0;
        c.ndistinct = (int64_t)nruns;
        c.N = (int64_t)N; c.Z = (int64_t)Z;
        c.kstats = st;
        c.have_counts = true;
        return;
    }
    if (msd_count_kmers(c, I, st)) {      // k <= 17 on inputs of some size: two-level value partition + LDS count tables (kmer_msd.hip)
        c.kstats = st;
        c.have_counts = true;
        return;
    }
    ELBA_REQUIRE(I < 0xFFFFFFF0ull, ELBA_ERR_UNSUPPORTED, "count_kmers: more than 2^32 k-mer instances on one GPU and the two-level partition does not take this input");
    {
        // ---- one-word k-mers: see the header of this file ----
        c.t_total.start(s);
        c.t_a.start(s);
        EnumParams e = make_enum(c);
        const uint64_t nblocks = (I + EN_PER_BLOCK - 1) / EN_PER_BLOCK;
        // One 64-bit word per instance: value << pb | (instance index >> drop).  The index is the payload — (read, pos) follow from it
        // through the instance offsets, for the entries that survive the count filter only — and when value and index are a few bits too
        // many for one word (k = 17 beyond 2^30 instances) its `drop` lowest bits are left out: the candidates they stand for are told
        // apart afterwards by recomputing their k-mers (k_instance_entries).  The sort then moves 8 bytes per instance, not 16.
        int ib = 1;
        while (ib < 63 && (I >> ib)) ++ib;
        int drop = 2 * k + ib > 64 ? 2 * k + ib - 64 : 0;
        if (c.opt.kmer_drop) { const int want = c.opt.kmer_drop; if (want > drop && want <= 3 && want < ib) drop = want; }      // (test hook: small inputs through the dropped-index-bit path)
        const bool packed_words = drop <= 3 && !c.opt.kmer_pairs;
        if (!packed_words) { ib = 0; drop = 0; }
        const int pb = ib - drop;                  // payload bits below the value
        // workspaces: the words and their ping-pong copy; (value, payload) pairs need two more arrays (the other buffers are sized where they are used)
        c.ws_a.reserve((size_t)(I + 2) * 8); c.ws_c.reserve((size_t)(I + 2) * 8);
        if (!packed_words) { c.ws_b.reserve((size_t)(I + 2) * 8); c.ws_d.reserve((size_t)(I + 2) * 8); }
        int where = 0;
        if (packed_words) {
            int sh0 = 0, b0 = 0, tile = 0;
            uint32_t *hist0 = I > 1 ? radix_first_histogram((int64_t)I, pb, pb + 2 * k, c.ws_sort, &sh0, &b0, &tile) : nullptr;
            const bool fuse_hist = hist0 && tile == 4 * EN_PER_BLOCK && b0 <= 9 && !c.opt.emit_plain;      // (the emit writes whole tiles of the sort and counts their first digit)
            if (fuse_hist) {
                const uint64_t nib = (I >> IB_SHIFT) + 1;
                c.ws_b.reserve((size_t)(nib + 1) * sizeof(BlockInfo));
                hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((nib + 255) / 256)), dim3(256), 0, s, e.inst_off, e.byte_off, e.nreads, nib, c.ws_b.as<BlockInfo>());
                hipLaunchKernelGGL((k_kmer_emit_packed_hist<4>), dim3((unsigned)((I + tile - 1) / tile)), dim3(EN_THREADS), 0, s, e, (const BlockInfo *)c.ws_b.as<BlockInfo>(), pb, drop, c.ws_a.as<uint64_t>(), sh0, b0, hist0);
            }
            else if (I > 0) hipLaunchKernelGGL(k_kmer_emit_packed, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, pb, drop, c.ws_a.as<uint64_t>());
            where = radix_sort_keys(s, c.ws_a.as<uint64_t>(), c.ws_c.as<uint64_t>(), (int64_t)I, pb, pb + 2 * k, c.ws_sort, fuse_hist);
        } else {
            if (I > 0) hipLaunchKernelGGL(k_kmer_emit, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>());
            where = radix_sort_pairs(s, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>(), c.ws_c.as<uint64_t>(), c.ws_d.as<uint64_t>(), (int64_t)I, 64 - 2 * k, 64, c.ws_sort);
        }
        const uint64_t *skeys = where ? c.ws_c.as<uint64_t>() : c.ws_a.as<uint64_t>(), *svals = packed_words ? (const uint64_t *)nullptr : (where ? c.ws_d.as<uint64_t>() : c.ws_b.as<uint64_t>());
        c.t_a.stop(s);
        c.t_b.start(s);
        uint64_t nruns = 0, N = 0, Z = 0;
        runs_to_columns(c, skeys, svals, I, nruns, N, Z, nullptr, nullptr, pb, drop);
        c.t_b.stop(s);
        c.t_total.stop(s);
        ELBA_HIP(hipStreamSynchronize(s));
        st.instances = (int64_t)I; st.distinct = (int64_t)nruns; st.reliable = (int64_t)N; st.entries = (int64_t)Z;
        st.ms_total = c.t_total.ms(); st.ms_count = c.t_a.ms(); st.ms_sort = c.t_b.ms(); st.ms_lookup = 0;
        c.ndistinct = (int64_t)nruns;
        c.N = (int64_t)N; c.Z = (int64_t)Z;
        c.kstats = st;
        c.have_counts = true;
        return;
    }

}

// column id of every entry, from the column pointers (one lane per column: columns hold at most UPPER entries)
__global__ void k_expand_colptr(const uint32_t *colptr, uint64_t N, uint64_t *kid_of_entry)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    for (uint32_t z = colptr[k], z1 = colptr[k + 1]; z < z1; ++z) kid_of_entry[z] = k;
}

void stage_create_kmer_matrix(Ctx &c)
{
    ELBA_REQUIRE(c.have_counts, ELBA_ERR_STATE, "create_kmer_matrix: no k-mer counts (call elba_count_kmers)");
    ELBA_REQUIRE(!c.dist_owner, ELBA_ERR_STATE, "create_kmer_matrix: this context counted exchanged records (elba_dist_count_records); its matrix is a panel (elba_dist_set_panel)");
    hipStream_t s = c.stream;
    const int64_t M = c.nreads, N = c.N, Z = c.Z;
    c.have_A = false; c.have_B = false;
    // CSC(A) came out of the counting sort already; the CSR build needs the column id of every entry.  The fused column pass (k_runs_emit) left
    // them as ready-made sort keys (csr_words), which the build CONSUMES (hint bits are ORed in, the sort ping-pongs over them): a second call
    // after one elba_count_kmers rebuilds the column ids from the column pointers instead.
    const bool pre = c.pre_ready;
    if (!pre && c.pre_consumed && N > 0) {
        c.kid_of_entry.reserve((size_t)(Z + 8) * 8);
        hipLaunchKernelGGL(k_expand_colptr, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const uint32_t *)c.a_colptr.as<uint32_t>(), (uint64_t)N, c.kid_of_entry.as<uint64_t>());
    }
    c.t_c.start(s);
    c.t_c.stop(s);
    c.A_has_kmers = true;
    finish_matrix_from_sorted_csc(c, M, N, Z, c.kid_of_entry.as<uint64_t>(), 0, c.a_csc.as<uint64_t>(), 0, -1, pre);      // (column ids — or the CSR sort keys — were written with the columns: k_runs / k_runs_emit)
    if (pre) { c.pre_ready = false; c.pre_consumed = true; }
    c.kstats.ms_lookup = c.t_c.ms();
}

}  // namespace elba

// =====================================================================================================================
// Distributed building blocks: 1D read-row shards x hash-owned k-mer columns.
//   exchange #1  every k-mer instance (kmer, global read, pos) goes to the rank that owns the k-mer        [all-to-all]
//   owner        exact count + [LOWER, UPPER] filter on the received records, columns sorted by (read, pos)
//   ids          reliable k-mers of all owners are all-gathered; global k-mer id = rank of the packed value  [all-gather]
//   exchange #2  each column goes, whole, to every rank that owns at least one of its reads (the column panel) [all-to-all]
// The collectives themselves are issued by the host driver (torch.distributed: RCCL on GPUs, gloo in the CPU tests); the
// kernels below only produce / consume the device buffers.  A record of exchange #1 is W + 1 u64 words (W words of k-mer, 1 for k <= 31), of exchange #2 two.
// =====================================================================================================================
namespace elba {

namespace {

constexpr int MAX_RANKS = 64;

// Owner of a k-mer = the rank whose VALUE RANGE holds it.  The value space is cut into 2^OWNER_BITS equal bins by the leading bits of
// the packed canonical k-mer (first word); the driver picks, from the all-reduced bin histogram of the instances, OWNER boundaries
// upper[r] (exclusive, in bins) that balance the instances, and rank r owns the bins [upper[r-1], upper[r]).  Ranges instead of the
// reference's hash (GetKmerOwner, src/KmerOps.cpp:352-359) because the owners' reliable k-mers are then disjoint ascending value
// ranges: the global k-mer id — rank of the value, SURVEY.md §8c-2 — is the owner's local index plus an exclusive scan of the owners'
// counts (the reference's Exscan, src/KmerOps.cpp:371-375), and no rank ever needs the other ranks' k-mers.
constexpr int OWNER_BITS = 12;
struct OwnerMap { uint32_t nranks; uint32_t upper[MAX_RANKS]; };
__device__ __forceinline__ uint32_t owner_of(uint64_t first_word, const OwnerMap &om)
{
    const uint32_t bin = (uint32_t)(first_word >> (64 - OWNER_BITS));
    uint32_t lo = 0, hi = om.nranks - 1u;             // first r with bin < upper[r]
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (bin < om.upper[mid]) hi = mid; else lo = mid + 1u; }
    return lo;
}

// W = 64-bit words per k-mer (1 for k <= 31, 2 up to 63, 3 up to 95).  A record of exchange #1 is W + 1 words: the k-mer, most
// significant word first, then global read << 32 | pos.  The owner of a multi-word k-mer hashes all its words.
template <int W, class F>
__device__ __forceinline__ void for_each_kmer_words(const EnumParams &e, F &&f)
{
    if constexpr (W == 1) for_each_instance(e, [&](uint64_t g, uint32_t r, uint32_t p, uint64_t km) { f(g, r, p, km, 0ull, 0ull); });
    else for_each_position(e, [&](uint64_t g, uint32_t r, uint32_t p) {
        uint64_t a, b, c2 = 0;
        if constexpr (W == 2) canonical2_at(e, r, p, a, b); else canonical3_at(e, r, p, a, b, c2);
        f(g, r, p, a, b, c2);
    });
}
// histogram of the instances over the 2^OWNER_BITS value bins (what the driver all-reduces to place the owners' boundaries)
template <int W>
__global__ __launch_bounds__(EN_THREADS) void k_dist_value_hist(EnumParams e, unsigned long long *hist)
{
    __shared__ uint32_t h[1 << OWNER_BITS];
    for (uint32_t b = threadIdx.x; b < (1u << OWNER_BITS); b += EN_THREADS) h[b] = 0;
    __syncthreads();
    for_each_kmer_words<W>(e, [&](uint64_t, uint32_t, uint32_t, uint64_t a, uint64_t, uint64_t) { atomicAdd(&h[(uint32_t)(a >> (64 - OWNER_BITS))], 1u); });
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < (1u << OWNER_BITS); b += EN_THREADS) if (h[b]) atomicAdd(&hist[b], (unsigned long long)h[b]);
}

template <int W>
__global__ __launch_bounds__(EN_THREADS) void k_dist_count_owners(EnumParams e, OwnerMap om, unsigned long long *counts)
{
    __shared__ uint32_t hist[MAX_RANKS];
    if (threadIdx.x < MAX_RANKS) hist[threadIdx.x] = 0;
    __syncthreads();
    for_each_kmer_words<W>(e, [&](uint64_t, uint32_t, uint32_t, uint64_t a, uint64_t, uint64_t) { atomicAdd(&hist[owner_of(a, om)], 1u); });
    __syncthreads();
    if (threadIdx.x < om.nranks && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

template <int W>
__global__ __launch_bounds__(EN_THREADS) void k_dist_fill_send(EnumParams e, OwnerMap om, uint64_t first_global_id, uint64_t *send, unsigned long long *cursors)
{
    __shared__ uint32_t hist[MAX_RANKS];
    __shared__ unsigned long long base[MAX_RANKS];
    if (threadIdx.x < MAX_RANKS) hist[threadIdx.x] = 0;
    __syncthreads();
    for_each_kmer_words<W>(e, [&](uint64_t, uint32_t, uint32_t, uint64_t a, uint64_t, uint64_t) { atomicAdd(&hist[owner_of(a, om)], 1u); });
    __syncthreads();
    if (threadIdx.x < om.nranks) {
        base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], (unsigned long long)hist[threadIdx.x]) : 0ull;
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
    for_each_kmer_words<W>(e, [&](uint64_t, uint32_t r, uint32_t p, uint64_t a, uint64_t b, uint64_t c2) {
        const uint32_t o = owner_of(a, om);
        const unsigned long long at = base[o] + atomicAdd(&hist[o], 1u);
        uint64_t *rec = send + (size_t)(W + 1) * at;
        rec[0] = a;
        if constexpr (W >= 2) rec[1] = b;
        if constexpr (W >= 3) rec[2] = c2;
        rec[W] = ((first_global_id + r) << 32) | p;
    });
}

// ---- exchange #1 with 8-byte records (one-word k-mers; round 5) -------------------------------------------------------------------------
// A record of exchange #1 was (k-mer, global read << 32 | pos): 16 bytes per instance, 2 * 10^9 instances on BASELINE config 3 — at 8 GPUs ~3.5 GB per
// rank over xGMI against a ~7 ms local k-mer stage.  Packed: (value - first value of the OWNER's range) << IB | instance index in the SENDER's reads —
// the owner's range is 1/nranks of the value space and an index needs bits(instances per rank): 31 + 28 bits at eight ranks, 33 + 30 at two.  The source
// rank is known from the segment of the receive buffer a record arrives in; every rank holds every rank's read lengths (one all-gather of 4 bytes per
// read), so the owner turns (source, index) back into (global read, pos) by a search in the source's instance offsets — neighbouring records name
// neighbouring instances: the searches of a wavefront walk the same few cache lines — and hands elba_dist_count_records the 16-byte records it always took.
struct PackFmt { int vb, ib, k2; uint32_t nranks; uint64_t lo[MAX_RANKS]; };      // lo[r] = first value (right-aligned) of owner r's range
__global__ __launch_bounds__(EN_THREADS) void k_dist_fill_send_packed(EnumParams e, OwnerMap om, PackFmt f, uint64_t *send, unsigned long long *cursors)
{
    __shared__ uint32_t hist[MAX_RANKS];
    __shared__ unsigned long long base[MAX_RANKS];
    if (threadIdx.x < MAX_RANKS) hist[threadIdx.x] = 0;
    __syncthreads();
    for_each_instance(e, [&](uint64_t, uint32_t, uint32_t, uint64_t km) { atomicAdd(&hist[owner_of(km, om)], 1u); });
    __syncthreads();
    if (threadIdx.x < om.nranks) {
        base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], (unsigned long long)hist[threadIdx.x]) : 0ull;
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
    for_each_instance(e, [&](uint64_t g, uint32_t, uint32_t, uint64_t km) {
        const uint32_t o = owner_of(km, om);
        const unsigned long long at = base[o] + atomicAdd(&hist[o], 1u);
        send[at] = (((km >> (64 - f.k2)) - f.lo[o]) << f.ib) | g;
    });
}
// segment `src` of the receive buffer: n packed records of rank src -> 16-byte records; off = the instance offsets of src's reads (nr + 1 of them), gid0 = its first global read
__global__ __launch_bounds__(256) void k_dist_unpack(const uint64_t *packed, uint64_t n, const uint64_t *off, uint32_t nr, uint64_t gid0, uint64_t lo_self, int ib, int k2, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t w = packed[i], g = w & ((1ull << ib) - 1ull), value = (w >> ib) + lo_self;
    uint32_t a = 0, b = nr;      // last read with off[read] <= g (reads shorter than k have empty ranges: the LAST of equal offsets is the one that holds g)
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (off[mid] <= g) a = mid; else b = mid; }
    out[2 * i] = value << (64 - k2);
    out[2 * i + 1] = ((gid0 + a) << 32) | (g - off[a]);
}

// records of W + 1 words -> W word arrays + the payload
__global__ void k_split_records(const uint64_t *rec, uint64_t n, int rw, uint64_t *w0, uint64_t *w1, uint64_t *w2, uint64_t *val, uint64_t *idx)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *r = rec + (size_t)rw * i;
    w0[i] = r[0];
    if (w1) w1[i] = r[1];
    if (w2) w2[i] = r[2];
    if (val) val[i] = r[rw - 1];
    if (idx) idx[i] = i;
}

// W word arrays -> interleaved words (what travels in the all-gather of the reliable k-mers)
__global__ void k_join_words(const uint64_t *w0, const uint64_t *w1, const uint64_t *w2, uint64_t n, int W, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[(size_t)W * i] = w0[i];
    if (W >= 2) out[(size_t)W * i + 1] = w1[i];
    if (W >= 3) out[(size_t)W * i + 2] = w2[i];
}

// global id of a multi-word k-mer: its rank in the sorted union (word arrays, lexicographic)
__global__ void k_global_ids_words(const uint64_t *l0, const uint64_t *l1, const uint64_t *l2, uint64_t nlocal, const uint64_t *a0, const uint64_t *a1, const uint64_t *a2, uint64_t nall,
                                   uint32_t *gid)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlocal) return;
    const uint64_t x0 = l0[i], x1 = l1[i], x2 = l2 ? l2[i] : 0;
    uint64_t lo = 0, hi = nall;               // first index whose k-mer is >= (x0, x1, x2)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t y0 = a0[mid], y1 = a1[mid], y2 = a2 ? a2[mid] : 0;
        const bool less = y0 < x0 || (y0 == x0 && (y1 < x1 || (y1 == x1 && y2 < x2)));
        if (less) lo = mid + 1; else hi = mid;
    }
    gid[i] = (uint32_t)lo;
}

// global id of each local reliable k-mer = its rank in the sorted union of all owners' reliable k-mers
__global__ void k_global_ids(const uint64_t *local, uint64_t nlocal, const uint64_t *all_sorted, uint64_t nall, uint32_t *gid)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlocal) return;
    const uint64_t km = local[i];
    uint64_t lo = 0, hi = nall;               // first index with all_sorted[idx] >= km (k-mers are distinct across owners)
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (all_sorted[mid] < km) lo = mid + 1; else hi = mid; }
    gid[i] = (uint32_t)lo;
}

__device__ __forceinline__ uint32_t rank_of_read(const uint64_t *bounds, uint32_t nranks, uint64_t read)
{
    uint32_t lo = 0, hi = nranks;             // last r with bounds[r] <= read
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (bounds[mid] <= read) lo = mid; else hi = mid; }
    return lo;
}

// one lane per column: the set of ranks that own at least one read of the column; FILL = false counts, true writes
// (win != nullptr: a rank is a destination only if the column has a read inside that rank's current row block [win[r], win[nranks + r]))
template <bool FILL>
__global__ void k_panel(const uint32_t *colptr, const uint64_t *csc, const uint32_t *gid, uint64_t N, const uint64_t *bounds, uint32_t nranks, const uint64_t *win,
                        unsigned long long *counts_or_cursors, uint64_t *send)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c0 = k < N ? colptr[k] : 0u, c1 = k < N ? colptr[k + 1] : 0u;      // (lanes behind the last column stay for the wave scans: empty columns)
    uint64_t mask = 0;
    for (uint32_t a = c0; a < c1; ++a) {
        const uint64_t read = csc[a] >> 32;
        const uint32_t r = rank_of_read(bounds, nranks, read);
        if (!win || (read >= win[r] && read < win[nranks + r])) mask |= 1ull << r;
    }
    // one atomic per wavefront and destination (its lanes' columns take consecutive places): a lane per (column, destination) sent
    // 10^8 adds to at most 64 words — 334 ms for 58 M columns, of a 0.5 s distributed build
    const uint32_t lane = threadIdx.x & 63;
    uint64_t any = mask;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) any |= __shfl_xor(any, dd, 64);
    while (any) {                                  // (wave-uniform)
        const uint32_t d = (uint32_t)__ffsll((unsigned long long)any) - 1;
        any &= any - 1;
        const uint32_t mine = ((mask >> d) & 1ull) ? c1 - c0 : 0u;
        uint32_t inc = mine;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = __shfl_up(inc, dd, 64); if ((int)lane >= dd) inc += o; }
        const uint32_t total = __shfl(inc, 63, 64);
        unsigned long long base = 0;
        if (lane == 63) base = atomicAdd(&counts_or_cursors[d], (unsigned long long)total);
        base = __shfl(base, 63, 64);
        if (FILL && mine) {
            const unsigned long long at = base + (inc - mine);
            const uint64_t g = gid[k];
            for (uint32_t a = c0; a < c1; ++a) { send[2 * (at + (a - c0))] = g; send[2 * (at + (a - c0)) + 1] = csc[a]; }
        }
    }
}

__global__ void k_iota_u32(uint32_t *out, uint64_t n, uint32_t base)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base + (uint32_t)i;
}

// sorted global column ids -> local ones: id of a record = number of distinct ids before it (flag: 1 where a new id starts; excl: its exclusive scan)
__global__ void k_local_ids(const uint32_t *flag, const uint32_t *excl, uint64_t n, uint64_t *lid)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lid[i] = (uint64_t)(excl[i] + flag[i] - 1u);
}

__global__ void k_deinterleave(const uint64_t *rec, uint64_t n, uint64_t *k0, uint64_t *v0)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    k0[i] = rec[2 * i];
    v0[i] = rec[2 * i + 1];
}

}  // namespace

static int kmer_words(int k) { return k > 64 ? 3 : (k > 32 ? 2 : 1); }      // (k is odd: 31 is the last one-word k, 63 the last two-word k)

// Stable LSD sort of n multi-word keys (src[0] most significant) through an index permutation, last word first; `perm` receives the
// order (perm[j] = index of the j-th smallest key).  Scratch: four n-word buffers.
static const uint64_t *sort_words_permutation(Ctx &c, int words, const uint64_t *const src[3], uint64_t n, int k, uint64_t *ia, uint64_t *ib, uint64_t *ka, uint64_t *kb)
{
    hipStream_t s = c.stream;
    const unsigned nb = (unsigned)((n + 255) / 256);
    for (int wd = words - 1; wd >= 0; --wd) {
        if (wd == words - 1) ELBA_HIP(hipMemcpyAsync(ka, src[wd], (size_t)n * 8, hipMemcpyDeviceToDevice, s));     // ia is the identity here
        else hipLaunchKernelGGL(k_gather_u64, dim3(nb), dim3(256), 0, s, ia, src[wd], n, ka);
        const int lo_bit = wd == words - 1 ? 64 - 2 * (k - 32 * (words - 1)) : 0;
        const int w = radix_sort_pairs(s, ka, ia, kb, ib, (int64_t)n, lo_bit, 64, c.ws_sort);
        if (w) { uint64_t *t; t = ia; ia = ib; ib = t; t = ka; ka = kb; kb = t; }
    }
    return ia;
}

// the owners' boundaries set by elba_dist_set_owner_ranges (one rank: everything is rank 0's; otherwise they must have been set for this world size)
static OwnerMap owner_map(Ctx &c, int nranks)
{
    OwnerMap om{};
    om.nranks = (uint32_t)nranks;
    if (nranks == 1) { om.upper[0] = 1u << OWNER_BITS; return om; }
    ELBA_REQUIRE((int)c.owner_upper.size() == nranks, ELBA_ERR_STATE, "owner value ranges not set for this many ranks (call elba_dist_set_owner_ranges)");
    for (int r = 0; r < nranks; ++r) om.upper[r] = c.owner_upper[(size_t)r];
    return om;
}

static uint64_t upload_instance_offsets(Ctx &c)
{
    const int k = c.cfg.k;
    const int64_t M = c.nreads;
    std::vector<uint64_t> off((size_t)M + 1);
    uint64_t I = 0;
    for (int64_t r = 0; r < M; ++r) { off[(size_t)r] = I; if ((int64_t)c.h_len[(size_t)r] >= k) I += (uint64_t)c.h_len[(size_t)r] - k + 1; }
    off[(size_t)M] = I;
    ELBA_REQUIRE(I < 0xFFFFFFF0ull, ELBA_ERR_UNSUPPORTED, "more than 2^32 k-mer instances on one GPU");
    c.I = (int64_t)I;
    c.inst_off.reserve((size_t)(M + 1) * 8);
    ELBA_HIP(hipMemcpyAsync(c.inst_off.p, off.data(), (size_t)(M + 1) * 8, hipMemcpyHostToDevice, c.stream));
    ELBA_HIP(hipStreamSynchronize(c.stream));      // `off` goes out of scope
    return I;
}

void stage_dist_value_histogram(Ctx &c, uint64_t *hist_host, int64_t nbins)
{
    ELBA_REQUIRE(c.have_reads, ELBA_ERR_STATE, "dist_value_histogram: no reads");
    ELBA_REQUIRE(nbins == (1 << OWNER_BITS) && hist_host, ELBA_ERR_INVALID_ARG, "dist_value_histogram: the histogram has 4096 bins");
    hipStream_t s = c.stream;
    const uint64_t I = upload_instance_offsets(c);
    c.ws_scan.reserve((size_t)nbins * 8);
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, (size_t)nbins * 8, s));
    EnumParams e = make_enum(c);
    const uint64_t nblocks = (I + EN_PER_BLOCK - 1) / EN_PER_BLOCK;
    if (I > 0) {
        const int W = kmer_words(c.cfg.k);
        if (W == 1) hipLaunchKernelGGL(k_dist_value_hist<1>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, c.ws_scan.as<unsigned long long>());
        else if (W == 2) hipLaunchKernelGGL(k_dist_value_hist<2>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, c.ws_scan.as<unsigned long long>());
        else hipLaunchKernelGGL(k_dist_value_hist<3>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, c.ws_scan.as<unsigned long long>());
    }
    ELBA_HIP(hipMemcpyAsync(hist_host, c.ws_scan.p, (size_t)nbins * 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
}

void stage_dist_set_owner_ranges(Ctx &c, int nranks, const uint32_t *upper_bins)
{
    ELBA_REQUIRE(nranks >= 1 && nranks <= MAX_RANKS && upper_bins, ELBA_ERR_INVALID_ARG, "dist_set_owner_ranges: 1..64 ranks");
    uint32_t prev = 0;
    for (int r = 0; r < nranks; ++r) { ELBA_REQUIRE(upper_bins[r] >= prev && upper_bins[r] <= (1u << OWNER_BITS), ELBA_ERR_INVALID_ARG, "dist_set_owner_ranges: boundaries must ascend within [0, 4096]"); prev = upper_bins[r]; }
    ELBA_REQUIRE(upper_bins[nranks - 1] == (1u << OWNER_BITS), ELBA_ERR_INVALID_ARG, "dist_set_owner_ranges: the last rank's range must end at 4096");
    c.owner_upper.assign(upper_bins, upper_bins + nranks);
}

// global k-mer ids of this owner's columns: base + local index (the owners hold ascending value ranges)
void stage_dist_set_kmer_id_base(Ctx &c, int64_t base, int64_t nall)
{
    ELBA_REQUIRE(c.have_counts && c.dist_owner, ELBA_ERR_STATE, "dist_set_kmer_id_base: call dist_count_records first");
    ELBA_REQUIRE(base >= 0 && base + c.own_N <= nall && nall < 0xFFFFFFF0ll, ELBA_ERR_INVALID_ARG, "dist_set_kmer_id_base: bad id range");
    c.dist_gid.reserve((size_t)(c.own_N + 1) * 4);
    if (c.own_N > 0) hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((c.own_N + 255) / 256)), dim3(256), 0, c.stream, c.dist_gid.as<uint32_t>(), (uint64_t)c.own_N, (uint32_t)base);
    ELBA_HIP(hipStreamSynchronize(c.stream));
    c.dist_nall = nall;
}

// ---- the reference's own k-mer hash and owner, on the device -------------------------------------------------------
// Kmer::GetHash (src/Kmer.cpp:207-213) = h1 of murmurhash3_x64_128 (src/HashFuncs.cpp:40-117) over the 8 * NLONGS key bytes (the words
// of `longs`, little-endian, first word first), seed 313; GetKmerOwner (src/KmerOps.cpp:352-359) = (size_t)(double(h) * double(p) /
// double(UINT64_MAX)).  This build places k-mers by value range instead (the results do not depend on the placement); the reference's
// placement is offered for callers that want to reproduce it, and checked against the reference's own vectors (tests/golden/kmer_vectors_*).
namespace {
__device__ __forceinline__ uint64_t rotl64d(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__global__ void k_ref_hash_owner(const uint64_t *kmers, uint64_t n, int W, int nprocs, uint64_t *hash, int32_t *owner)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    const uint64_t *w = kmers + (size_t)W * i;
    uint64_t h1 = 313, h2 = 313;
    if (W >= 2) {                                   // one 16-byte block
        uint64_t k1 = w[0], k2 = w[1];
        k1 *= c1; k1 = rotl64d(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64d(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64d(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64d(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    if (W != 2) {                                   // an 8-byte tail: the only word (W == 1) or the third (W == 3)
        uint64_t k1 = w[W - 1];
        k1 *= c1; k1 = rotl64d(k1, 31); k1 *= c2; h1 ^= k1;
    }
    const uint64_t len = 8ull * (uint64_t)W;
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = mix64(h1); h2 = mix64(h2);                 // fmix64
    h1 += h2;
    if (hash) hash[i] = h1;
    if (owner) owner[i] = (int32_t)(size_t)((double)h1 * (double)nprocs / (double)0xFFFFFFFFFFFFFFFFull);
}
}  // namespace

void stage_ref_hash_owner(Ctx &c, const uint64_t *kmers_host, int64_t n, int nprocs, uint64_t *hash_host, int32_t *owner_host)
{
    ELBA_REQUIRE(n >= 0 && (n == 0 || kmers_host) && nprocs >= 1, ELBA_ERR_INVALID_ARG, "kmer_hash_owner: bad argument");
    if (n == 0) return;
    hipStream_t s = c.stream;
    const int W = kmer_words(c.cfg.k);
    DevBuf dk, dh, dow;
    dk.reserve((size_t)n * W * 8); dh.reserve((size_t)n * 8); dow.reserve((size_t)n * 4);
    ELBA_HIP(hipMemcpyAsync(dk.p, kmers_host, (size_t)n * W * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ref_hash_owner, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dk.as<uint64_t>(), (uint64_t)n, W, nprocs, dh.as<uint64_t>(), dow.as<int32_t>());
    if (hash_host) ELBA_HIP(hipMemcpyAsync(hash_host, dh.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (owner_host) ELBA_HIP(hipMemcpyAsync(owner_host, dow.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
}

void stage_dist_count_owners(Ctx &c, int nranks, uint64_t *counts_host)
{
    ELBA_REQUIRE(c.have_reads, ELBA_ERR_STATE, "dist_count_owners: no reads");
    ELBA_REQUIRE(nranks >= 1 && nranks <= MAX_RANKS, ELBA_ERR_INVALID_ARG, "dist_count_owners: 1..64 ranks");
    hipStream_t s = c.stream;
    const int k = c.cfg.k;
    const uint64_t I = upload_instance_offsets(c);
    c.ws_scan.reserve(MAX_RANKS * 8);
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, MAX_RANKS * 8, s));
    EnumParams e = make_enum(c);
    const uint64_t nblocks = (I + EN_PER_BLOCK - 1) / EN_PER_BLOCK;
    const OwnerMap om = owner_map(c, nranks);
    if (I > 0) {
        const int W = kmer_words(k);
        if (W == 1) hipLaunchKernelGGL(k_dist_count_owners<1>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, om, c.ws_scan.as<unsigned long long>());
        else if (W == 2) hipLaunchKernelGGL(k_dist_count_owners<2>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, om, c.ws_scan.as<unsigned long long>());
        else hipLaunchKernelGGL(k_dist_count_owners<3>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, om, c.ws_scan.as<unsigned long long>());
    }
    ELBA_HIP(hipMemcpyAsync(counts_host, c.ws_scan.p, (size_t)nranks * 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
}

void stage_dist_fill_send(Ctx &c, int nranks, void *d_send, const uint64_t *offsets_host)
{
    ELBA_REQUIRE(c.have_reads && c.inst_off.p, ELBA_ERR_STATE, "dist_fill_send: call dist_count_owners first");
    ELBA_REQUIRE(nranks >= 1 && nranks <= MAX_RANKS, ELBA_ERR_INVALID_ARG, "dist_fill_send: 1..64 ranks");
    hipStream_t s = c.stream;
    c.ws_scan.reserve(MAX_RANKS * 8);
    ELBA_HIP(hipMemcpyAsync(c.ws_scan.p, offsets_host, (size_t)nranks * 8, hipMemcpyHostToDevice, s));
    EnumParams e = make_enum(c);
    const uint64_t nblocks = ((uint64_t)c.I + EN_PER_BLOCK - 1) / EN_PER_BLOCK;
    if (c.I > 0) {
        const int W = kmer_words(c.cfg.k);
        uint64_t *snd = static_cast<uint64_t *>(d_send);
        unsigned long long *cur = c.ws_scan.as<unsigned long long>();
        const OwnerMap om = owner_map(c, nranks);
        if (W == 1) hipLaunchKernelGGL(k_dist_fill_send<1>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, om, (uint64_t)c.first_global_id, snd, cur);
        else if (W == 2) hipLaunchKernelGGL(k_dist_fill_send<2>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, om, (uint64_t)c.first_global_id, snd, cur);
        else hipLaunchKernelGGL(k_dist_fill_send<3>, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, om, (uint64_t)c.first_global_id, snd, cur);
    }
    ELBA_HIP(hipStreamSynchronize(s));
}

// ---- exchange #1 with 8-byte records: format, fill, unpack (the kernels above k_split_records) ---------------------------------------------------
static PackFmt pack_format(Ctx &c, int nranks)
{
    PackFmt f{};
    f.k2 = 2 * c.cfg.k; f.nranks = (uint32_t)nranks; f.vb = c.dist_pack_vb; f.ib = c.dist_pack_ib;
    const OwnerMap om = owner_map(c, nranks);
    for (int r = 0; r < nranks; ++r) f.lo[r] = r == 0 ? 0ull : ((uint64_t)om.upper[r - 1] << (f.k2 - OWNER_BITS));
    return f;
}
// Every rank's read lengths (bounds[r] .. bounds[r + 1] = the reads of rank r): decides the packed format — the same on every rank, a pure function of
// the lengths, k and the owner ranges — and uploads every rank's instance offsets.  Returns false where an instance does not fit 64 bits.
bool stage_dist_packed_format(Ctx &c, int nranks, const int64_t *bounds, const uint32_t *all_lens, int *value_bits, int *index_bits)
{
    ELBA_REQUIRE(nranks >= 1 && nranks <= MAX_RANKS && bounds && (all_lens || bounds[nranks] == 0), ELBA_ERR_INVALID_ARG, "dist_packed_format: 1..64 ranks, bounds and lengths");
    const int k = c.cfg.k, k2 = 2 * k;
    c.dist_pack_vb = c.dist_pack_ib = 0;
    if (kmer_words(k) != 1 || k2 < OWNER_BITS) return false;
    const int64_t Mt = bounds[nranks];
    std::vector<uint64_t> off((size_t)Mt + (size_t)nranks + 1);      // rank r's offsets (reads of r + 1 entries) start at bounds[r] + r
    uint64_t maxI = 0;
    for (int r = 0; r < nranks; ++r) {
        ELBA_REQUIRE(bounds[r] <= bounds[r + 1], ELBA_ERR_INVALID_ARG, "dist_packed_format: bounds must ascend");
        uint64_t I = 0;
        uint64_t *o = off.data() + bounds[r] + r;
        for (int64_t q = bounds[r]; q < bounds[r + 1]; ++q) { o[q - bounds[r]] = I; if ((int64_t)all_lens[q] >= k) I += (uint64_t)all_lens[q] - k + 1; }
        o[bounds[r + 1] - bounds[r]] = I;
        maxI = std::max(maxI, I);
    }
    const OwnerMap om = owner_map(c, nranks);
    uint64_t maxw = 0;
    for (int r = 0; r < nranks; ++r) maxw = std::max<uint64_t>(maxw, (uint64_t)(om.upper[r] - (r ? om.upper[r - 1] : 0u)) << (k2 - OWNER_BITS));
    const int ib = bits_needed(maxI > 0 ? maxI - 1 : 0), vb = bits_needed(maxw > 0 ? maxw - 1 : 0);
    if (value_bits) *value_bits = vb;
    if (index_bits) *index_bits = ib;
    if (vb + ib > 64 || c.opt.tune[2] == 1) return false;      // (tune2 = 1: 16-byte records — A/B, tests)
    c.dist_all_off.reserve(off.size() * 8);
    ELBA_HIP(hipMemcpyAsync(c.dist_all_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, c.stream));
    ELBA_HIP(hipStreamSynchronize(c.stream));
    c.dist_bounds.assign(bounds, bounds + nranks + 1);
    c.dist_pack_vb = vb; c.dist_pack_ib = ib;
    return true;
}

void stage_dist_fill_send_packed(Ctx &c, int nranks, void *d_send, const uint64_t *offsets_host)
{
    ELBA_REQUIRE(c.have_reads && c.inst_off.p, ELBA_ERR_STATE, "dist_fill_send_packed: call dist_count_owners first");
    ELBA_REQUIRE(c.dist_pack_ib > 0 && (int)c.dist_bounds.size() == nranks + 1, ELBA_ERR_STATE, "dist_fill_send_packed: call dist_packed_format first (it must have returned 1)");
    hipStream_t s = c.stream;
    c.ws_scan.reserve(MAX_RANKS * 8);
    ELBA_HIP(hipMemcpyAsync(c.ws_scan.p, offsets_host, (size_t)nranks * 8, hipMemcpyHostToDevice, s));
    EnumParams e = make_enum(c);
    const uint64_t nblocks = ((uint64_t)c.I + EN_PER_BLOCK - 1) / EN_PER_BLOCK;
    if (c.I > 0)
        hipLaunchKernelGGL(k_dist_fill_send_packed, dim3((unsigned)nblocks), dim3(EN_THREADS), 0, s, e, owner_map(c, nranks), pack_format(c, nranks), static_cast<uint64_t *>(d_send), c.ws_scan.as<unsigned long long>());
    ELBA_HIP(hipStreamSynchronize(s));
}

// recv_counts[p] packed records of rank p, segment after segment in d_packed -> 2 words per record in d_out (what elba_dist_count_records takes); `rank` = this owner
void stage_dist_unpack_records(Ctx &c, int nranks, int rank, const void *d_packed, const uint64_t *recv_counts_host, void *d_out)
{
    ELBA_REQUIRE(c.dist_pack_ib > 0 && (int)c.dist_bounds.size() == nranks + 1 && rank >= 0 && rank < nranks, ELBA_ERR_STATE, "dist_unpack_records: call dist_packed_format first (it must have returned 1)");
    hipStream_t s = c.stream;
    const PackFmt f = pack_format(c, nranks);
    uint64_t at = 0;
    for (int p = 0; p < nranks; ++p) {
        const uint64_t n = recv_counts_host[p];
        if (n) {
            ELBA_REQUIRE(d_packed && d_out, ELBA_ERR_INVALID_ARG, "dist_unpack_records: null buffers");
            const uint32_t nr = (uint32_t)(c.dist_bounds[(size_t)p + 1] - c.dist_bounds[(size_t)p]);
            hipLaunchKernelGGL(k_dist_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const uint64_t *>(d_packed) + at, n,
                               (const uint64_t *)(c.dist_all_off.as<uint64_t>() + c.dist_bounds[(size_t)p] + p), nr, (uint64_t)c.dist_bounds[(size_t)p], f.lo[rank], f.ib, f.k2,
                               static_cast<uint64_t *>(d_out) + 2 * at);
        }
        at += n;
    }
    ELBA_HIP(hipStreamSynchronize(s));
}

// Owner side of exchange #1: count the received records exactly, keep LOWER <= count <= UPPER, sort the kept k-mers.
void stage_dist_count_records(Ctx &c, const void *d_rec, int64_t nrec)
{
    ELBA_REQUIRE(nrec >= 0 && (nrec == 0 || d_rec), ELBA_ERR_INVALID_ARG, "dist_count_records: null records");
    ELBA_REQUIRE(nrec < 0xFFFFFFF0ll, ELBA_ERR_UNSUPPORTED, "more than 2^32 records on one GPU");
    hipStream_t s = c.stream;
    const int k = c.cfg.k;
    c.have_counts = false; c.have_A = false; c.have_B = false;
    c.d_records = static_cast<const uint64_t *>(d_rec); c.nrecords = nrec;
    const uint64_t I = (uint64_t)nrec;
    // The owner counts exactly like the single-GPU path, by sorting (file header): records -> (k-mer, read << 32 | pos) pairs -> stable radix
    // sort on the k-mer -> runs -> reliable columns.  The records arrive in no particular order, so every column (<= UPPER entries) is
    // sorted by (read, pos) afterwards.
    c.ws_a.reserve((size_t)(I + 2) * 8); c.ws_b.reserve((size_t)(I + 2) * 8); c.ws_c.reserve((size_t)(I + 2) * 8); c.ws_d.reserve((size_t)(I + 2) * 8);
    c.ws_e.reserve((size_t)(I + 2) * 4); c.ws_f.reserve((size_t)(I + 2) * 8);
    uint64_t nruns = 0, N = 0, Z = 0;
    const int W = kmer_words(k);
    if (W > 1) {
        // multi-word k-mers: records of W + 1 words -> word arrays, an index permutation sorted last word first, everything gathered in that order
        DevBuf w0, w1, w2, val, i0, i1, t0, t1, s2;
        for (DevBuf *b : {&w0, &w1, &val, &i0, &i1, &t0, &t1}) b->reserve((size_t)(I + 2) * 8);
        if (W == 3) { w2.reserve((size_t)(I + 2) * 8); s2.reserve((size_t)(I + 2) * 8); }
        const unsigned nbI = (unsigned)((I + 255) / 256);
        uint64_t *shi = c.ws_a.as<uint64_t>(), *slo = c.ws_b.as<uint64_t>(), *sval = c.ws_c.as<uint64_t>(), *slo2 = W == 3 ? s2.as<uint64_t>() : nullptr;
        if (I > 0) {
            hipLaunchKernelGGL(k_split_records, dim3(nbI), dim3(256), 0, s, c.d_records, I, W + 1, w0.as<uint64_t>(), w1.as<uint64_t>(), W == 3 ? w2.as<uint64_t>() : (uint64_t *)nullptr,
                               val.as<uint64_t>(), i0.as<uint64_t>());
            const uint64_t *src[3] = {w0.as<uint64_t>(), w1.as<uint64_t>(), W == 3 ? w2.as<uint64_t>() : nullptr};
            const uint64_t *perm = sort_words_permutation(c, W, src, I, k, i0.as<uint64_t>(), i1.as<uint64_t>(), t0.as<uint64_t>(), t1.as<uint64_t>());
            hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, perm, w0.as<uint64_t>(), I, shi);
            hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, perm, w1.as<uint64_t>(), I, slo);
            if (W == 3) hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, perm, w2.as<uint64_t>(), I, slo2);
            hipLaunchKernelGGL(k_gather_u64, dim3(nbI), dim3(256), 0, s, perm, val.as<uint64_t>(), I, sval);
        }
        // (the permutation buffers are dead once the gathers are queued: runs_to_columns may use two of them as scratch)
        runs_to_columns(c, shi, sval, I, nruns, N, Z, slo, slo2);
    } else {
    if (I > 0) hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, s, c.d_records, I, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>());
    const int where = radix_sort_pairs(s, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>(), c.ws_c.as<uint64_t>(), c.ws_d.as<uint64_t>(), (int64_t)I, 64 - 2 * k, 64, c.ws_sort);
    const uint64_t *skeys = where ? c.ws_c.as<uint64_t>() : c.ws_a.as<uint64_t>(), *svals = where ? c.ws_d.as<uint64_t>() : c.ws_b.as<uint64_t>();
    runs_to_columns(c, skeys, svals, I, nruns, N, Z);
    }
    c.ws_f.reserve((size_t)(Z + 1) * 8);
    if (N > 0)
        hipLaunchKernelGGL(k_sort_columns, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, c.a_colptr.as<uint32_t>(), c.a_csc.as<uint64_t>(), c.ws_f.as<uint64_t>(), N);
    ELBA_HIP(hipStreamSynchronize(s));
    struct { uint64_t distinct; } hc{nruns};
    c.I = nrec; c.ndistinct = (int64_t)hc.distinct; c.N = (int64_t)N; c.Z = (int64_t)Z;
    c.kstats = elba_kmer_stats{};
    c.kstats.instances = nrec; c.kstats.distinct = (int64_t)hc.distinct; c.kstats.reliable = (int64_t)N; c.kstats.entries = (int64_t)Z;
    c.have_counts = true;
    c.dist_owner = true;
    // the owner's columns leave the context's A (which every panel this rank RECEIVES overwrites) for buffers of their own
    c.a_colptr.swap(c.own_colptr); c.a_csc.swap(c.own_csc);
    c.own_N = (int64_t)N; c.own_Z = (int64_t)Z;
    c.have_A = false;
}

void stage_dist_copy_reliable_kmers(Ctx &c, void *d_dst)
{
    hipStream_t s = c.stream;
    const int W = kmer_words(c.cfg.k);
    if (c.own_N > 0) {
        if (W == 1) ELBA_HIP(hipMemcpyAsync(d_dst, c.rel_kmers.p, (size_t)c.own_N * 8, hipMemcpyDeviceToDevice, s));
        else hipLaunchKernelGGL(k_join_words, dim3((unsigned)((c.own_N + 255) / 256)), dim3(256), 0, s, c.rel_kmers.as<uint64_t>(), c.rel_kmers_lo.as<uint64_t>(),
                                W == 3 ? c.rel_kmers_lo2.as<uint64_t>() : (const uint64_t *)nullptr, (uint64_t)c.own_N, W, static_cast<uint64_t *>(d_dst));
    }
    ELBA_HIP(hipStreamSynchronize(s));
}

// After the all-gather of every owner's sorted reliable k-mers: global k-mer ids of the local columns.
void stage_dist_set_global_kmers(Ctx &c, const void *d_all, int64_t nall)
{
    ELBA_REQUIRE(c.have_counts && c.dist_owner, ELBA_ERR_STATE, "dist_set_global_kmers: call dist_count_records first");
    ELBA_REQUIRE(nall >= c.own_N && nall < 0xFFFFFFF0ll, ELBA_ERR_INVALID_ARG, "dist_set_global_kmers: bad global k-mer count");
    hipStream_t s = c.stream;
    const int W = kmer_words(c.cfg.k);
    if (W > 1) {
        // d_all: nall k-mers of W interleaved words each
        DevBuf w0, w1, w2, i0, i1, t0, t1, s0, s1, s2;
        for (DevBuf *b : {&w0, &w1, &i0, &i1, &t0, &t1, &s0, &s1}) b->reserve((size_t)(nall + 2) * 8);
        if (W == 3) { w2.reserve((size_t)(nall + 2) * 8); s2.reserve((size_t)(nall + 2) * 8); }
        const unsigned nbA = (unsigned)((nall + 255) / 256);
        c.dist_gid.reserve((size_t)(c.own_N + 1) * 4);
        if (nall > 0) {
            hipLaunchKernelGGL(k_split_records, dim3(nbA), dim3(256), 0, s, static_cast<const uint64_t *>(d_all), (uint64_t)nall, W, w0.as<uint64_t>(), w1.as<uint64_t>(),
                               W == 3 ? w2.as<uint64_t>() : (uint64_t *)nullptr, (uint64_t *)nullptr, i0.as<uint64_t>());
            const uint64_t *src[3] = {w0.as<uint64_t>(), w1.as<uint64_t>(), W == 3 ? w2.as<uint64_t>() : nullptr};
            const uint64_t *perm = sort_words_permutation(c, W, src, (uint64_t)nall, c.cfg.k, i0.as<uint64_t>(), i1.as<uint64_t>(), t0.as<uint64_t>(), t1.as<uint64_t>());
            hipLaunchKernelGGL(k_gather_u64, dim3(nbA), dim3(256), 0, s, perm, w0.as<uint64_t>(), (uint64_t)nall, s0.as<uint64_t>());
            hipLaunchKernelGGL(k_gather_u64, dim3(nbA), dim3(256), 0, s, perm, w1.as<uint64_t>(), (uint64_t)nall, s1.as<uint64_t>());
            if (W == 3) hipLaunchKernelGGL(k_gather_u64, dim3(nbA), dim3(256), 0, s, perm, w2.as<uint64_t>(), (uint64_t)nall, s2.as<uint64_t>());
        }
        if (c.own_N > 0)
            hipLaunchKernelGGL(k_global_ids_words, dim3((unsigned)((c.own_N + 255) / 256)), dim3(256), 0, s, c.rel_kmers.as<uint64_t>(), c.rel_kmers_lo.as<uint64_t>(),
                               W == 3 ? c.rel_kmers_lo2.as<uint64_t>() : (const uint64_t *)nullptr, (uint64_t)c.own_N, s0.as<uint64_t>(), s1.as<uint64_t>(),
                               W == 3 ? s2.as<uint64_t>() : (const uint64_t *)nullptr, (uint64_t)nall, c.dist_gid.as<uint32_t>());
        ELBA_HIP(hipStreamSynchronize(s));
        c.dist_nall = nall;
        return;
    }
    c.ws_a.reserve((size_t)(nall + 1) * 8); c.ws_b.reserve((size_t)(nall + 1) * 8); c.ws_c.reserve((size_t)(nall + 1) * 8); c.ws_d.reserve((size_t)(nall + 1) * 8);
    if (nall > 0) ELBA_HIP(hipMemcpyAsync(c.ws_a.p, d_all, (size_t)nall * 8, hipMemcpyDeviceToDevice, s));
    int where = radix_sort_pairs(s, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>(), c.ws_c.as<uint64_t>(), c.ws_d.as<uint64_t>(), nall, 64 - 2 * c.cfg.k, 64, c.ws_sort);
    const uint64_t *sorted = where ? c.ws_c.as<uint64_t>() : c.ws_a.as<uint64_t>();
    c.dist_gid.reserve((size_t)(c.own_N + 1) * 4);
    if (c.own_N > 0)
        hipLaunchKernelGGL(k_global_ids, dim3((unsigned)((c.own_N + 255) / 256)), dim3(256), 0, s, c.rel_kmers.as<uint64_t>(), (uint64_t)c.own_N, sorted, (uint64_t)nall, c.dist_gid.as<uint32_t>());
    ELBA_HIP(hipStreamSynchronize(s));
    c.dist_nall = nall;
}

void stage_dist_panel(Ctx &c, int nranks, const uint64_t *bounds_host, const uint64_t *win_lo_host, const uint64_t *win_hi_host, bool fill, void *d_send, uint64_t *counts_or_offsets_host)
{
    ELBA_REQUIRE(c.own_N >= 0 && c.dist_nall >= 0 && c.own_colptr.p, ELBA_ERR_STATE, "dist_panel: call dist_count_records and dist_set_kmer_id_base (or dist_set_global_kmers) first");
    ELBA_REQUIRE(nranks >= 1 && nranks <= MAX_RANKS, ELBA_ERR_INVALID_ARG, "dist_panel: 1..64 ranks");
    ELBA_REQUIRE((win_lo_host == nullptr) == (win_hi_host == nullptr), ELBA_ERR_INVALID_ARG, "dist_panel: both window arrays or none");
    hipStream_t s = c.stream;
    c.ws_scan.reserve((size_t)(4 * MAX_RANKS + 2) * 8);
    unsigned long long *dcnt = c.ws_scan.as<unsigned long long>();
    uint64_t *dbounds = c.ws_scan.as<uint64_t>() + MAX_RANKS, *dwin = c.ws_scan.as<uint64_t>() + 2 * MAX_RANKS + 2;
    if (fill) ELBA_HIP(hipMemcpyAsync(dcnt, counts_or_offsets_host, (size_t)nranks * 8, hipMemcpyHostToDevice, s));
    else ELBA_HIP(hipMemsetAsync(dcnt, 0, MAX_RANKS * 8, s));
    ELBA_HIP(hipMemcpyAsync(dbounds, bounds_host, (size_t)(nranks + 1) * 8, hipMemcpyHostToDevice, s));
    if (win_lo_host) {
        for (int r = 0; r < nranks; ++r)
            ELBA_REQUIRE(win_lo_host[r] >= bounds_host[r] && win_lo_host[r] <= win_hi_host[r] && win_hi_host[r] <= bounds_host[r + 1], ELBA_ERR_INVALID_ARG, "dist_panel: a row block must lie inside its rank's rows");
        ELBA_HIP(hipMemcpyAsync(dwin, win_lo_host, (size_t)nranks * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(dwin + nranks, win_hi_host, (size_t)nranks * 8, hipMemcpyHostToDevice, s));
    }
    const uint64_t *win = win_lo_host ? dwin : nullptr;
    if (c.own_N > 0) {
        const unsigned nb = (unsigned)((c.own_N + 255) / 256);
        if (fill) hipLaunchKernelGGL((k_panel<true>), dim3(nb), dim3(256), 0, s, c.own_colptr.as<uint32_t>(), c.own_csc.as<uint64_t>(), c.dist_gid.as<uint32_t>(), (uint64_t)c.own_N, dbounds, (uint32_t)nranks, win, dcnt, static_cast<uint64_t *>(d_send));
        else hipLaunchKernelGGL((k_panel<false>), dim3(nb), dim3(256), 0, s, c.own_colptr.as<uint32_t>(), c.own_csc.as<uint64_t>(), c.dist_gid.as<uint32_t>(), (uint64_t)c.own_N, dbounds, (uint32_t)nranks, win, dcnt, (uint64_t *)nullptr);
    }
    if (!fill) ELBA_HIP(hipMemcpyAsync(counts_or_offsets_host, dcnt, (size_t)nranks * 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
}

// Receiver side of exchange #2: records (global kid, read << 32 | pos), each column contiguous and internally ordered.
// Builds the CSC panel (indexed by GLOBAL k-mer id) and the CSR of every read that appears in it; B is then computed for
// rows [row_lo, row_hi) only — the rows this rank owns, whose columns are complete by construction.
void stage_dist_set_panel(Ctx &c, const void *d_rec, int64_t nrec, int64_t M_total, int64_t N_total, int64_t row_lo, int64_t row_hi)
{
    ELBA_REQUIRE(nrec >= 0 && (nrec == 0 || d_rec), ELBA_ERR_INVALID_ARG, "dist_set_panel: null records");
    ELBA_REQUIRE(M_total >= 0 && N_total >= 0 && M_total < 0xFFFFFFFFll && N_total < 0xFFFFFFFFll && nrec < 0xFFFFFFF0ll, ELBA_ERR_UNSUPPORTED, "dist_set_panel: dimension beyond 32 bits");
    ELBA_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= M_total, ELBA_ERR_INVALID_ARG, "dist_set_panel: bad row window");
    hipStream_t s = c.stream;
    c.ws_a.reserve((size_t)(nrec + 1) * 8); c.ws_b.reserve((size_t)(nrec + 1) * 8); c.ws_c.reserve((size_t)(nrec + 1) * 8); c.ws_d.reserve((size_t)(nrec + 1) * 8);
    c.ws_e.reserve((size_t)(nrec + 1) * 8); c.ws_f.reserve((size_t)(nrec + 1) * 8);
    if (nrec > 0)
        hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, s, static_cast<const uint64_t *>(d_rec), (uint64_t)nrec, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>());
    int where = radix_sort_pairs(s, c.ws_a.as<uint64_t>(), c.ws_b.as<uint64_t>(), c.ws_c.as<uint64_t>(), c.ws_d.as<uint64_t>(), nrec, 0, bits_needed((uint64_t)(N_total > 0 ? N_total - 1 : 0)), c.ws_sort);
    // stable: a column's entries stay contiguous and ordered by (read, pos).  Move out of the ws_a..d pool (finish reuses it).
    if (nrec > 0) {
        ELBA_HIP(hipMemcpyAsync(c.ws_e.p, where ? c.ws_c.p : c.ws_a.p, (size_t)nrec * 8, hipMemcpyDeviceToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.ws_f.p, where ? c.ws_d.p : c.ws_b.p, (size_t)nrec * 8, hipMemcpyDeviceToDevice, s));
    }
    // Columns are renumbered by their rank among the columns PRESENT in this panel (ascending global id: the canonical order of a row's
    // entries, hence every seed, is unchanged): the padded column store then holds the panel's columns only, not all N_total of the run.
    int64_t N_local = 0;
    if (nrec > 0) {
        DevBuf flagbuf, exclbuf;
        flagbuf.reserve((size_t)(nrec + 2) * 4); exclbuf.reserve((size_t)(nrec + 2) * 4);
        const unsigned nb = (unsigned)((nrec + 1 + 255) / 256);
        hipLaunchKernelGGL(k_run_flags, dim3(nb), dim3(256), 0, s, c.ws_e.as<uint64_t>(), (const uint64_t *)nullptr, (const uint64_t *)nullptr, (uint64_t)nrec, flagbuf.as<uint32_t>(), 0);
        exclusive_scan_u32(s, flagbuf.as<uint32_t>(), exclbuf.as<uint32_t>(), nrec + 1, c.ws_scan);
        uint32_t nd = 0;
        ELBA_HIP(hipMemcpyAsync(&nd, exclbuf.as<uint32_t>() + nrec, 4, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(k_local_ids, dim3(nb), dim3(256), 0, s, flagbuf.as<uint32_t>(), exclbuf.as<uint32_t>(), (uint64_t)nrec, c.ws_e.as<uint64_t>());
        ELBA_HIP(hipStreamSynchronize(s));
        N_local = (int64_t)nd;
    }
    c.A_has_kmers = false;
    finish_matrix_from_sorted_csc(c, M_total, N_local, nrec, c.ws_e.as<uint64_t>(), 0, c.ws_f.as<uint64_t>(), row_lo, row_hi);
    c.N_global = N_total;
}

}  // namespace elba
