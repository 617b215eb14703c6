// kmer_msd.hip — k-mer counting for one-word k-mers (k <= 31) by a two-level value partition and per-bucket counting in LDS.
//
// Replaces the sort of kmer.hip as the way to compute what get_kmer_count_map_keys / get_kmer_count_map_values compute
// (src/KmerOps.cpp:18-350): the canonical k-mers with LOWER <= count <= UPPER and the (read, pos) of each of their instances.
//
// k <= 17 (every run recipe of the reference's README, README.md:104-109).  The LSD sort of kmer.hip moves every instance four times over HBM
// (2k = 34 value bits, 9-bit digits) and, because 34 value bits + 31 index bits do not fit one word, has to recompute k-mers afterwards.
// Here an instance moves twice and is read twice more:
//   k_msd_hist1     enumerate the k-mers of every tile of 16384 instances, count the tile's FIRST digit (the top b1 value bits)
//   k_msd_scatter   <ENUM> enumerate again and write every instance as ONE word straight into its first-digit bucket: the word
//                   no longer holds the first digit — word = remaining value bits << PB | read << pbits | pos — so read and position travel
//                   with the instance and nothing is recomputed or looked up afterwards;
//   k_msd_hist2 / k_msd_segscan / k_msd_scatter<MEM>   the same on the SECOND digit inside every first-digit bucket (tiles never straddle buckets):
//                   after it the instances sit grouped by their top b1 + b2 = 2k - 16 value bits, 2^(2k-16) buckets;
//   k_msd_count     one workgroup per bucket: the 16 value bits left index a table of 16-bit counters IN LDS (128 KB): one LDS atomic per
//                   instance gives the exact counts; the instances of reliable k-mers — the entries of A — are compacted to the front of the bucket;
//   k_msd_emit_small<8, 256 | 512 | 1024>   (buckets of up to 2048 / 4096 / 8192 entries: eight per lane) sort the bucket's entries in LDS by (value, read, pos) —
//                   by ranges of the value bits, or of (the column's rank in the bucket, the read) where UPPER allows long columns (MsdParams::rk) —: columns
//                   in value order (k-mer id = rank of the value), each ordered by (read, pos); the k-mers, the column pointers,
//                   the columns, the padded column store the SpGEMM gathers from — for the columns some row entry still fetches only: "gather
//                   slots", BucketOut —, and the CSR build's sort keys with their ownership hints and inline partners leave from LDS: everything
//                   k_runs / k_runs_emit / k_add_hints / k_fill_ell of the sort path produce.  k_msd_bucket<true>: buckets beyond that.
// HBM traffic per instance: 8 B written + 8 B read (hist2) + 8 B read + 8 B written + 2 x 8 B read = 48 B, against ~112 B on the sort path.
//
// 19 <= k <= 31 (the reference's default build is k = 31, Makefile:1-3; round 4): the section above k31_hist1 — 16-byte records, the same two-level
// partition on as many leading value bits as make buckets of ~500-2000 instances (cut by the flattened density of canonical k-mers), an LDS
// hash table per bucket for the counts; from the compacted entries on, the k <= 17 kernels.
#include "common.hpp"
#include "matrix.hpp"
#include <algorithm>
#include <functional>

namespace elba {

namespace {

#include "kmer_enum.hpp"

// (16384-key tiles, 139 KB of LDS, one workgroup per CU: on 1024 lanes x 16 keys — sixteen wavefronts to hide the barriers — the partition of
//  config 3 takes 26.4 ms, on 512 x 32 27.6 ms)
#ifndef ELBA_MT_THREADS
#define ELBA_MT_THREADS 1024
#endif
#ifndef ELBA_MT_ITEMS
#define ELBA_MT_ITEMS 16
#endif
constexpr int MT_THREADS = ELBA_MT_THREADS, MT_ITEMS = ELBA_MT_ITEMS, MT_TILE = MT_THREADS * MT_ITEMS;      // instances per tile; a wavefront's share is at most one block of the instance -> read table
constexpr int MT_MAXBITS = 9, MT_MAXBINS = 1 << MT_MAXBITS;
constexpr int VBITS = 16;                     // value bits left to the bucket kernel (two halves of 2^15 values)
constexpr int BK_THREADS = 1024;
constexpr uint32_t EW = 2048;                 // entries of a bucket half staged at a time (a window of its columns)
constexpr uint32_t KW = EW / 2 + 1;           // columns such a window can hold (a reliable column has >= 2 entries)
constexpr uint32_t EPAD = 256;                // the last column of a window may reach this far beyond it (UPPER <= 255)
constexpr uint32_t BK_TAB = 16384, BK_ENT = BK_TAB + 2048 + 128 + 2 * (KW + 1);      // word offsets in the bucket kernel's LDS (BK_ENT even: 8-byte aligned)
static_assert(BK_ENT % 2 == 0, "staged entries are 8-byte words");
constexpr size_t BK_LDS_EMIT = (size_t)BK_ENT * 4 + (size_t)(EW + EPAD) * 10;
constexpr int KPT = 8;                        // instances of a bucket a lane keeps in registers (8192 per workgroup; beyond: re-read from L2)
static_assert(MT_ITEMS * 64 <= (1 << IB_SHIFT) && MT_TILE % 4096 == 0 && MT_TILE <= 65536, "a wavefront's share of a tile lies inside one block of the instance -> read table; run-start bitmap words per lane; 16-bit ranks");

struct MsdParams {
    int k2;                 // 2k value bits
    int b1, b2;             // digit widths: b1 + b2 + VBITS == k2
    int pbits, PB;          // payload = read << pbits | pos, PB = bits of the payload
    uint64_t I;
    int rk;                 // entries carry their column's rank inside the bucket from this bit up (the count kernels write it, the emit kernels sort by it); 0: they do not
    uint32_t rkmask;        // ... in these bits (triples: the rank is the column id's low bits, the rest of the id sits above them)
    uint32_t dup;           // the bucket's words need not be distinct (triples: duplicates are kept, include/elba_amd.h) — equal words are ranked by where the first scatter put them
};

// Enumeration for the partition kernels: a lane takes 32 CONSECUTIVE instances (g = wbase + 32 * lane + it) and rolls its k-mer window along
// the read — three aligned 8-byte loads per lane instead of two per instance, ~20 integer operations per instance instead of ~40: the forward
// k-mer is the top 2k bits of a 192-bit window shifted left by one base per step, the twin takes the complement of the entering base at its
// front (src/Kmer.cpp:149-165 rolls the same way; :167-198 is the twin).  Nothing downstream of the partition needs the instances in (read, pos)
// order — the bucket kernels sort every column — so the tile may hold them in any order.  f(it, canonical k-mer, read, pos), `it` a constant.
// Where a lane's ITEMS consecutive k-mers fit ONE 64-bit window (ITEMS + k - 1 <= 32 bases: k <= 17 with sixteen instances per lane) nothing is
// rolled at all: W = the 32 bases from the lane's first position, RC = the reverse complement of all of W, and the j-th k-mer and its twin
// are two shifts and two masks — (W << 2j) & kmask, (RC << 2 (32 - k - j)) & kmask (the twin of bases j .. j+k-1 starts at comp(base j+k-1),
// which sits 31 - (j+k-1) places from the top of RC).  Two aligned 8-byte loads and ~25 operations per lane, then ~9 per instance where the
// rolling window of three words below spends ~22 (the partition's first scatter issued 82 vector instructions per instance: profiles/r04_kmer_pmc.txt).
template <int ITEMS, class F>
__device__ __forceinline__ void enum_consecutive_one_window(const EnumParams &e, const BlockInfo *block_read, uint64_t wbase, F &&f)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t g0 = wbase + (uint64_t)lane * ITEMS;
    const int k = e.k;
    const uint64_t kmask = ~0ull << (64 - 2 * k);
    if (g0 >= e.I) return;
    const uint64_t leftI = e.I - g0;
    const uint32_t nvalid = leftI < (uint64_t)ITEMS ? (uint32_t)leftI : (uint32_t)ITEMS;
    const ReadCursor rc = cursor_at(e, block_read, wbase);
    uint32_t r = rc.lo;
    uint64_t off_lo = rc.off_lo, off_hi = rc.off_hi, boff = rc.boff;
    while (g0 >= off_hi) { ++r; off_lo = off_hi; off_hi = e.inst_off[r + 1]; boff = e.byte_off[r]; }      // (g0 < I: ends inside the reads; reads shorter than k have empty ranges)
    uint32_t p = (uint32_t)(g0 - off_lo), j = 0;
    uint32_t rem = off_hi - g0 < (uint64_t)ITEMS ? (uint32_t)(off_hi - g0) : (uint32_t)ITEMS;      // instances of this read from here on (all a lane can use)
    uint64_t W = 0, RC = 0;
    auto load = [&]() {
        const uint64_t b = boff + (p >> 2), a = b & ~7ull;
        const uint64_t *w = reinterpret_cast<const uint64_t *>(e.packed + a);
        const uint64_t w0 = __builtin_bswap64(w[0]), w1 = __builtin_bswap64(w[1]);      // (16 guard bytes follow the reads)
        const uint32_t sh = (uint32_t)(b - a) * 8 + 2 * (p & 3);                       // 0..62
        W = sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
        RC = rev2bit64(~W);
    };
    load();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        if ((uint32_t)it < nvalid) {
            if (rem == 0) {      // the next read that holds instances
                off_lo = off_hi;
                while (off_hi == off_lo) { ++r; off_hi = e.inst_off[r + 1]; }
                boff = e.byte_off[r];
                p = 0; j = 0;
                rem = off_hi - off_lo < (uint64_t)ITEMS ? (uint32_t)(off_hi - off_lo) : (uint32_t)ITEMS;
                load();
            }
            const uint64_t fwd = (W << (2u * j)) & kmask, tw = (RC << (2u * (32u - (uint32_t)k - j))) & kmask;
            f(it, tw < fwd ? tw : fwd, r, p);
            ++p; ++j; --rem;
        }
    }
}

template <int ITEMS = MT_ITEMS, class F>
__device__ __forceinline__ void enum_consecutive(const EnumParams &e, const BlockInfo *block_read, uint64_t wbase, F &&f)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t g0 = wbase + (uint64_t)lane * ITEMS;
    const int k = e.k;
    const uint64_t kmask = ~0ull << (64 - 2 * k);
    if (wbase >= e.I) return;
#ifndef ELBA_ENUM_ROLL
    if (ITEMS + k - 1 <= 32) { enum_consecutive_one_window<ITEMS>(e, block_read, wbase, f); return; }
#endif
    const ReadCursor rc = cursor_at(e, block_read, wbase);
    uint32_t r = rc.lo;
    uint64_t off_lo = rc.off_lo, off_hi = rc.off_hi, boff = rc.boff;
    uint64_t hi = 0, mid = 0, lo = 0, tw = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint64_t g = g0 + (uint64_t)it;
        if (g >= e.I) break;
        uint64_t fwd;
        if (it == 0 || g >= off_hi) {
            while (g >= off_hi) { ++r; off_lo = off_hi; off_hi = e.inst_off[r + 1]; boff = e.byte_off[r]; }      // reads shorter than k have empty ranges and are skipped here
            const uint32_t p = (uint32_t)(g - off_lo);
            const uint64_t left = off_hi - g;                     // instances of this read from here on: the window must reach left (<= 32 - it) + k - 1 bases
            const uint32_t nbases = (uint32_t)(left < (uint64_t)(ITEMS - it) ? left : (uint64_t)(ITEMS - it)) + (uint32_t)k - 1u;
            const uint64_t b = boff + (p >> 2), a = b & ~7ull, last = boff + ((p + nbases - 1u) >> 2);
            const uint64_t *w = reinterpret_cast<const uint64_t *>(e.packed + a);
            const uint64_t w0 = __builtin_bswap64(w[0]), w1 = __builtin_bswap64(w[1]), w2 = a + 16 <= last ? __builtin_bswap64(w[2]) : 0ull;      // (16 guard bytes follow the reads: the third word is read only where the read itself reaches it)
            const uint32_t sh = (uint32_t)(b - a) * 8 + 2 * (p & 3);              // 0..62
            hi = sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
            mid = sh ? (w1 << sh) | (w2 >> (64 - sh)) : w1;
            lo = w2 << sh;
            fwd = hi & kmask;
            tw = twin64(fwd, k);
        } else {
            const uint64_t nb = (hi >> (62 - 2 * k)) & 3ull;      // the base that enters the window
            hi = (hi << 2) | (mid >> 62); mid = (mid << 2) | (lo >> 62); lo <<= 2;
            fwd = hi & kmask;
            tw = ((tw >> 2) & kmask) | ((3ull - nb) << 62);
        }
        f(it, tw < fwd ? tw : fwd, r, (uint32_t)(g - off_lo));
    }
}

// ---- first digit: count ---------------------------------------------------------------------------------------------------------
// (dlo, dhi: the first digits of this PASS — value-range batching, msd_run: instances of other first digits are not counted; one pass: 0, all of them)
__global__ __launch_bounds__(MT_THREADS) void k_msd_hist1(EnumParams e, const BlockInfo *block_read, MsdParams m, uint32_t *hist, uint32_t dlo, uint32_t dhi)
{
    __shared__ uint32_t h[MT_MAXBINS];
    const uint32_t nbins = 1u << m.b1;
    for (uint32_t i = threadIdx.x; i < nbins; i += MT_THREADS) h[i] = 0;
    __syncthreads();
    const uint64_t base = ((uint64_t)blockIdx.x * (MT_THREADS / 64) + (threadIdx.x >> 6)) * (uint64_t)(MT_ITEMS * 64);
    enum_consecutive(e, block_read, base, [&](int, uint64_t km, uint32_t, uint32_t) { const uint32_t d = (uint32_t)(km >> (64 - m.b1)); if (d >= dlo && d < dhi) atomicAdd(&h[d], 1u); });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += MT_THREADS) hist[(size_t)blockIdx.x * nbins + i] = h[i];
}

// per-digit totals of a tile-major histogram in 64 bits (value-range batching: more than 2^32 instances in all); blockDim.x = nbins
__global__ void k_msd_digit_totals(const uint32_t *hist, uint64_t ntiles, uint32_t nbins, unsigned long long *tot)
{
    const uint32_t d = threadIdx.x;
    unsigned long long sum = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * 256u; t < ntiles && t < (uint64_t)(blockIdx.x + 1) * 256u; ++t) sum += hist[t * nbins + d];
    if (sum) atomicAdd(&tot[d], sum);
}
__global__ void k_add_u32(uint32_t *a, uint32_t n, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += v;
}

// ---- tiles of the second pass: none of them straddles two first-digit buckets ------------------------------------------------------------
// b1start[d] = place of bucket d (row 0 of the scanned first histogram), tile0[d] = first tile of bucket d; one workgroup of 512 threads
__global__ __launch_bounds__(MT_MAXBINS) void k_msd_tiles(const uint32_t *hist1_row0, uint32_t nb1, uint64_t I, uint32_t *b1start, uint32_t *tile0)
{
    __shared__ uint32_t wsum[MT_MAXBINS / 64];
    const uint32_t d = threadIdx.x, lane = d & 63, w = d >> 6;
    const uint32_t st = d < nb1 ? hist1_row0[d] : (uint32_t)I, en = d + 1 < nb1 ? hist1_row0[d + 1] : (uint32_t)I;
    const uint32_t nt = d < nb1 ? (en - st + MT_TILE - 1) / MT_TILE : 0u;
    uint32_t inc = nt;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= (uint32_t)s2) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t run = inc - nt;
    for (uint32_t ww = 0; ww < w; ++ww) run += wsum[ww];
    if (d < nb1) { b1start[d] = st; tile0[d] = run; }
    if (d == nb1 - 1) { b1start[nb1] = (uint32_t)I; tile0[nb1] = run + nt; }
}

struct SegTiles { const uint32_t *b1start, *tile0; uint32_t nb1; const uint2 *tinfo; };      // tinfo (or null): every tile's (first key, keys), written once by k_msd_tile_info
// tile t of the second pass: its bucket (last b with tile0[b] <= t: empty buckets share their successor's first tile), first key, keys
__device__ __forceinline__ void seg_tile(const SegTiles &sg, uint32_t t, uint32_t &bucket, uint32_t &start, uint32_t &count)
{
    if (sg.tinfo) { const uint2 ti = sg.tinfo[t]; bucket = 0; start = ti.x; count = ti.y; return; }      // (ONE load: the search below is nine dependent ones, in front of every tile of a persistent workgroup)
    uint32_t lo = 0, hi = sg.nb1;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sg.tile0[mid] <= t) lo = mid; else hi = mid; }
    bucket = lo;
    start = sg.b1start[lo] + (t - sg.tile0[lo]) * (uint32_t)MT_TILE;
    const uint32_t end = sg.b1start[lo + 1];
    count = end - start < (uint32_t)MT_TILE ? end - start : (uint32_t)MT_TILE;
}

__global__ __launch_bounds__(256) void k_msd_tile_info(SegTiles sg, uint2 *tinfo)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= sg.tile0[sg.nb1]) return;
    SegTiles plain = sg; plain.tinfo = nullptr;
    uint32_t bucket, start, count;
    seg_tile(plain, t, bucket, start, count);
    tinfo[t] = make_uint2(start, count);
}

__global__ __launch_bounds__(MT_THREADS) void k_msd_hist2(const uint64_t *words, SegTiles sg, int shift, int bits, uint32_t *hist)
{
    __shared__ uint32_t h[MT_MAXBINS];
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    if (blockIdx.x >= sg.tile0[sg.nb1]) return;
    for (uint32_t i = threadIdx.x; i < nbins; i += MT_THREADS) h[i] = 0;
    __syncthreads();
    uint32_t bucket, start, count;
    seg_tile(sg, blockIdx.x, bucket, start, count);
    uint64_t k[MT_ITEMS];
#pragma unroll
    for (int r = 0; r < MT_ITEMS; ++r) { const uint32_t q = (uint32_t)r * MT_THREADS + threadIdx.x; k[r] = q < count ? words[start + q] : 0; }
#pragma unroll
    for (int r = 0; r < MT_ITEMS; ++r) { const uint32_t q = (uint32_t)r * MT_THREADS + threadIdx.x; if (q < count) atomicAdd(&h[(uint32_t)(k[r] >> shift) & dmask], 1u); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += MT_THREADS) hist[(size_t)blockIdx.x * nbins + i] = h[i];
}

// Second-digit places: one workgroup per first-digit bucket, thread d walks column d of the bucket's tile rows.  hist[t][d] becomes the place
// of tile t's first key with digit d; b2start[bucket * nb2 + d] the place of the (bucket, d) sub-bucket — the buckets the count kernel walks.
__global__ __launch_bounds__(MT_MAXBINS) void k_msd_segscan(uint32_t *hist, SegTiles sg, uint32_t nb2, uint32_t *b2start, uint64_t I)
{
    __shared__ uint32_t wsum[MT_MAXBINS / 64];
    const uint32_t b = blockIdx.x, d = threadIdx.x, lane = d & 63, w = d >> 6;
    const uint32_t t0 = sg.tile0[b], t1 = sg.tile0[b + 1];
    uint32_t run = 0;
    if (d < nb2) {
        for (uint32_t t = t0; t < t1; t += 8) {
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = t + u < t1 ? hist[(size_t)(t + u) * nb2 + d] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) { if (t + u < t1) hist[(size_t)(t + u) * nb2 + d] = run; run += x[u]; }
        }
    }
    uint32_t inc = run;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= (uint32_t)s2) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = sg.b1start[b] + inc - run;
    for (uint32_t ww = 0; ww < w; ++ww) base += wsum[ww];
    if (d < nb2) {
        b2start[(size_t)b * nb2 + d] = base;
        for (uint32_t t = t0; t < t1; t += 8) {
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = t + u < t1 ? hist[(size_t)(t + u) * nb2 + d] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (t + u < t1) hist[(size_t)(t + u) * nb2 + d] = x[u] + base;
        }
    }
    if (b == gridDim.x - 1 && d == 0) b2start[(size_t)gridDim.x * nb2] = (uint32_t)I;
}

#ifdef ELBA_SCATTER_CLOCK      // diagnostic build only: shader-clock cycles of wavefront 0 (and 1) per phase of k_msd_scatter, summed over workgroups and tiles
__device__ unsigned long long g_sc_phase[2][2][12];
#define ELBA_SSTAMP(k) do { if (lane == 0 && w < 2) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); sph[k] += tn - stp; stp = tn; } } while (0)
#else
#define ELBA_SSTAMP(k) do { } while (0)
#endif
__device__ __forceinline__ void lds_sync_fwd() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // LDS-only workgroup barrier (lds_sync below)
// ---- stable scatter of one tile by one digit --------------------------------------------------------------------------------------
// ENUM: the tile's keys are enumerated from the reads (tile = 8192 consecutive instances) and the digit is the top b1 value bits, which the word
// written does not hold any more.  !ENUM: the tile's keys are read from `in` (a bucket-aligned tile) and the digit is (word >> shift) & mask.
// The tile is ordered by digit in LDS and written out by consecutive lanes (a digit's keys of one tile are one contiguous run in the output).
// The grid is PERSISTENT (ntiles tiles dealt round-robin to the workgroups, one workgroup per CU: the tile fills its LDS): a workgroup per tile paid the
// dispatch of sixteen wavefronts and 139 KB of LDS every ~20 us.
template <bool ENUM, bool FILTER = false>      // FILTER: value-range batching — only the instances whose first digit lies in [dlo, dhi) (an instantiation of its own: the test costs the one-pass kernel 7 %)
__global__ __launch_bounds__(MT_THREADS) void k_msd_scatter(EnumParams e, const BlockInfo *block_read, MsdParams m, const uint64_t *in, SegTiles sg, int shift, int bits,
                                                           const uint32_t *hist_scanned, uint64_t *out, uint32_t ntiles, uint32_t dlo, uint32_t dhi)
{
    __shared__ uint32_t kept_s;      // keys of the tile that belong to this pass (value-range batching: k_msd_hist1)
    constexpr int WAVES = MT_THREADS / 64, DPT = MT_MAXBINS / MT_THREADS > 0 ? MT_MAXBINS / MT_THREADS : 1;
    __shared__ uint32_t lcnt[MT_MAXBINS], lstart[MT_MAXBINS], gbase[MT_MAXBINS], wsum[WAVES];
    __shared__ uint64_t lkey[MT_TILE];
    __shared__ unsigned long long hbits[MT_TILE / 64];
    __shared__ uint32_t hpre[MT_TILE / 64], delta[MT_MAXBINS];
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    if (!ENUM) ntiles = sg.tile0[sg.nb1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long lane_le = (2ull << lane) - 1ull;      // (a place t = it * MT_THREADS + tid of the write-out has t & 63 == lane)
    // The keys of a tile live in registers from where they are fetched — enumerated from the reads, or loaded from the first pass's output — to where they
    // are placed in LDS.  Round 5 (ELBA_SCATTER_NO_PIPE: off): the NEXT tile is fetched in the shadow of this tile's write-out — the key registers are dead by
    // then — so that the loads' latency (MEM) hides behind the write-out, and, ENUM, half the wavefronts enumerate (vector ALU) while the other half
    // write out (LDS reads + global stores) instead of all sixteen doing the one and then the other: partition of config 3 26.4 -> 24.1 ms.
#ifdef ELBA_SCATTER_CLOCK
    unsigned long long sph[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, stp = __builtin_amdgcn_s_memtime();
#endif
    uint64_t key[MT_ITEMS];
    uint32_t dig2[MT_ITEMS / 2];    // the items' digits, two per register (0xFFFF: no key)
    uint32_t count = 0, ncount = 0; (void)ncount;
    auto digit = [&](int it) -> uint32_t { return (dig2[it >> 1] >> ((it & 1) * 16)) & 0xFFFFu; };
    auto set_digit = [&](int it, uint32_t d) { if (it & 1) dig2[it >> 1] |= d << 16; else dig2[it >> 1] = d; };
    auto fetch = [&](uint32_t t, uint32_t &cnt) {
        if (ENUM) {
            const uint64_t tbase = (uint64_t)t * MT_TILE, base = tbase + (uint64_t)w * (MT_ITEMS * 64);
            const uint64_t left = e.I - tbase;
            cnt = left < (uint64_t)MT_TILE ? (uint32_t)left : (uint32_t)MT_TILE;
            const uint64_t vmask = (1ull << (m.k2 - m.b1)) - 1;
#pragma unroll
            for (int it = 0; it < MT_ITEMS; ++it) key[it] = 0;
#pragma unroll
            for (int q = 0; q < MT_ITEMS / 2; ++q) dig2[q] = 0xFFFFFFFFu;
            enum_consecutive(e, block_read, base, [&](int it, uint64_t km, uint32_t r, uint32_t p) {
                const uint32_t dg = (uint32_t)(km >> (64 - m.b1));
                if (FILTER && (dg < dlo || dg >= dhi)) return;      // (another pass's instance: no key)
                key[it] = (((km >> (64 - m.k2)) & vmask) << m.PB) | ((uint64_t)r << m.pbits) | p;
                dig2[it >> 1] = (it & 1) ? (dig2[it >> 1] & 0xFFFFu) | (dg << 16) : (dig2[it >> 1] & 0xFFFF0000u) | dg;
            });
        } else {
            uint32_t bucket, start;
            seg_tile(sg, t, bucket, start, cnt);
            const uint32_t wb = (uint32_t)w * (MT_ITEMS * 64);
#pragma unroll
            for (int it = 0; it < MT_ITEMS; ++it) {
                const uint32_t q = wb + (uint32_t)it * 64 + lane;
                key[it] = q < cnt ? in[start + q] : 0;
            }
        }
    };
    auto digits_of_loaded = [&](uint32_t cnt) {      // (MEM: the digits follow from the words once they have arrived)
        if (ENUM) return;
        const uint32_t wb = (uint32_t)w * (MT_ITEMS * 64);
#pragma unroll
        for (int it = 0; it < MT_ITEMS; ++it) {
            const uint32_t q = wb + (uint32_t)it * 64 + lane;
            set_digit(it, q < cnt ? (uint32_t)(key[it] >> shift) & dmask : 0xFFFFu);
        }
    };
#ifndef ELBA_SCATTER_ENUM_MODE
#define ELBA_SCATTER_ENUM_MODE 2      // ENUM: 0 = the tile is enumerated at the top of its own turn, 1 = every wavefront enumerates the next tile before it writes this one out, 2 = half of them before, half after
#endif
    constexpr bool PIPE = !ENUM || ELBA_SCATTER_ENUM_MODE != 0;
    // Which tiles a workgroup takes.  A digit's runs of CONSECUTIVE tiles lie one behind the other in the output, 256 bytes each at any 8-byte alignment: the
    // 64-byte lines at their seams are written half by one tile and half by the next.  Workgroups go to the XCDs round-robin (blockIdx mod 8), each XCD
    // has its own L2: dealt round-robin, neighbouring tiles never share an L2 and every seam line leaves two L2s as a partial line.  Each XCD therefore takes a
    // BLOCK of consecutive tiles per round (its 32 workgroups walk 32 neighbouring tiles at the same time): the seams inside the block merge in its L2.
#ifndef ELBA_SCATTER_RR
    const uint32_t first_tile = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#else
    const uint32_t first_tile = blockIdx.x;
#endif
#ifndef ELBA_SCATTER_NO_PIPE
    if (PIPE && first_tile < ntiles) fetch(first_tile, count);
#endif
#ifndef ELBA_SCATTER_ONE_TILE      // (the persistent grid costs the ENUM kernel 50 registers — 228 bytes of scratch — and is still 0.4 ms ahead of a workgroup per tile: 25.1 vs 25.5 ms for the partition)
    for (uint32_t tile = first_tile; tile < ntiles; tile += gridDim.x) {
#else
    const uint32_t tile = blockIdx.x;
    if (tile < ntiles) {
#endif
    ELBA_SSTAMP(0);
    for (int i = threadIdx.x; i < MT_MAXBINS; i += MT_THREADS) lcnt[i] = 0;
    for (int i = threadIdx.x; i < MT_TILE / 64; i += MT_THREADS) hbits[i] = 0;
    __syncthreads();
    ELBA_SSTAMP(1);
#ifdef ELBA_SCATTER_NO_PIPE
    fetch(tile, count);
#else
    if (!PIPE) fetch(tile, count);
#endif
    digits_of_loaded(count);
    // (the tile's row of output places: one coalesced load, in flight while the ranks are computed)
    uint32_t gb[DPT];
#pragma unroll
    for (int u = 0; u < DPT; ++u) { const uint32_t d = threadIdx.x + u * MT_THREADS; gb[u] = d < nbins ? hist_scanned[(size_t)tile * nbins + d] : 0u; }
    // rank of every key among the tile's keys with its digit: ONE returning LDS atomic per key on the workgroup's digit counters.  (Nothing
    // downstream needs the partition to be stable — the bucket kernels sort every column — so the keys of a digit may land in any order; the
    // ballot ranking of the radix sort, ~70 instructions per key, bought an order nobody reads.)
    uint16_t rank[MT_ITEMS];
#pragma unroll
    for (int it = 0; it < MT_ITEMS; ++it) {
        const uint32_t d = digit(it);
        rank[it] = 0;
        if (d != 0xFFFFu) rank[it] = (uint16_t)atomicAdd(&lcnt[d], 1u);
    }
    ELBA_SSTAMP(2);
    __syncthreads();
    ELBA_SSTAMP(3);
    {
        // exclusive scan of the digit counts: the digits' places in the tile
        uint32_t tot[DPT], both = 0;
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = DPT * threadIdx.x + u; tot[u] = d < nbins ? lcnt[d] : 0u; both += tot[u]; }
        uint32_t inc = both;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= s2) inc += o; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t run = inc - both;
        for (int ww = 0; ww < w; ++ww) run += wsum[ww];
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = DPT * threadIdx.x + u; if (d < nbins) lstart[d] = run; run += tot[u]; }
        if (threadIdx.x == MT_THREADS - 1) kept_s = run;      // (the inclusive sum at the last thread: every key of the tile that has a digit)
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = threadIdx.x + u * MT_THREADS; if (d < nbins) gbase[d] = gb[u]; }
    }
    __syncthreads();
    ELBA_SSTAMP(4);
    if (ENUM && FILTER) count = kept_s;
#pragma unroll
    for (int it = 0; it < MT_ITEMS; ++it) {
        const uint32_t d = digit(it);
        if (d != 0xFFFFu) lkey[lstart[d] + rank[it]] = key[it];
    }
    ELBA_SSTAMP(5);
#ifdef ELBA_SCATTER_MEM_DIRECT
    if (!ENUM) {
        // a word that still holds its digit finds its run by it: delta[digit] = the run's place in the output - its place in the tile; the places of
        // the tile are walked with a compile-time trip count, so that the LDS reads of all of a lane's places are in flight together (a loop with a
        // run-time bound exposed three dependent LDS round trips per key: ~4 us of a tile's ~28)
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = DPT * threadIdx.x + u; if (d < nbins) delta[d] = gbase[d] - lstart[d]; }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MT_ITEMS; ++it) {
            const uint32_t t = (uint32_t)it * MT_THREADS + threadIdx.x;
            if (t < count) { const uint64_t kv = lkey[t]; out[delta[(uint32_t)(kv >> shift) & dmask] + t] = kv; }
        }
        __syncthreads();
#ifndef ELBA_SCATTER_ONE_TILE
        continue;
#else
        return;
#endif
    }
#endif
    // The tile now lies ordered by digit in LDS, but an ENUM word does not hold its digit any more.  A place finds its digit's run from a
    // bitmap of the run starts: run number = set bits at or before the place, delta[run] = the run's place in the output - its place in the tile.
    {
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const uint32_t d = DPT * threadIdx.x + u;
            if (d < nbins) { const uint32_t ls = lstart[d], le = d + 1 < nbins ? lstart[d + 1] : count; if (le > ls) atomicOr(&hbits[ls >> 6], 1ull << (ls & 63u)); }
        }
    }
    __syncthreads();
    if (w == 0) {
        constexpr int HPL = MT_TILE / 64 / 64;      // bitmap words per lane
        uint32_t cw[HPL], tot = 0;
#pragma unroll
        for (int q = 0; q < HPL; ++q) { cw[q] = (uint32_t)__popcll(hbits[HPL * lane + q]); tot += cw[q]; }
        uint32_t inc = tot;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= s2) inc += o; }
        uint32_t run = inc - tot;
#pragma unroll
        for (int q = 0; q < HPL; ++q) { hpre[HPL * lane + q] = run; run += cw[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
        const uint32_t d = DPT * threadIdx.x + u;
        if (d < nbins) {
            const uint32_t ls = lstart[d], le = d + 1 < nbins ? lstart[d + 1] : count;
            if (le > ls) delta[hpre[ls >> 6] + (uint32_t)__popcll(hbits[ls >> 6] & ((1ull << (ls & 63u)) - 1ull))] = gbase[d] - ls;
        }
    }
    __syncthreads();
    ELBA_SSTAMP(6);
    auto writeout = [&]() {
#ifdef ELBA_SCATTER_UNROLL
#pragma unroll
    for (int it = 0; it < MT_ITEMS; ++it) {      // (compile-time trip count: the LDS reads of a lane's places are in flight together)
        const uint32_t t = (uint32_t)it * MT_THREADS + threadIdx.x;
        if (t < count) {
            const uint32_t run = hpre[t >> 6] + (uint32_t)__popcll(hbits[t >> 6] & ((2ull << (t & 63u)) - 1ull)) - 1u;
            out[delta[run] + t] = lkey[t];
        }
    }
#else
    // (a compile-time trip count — all of a lane's LDS reads in flight together — was measured: partition 28.0 -> 30.0 ms; the 32 keys it keeps
    //  in registers beside the tile's cost more than the round trips they hide)
#if defined(ELBA_X_SCATTER) && ELBA_X_SCATTER == 1      // timing experiment (wrong results): the tile leaves as ONE contiguous run
    for (uint32_t t = threadIdx.x; t < count; t += MT_THREADS) out[(size_t)tile * MT_TILE + t] = lkey[t];
#elif defined(ELBA_X_SCATTER) && ELBA_X_SCATTER == 2    // timing experiment (wrong results): nothing leaves
    for (uint32_t t = threadIdx.x; t < count; t += MT_THREADS) {
        const uint32_t run = hpre[t >> 6] + (uint32_t)__popcll(hbits[t >> 6] & lane_le) - 1u;
        if (delta[run] == 0xFFFFFFFEu) out[t] = lkey[t];
    }
#else
    // (a place's run costs two dependent LDS round trips — bitmap word, then the run's delta — in a loop with a run-time trip count: two places per trip keep two chains in
    //  flight: ENUM 10.25 -> 9.87 ms; MEM 8.54 -> 8.76: its trips stay single; four: no better)
    if (ENUM) {
#pragma unroll 2
        for (uint32_t t = threadIdx.x; t < count; t += MT_THREADS) {
            const uint32_t run = hpre[t >> 6] + (uint32_t)__popcll(hbits[t >> 6] & lane_le) - 1u;
            out[delta[run] + t] = lkey[t];
        }
    } else
    for (uint32_t t = threadIdx.x; t < count; t += MT_THREADS) {
        const uint32_t run = hpre[t >> 6] + (uint32_t)__popcll(hbits[t >> 6] & lane_le) - 1u;
        out[delta[run] + t] = lkey[t];
    }
#endif
#endif
    };
#if !defined(ELBA_SCATTER_NO_PIPE) && !defined(ELBA_SCATTER_ONE_TILE)
    {
        const uint32_t nxt = tile + gridDim.x;
        const bool has = nxt < ntiles;
        if (!PIPE) writeout();
        else if (ENUM && ELBA_SCATTER_ENUM_MODE == 2 && (w & 1)) { writeout(); ELBA_SSTAMP(7); if (has) fetch(nxt, ncount); ELBA_SSTAMP(8); }
        else { if (has) fetch(nxt, ncount); ELBA_SSTAMP(8); writeout(); ELBA_SSTAMP(7); }
    }
#else
    writeout();
#endif
    lds_sync_fwd();      // (the tile's LDS is reused by the workgroup's next tile; its global stores stay in flight)
    ELBA_SSTAMP(9);
#ifndef ELBA_SCATTER_NO_PIPE
    if (PIPE) count = ncount;
#endif
    }
#ifdef ELBA_SCATTER_CLOCK
    if (lane == 0 && w < 2) for (int q = 0; q < 12; ++q) atomicAdd(&g_sc_phase[ENUM ? 1 : 0][w][q], sph[q]);
#endif
}

// LDS-only workgroup barrier: the global stores of a bucket (never read back by the workgroup) stay in flight
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BucketStats { unsigned long long distinct, sumsq; unsigned int maxcol, ncrowded, nmid, nbig; };      // nmid / nbig: buckets of 4097..8192 / 8193..ES_CAP_MAX entries
constexpr uint32_t ES_CAP_MAX = 12288;      // entries of a bucket the widest emit kernel sorts in LDS
struct BucketOut {
    uint64_t *rel_kmers; uint32_t *rel_counts, *colptr;
    uint64_t *csc, *csr_words, *kid_of_entry, *ell;
    uint32_t ell_stride;      // 0: no padded column store
    int nb, pb, rs, mb;       // CSR sort key: read << rs | kid << (pb + 2) | hint << pb | pos   (rs >= nb + pb + 2; mb = bits of a read id)
    uint32_t hints;           // write the ownership bits (Ctx::csr_hints)
    uint32_t inl;             // Ctx::csr_inline (0: off, else pbi = position bits of an inline key): the owning row's key of a two-read column is
                              // 1 << 63 | read << rs | (partner >> 1) << 2 pbi | posQ << pbi | posT — when both positions fit pbi bits
    // Gather slots (Ctx::ell_compact, with inline partners): four columns in five are never fetched by the SpGEMM — every entry of theirs either
    // carries its product inline or is hinted "no pair owned".  The padded store then holds only the columns some entry still gathers, one after
    // the other in the order the buckets draw their slots (slot_cursor; a workgroup draws SLOT_CHUNK at a time), and a row entry that gathers
    // names its column's SLOT where the others name the k-mer id (the SpGEMM multiplies whichever it finds by the stride; a sequence number is
    // the entry's rank in its row, never the id).  slot_kid[slot] = k-mer id, for exports and tests.  8 N S bytes were written before (14.8 GB on
    // BASELINE config 3), 8 S per gathered column now.
    unsigned long long *slot_cursor; uint32_t *slot_kid; uint32_t compact, slot_chunk;      // slot_cursor[1] != 0: a draw ran past slot_cap (nothing was written there; the host emits again without slots)
    unsigned long long slot_cap;
    uint32_t *pair_key; uint64_t *pair_val; // dense matrices (Ctx::csr_suffix): the pairs the CSR build sorts by read — read; k-mer id << 32 | column length << 23 | place in the column << 16 | pos — instead of sort words
    const uint64_t *kmer_src;      // k > 17 (k31_count): the bucket's reliable k-mers, left-aligned, at [bucket's first instance + column]; null: the k-mer is bucket << 16 | the entry's 16 value bits
    const uint32_t *ncols;         // columns per bucket where consecutive buckets' k-mer id bases do not follow one another (pseudo-buckets); null: kidbase[b + 1] - kidbase[b]
    const uint64_t *kmer_dist; const uint32_t *dist_base;      // pseudo-buckets of a crowded wide bucket (k31_crowded_*): the entry's 16 value bits are the k-mer's rank among the pseudo-bucket's DISTINCT k-mers — the k-mer is kmer_dist[dist_base[bucket] + those bits]
};

// ---- buckets: count ------------------------------------------------------------------------------------------------------------------
// One workgroup per bucket (the instances whose top 2k - 16 value bits agree).  The 16 value bits left index a table of 16-bit counters in LDS
// (two per word, 128 KB): one LDS atomic per instance gives the exact counts.  Every instance then looks its value's count up; reliable values
// (LOWER <= count <= UPPER) are marked in a bitmap, the values seen at all in another (the number of distinct k-mers), and the instances of
// reliable values — the entries of A — are copied, in no particular order, to the front of the bucket's place in `wrel`: the emit kernels read
// those alone (a quarter of the instances on 15 %-error reads).  Out: reliable k-mers and entries per bucket (the scan over the buckets gives the
// k-mer ids and column pointers), buckets too crowded for the small emit kernel.
constexpr uint32_t CT_TAB = 32768, CT_BITS = 2048;      // words
constexpr size_t CT_LDS = (size_t)(CT_TAB + 3 * CT_BITS + 64) * 4;
// (the count kernel keeps TWELVE instances per lane: a bucket of BASELINE config 3 holds 7600 on average, more than 8192 often enough — those were
//  read three times — bucket kernels 17.6 -> 17.1 ms; sixteen spill)
constexpr int CT_KPT = 12;
template <bool RANK>      // RANK: the entries leave with their column's rank in the bucket (MsdParams::rk) — one more pass over the bucket's instances, which only long columns repay
__global__ __launch_bounds__(BK_THREADS) void k_msd_count(const uint64_t *words, const uint32_t *b2start, uint32_t nbuckets, MsdParams m, uint32_t lower, uint32_t upper, uint32_t small_cap,
                                                         uint32_t *bN, uint32_t *bZ, BucketStats *gstat, uint32_t *crowded, uint64_t *wrel)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *tab = smem;
    const uint16_t *tab16 = reinterpret_cast<const uint16_t *>(smem);
    uint32_t *relbits = smem + CT_TAB, *seenbits = relbits + CT_BITS, *misc = seenbits + CT_BITS, *pre = misc + 64;      // pre[w]: reliable values below word w of the bitmap
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    unsigned long long st_distinct = 0, st_sumsq = 0;
    uint32_t st_maxcol = 0;
    const uint64_t lowmask = m.rk ? (1ull << m.rk) - 1 : ~0ull;
    // the instances of the NEXT bucket are requested before this one is processed (one workgroup per CU: nobody else hides the round trip)
    uint32_t b = blockIdx.x, s0 = 0, n = 0;
    // (a lane holds PAIRS of neighbouring instances: 16-byte requests)
    struct __attribute__((aligned(8))) Two { uint64_t a, c; };
    auto ld2 = [&](const uint64_t *src, uint32_t cnt, uint32_t u, uint64_t &a, uint64_t &c) {
        const uint32_t i = ((u >> 1) * BK_THREADS + tid) * 2u;
        if (i + 1u < cnt) { const Two t = *reinterpret_cast<const Two *>(src + i); a = t.a; c = t.c; }
        else if (i < cnt) a = src[i];
    };
    uint64_t kreg[CT_KPT];
#pragma unroll
    for (int u = 0; u < CT_KPT; ++u) kreg[u] = 0;
    if (b < nbuckets) {
        s0 = b2start[b]; n = b2start[b + 1] - s0;
#pragma unroll
        for (int u = 0; u < CT_KPT; u += 2) ld2(words + s0, n, (uint32_t)u, kreg[u], kreg[u + 1]);
    }
    for (; b < nbuckets;) {
        const uint32_t bnext = b + gridDim.x;
        uint32_t s0n = 0, nn = 0;
        uint64_t knext[CT_KPT];
#pragma unroll
        for (int u = 0; u < CT_KPT; ++u) knext[u] = 0;
        if (bnext < nbuckets) {
            s0n = b2start[bnext]; nn = b2start[bnext + 1] - s0n;
#pragma unroll
            for (int u = 0; u < CT_KPT; u += 2) ld2(words + s0n, nn, (uint32_t)u, knext[u], knext[u + 1]);
        }
        if (n == 0) { if (tid == 0) { bN[b] = 0; bZ[b] = 0; } }
        else {
        {   // zero the table and the two bitmaps (36 K words = 9 uint4 per lane), and the counters
            uint4 *t4 = reinterpret_cast<uint4 *>(smem);
#pragma unroll
            for (int u = 0; u < 9; ++u) t4[(uint32_t)u * BK_THREADS + tid] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < 32) misc[tid] = 0;
        }
        lds_sync();
        const bool guard = n > 65535u;          // a value's count could run over its 16 bits: stop counting beyond 2^15 (UPPER <= 255: unreliable anyway)
        auto for_keys = [&](auto &&f) {
#pragma unroll
            for (int u = 0; u < CT_KPT; ++u) if (((uint32_t)(u >> 1) * BK_THREADS + tid) * 2u + (uint32_t)(u & 1) < n) f(kreg[u]);
            for (uint32_t i = (uint32_t)CT_KPT * BK_THREADS + tid; i < n; i += BK_THREADS) f(words[s0 + i]);
        };
        for_keys([&](uint64_t wd) {
            const uint32_t v = (uint32_t)(wd >> m.PB) & 0xFFFFu;
            if (guard && tab16[v] >= 0x8000u) return;
            atomicAdd(&tab[v >> 1], 1u << ((v & 1u) * 16u));
        });
        lds_sync();
        if (!RANK) {
        for_keys([&](uint64_t wd) {
            const uint32_t v = (uint32_t)(wd >> m.PB) & 0xFFFFu, cnt = tab16[v];
            atomicOr(&seenbits[v >> 5], 1u << (v & 31u));
            if (cnt >= lower && cnt <= upper) {
                atomicOr(&relbits[v >> 5], 1u << (v & 31u));
                wrel[s0 + atomicAdd(&misc[0], 1u)] = wd;
            }
        });
        lds_sync();
        uint32_t nrel = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t wi = (uint32_t)q * BK_THREADS + tid, mybits = relbits[wi];
            nrel += (uint32_t)__popc(mybits);
            st_distinct += (unsigned long long)__popc(seenbits[wi]);
            for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                const uint32_t cnt = tab16[wi * 32u + (uint32_t)__ffs((int)bits) - 1u];
                st_sumsq += (unsigned long long)cnt * cnt; st_maxcol = cnt > st_maxcol ? cnt : st_maxcol;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nrel += __shfl_xor(nrel, d, 64);
        if (lane == 0 && nrel) atomicAdd(&misc[1], nrel);
        lds_sync();
        } else {
        for_keys([&](uint64_t wd) {
            const uint32_t v = (uint32_t)(wd >> m.PB) & 0xFFFFu, cnt = tab16[v];
            atomicOr(&seenbits[v >> 5], 1u << (v & 31u));
            if (cnt >= lower && cnt <= upper) atomicOr(&relbits[v >> 5], 1u << (v & 31u));
        });
        lds_sync();
        // the reliable values in value order: lane t owns words 2t, 2t + 1 of the bitmap; a scan over the lanes numbers them (the column's rank in the bucket)
        uint32_t nrel = 0, c0 = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t wi = 2u * tid + (uint32_t)q, mybits = relbits[wi];
            if (q == 0) c0 = (uint32_t)__popc(mybits);
            nrel += (uint32_t)__popc(mybits);
            st_distinct += (unsigned long long)__popc(seenbits[wi]);
            for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                const uint32_t cnt = tab16[wi * 32u + (uint32_t)__ffs((int)bits) - 1u];
                st_sumsq += (unsigned long long)cnt * cnt; st_maxcol = cnt > st_maxcol ? cnt : st_maxcol;
            }
        }
        {
            uint32_t inc = nrel;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
            if (lane == 63) misc[16 + (tid >> 6)] = inc;
            lds_sync();
            uint32_t ex = inc - nrel;
            for (uint32_t ww = 0; ww < (tid >> 6); ++ww) ex += misc[16 + ww];
            pre[2u * tid] = ex; pre[2u * tid + 1u] = ex + c0;
            if (tid == BK_THREADS - 1) misc[1] = ex + nrel;
        }
        lds_sync();
        for_keys([&](uint64_t wd) {
            const uint32_t v = (uint32_t)(wd >> m.PB) & 0xFFFFu, bits = relbits[v >> 5];
            if ((bits >> (v & 31u)) & 1u) {
                const uint64_t rc = pre[v >> 5] + (uint32_t)__popc(bits & ((1u << (v & 31u)) - 1u));
                wrel[s0 + atomicAdd(&misc[0], 1u)] = m.rk ? (wd & lowmask) | (rc << m.rk) : wd;
            }
        });
        lds_sync();
        }
        if (tid == 0) {
            const uint32_t Zb = misc[0], Nb = misc[1];
            bN[b] = Nb; bZ[b] = Zb;
            if (Zb > small_cap) crowded[atomicAdd(&gstat->ncrowded, 1u)] = b;
            else if (Zb > 8192u) atomicAdd(&gstat->nbig, 1u);
            else if (Zb > 4096u) atomicAdd(&gstat->nmid, 1u);
        }
        lds_sync();                               // (misc is zeroed for the next bucket)
        }
#pragma unroll
        for (int u = 0; u < CT_KPT; ++u) kreg[u] = knext[u];
        b = bnext; s0 = s0n; n = nn;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        st_distinct += __shfl_xor(st_distinct, d, 64); st_sumsq += __shfl_xor(st_sumsq, d, 64);
        const uint32_t o2 = __shfl_xor(st_maxcol, d, 64); st_maxcol = o2 > st_maxcol ? o2 : st_maxcol;
    }
    if (lane == 0) { if (st_distinct) atomicAdd(&gstat->distinct, st_distinct); if (st_sumsq) atomicAdd(&gstat->sumsq, st_sumsq); if (st_maxcol) atomicMax(&gstat->maxcol, st_maxcol); }
}

// ---- buckets: emit (the usual case: at most ES_CAP entries) --------------------------------------------------------------------------------
// The bucket's entries (wrel: instances of reliable k-mers, word = ... value16 << PB | read << pbits | pos, all with the same bits above) are
// SORTED as 64-bit words in LDS — that is by (k-mer value, read, pos): columns in value order, each ordered by (read, pos) — with a bucket + rank
// sort (the 16 value bits are spread evenly over a bucket: ~Z/8 equal value ranges, count, scan, scatter, then every entry ranks itself among
// the handful that share its range).  A run of equal values is a column: its number (k-mer id = rank of the value) is the
// number of run heads before it, its pointer the head's place.  Everything the sort path's k_runs_emit / k_add_hints / k_fill_ell produce
// leaves from here as coalesced streams.  Eight entries per lane; instantiated on 256 lanes (buckets of up to 2048 entries: 27 KB of LDS, five
// workgroups per CU), 512 (up to 4096: 50 KB, three) and 1024 (up to 8192 — the canonical k-mer is the smaller of two: low values are twice as
// dense as the average, and the fullest buckets of a large input, or most buckets of deep low-error coverage, land here —: 99 KB, one workgroup of
// sixteen wavefronts).  History of the widest class on BASELINE config 5 at one GPU's share (average bucket 4400 entries): 256 lanes x 32 entries
// (256 VGPRs, four wavefronts per CU) 85 ms; 512 x 16 44 ms; with ranges by column rank 20 ms; 1024 x 8 13 ms.
// SPEC: the instantiation for what the reads path of k <= 17 always builds — no column ranks, distinct words, gather slots, hints, inline partners, one-word CSR sort
// keys, k-mers from the bucket number — with those switches compiled in: the general kernel keeps ~75 kernel-argument scalars alive and spilled them to vector lanes
// (472 v_readlane of 4000 vector instructions per bucket and lane: the emit kernels are bound by instruction issue, profiles/r05_notes.md)
template <int ES_KPT, int ES_THREADS = 256, bool SPEC = false>
#ifndef ELBA_ES_OCC256
#define ELBA_ES_OCC256 5      // (92 VGPRs instead of 100: five workgroups per CU instead of four)
#endif
#ifndef ELBA_ES_OCC512
#define ELBA_ES_OCC512 6      // (80 VGPRs + 32 bytes of scratch instead of 99: THREE workgroups per CU — what the 49 KB of LDS allow — instead of two: bucket kernels of config 3 17.6 -> 16.1 ms)
#endif
// (second launch bound = wavefronts per SIMD the register allocation leaves room for; 0: whatever the kernel needs)
__global__ __launch_bounds__(ES_THREADS, (ES_THREADS == 256 ? ELBA_ES_OCC256 : (ES_THREADS == 512 ? ELBA_ES_OCC512 : 0))) void k_msd_emit_small(const uint64_t *wrel, const uint32_t *b2start, const uint32_t *bZ, uint32_t nbuckets, MsdParams m, uint32_t cap_lo, uint32_t cap_hi,
                                                              const uint32_t *kidbase, const uint32_t *entbase, BucketOut o)
{
    constexpr uint32_t ES_CAP = ES_THREADS * ES_KPT, NW = ES_THREADS / 64, NH = ES_KPT * NW;      // entries; wavefronts; (u, wavefront) head counts
    // (ranges of the sort; a quarter of the capacity — 1024 / 2048 ranges for the larger classes, ~4 entries each instead of ~8 — was measured in round 5: no difference)
    constexpr uint32_t ES_NSB = ES_CAP >= 8192 ? 1024 : 512;
    constexpr int NSB_BITS = ES_CAP >= 8192 ? 10 : 9;
    constexpr int PER = (int)(ES_NSB / ES_THREADS);      // ranges per lane in their scan
    static_assert(PER >= 1 && ES_NSB % ES_THREADS == 0, "one value range per lane at least");
    __shared__ uint64_t A[ES_CAP];
    __shared__ uint32_t sbcnt[ES_NSB], sbstart[ES_NSB + 1], H[ES_CAP / 2 + 2], hcnt[NH + 1], wsum[NW];
    __shared__ uint16_t GL[ES_CAP / 2 + 2];            // gather slots: the bucket's columns that some entry still fetches, in the order their heads drew a place
    __shared__ uint32_t gmisc[4];                      // 0: columns that need a slot, 1: the bucket's first slot
    constexpr uint32_t HPOS = 0x3FFFu, HNEED = 0x80000000u;      // H[column] = head place | local slot << 14 | HNEED
    unsigned long long slot_next = 0; uint32_t slot_left = 0;    // (thread 0) what is left of the chunk of slots this workgroup drew
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t lt = (1ull << lane) - 1;
    const uint64_t paymask = (1ull << m.PB) - 1, posmask = (1ull << m.pbits) - 1;
    const int m_rk = SPEC ? 0 : m.rk;
    const bool m_dup = SPEC ? false : m.dup != 0u;
    const bool o_hints = SPEC ? true : o.hints != 0u, o_compact = SPEC ? true : o.compact != 0u;
    uint64_t *const o_pair_val = SPEC ? nullptr : o.pair_val;
    uint64_t *const o_csr_words = o.csr_words;
    const bool has_words = SPEC ? true : o.csr_words != nullptr;
    const uint64_t *const o_kmer_src = SPEC ? nullptr : o.kmer_src, *const o_kmer_dist = SPEC ? nullptr : o.kmer_dist;
    const uint32_t *const o_ncols = SPEC ? nullptr : o.ncols;
    const bool has_relk = SPEC ? true : o.rel_kmers != nullptr;
#ifdef ELBA_X_EMIT
    const bool xst = o.nb == 12345;      // timing experiment (wrong results): nothing leaves the kernel
#else
    constexpr bool xst = true;
#endif
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        const uint32_t Z = bZ[b];
        if (Z <= cap_lo || Z > cap_hi) continue;      // (other sizes: the other instantiation, or k_msd_bucket)
        const uint32_t s0 = b2start[b];
        uint64_t key[ES_KPT];
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) { const uint32_t i = (uint32_t)u * ES_THREADS + tid; key[u] = i < Z ? wrel[s0 + i] : ~0ull; }
        // Ranges of the sort, monotone in the word.  Without column ranks: 2^(16 - sh) ranges of the 16 value bits, 4-8 entries each while the
        // values are spread evenly — but a column of 30 entries (deep coverage, BASELINE config 5) is ONE value: every entry then ranks itself among 30.
        // With ranks (MsdParams::rk): the range is (column rank, leading bits of the read) cut to NSB_BITS bits — each column has its own ranges, and
        // a long column is split by read.
        uint32_t sh = 16 - NSB_BITS;
        while (sh < 16 && (Z >> (16 - sh)) < 4u) ++sh;
        int rs_s = 0, rs_t = 0;
        if (m_rk) {
            const uint32_t ncol = o_ncols ? o_ncols[b] : kidbase[b + 1] - kidbase[b];
            int cb = 0;
            while (cb < 16 && (ncol >> cb)) ++cb;                     // bits of a column rank
            int want = 2;
            while (want < NSB_BITS && (Z >> (want + 2)) != 0u) ++want;      // ~4 entries per range
            if (cb >= want) rs_t = cb - want;
            else { rs_s = want - cb; const int mbits = m.PB - m.pbits; if (rs_s > mbits) rs_s = mbits; }
        }
        auto range_of = [&](uint64_t x) -> uint32_t {
            return m_rk ? (uint32_t)(((((x >> m_rk) & m.rkmask) << rs_s) | ((x & paymask) >> (m.PB - rs_s))) >> rs_t) : ((uint32_t)(x >> m.PB) & 0xFFFFu) >> sh;
        };
        auto col_of = [&](uint64_t x) -> uint32_t { return m_rk ? (uint32_t)(x >> m_rk) & m.rkmask : (uint32_t)(x >> m.PB) & 0xFFFFu; };
#pragma unroll
        for (int q = 0; q < PER; ++q) sbcnt[tid + (uint32_t)q * ES_THREADS] = 0;
        lds_sync();
        uint32_t slot[ES_KPT];
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) {
            slot[u] = 0;
            if ((uint32_t)u * ES_THREADS + tid < Z) slot[u] = atomicAdd(&sbcnt[range_of(key[u])], 1u);
        }
        lds_sync();
        {   // exclusive scan of the <= 512 range counts: PER consecutive ones per lane
            uint32_t cc[PER], tot = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { cc[q] = sbcnt[PER * tid + q]; tot += cc[q]; }
            uint32_t inc = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
            if (lane == 63) wsum[wv] = inc;
            lds_sync();
            uint32_t ex = inc - tot;
            for (uint32_t ww = 0; ww < wv; ++ww) ex += wsum[ww];
#pragma unroll
            for (int q = 0; q < PER; ++q) { sbstart[PER * tid + q] = ex; ex += cc[q]; }
            if (tid == ES_THREADS - 1) sbstart[ES_NSB] = ex;
        }
        lds_sync();
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u)
            if ((uint32_t)u * ES_THREADS + tid < Z) A[sbstart[range_of(key[u])] + slot[u]] = key[u];
        lds_sync();
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) {
            if ((uint32_t)u * ES_THREADS + tid < Z) {
                const uint32_t sb = range_of(key[u]), lo = sbstart[sb], hi = sbstart[sb + 1];
                // (the range's first eight words requested at once: a loop with a per-lane trip count is one LDS round trip per word — ~100 per lane and bucket)
                uint32_t rank = 0;
                uint64_t kk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) kk[q] = A[lo + (uint32_t)q < hi ? lo + (uint32_t)q : lo];
                if (m_dup) {      // (equal words — an entry handed over twice — keep the order of their places in the range: every word gets a place of its own)
                    const uint32_t mine = lo + slot[u];
#pragma unroll
                    for (int q = 0; q < 8; ++q) rank += (lo + (uint32_t)q < hi && (kk[q] < key[u] || (kk[q] == key[u] && lo + (uint32_t)q < mine))) ? 1u : 0u;
                    for (uint32_t x = lo + 8u; x < hi; ++x) { const uint64_t a = A[x]; rank += (a < key[u] || (a == key[u] && x < mine)) ? 1u : 0u; }
                } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) rank += (lo + (uint32_t)q < hi && kk[q] < key[u]) ? 1u : 0u;
                for (uint32_t x = lo + 8u; x < hi; ++x) rank += A[x] < key[u] ? 1u : 0u;
                }
                slot[u] = lo + rank;
            }
        }
        lds_sync();
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) if ((uint32_t)u * ES_THREADS + tid < Z) A[slot[u]] = key[u];
        lds_sync();
        // run heads, in place order p = u * 256 + tid; the column of place p = heads at or before it - 1
        uint32_t headmask = 0;      // bit u: place u * 256 + tid heads a column
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) {
            const uint32_t p = (uint32_t)u * ES_THREADS + tid;
            const bool head = p < Z && (p == 0 || col_of(A[p - 1]) != col_of(A[p]));
            const uint64_t bal = __ballot(head);
            if (head) headmask |= 1u << u;
            slot[u] = (uint32_t)__popcll(bal & lt) + (head ? 1u : 0u);      // heads at or before this place within its wavefront's 64 places
            if (lane == 0) hcnt[u * NW + wv] = (uint32_t)__popcll(bal);
        }
        lds_sync();
        if (wv == 0) {      // exclusive scan of the NH (u, wave) head counts, in place order: NH / 64 consecutive ones per lane
            constexpr int PL = NH >= 64 ? NH / 64 : 1;      // (NH = 32 with 8 entries per lane: the upper half of the lanes holds nothing)
            uint32_t c[PL], sum = 0;
#pragma unroll
            for (int q = 0; q < PL; ++q) { c[q] = lane * PL + q < NH ? hcnt[lane * PL + q] : 0u; sum += c[q]; }
            uint32_t inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
            uint32_t run = inc - sum;
#pragma unroll
            for (int q = 0; q < PL; ++q) { if (lane * PL + q < NH) hcnt[lane * PL + q] = run; run += c[q]; }
            if (lane == 63) hcnt[NH] = inc;
        }
        lds_sync();
        const uint32_t Nb = hcnt[NH], kb = kidbase[b], eb = entbase[b];
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) {
            slot[u] = hcnt[u * NW + wv] + slot[u] - 1u;                     // the column of place u * ES_THREADS + tid
            if ((headmask >> u) & 1u) H[slot[u]] = (uint32_t)u * ES_THREADS + tid;
        }
        if (tid == 0) { H[Nb] = Z; gmisc[0] = 0; }
        lds_sync();
        uint32_t needmask = 0;      // bit u: the entry at place u * 256 + tid fetches its column (its key waits for the column's slot)
#pragma unroll
        for (int u = 0; u < ES_KPT; ++u) {
            const uint32_t p = (uint32_t)u * ES_THREADS + tid;
            if (p < Z) {
                const uint64_t x = A[p];
                const uint32_t kl = slot[u], h0 = H[kl] & HPOS, L = (H[kl + 1] & HPOS) - h0;
                const uint64_t read = (x & paymask) >> m.pbits, pos = x & posmask;
                uint64_t hint = 0, oread = 0, opos = 0;
                uint32_t nown = 0, mult = 0;
                if (o_hints && L <= HINT_MAX_COL) {
                    // Ctx::csr_hints: an entry whose row accumulates no pair of this column under the parity rule (and occurs in it once) is marked;
                    // the entries of the column this row does accumulate a pair with are counted, the last of them remembered
                    const uint32_t i = (uint32_t)read;
                    auto look = [&](uint64_t y) {
                        const uint32_t j = (uint32_t)((y & paymask) >> m.pbits);
                        if (j == i) { ++mult; return; }
                        if (((i ^ j) & 1u) ? j < i : j > i) { ++nown; oread = j; opos = y & posmask; }
                    };
                    // (the column's first four entries requested at once — nine columns in ten are no longer —: a loop with a per-lane trip count is an LDS round trip per entry)
                    uint64_t y4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) y4[q] = A[h0 + ((uint32_t)q < L ? (uint32_t)q : 0u)];
#pragma unroll
                    for (int q = 0; q < 4; ++q) if ((uint32_t)q < L) look(y4[q]);
                    for (uint32_t t = 4; t < L; ++t) look(A[h0 + t]);
                    if (mult < 2 && nown == 0) hint = 3;
                }
                const uint32_t z = eb + p, kid = kb + kl;
                if (xst) o.csc[z] = (read << 32) | pos;
                uint64_t word = (read << o.rs) | ((uint64_t)kid << (o.pb + 2)) | (hint << o.pb) | pos;
                // Ctx::csr_inline: a row that accumulates exactly ONE pair of this column (always so for the owner of a two-read column) carries that
                // pair in its own entry — the SpGEMM then fetches no column for it
                const bool isinl = o.inl && mult == 1u && nown == 1u && ((pos | opos) >> o.inl) == 0;
                if (isinl) word = (1ull << 63) | (read << o.rs) | ((oread >> 1) << (2 * o.inl)) | (pos << o.inl) | opos;
                if (!xst) {} else if (o_pair_val) { o.pair_key[z] = (uint32_t)read; o_pair_val[z] = ((uint64_t)kid << 32) | ((uint64_t)L << 23) | ((uint64_t)(p - h0) << 16) | pos; }
                else if (o_compact && !isinl && hint == 0) {      // this entry fetches its column: the column needs a gather slot, the key names it (below)
                    needmask |= 1u << u;
                    atomicOr(&H[kl], HNEED);
                }
                else if (has_words) o_csr_words[z] = word;
                else o.kid_of_entry[z] = kid;
                if (xst && ((headmask >> u) & 1u)) {
                    const uint64_t value = ((uint64_t)b << VBITS) | ((uint32_t)(x >> m.PB) & 0xFFFFu);
                    if (has_relk) o.rel_kmers[kid] = o_kmer_dist ? o_kmer_dist[o.dist_base[b] + ((uint32_t)(x >> m.PB) & 0xFFFFu)] : (o_kmer_src ? o_kmer_src[s0 + kl] : value << (64 - m.k2));
                    o.rel_counts[kid] = L; o.colptr[kid] = z;
                }
            }
        }
        if (o_compact) {
            lds_sync();                                                  // every entry has flagged its column
#pragma unroll
            for (int u = 0; u < ES_KPT; ++u)
                if (((headmask >> u) & 1u) && (H[slot[u]] & HNEED)) {    // the head of a column that is fetched draws the column's place among the bucket's slots
                    const uint32_t ls = atomicAdd(&gmisc[0], 1u);
                    H[slot[u]] |= ls << 14; GL[ls] = (uint16_t)slot[u];
                }
            lds_sync();
            if (tid == 0) {
                const uint32_t n = gmisc[0];
                unsigned long long base = slot_next;
                if (n > slot_left) {                                     // a new chunk (what is left of the old one stays unused: slots need not be dense)
                    const uint32_t take = n > o.slot_chunk ? n : o.slot_chunk;
                    base = atomicAdd(o.slot_cursor, (unsigned long long)take);
                    slot_next = base; slot_left = take;
                }
                slot_next += n; slot_left -= n;
                gmisc[1] = (uint32_t)base;
                gmisc[2] = base + n > o.slot_cap ? 1u : 0u;
                if (gmisc[2]) atomicOr(&o.slot_cursor[1], 1ull);
            }
            lds_sync();
            const uint32_t gb = gmisc[1], ng = gmisc[2] ? 0u : gmisc[0];      // (past the store: this bucket's slots are not written — the whole emit is repeated)
            if (gmisc[2]) needmask = 0;
#pragma unroll
            for (int u = 0; u < ES_KPT; ++u)
                if ((needmask >> u) & 1u) {
                    const uint32_t p = (uint32_t)u * ES_THREADS + tid;
                    const uint64_t x = A[p];
                    const uint64_t sid = (uint64_t)gb + ((H[slot[u]] >> 14) & 0x1FFFu);
                    if (xst) o_csr_words[eb + p] = (((x & paymask) >> m.pbits) << o.rs) | (sid << (o.pb + 2)) | (x & posmask);
                }
            const uint32_t S = o.ell_stride, nq = ng * S, sl = (S & (S - 1u)) ? 0u : (uint32_t)__ffs((int)S) - 1u;
            uint64_t *dst = o.ell + (uint64_t)gb * S;
            for (uint32_t q = tid; q < nq; q += ES_THREADS) {
                const uint32_t ls = (S & (S - 1u)) ? q / S : q >> sl, j = q - ls * S, kl = GL[ls], h0 = H[kl] & HPOS;
                uint64_t v = ~0ull;
                if (h0 + j < (H[kl + 1] & HPOS)) { const uint64_t x = A[h0 + j]; v = (((x & paymask) >> m.pbits) << 32) | (x & posmask); }
                if (xst) dst[q] = v;
                if (xst && j == 0) o.slot_kid[gb + ls] = kb + kl;
            }
        } else if (o.ell_stride) {
            const uint32_t S = o.ell_stride, nq = Nb * S, sl = (S & (S - 1u)) ? 0u : (uint32_t)__ffs((int)S) - 1u;
            uint64_t *dst = o.ell + (uint64_t)kb * S;
            for (uint32_t q = tid; q < nq; q += ES_THREADS) {
                const uint32_t kl = (S & (S - 1u)) ? q / S : q >> sl, j = q - kl * S, h0 = H[kl];
                // (not writing the slots of two-read columns whose pair travels inline — never fetched, 80 % of this store — was measured: 23.0 vs 22.2 ms for
                //  the bucket kernels: the stores are fire-and-forget, the test is not)
                uint64_t v = ~0ull;
                if (h0 + j < H[kl + 1]) { const uint64_t x = A[h0 + j]; v = (((x & paymask) >> m.pbits) << 32) | (x & posmask); }
                if (xst) dst[q] = v;
            }
        }
        lds_sync();                               // A, H and the counters are reused by the next bucket
    }
}

// ---- buckets: emit, crowded buckets (more than ES_CAP entries: repeats, very deep coverage) -------------------------------------------------------
// One workgroup of 1024 lanes per bucket of the `crowded` list, reading ALL the bucket's instances again, in two halves of 2^15 values each:
//   count, classify   as k_msd_count, on a 64 KB table;
//   number     lane t walks the set bits of word t of the reliable bitmap — in value order — and sums their counts; a scan over the lanes gives
//              the k-mer id and the column pointer of every reliable value;
//   windows    the reliable columns are staged EW entries at a time: the lanes write their values' window-local column numbers into the
//              table (16-bit stores over the counts), every instance of such a value draws a slot in its column (LDS atomic) and leaves its
//              payload there; one lane per column sorts it by (read, pos) and computes the ownership hints of its entries; the staged window
//              leaves as coalesced streams: columns, CSR sort keys, padded columns.
// Any number of entries per bucket; ~45 us per bucket, which is why the usual buckets go through k_msd_emit_small.
template <bool EMIT>
__global__ __launch_bounds__(BK_THREADS) void k_msd_bucket(const uint64_t *words, const uint32_t *b2start, uint32_t nbuckets, MsdParams m, uint32_t lower, uint32_t upper,
                                                          const uint32_t *crowded, const BucketStats *gstat, const uint32_t *kidbase, const uint32_t *entbase, BucketOut o)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // LDS, in words: table 16384 | reliable bitmap 1024 | seen bitmap 1024 | misc 64 | scan partials 64 | headpos KW + 1 | fill KW + 1 | staged entries | their columns
    uint32_t *tab = smem;                                   // 2^15 counters of 16 bits; in a window: the window-local column number of the reliable values
    uint16_t *tab16 = reinterpret_cast<uint16_t *>(smem);
    uint32_t *relbits = smem + BK_TAB;      // (+ 1024 spare words)
    uint32_t *misc = smem + BK_TAB + 2048;
    uint32_t *wsc = misc + 64;
    uint32_t *headpos = wsc + 64;
    uint32_t *fill = headpos + (KW + 1);
    uint64_t *ent = reinterpret_cast<uint64_t *>(smem + BK_ENT);
    uint16_t *entk = reinterpret_cast<uint16_t *>(ent + (EW + EPAD));
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t paymask = (1ull << m.PB) - 1, posmask = (1ull << m.pbits) - 1;

    const uint32_t ncrowded = gstat->ncrowded;
    for (uint32_t li = blockIdx.x; li < ncrowded; li += gridDim.x) {
        const uint32_t b = crowded[li];
        const uint32_t s0 = b2start[b], n = b2start[b + 1] - s0;
        if (n == 0 || b >= nbuckets) continue;
        // the bucket's instances: the first KPT * 1024 live in registers, the rest (crowded buckets) is re-read from L2 in every pass
        uint64_t kreg[KPT];
#pragma unroll
        for (int u = 0; u < KPT; ++u) { const uint32_t i = (uint32_t)u * BK_THREADS + tid; kreg[u] = i < n ? words[s0 + i] : 0; }
        const bool guard = n > 65535u;          // a value's count could run over its 16 bits: stop counting beyond 2^15 (UPPER <= 255: unreliable anyway)
        uint32_t n0 = 0, z0 = 0;
#pragma unroll 1
        for (uint32_t h = 0; h < 2; ++h) {
            {   // zero the table and the two bitmaps (18 K words: 4.5 uint4 per lane)
                uint4 *t4 = reinterpret_cast<uint4 *>(smem);
#pragma unroll
                for (int u = 0; u < 4; ++u) t4[(uint32_t)u * BK_THREADS + tid] = make_uint4(0u, 0u, 0u, 0u);
                if (tid < 512) t4[4096 + tid] = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
            auto for_keys = [&](auto &&f) {      // f(word) for every instance of this half
#pragma unroll
                for (int u = 0; u < KPT; ++u) if ((uint32_t)u * BK_THREADS + tid < n && (((uint32_t)(kreg[u] >> (m.PB + 15)) & 1u) == h)) f(kreg[u]);
                for (uint32_t i = (uint32_t)KPT * BK_THREADS + tid; i < n; i += BK_THREADS) { const uint64_t wd = words[s0 + i]; if ((((uint32_t)(wd >> (m.PB + 15)) & 1u) == h)) f(wd); }
            };
            for_keys([&](uint64_t wd) {
                const uint32_t idx = (uint32_t)(wd >> m.PB) & 0x7FFFu;
                if (guard && tab16[idx] >= 0x8000u) return;
                atomicAdd(&tab[idx >> 1], 1u << ((idx & 1u) * 16u));
            });
            __syncthreads();
            for_keys([&](uint64_t wd) {
                const uint32_t idx = (uint32_t)(wd >> m.PB) & 0x7FFFu, cnt = tab16[idx];
                if (cnt >= lower && cnt <= upper) atomicOr(&relbits[idx >> 5], 1u << (idx & 31u));
            });
            __syncthreads();
            // lane t owns the values 32 t .. 32 t + 31: its reliable ones, in value order, are the set bits of word t
            const uint32_t mybits = relbits[tid];
            uint32_t nrel = (uint32_t)__popc(mybits), nent = 0;
            for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                const uint32_t cnt = tab16[tid * 32u + (uint32_t)__ffs((int)bits) - 1u];
                nent += cnt;
            }
            // exclusive scan of (nrel, nent) over the 1024 lanes
            uint32_t ir = nrel, ie = nent;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t a = __shfl_up(ir, d, 64), c2 = __shfl_up(ie, d, 64); if ((int)lane >= d) { ir += a; ie += c2; } }
            if (lane == 63) { wsc[wv] = ir; wsc[16 + wv] = ie; }
            __syncthreads();
            uint32_t R = ir - nrel, E = ie - nent, totR = 0, totE = 0;
#pragma unroll
            for (uint32_t ww = 0; ww < BK_THREADS / 64; ++ww) { const uint32_t a = wsc[ww], c2 = wsc[16 + ww]; if (ww < wv) { R += a; E += c2; } totR += a; totE += c2; }
            if (h == 0) { n0 = totR; z0 = totE; }
            if (EMIT && totR != 0) {
                const uint32_t kb = kidbase[b] + (h ? n0 : 0u), eb = entbase[b] + (h ? z0 : 0u);
                {   // the reliable k-mers of this lane's values: k-mer (left-aligned, src/Kmer.cpp:78-86), count, column pointer
                    uint32_t r2 = R, e2 = E;
                    for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                        const uint32_t idx = tid * 32u + (uint32_t)__ffs((int)bits) - 1u, cnt = tab16[idx];
                        const uint64_t value = ((uint64_t)b << VBITS) | (h << 15) | idx;
                        if (o.rel_kmers) o.rel_kmers[kb + r2] = o.kmer_dist ? o.kmer_dist[o.dist_base[b] + ((h << 15) | idx)] : value << (64 - m.k2);
                        o.rel_counts[kb + r2] = cnt; o.colptr[kb + r2] = eb + e2;
                        ++r2; e2 += cnt;
                    }
                }
                // Windows of EW staged entries.  A window's values carry their window-local column number in the table while it is staged (bit 15
                // set: a reliable count is <= 255) and get their count back — fill[], every instance of the column placed — when it is done.
                const uint32_t nwin = (totE + EW - 1) / EW;
#pragma unroll 1
                for (uint32_t wn = 0; wn < nwin; ++wn) {
                    const uint32_t wlo = wn * EW;
                    __syncthreads();                       // (the previous window's staging area is free; wsc has been read)
                    if (tid == 0) { misc[0] = 0xFFFFFFFFu; misc[1] = 0; misc[2] = 0; }
                    __syncthreads();
                    {   // first column of the window
                        uint32_t r2 = R, e2 = E;
                        for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                            const uint32_t cnt = tab16[tid * 32u + (uint32_t)__ffs((int)bits) - 1u];
                            if (e2 >= wlo && e2 < wlo + EW) { atomicMin(&misc[0], r2); break; }
                            ++r2; e2 += cnt;
                        }
                    }
                    __syncthreads();
                    const uint32_t klo = misc[0];
                    {   // the table now maps the window's values to their column's number inside the window; the columns' places in the staging area
                        uint32_t r2 = R, e2 = E, kmax = 0, emax = 0;
                        for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                            const uint32_t idx = tid * 32u + (uint32_t)__ffs((int)bits) - 1u, cnt = tab16[idx];
                            if (e2 >= wlo && e2 < wlo + EW) {
                                const uint32_t kl = r2 - klo;
                                tab16[idx] = (uint16_t)(0x8000u | kl); headpos[kl] = e2 - wlo; fill[kl] = 0;
                                kmax = kl + 1; emax = e2 - wlo + cnt;
                            }
                            ++r2; e2 += cnt;
                        }
                        if (kmax) { atomicMax(&misc[1], kmax); atomicMax(&misc[2], emax); }
                    }
                    __syncthreads();
                    const uint32_t ncolw = misc[1], nentw = misc[2];
                    if (o.compact && tid == 0) {      // gather slots (BucketOut): every column of a crowded bucket's window gets one
                        const unsigned long long wb = atomicAdd(o.slot_cursor, (unsigned long long)ncolw);
                        misc[3] = (uint32_t)wb; misc[4] = wb + ncolw > o.slot_cap ? 1u : 0u;
                        if (misc[4]) atomicOr(&o.slot_cursor[1], 1ull);
                    }
                    for_keys([&](uint64_t wd) {
                        const uint32_t idx = (uint32_t)(wd >> m.PB) & 0x7FFFu;
                        if (!((relbits[idx >> 5] >> (idx & 31u)) & 1u)) return;
                        const uint32_t code = tab16[idx];
                        if (!(code & 0x8000u)) return;       // a reliable value of another window (its count, <= 255, sits there)
                        const uint32_t kl = code & 0x7FFFu;
                        const uint32_t at = headpos[kl] + atomicAdd(&fill[kl], 1u);
                        ent[at] = wd & paymask; entk[at] = (uint16_t)kl;
                    });
                    __syncthreads();
                    // the window's values get their counts back (the next window tells its columns from the others by the code bit)
                    for (uint32_t bits = mybits; bits; bits &= bits - 1u) {
                        const uint32_t idx = tid * 32u + (uint32_t)__ffs((int)bits) - 1u, code = tab16[idx];
                        if (code & 0x8000u) tab16[idx] = (uint16_t)fill[code & 0x7FFFu];
                    }
                    // every column of the window is sorted by (read, pos) = by payload, by one lane; the ownership hints of its entries with it
                    for (uint32_t kl = tid; kl < ncolw; kl += BK_THREADS) {
                        const uint32_t p0 = headpos[kl], L = fill[kl];
                        for (uint32_t a = 1; a < L; ++a) {
                            const uint64_t v = ent[p0 + a];
                            uint32_t q = a;
                            while (q > 0 && ent[p0 + q - 1] > v) { ent[p0 + q] = ent[p0 + q - 1]; --q; }
                            ent[p0 + q] = v;
                        }
                        if (o.hints && L <= HINT_MAX_COL) {
                            // Ctx::csr_hints: an entry whose row accumulates no pair of this column under the parity rule (and occurs in it once) is marked
                            for (uint32_t a = 0; a < L; ++a) {
                                const uint32_t i = (uint32_t)(ent[p0 + a] >> m.pbits);
                                bool own = false; uint32_t mult = 0;
                                for (uint32_t t = 0; t < L; ++t) {
                                    const uint32_t j = (uint32_t)((ent[p0 + t] & paymask) >> m.pbits);
                                    if (j == i) { ++mult; continue; }
                                    own |= ((i ^ j) & 1u) ? j < i : j > i;
                                }
                                if (mult < 2 && !own) ent[p0 + a] |= 3ull << 62;
                            }
                        }
                    }
                    __syncthreads();
                    const uint32_t wslot = o.compact ? misc[3] : 0u;
                    const bool past = o.compact && misc[4] != 0u;      // (past the store: nothing slot-addressed is written, the emit is repeated without slots)
                    for (uint32_t p = headpos[0] + tid; p < nentw; p += BK_THREADS) {      // (the window's first column may start a few places in: the previous window's last column reaches that far)
                        const uint64_t x = ent[p];
                        const uint64_t read = (x & paymask) >> m.pbits, pos = x & posmask, hint = x >> 62;
                        const uint32_t z = eb + wlo + p, kid = kb + klo + entk[p];
                        const uint32_t id = o.compact && hint == 0 ? wslot + entk[p] : kid;      // (an entry that fetches its column names the column's gather slot)
                        o.csc[z] = (read << 32) | pos;
                        if (o.pair_val) { const uint32_t kl = entk[p]; o.pair_key[z] = (uint32_t)read; o.pair_val[z] = ((uint64_t)kid << 32) | ((uint64_t)fill[kl] << 23) | ((uint64_t)(p - headpos[kl]) << 16) | pos; }
                        else if (o.csr_words) o.csr_words[z] = (read << o.rs) | ((uint64_t)id << (o.pb + 2)) | (hint << o.pb) | pos;
                        else o.kid_of_entry[z] = kid;
                    }
                    if (o.ell_stride && !past) {
                        const uint32_t S = o.ell_stride, nq = ncolw * S;
                        uint64_t *dst = o.ell + (uint64_t)(o.compact ? wslot : kb + klo) * S;
                        for (uint32_t q = tid; q < nq; q += BK_THREADS) {
                            const uint32_t kl = q / S, j = q - kl * S;
                            uint64_t v = ~0ull;
                            if (j < fill[kl]) { const uint64_t x = ent[headpos[kl] + j]; v = (((x & paymask) >> m.pbits) << 32) | (x & posmask); }
                            dst[q] = v;
                            if (o.compact && j == 0) o.slot_kid[wslot + kl] = kb + klo + kl;
                        }
                    }
                }
            }
            __syncthreads();                               // wsc, the table and the staging area are reused by the next half / bucket
        }
    }
}

// =====================================================================================================================================
// One-word k-mers beyond k = 17 (19 <= k <= 31: the reference's default build is k = 31, Makefile:1-3).  2k - 16 value bits do not fit a
// two-level partition any more, and value + read + position do not fit one word (62 + 32 bits), so
//   * an instance travels as a RECORD of 16 bytes, hi = the canonical k-mer (2k bits, right-aligned), lo = read << pbits | pos;
//   * the two-level partition takes the top T = b1 + b2 value bits (T chosen so that a bucket holds ~1500 instances; up to 10 bits a level);
//   * k31_count sorts a bucket's records by (k-mer, read, pos) IN LDS (<= W2_CAP records), finds the runs of equal k-mers, keeps those of
//     LOWER..UPPER instances and writes their entries — compacted, in order, as the one-word entries the emit kernels of the k <= 17 path
//     read: (column of the bucket scaled to 16 bits) << PB | read << pbits | pos — and the bucket's reliable k-mers beside them.
// From there on the path is the k <= 17 one (k_msd_emit_small, the CSR build).  A bucket of more DISTINCT k-mers than the count table takes, or of
// more kept entries than the emit kernels sort, sends the whole input to the sort of kmer.hip: correct, slower; profiles/r04_notes.md.
// HBM traffic per instance: 16 B written + 16 read (hist2) + 16 read + 16 written + 16 read = 80 B, against 7 passes x 32 B + 3 x 16 B on the sort path.
#ifndef ELBA_W2_THREADS
#define ELBA_W2_THREADS 512
#endif
#ifndef ELBA_W2_ITEMS
#define ELBA_W2_ITEMS 16
#endif
constexpr int W2_THREADS = ELBA_W2_THREADS, W2_ITEMS = ELBA_W2_ITEMS, W2_TILE = W2_THREADS * W2_ITEMS;      // 8192 records of 16 bytes: 128 KB of LDS (runs of 8 records per digit and tile; 4096-record tiles: partition 78 -> 58 ms on 2.0 G instances)
constexpr int W2_MAXBITS = 10, W2_MAXBINS = 1 << W2_MAXBITS;
constexpr uint32_t W2_CAP = 4096;                // records of a bucket k31_count takes (eight per lane, in registers)
constexpr int W2C_THREADS = 512, W2C_KPT = (int)(W2_CAP / W2C_THREADS);
static_assert(W2_ITEMS * 64 <= (1 << IB_SHIFT), "a wavefront's share of a tile lies inside one block of the instance -> read table");
struct alignas(16) Rec2 { uint64_t hi, lo; };
// The canonical k-mer is the smaller of a k-mer and its reverse complement: as a fraction x of the value range its density is 2 (1 - x) — the lowest
// buckets of an even split hold twice the average (and run past W2_CAP).  Buckets are therefore cut by the leading bits of G(x) = 2x - x^2, the
// distribution function of that density, taken on the value's leading 32 bits: monotone in the value (k-mer ids follow the buckets' order),
// and every bucket receives the same share of a random genome's k-mers.
__device__ __forceinline__ uint32_t w2_flat(uint32_t x32)
{
    const uint64_t g = ((uint64_t)x32 << 1) - (((uint64_t)x32 * x32) >> 32);
    return g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g;
}

__global__ __launch_bounds__(W2_THREADS) void k31_hist1(EnumParams e, const BlockInfo *block_read, int b1, uint32_t *hist)
{
    __shared__ uint32_t h[W2_MAXBINS];
    const uint32_t nbins = 1u << b1;
    for (uint32_t i = threadIdx.x; i < nbins; i += W2_THREADS) h[i] = 0;
    __syncthreads();
    const uint64_t base = ((uint64_t)blockIdx.x * (W2_THREADS / 64) + (threadIdx.x >> 6)) * (uint64_t)(W2_ITEMS * 64);
    enum_consecutive<W2_ITEMS>(e, block_read, base, [&](int, uint64_t km, uint32_t, uint32_t) { atomicAdd(&h[w2_flat((uint32_t)(km >> 32)) >> (32 - b1)], 1u); });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += W2_THREADS) hist[(size_t)blockIdx.x * nbins + i] = h[i];
}

// tiles of the second pass (none straddles two first-digit buckets): as k_msd_tiles, up to 1024 first digits, tiles of W2_TILE records
__global__ __launch_bounds__(W2_MAXBINS) void k31_tiles(const uint32_t *hist1_row0, uint32_t nb1, uint64_t I, uint32_t *b1start, uint32_t *tile0)
{
    __shared__ uint32_t wsum[W2_MAXBINS / 64];
    const uint32_t d = threadIdx.x, lane = d & 63, w = d >> 6;
    const uint32_t st = d < nb1 ? hist1_row0[d] : (uint32_t)I, en = d + 1 < nb1 ? hist1_row0[d + 1] : (uint32_t)I;
    const uint32_t nt = d < nb1 ? (en - st + W2_TILE - 1) / W2_TILE : 0u;
    uint32_t inc = nt;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= (uint32_t)s2) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t run = inc - nt;
    for (uint32_t ww = 0; ww < w; ++ww) run += wsum[ww];
    if (d < nb1) { b1start[d] = st; tile0[d] = run; }
    if (d == nb1 - 1) { b1start[nb1] = (uint32_t)I; tile0[nb1] = run + nt; }
}
__device__ __forceinline__ void seg_tile2(const SegTiles &sg, uint32_t t, uint32_t &bucket, uint32_t &start, uint32_t &count)
{
    uint32_t lo = 0, hi = sg.nb1;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sg.tile0[mid] <= t) lo = mid; else hi = mid; }
    bucket = lo;
    start = sg.b1start[lo] + (t - sg.tile0[lo]) * (uint32_t)W2_TILE;
    const uint32_t end = sg.b1start[lo + 1];
    count = end - start < (uint32_t)W2_TILE ? end - start : (uint32_t)W2_TILE;
}

// (shift / bits of the digit refer to the flattened leading 32 bits, w2_flat; k2 - 32 = the value bits below them)
__global__ __launch_bounds__(W2_THREADS) void k31_hist2(const Rec2 *recs, SegTiles sg, int k2, int shift, int bits, uint32_t *hist)
{
    __shared__ uint32_t h[W2_MAXBINS];
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    if (blockIdx.x >= sg.tile0[sg.nb1]) return;
    for (uint32_t i = threadIdx.x; i < nbins; i += W2_THREADS) h[i] = 0;
    __syncthreads();
    uint32_t bucket, start, count;
    seg_tile2(sg, blockIdx.x, bucket, start, count);
    uint64_t k[W2_ITEMS];
#pragma unroll
    for (int r = 0; r < W2_ITEMS; ++r) { const uint32_t q = (uint32_t)r * W2_THREADS + threadIdx.x; k[r] = q < count ? recs[start + q].hi : 0; }
#pragma unroll
    for (int r = 0; r < W2_ITEMS; ++r) { const uint32_t q = (uint32_t)r * W2_THREADS + threadIdx.x; if (q < count) atomicAdd(&h[(w2_flat((uint32_t)(k[r] >> (k2 - 32))) >> shift) & dmask], 1u); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += W2_THREADS) hist[(size_t)blockIdx.x * nbins + i] = h[i];
}

// second-digit places: as k_msd_segscan, up to 1024 second digits
__global__ __launch_bounds__(W2_MAXBINS) void k31_segscan(uint32_t *hist, SegTiles sg, uint32_t nb2, uint32_t *b2start, uint64_t I)
{
    __shared__ uint32_t wsum[W2_MAXBINS / 64];
    const uint32_t b = blockIdx.x, d = threadIdx.x, lane = d & 63, w = d >> 6;
    const uint32_t t0 = sg.tile0[b], t1 = sg.tile0[b + 1];
    uint32_t run = 0;
    if (d < nb2) {
        for (uint32_t t = t0; t < t1; ++t) { const uint32_t x = hist[(size_t)t * nb2 + d]; hist[(size_t)t * nb2 + d] = run; run += x; }
    }
    uint32_t inc = run;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= (uint32_t)s2) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = sg.b1start[b] + inc - run;
    for (uint32_t ww = 0; ww < w; ++ww) base += wsum[ww];
    if (d < nb2) {
        b2start[(size_t)b * nb2 + d] = base;
        for (uint32_t t = t0; t < t1; ++t) hist[(size_t)t * nb2 + d] += base;
    }
    if (b == gridDim.x - 1 && d == 0) b2start[(size_t)gridDim.x * nb2] = (uint32_t)I;
}

// scatter of one tile of records by one digit of the k-mer (any order inside a digit: k31_count sorts every bucket).  ENUM: the tile's
// records are enumerated from the reads (4096 consecutive instances); else read from `in` (a bucket-aligned tile).
template <bool ENUM>
__global__ __launch_bounds__(W2_THREADS) void k31_scatter(EnumParams e, const BlockInfo *block_read, int k2, int pbits, const Rec2 *in, SegTiles sg, int shift, int bits,
                                                         const uint32_t *hist_scanned, Rec2 *out)
{
    constexpr int WAVES = W2_THREADS / 64, DPT = W2_MAXBINS / W2_THREADS;
    __shared__ uint32_t lcnt[W2_MAXBINS], lstart[W2_MAXBINS], delta[W2_MAXBINS], wsum[WAVES];
    __shared__ Rec2 lrec[W2_TILE];
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    if (!ENUM && blockIdx.x >= sg.tile0[sg.nb1]) return;
    for (int i = threadIdx.x; i < W2_MAXBINS; i += W2_THREADS) lcnt[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t khi[W2_ITEMS], klo[W2_ITEMS];
    uint32_t count = 0;
    if (ENUM) {
        const uint64_t tbase = (uint64_t)blockIdx.x * W2_TILE, base = tbase + (uint64_t)w * (W2_ITEMS * 64);
        const uint64_t left = e.I - tbase;
        count = left < (uint64_t)W2_TILE ? (uint32_t)left : (uint32_t)W2_TILE;
#pragma unroll
        for (int it = 0; it < W2_ITEMS; ++it) { khi[it] = ~0ull; klo[it] = 0; }
        enum_consecutive<W2_ITEMS>(e, block_read, base, [&](int it, uint64_t km, uint32_t r, uint32_t p) { khi[it] = km >> (64 - k2); klo[it] = ((uint64_t)r << pbits) | p; });
    } else {
        uint32_t bucket, start;
        seg_tile2(sg, blockIdx.x, bucket, start, count);
#pragma unroll
        for (int it = 0; it < W2_ITEMS; ++it) {
            const uint32_t q = (uint32_t)it * W2_THREADS + threadIdx.x;
            if (q < count) { const Rec2 r = in[start + q]; khi[it] = r.hi; klo[it] = r.lo; } else { khi[it] = ~0ull; klo[it] = 0; }
        }
    }
    uint32_t gb[DPT];
#pragma unroll
    for (int u = 0; u < DPT; ++u) { const uint32_t d = threadIdx.x + u * W2_THREADS; gb[u] = d < nbins ? hist_scanned[(size_t)blockIdx.x * nbins + d] : 0u; }
    uint16_t rank[W2_ITEMS];
#pragma unroll
    for (int it = 0; it < W2_ITEMS; ++it) {
        rank[it] = 0;
        if (khi[it] != ~0ull) rank[it] = (uint16_t)atomicAdd(&lcnt[(w2_flat((uint32_t)(khi[it] >> (k2 - 32))) >> shift) & dmask], 1u);      // (a k-mer of 2k <= 62 bits is never all ones)
    }
    __syncthreads();
    {
        uint32_t tot[DPT], both = 0;
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = DPT * threadIdx.x + u; tot[u] = d < nbins ? lcnt[d] : 0u; both += tot[u]; }
        uint32_t inc = both;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= s2) inc += o; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t run = inc - both;
        for (int ww = 0; ww < w; ++ww) run += wsum[ww];
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = DPT * threadIdx.x + u; if (d < nbins) lstart[d] = run; run += tot[u]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = threadIdx.x + u * W2_THREADS; if (d < nbins) delta[d] = gb[u] - lstart[d]; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < W2_ITEMS; ++it)
        if (khi[it] != ~0ull) lrec[lstart[(w2_flat((uint32_t)(khi[it] >> (k2 - 32))) >> shift) & dmask] + rank[it]] = Rec2{khi[it], klo[it]};
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < count; t += W2_THREADS) {      // the tile lies ordered by digit: a digit's records are one contiguous run of the output
        const Rec2 r = lrec[t];
        out[delta[(w2_flat((uint32_t)(r.hi >> (k2 - 32))) >> shift) & dmask] + t] = r;
    }
}

// One workgroup per bucket: count, filter, number, compact (see the header of this section).  The records are NOT sorted: an LDS hash table keyed by
// the k-mer counts them (HiFi reads at 40x hold every genomic k-mer ~34 times: a sort ranks every record against its whole run, the table takes
// one compare-and-swap + one add per record), the RELIABLE k-mers alone — a few hundred per bucket — are sorted by value (range + rank), and every
// record of a reliable k-mer then draws its place in the k-mer's column; the order inside a column is left to the emit kernels, which sort their
// entries by (column, read, pos) anyway.  (First version: a full sort of the bucket in LDS — 159 ms for 2.0 G instances, profiles/r04_notes.md.)
constexpr uint32_t W2_SLOTS = 4096, W2_RELMAX = W2_CAP / 2;      // table slots; reliable k-mers of a bucket (LOWER >= 2)
constexpr uint32_t W2_DISTINCT_MAX = W2_SLOTS - 2 * W2C_THREADS;  // distinct k-mers at which a bucket gives up (every lane may claim one more slot: the probe loop always finds an empty one)
static_assert(W2C_KPT * W2C_THREADS >= (int)W2_CAP, "every record has a register");
// (a bucket's size fluctuates with coverage x sqrt(distinct genomic k-mers in it): 1900 +- 280 instances on 40x reads of a 50 Mb genome cut into 2^20
//  buckets — 3300 at five sigma, which 2^20 buckets do reach)
__global__ __launch_bounds__(W2C_THREADS, 4) void k31_count(const Rec2 *recs, const uint32_t *b2start, uint32_t nbuckets, int k2, int T, int PB, int rk, uint32_t lower, uint32_t upper,
                                                        uint32_t *bN, uint32_t *bZ, BucketStats *gstat, uint64_t *wrel, uint64_t *ktmp, uint32_t *crowded)
{
    constexpr uint32_t NW = W2C_THREADS / 64, NSB = 512, IPT = (W2_RELMAX + W2C_THREADS - 1) / W2C_THREADS, SPT = W2_SLOTS / W2C_THREADS;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem31[];
    unsigned long long *K = reinterpret_cast<unsigned long long *>(smem31);      // W2_SLOTS keys (~0: empty)
    uint32_t *CNT = smem31 + 2 * W2_SLOTS;                                        // their counts; for reliable k-mers afterwards: the entries placed so far
    uint32_t *C = CNT + W2_SLOTS;                                                 // [W2_RELMAX + 1] count of the reliable k-mer number rc -> (scan) its first entry
    uint32_t *sbcnt = C + W2_RELMAX + 1, *sbstart = sbcnt + NSB, *wsum = sbstart + NSB + 1, *misc = wsum + 2 * NW;
    constexpr uint32_t RC_OFF = (2 * W2_SLOTS + W2_SLOTS + (W2_RELMAX + 1) + NSB + (NSB + 1) + 2 * NW + 8 + 3u) & ~3u;      // (16-byte aligned: the table is initialised with 16-byte stores; W2C_LDS has the slack)
    uint16_t *RC = reinterpret_cast<uint16_t *>(smem31 + RC_OFF);                        // [W2_SLOTS] number of the slot's k-mer among the bucket's reliable ones (0xFFFF: not reliable)
    uint16_t *RL = RC + W2_SLOTS, *SA = RL + W2_RELMAX;                           // [W2_RELMAX] the reliable slots, as found / ordered by value range
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int R = 32 - T;                                  // bits of the flattened leading 32 value bits (w2_flat) the partition has not used
    unsigned long long st_distinct = 0, st_sumsq = 0;
    uint32_t st_maxcol = 0;
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {      // (two workgroups per CU hide one another's round trips: no register prefetch of the next bucket)
        const uint32_t s0 = b2start[b], n = b2start[b + 1] - s0;
        if (n == 0) { if (tid == 0) { bN[b] = 0; bZ[b] = 0; } continue; }
        // A bucket of more than W2_CAP records (a homopolymer, a satellite: FEW k-mers, each far beyond UPPER) is walked in chunks of W2_CAP — the table
        // only holds the distinct k-mers — and read a second time for its kept entries; what gives a bucket up is more distinct k-mers than the
        // table takes, or more kept entries than the emit kernels sort.
        const uint32_t nch = (n + W2_CAP - 1u) / W2_CAP;
        Rec2 key[W2C_KPT];
        auto load_chunk = [&](uint32_t ch) {
#pragma unroll
            for (int u = 0; u < W2C_KPT; ++u) { const uint32_t i = ch * W2_CAP + (uint32_t)u * W2C_THREADS + tid; key[u] = i < n ? recs[s0 + i] : Rec2{~0ull, ~0ull}; }
        };
        load_chunk(0);
        {   // empty table: keys all ones, counts zero, no slot reliable — 16-byte stores (7 per lane instead of 24)
            static_assert(W2_SLOTS == 8 * W2C_THREADS, "the stores below cover the table exactly");
            uint4 *k4 = reinterpret_cast<uint4 *>(K), *c4 = reinterpret_cast<uint4 *>(CNT), *r4 = reinterpret_cast<uint4 *>(RC);
#pragma unroll
            for (int q = 0; q < 4; ++q) k4[(uint32_t)q * W2C_THREADS + tid] = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
            for (int q = 0; q < 2; ++q) c4[(uint32_t)q * W2C_THREADS + tid] = make_uint4(0u, 0u, 0u, 0u);
            r4[tid] = make_uint4(~0u, ~0u, ~0u, ~0u);
        }
        sbcnt[tid] = 0;
        if (tid < 8) misc[tid] = 0;
        lds_sync();
        // count: one compare-and-swap (+ a probe or two) and one add per record
        uint32_t slot[W2C_KPT];
#pragma unroll 1
        for (uint32_t ch = 0; ch < nch; ++ch) {
        if (ch) load_chunk(ch);
#ifdef ELBA_K31_SERIAL_CAS
#pragma unroll
        for (int u = 0; u < W2C_KPT; ++u) {
            slot[u] = 0;
            if (key[u].hi != ~0ull) {
                const unsigned long long hk = key[u].hi;
                uint32_t sl = (((uint32_t)hk ^ (uint32_t)(hk >> 27)) * 0x9E3779B1u) >> 20;      // 12 bits
                if (__hip_atomic_load(&misc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {      // (the table is filling up: the bucket is given up)
                    for (uint32_t probes = 0; probes < W2_SLOTS; ++probes) {      // (bounded whatever happens: a full table cannot hang the wavefront; a plain read in front of the compare-and-swap — most records find their k-mer there — was measured: 44.3 against 42.7 ms for the bucket kernels)
                        const unsigned long long old = atomicCAS(&K[sl], ~0ull, hk);
                        if (old == hk) break;
                        if (old == ~0ull) { if (atomicAdd(&misc[1], 1u) >= W2_DISTINCT_MAX) __hip_atomic_store(&misc[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                        sl = (sl + 1u) & (W2_SLOTS - 1u);
                        if (probes + 1u == W2_SLOTS) __hip_atomic_store(&misc[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    atomicAdd(&CNT[sl], 1u);
                }
                slot[u] = sl;
            }
        }
#else
        {   // (a lane's eight first attempts are in flight together — nine records in ten find their k-mer, or an empty slot, where their hash points —, then the rest probes on: the
            //  loop below used to be eight dependent LDS round trips per lane)
            const bool live = __hip_atomic_load(&misc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u;      // (the table is filling up: the bucket is given up)
            unsigned long long old[W2C_KPT];
#pragma unroll
            for (int u = 0; u < W2C_KPT; ++u) {
                slot[u] = 0; old[u] = 0;
                if (key[u].hi != ~0ull) {
                    const unsigned long long hk = key[u].hi;
                    slot[u] = (((uint32_t)hk ^ (uint32_t)(hk >> 27)) * 0x9E3779B1u) >> 20;      // 12 bits
                    if (live) old[u] = atomicCAS(&K[slot[u]], ~0ull, hk);
                }
            }
#pragma unroll
            for (int u = 0; u < W2C_KPT; ++u) {
                if (live && key[u].hi != ~0ull) {
                    const unsigned long long hk = key[u].hi;
                    uint32_t sl = slot[u];
                    if (old[u] == ~0ull) { if (atomicAdd(&misc[1], 1u) >= W2_DISTINCT_MAX) __hip_atomic_store(&misc[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    else if (old[u] != hk) {
                        for (uint32_t probes = 1; probes < W2_SLOTS; ++probes) {      // (bounded whatever happens: a full table cannot hang the wavefront)
                            if (__hip_atomic_load(&misc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;      // (given up meanwhile: a table that fills up is not probed to its end)
                            sl = (sl + 1u) & (W2_SLOTS - 1u);
                            const unsigned long long o = atomicCAS(&K[sl], ~0ull, hk);
                            if (o == hk) break;
                            if (o == ~0ull) { if (atomicAdd(&misc[1], 1u) >= W2_DISTINCT_MAX) __hip_atomic_store(&misc[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                            if (probes + 1u == W2_SLOTS) __hip_atomic_store(&misc[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        slot[u] = sl;
                    }
                    atomicAdd(&CNT[sl], 1u);
                }
            }
        }
#endif
        }
        lds_sync();
        if (__hip_atomic_load(&misc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {      // more distinct k-mers than the table takes: a crowded bucket like one beyond W2_CAP records
            if (tid == 0) { bN[b] = 0; bZ[b] = 0; crowded[atomicAdd(&gstat->ncrowded, 1u)] = b; }
            lds_sync();
            continue;
        }
        // the reliable k-mers (LOWER <= count <= UPPER), as found
#pragma unroll
        for (int q = 0; q < (int)SPT / 4; ++q) {
            const uint32_t s4 = (uint32_t)q * W2C_THREADS + tid;
            const uint4 c4 = reinterpret_cast<const uint4 *>(CNT)[s4];
            const uint32_t cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t cnt = cc[j];
                if (cnt) {
                    ++st_distinct;
                    if (cnt >= lower && cnt <= upper) { RL[atomicAdd(&misc[0], 1u)] = (uint16_t)(4u * s4 + (uint32_t)j); st_sumsq += (unsigned long long)cnt * cnt; st_maxcol = cnt > st_maxcol ? cnt : st_maxcol; }
                }
            }
        }
        lds_sync();
        const uint32_t Nb = misc[0];
        const bool few = Nb <= W2C_THREADS;
        uint32_t Zb = 0;
        if (few) {
            // ... numbered in value order.  The usual bucket holds a few dozen of them (a HiFi read set: ~1900 records of ~56 genomic k-mers): every lane
            // sums, over the bucket's reliable k-mers, those below its own and their counts — broadcast reads; no ranges, no scans, five barriers less (bucket kernels of the k = 31 workload: 47.6 -> 42.7 ms).
            // (The columns' fill counters are then sbcnt[column], zero since the bucket began.)
#ifdef ELBA_K31_FEW_ONE_LANE
            if (tid < Nb) {
                const uint32_t sl = RL[tid];
                const unsigned long long mine = K[sl];
                uint32_t rc = 0, first = 0, tot = 0;
#pragma unroll 4
                for (uint32_t x = 0; x < Nb; ++x) {
                    const uint32_t sx = RL[x], cx = CNT[sx];
                    const bool below = K[sx] < mine;
                    rc += below ? 1u : 0u; first += below ? cx : 0u; tot += cx;
                }
                RC[sl] = (uint16_t)rc;
                C[rc] = first;
                ktmp[s0 + rc] = mine << (64 - k2);
                if (tid == 0) misc[3] = tot;
            }
#else
            {   // (G lanes per reliable k-mer share the walk — as many as the workgroup has for Nb of them, 8 for the usual 56 — and add their parts up with shuffles:
                //  one wavefront used to walk all Nb alone while the other seven waited)
                uint32_t lg = 0;
                while (lg < 6u && ((Nb << (lg + 1u)) <= (uint32_t)W2C_THREADS)) ++lg;
                const uint32_t G = 1u << lg, i = tid >> lg, g = tid & (G - 1u);
                uint32_t sl = 0, rc = 0, first = 0, tot = 0;
                unsigned long long mine = 0;
                if (i < Nb) {
                    sl = RL[i]; mine = K[sl];
                    for (uint32_t x = g; x < Nb; x += G) {
                        const uint32_t sx = RL[x], cx = CNT[sx];
                        const bool below = K[sx] < mine;
                        rc += below ? 1u : 0u; first += below ? cx : 0u; tot += cx;
                    }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint32_t a = __shfl_xor(rc, d, 64), c2 = __shfl_xor(first, d, 64), e2 = __shfl_xor(tot, d, 64);
                    if ((uint32_t)d < G) { rc += a; first += c2; tot += e2; }
                }
                if (i < Nb && g == 0u) {
                    RC[sl] = (uint16_t)rc;
                    C[rc] = first;
                    ktmp[s0 + rc] = mine << (64 - k2);
                    if (i == 0u) misc[3] = tot;
                }
            }
#endif
            lds_sync();
            Zb = Nb ? misc[3] : 0u;
        } else {
        // ... numbered in value order: value ranges (monotone in the k-mer, like the buckets), count, scan, scatter, rank inside the range
        int rbits = R < 9 ? R : 9;
        while (rbits > 0 && (Nb >> rbits) < 4u) --rbits;
        const int rsh = R - rbits;
        auto range_of = [&](unsigned long long hi) -> uint32_t { return (w2_flat((uint32_t)(hi >> (k2 - 32))) >> rsh) & ((1u << rbits) - 1u); };
        uint32_t isl[IPT], ipos[IPT];
#pragma unroll
        for (int q = 0; q < (int)IPT; ++q) {
            const uint32_t i = (uint32_t)q * W2C_THREADS + tid;
            isl[q] = 0; ipos[q] = 0;
            if (i < Nb) { isl[q] = RL[i]; ipos[q] = atomicAdd(&sbcnt[range_of(K[isl[q]])], 1u); }
        }
        lds_sync();
        {   // exclusive scan of the 512 range counts: one per lane
            const uint32_t c0 = sbcnt[tid];
            uint32_t inc = c0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
            if (lane == 63) wsum[wv] = inc;
            lds_sync();
            uint32_t ex = inc - c0;
            for (uint32_t ww = 0; ww < wv; ++ww) ex += wsum[ww];
            sbstart[tid] = ex;
            if (tid == W2C_THREADS - 1) sbstart[NSB] = ex + c0;
        }
        lds_sync();
#pragma unroll
        for (int q = 0; q < (int)IPT; ++q) if ((uint32_t)q * W2C_THREADS + tid < Nb) SA[sbstart[range_of(K[isl[q]])] + ipos[q]] = (uint16_t)isl[q];
        lds_sync();
#pragma unroll
        for (int q = 0; q < (int)IPT; ++q) {
            if ((uint32_t)q * W2C_THREADS + tid < Nb) {
                const unsigned long long mine = K[isl[q]];
                const uint32_t sb = range_of(mine), lo = sbstart[sb], hi = sbstart[sb + 1];
                uint32_t rank = 0;
                for (uint32_t x = lo; x < hi; ++x) rank += K[SA[x]] < mine ? 1u : 0u;      // (distinct k-mers: a handful per range)
                const uint32_t rc = lo + rank;
                RC[isl[q]] = (uint16_t)rc;
                C[rc] = CNT[isl[q]];
                CNT[isl[q]] = 0;                                                            // (from here on: the column's entries placed so far)
                ktmp[s0 + rc] = mine << (64 - k2);
            }
        }
        lds_sync();
        // first entry of every reliable k-mer: exclusive scan of the counts in value order
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < Nb; c0 += W2C_THREADS) {
            const uint32_t c = c0 + tid, mine = c < Nb ? C[c] : 0u;
            uint32_t inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
            if (lane == 63) wsum[NW + wv] = inc;
            lds_sync();
            uint32_t ex = carry + inc - mine, tot = 0;
            for (uint32_t ww = 0; ww < NW; ++ww) { const uint32_t x = wsum[NW + ww]; if (ww < wv) ex += x; tot += x; }
            if (c < Nb) C[c] = ex;
            carry += tot;
            lds_sync();
        }
        Zb = carry;
        }
        if (Zb > ES_CAP_MAX) {      // (more kept entries than the emit kernels sort in LDS: given up like a bucket of too many distinct k-mers)
            if (tid == 0) { bN[b] = 0; bZ[b] = 0; crowded[atomicAdd(&gstat->ncrowded, 1u)] = b; }
            lds_sync();
            continue;
        }
        const uint32_t vscale = Nb > 1u ? 65535u / (Nb - 1u) : 0u;   // columns spread over the 16 value bits the emit kernels sort by (strictly increasing: vscale >= 1)
#pragma unroll 1
        for (uint32_t ch = 0; ch < nch; ++ch) {
            if (nch > 1u) {      // (the records again; their slots are found, not claimed: every k-mer of the bucket is in the table)
                load_chunk(ch);
#pragma unroll
                for (int u = 0; u < W2C_KPT; ++u) {
                    if (key[u].hi != ~0ull) {
                        const unsigned long long hk = key[u].hi;
                        uint32_t sl = (((uint32_t)hk ^ (uint32_t)(hk >> 27)) * 0x9E3779B1u) >> 20;
                        for (uint32_t probes = 0; probes < W2_SLOTS && K[sl] != hk; ++probes) sl = (sl + 1u) & (W2_SLOTS - 1u);
                        slot[u] = sl;
                    }
                }
            }
            uint32_t rcv[W2C_KPT], at[W2C_KPT];      // (in three steps, so that a lane's look-ups of one kind are in flight together)
#pragma unroll
            for (int u = 0; u < W2C_KPT; ++u) rcv[u] = key[u].hi != ~0ull ? (uint32_t)RC[slot[u]] : 0xFFFFu;
#pragma unroll
            for (int u = 0; u < W2C_KPT; ++u) { at[u] = 0; if (rcv[u] != 0xFFFFu) at[u] = C[rcv[u]] + atomicAdd(few ? &sbcnt[rcv[u]] : &CNT[slot[u]], 1u); }
#pragma unroll
            for (int u = 0; u < W2C_KPT; ++u)
                if (rcv[u] != 0xFFFFu) wrel[s0 + at[u]] = (rk ? (uint64_t)rcv[u] << rk : (uint64_t)(rcv[u] * vscale) << PB) | key[u].lo;
        }
        if (tid == 0) { bN[b] = Nb; bZ[b] = Zb; if (Zb > 8192u) atomicAdd(&gstat->nbig, 1u); else if (Zb > 4096u) atomicAdd(&gstat->nmid, 1u); }
        lds_sync();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        st_distinct += __shfl_xor(st_distinct, d, 64); st_sumsq += __shfl_xor(st_sumsq, d, 64);
        const uint32_t o2 = __shfl_xor(st_maxcol, d, 64); st_maxcol = o2 > st_maxcol ? o2 : st_maxcol;
    }
    if (lane == 0) { if (st_distinct) atomicAdd(&gstat->distinct, st_distinct); if (st_sumsq) atomicAdd(&gstat->sumsq, st_sumsq); if (st_maxcol) atomicMax(&gstat->maxcol, st_maxcol); }
}
constexpr size_t W2C_LDS = ((size_t)2 * W2_SLOTS + W2_SLOTS + (W2_RELMAX + 1) + 512 + 513 + 2 * (W2C_THREADS / 64) + 8) * 4 + ((size_t)W2_SLOTS + 2 * W2_RELMAX) * 2 + 64;

// ---- crowded buckets of the wide partition (round 5) -------------------------------------------------------------------------------------
// A bucket k31_count gives up — more DISTINCT k-mers than its LDS table takes (a satellite array: 10^5 k-mers that share their leading bits), or more
// kept entries than the emit kernels sort in LDS — used to send the WHOLE input to the sort of kmer.hip.  Now such buckets alone are taken out:
//   k31_gather_crowded   their records, contiguous, as (k-mer, payload) pairs;
//   radix_sort_pairs     sorted by the k-mer (prims.hip; buckets are value ranges: the sort keeps them together and in order);
//   k31_crowded_heads / k31_crowded_words   every record gets its k-mer's RANK among the DISTINCT k-mers of its bucket, and a bucket is cut into PSEUDO-BUCKETS
//                        of up to 2^16 distinct k-mers (as many as make ~4096 records): a record becomes the one-word instance of the k <= 17 path, rank's low 16 bits << PB | payload;
//   k_msd_count, k_msd_emit_small, k_msd_bucket   the k <= 17 kernels, unchanged, on the pseudo-buckets (any number of entries per pseudo-bucket: the windowed kernel);
// their counts are folded into their bucket's before the scan over the buckets (k31_fold_pseudo), their k-mer id / entry bases follow from the bucket's
// (k31_pseudo_bases).  A homopolymer (ONE k-mer, millions of records) never comes here: k31_count walks it in chunks.
__global__ __launch_bounds__(256) void k31_gather_crowded(const Rec2 *recs, const uint32_t *b2start, const uint32_t *clist, const uint64_t *coff, uint32_t nc, uint64_t *keys, uint64_t *vals)
{
    for (uint32_t p = blockIdx.y; p < nc; p += gridDim.y) {
        const uint32_t b = clist[p], s0 = b2start[b], n = b2start[b + 1] - s0;
        const uint64_t at = coff[p];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { const Rec2 r = recs[s0 + i]; keys[at + i] = r.hi; vals[at + i] = r.lo; }
    }
}
__global__ __launch_bounds__(256) void k31_crowded_heads(const uint64_t *keys, uint64_t n, uint32_t *head)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// dpos[i] = heads in front of record i (exclusive scan of head); coff[p] = first record of crowded bucket p (a head); pbase[p] = its first pseudo-bucket
__global__ __launch_bounds__(256) void k31_crowded_words(const uint64_t *keys, const uint64_t *vals, uint64_t n, const uint32_t *head, const uint32_t *dpos, const uint64_t *coff, const uint32_t *pbase,
                                                        uint32_t nc, int k2, int PB, uint32_t pshift, uint64_t *words, uint64_t *cdist, uint32_t *b2s, uint32_t *parent_of, uint32_t *dist_base)
{
    const uint32_t pmask = (1u << pshift) - 1u;      // a pseudo-bucket holds 2^pshift distinct k-mers (<= 2^16: the value field of the k <= 17 kernels)
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = nc;      // last p with coff[p] <= i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (coff[mid] <= i) lo = mid; else hi = mid; }
    const uint32_t h = head[i], d = dpos[i] + h - 1u, rank = d - dpos[coff[lo]];      // (the first record of a bucket is a head: its distinct index is dpos there)
    words[i] = ((uint64_t)(rank & pmask) << PB) | vals[i];
    if (h) {
        cdist[d] = keys[i] << (64 - k2);
        if ((rank & pmask) == 0u) { const uint32_t j = pbase[lo] + (rank >> pshift); b2s[j] = (uint32_t)i; parent_of[j] = lo; dist_base[j] = d; }
    }
}
__global__ void k_gather_u32_at(const uint32_t *src, const uint64_t *at, uint32_t n, uint32_t *dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[at[i]];
}
__global__ void k31_fold_pseudo(const uint32_t *bN2, const uint32_t *bZ2, const uint32_t *parent_of, const uint32_t *clist, uint32_t np, uint32_t *bN, uint32_t *bZ)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < np) { const uint32_t b = clist[parent_of[j]]; atomicAdd(&bN[b], bN2[j]); atomicAdd(&bZ[b], bZ2[j]); }
}
__global__ void k31_pseudo_bases(const uint32_t *kidbase, const uint32_t *entbase, const uint32_t *clist, const uint32_t *pbase, uint32_t nc, const uint32_t *bN2, const uint32_t *bZ2, uint32_t *kidbase2, uint32_t *entbase2)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nc) return;
    const uint32_t b = clist[p];
    uint32_t kn = kidbase[b], en = entbase[b];
    for (uint32_t j = pbase[p]; j < pbase[p + 1]; ++j) { kidbase2[j] = kn; entbase2[j] = en; kn += bN2[j]; en += bZ2[j]; }
}

int bits_needed_u(uint64_t maxval)
{
    int b = 1;
    while (b < 64 && (maxval >> b)) ++b;
    return b;
}

}  // namespace

// largest position among the triples (what the layout of a partition word depends on)
__global__ void k_tri_maxpos(const uint32_t *vals, uint64_t Z, unsigned long long *out)
{
    uint32_t mx = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Z; i += (uint64_t)gridDim.x * blockDim.x) { const uint32_t v = vals[i]; mx = v > mx ? v : mx; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o2 = __shfl_xor(mx, d, 64); mx = o2 > mx ? o2 : mx; }
    if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, (unsigned long long)mx);
}
// device triples -> one partition word each: col << PB | row << pbits | pos (msd_matrix_from_triples); indices out of range are counted
__global__ void k_pack_triple_msd(const int64_t *rows, const int64_t *cols, const uint32_t *vals, uint64_t Z, int64_t M, int64_t N, int pbits, int PB, uint64_t *w, unsigned long long *bad)
{
    const uint64_t z = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool b = false;
    if (z < Z) {
        const int64_t r = rows[z], cc = cols[z];
        b = r < 0 || r >= M || cc < 0 || cc >= N;
        w[z] = b ? 0ull : ((uint64_t)cc << PB) | ((uint64_t)r << pbits) | (uint64_t)vals[z];
    }
    const uint64_t bal = __ballot(b);
    if (bal && (threadIdx.x & 63) == 0) atomicAdd(bad, (unsigned long long)__popcll(bal));
}
// triples: entries and columns of every bucket (2^vb consecutive columns, the rank of a column = the low bits of its id), the longest column,
// the sum of the squared column lengths (= the products of A x A^T).  Nothing is rewritten: the emit kernels take the partitioned words.
__global__ __launch_bounds__(256) void k_tri_stats(const uint64_t *words, const uint32_t *b2start, uint32_t nbuckets, int PB, uint32_t vmask, uint32_t small_cap,
                                                  uint32_t *bN, uint32_t *bZ, BucketStats *gstat)
{
    __shared__ uint32_t cnt[8192], red[4];      // (counters per column of a bucket: 2^vb <= 8192)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    unsigned long long st_distinct = 0, st_sumsq = 0;
    uint32_t st_maxcol = 0;
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        const uint32_t s0 = b2start[b], n = b2start[b + 1] - s0;
        if (n == 0 || n > small_cap) {
            if (tid == 0) { bN[b] = 0; bZ[b] = n; if (n) atomicAdd(&gstat->ncrowded, 1u); }
            continue;
        }
        for (uint32_t v = tid; v <= vmask; v += 256) cnt[v] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) atomicAdd(&cnt[(uint32_t)(words[s0 + i] >> PB) & vmask], 1u);
        __syncthreads();
        uint32_t nb = 0;
        for (uint32_t v = tid; v <= vmask; v += 256) {
            const uint32_t cc = cnt[v];
            if (cc) { ++nb; st_sumsq += (unsigned long long)cc * cc; st_maxcol = cc > st_maxcol ? cc : st_maxcol; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nb += __shfl_xor(nb, d, 64);
        if (lane == 0) red[tid >> 6] = nb;
        __syncthreads();
        if (tid == 0) {
            const uint32_t Nb = red[0] + red[1] + red[2] + red[3];
            bN[b] = Nb; bZ[b] = n; st_distinct += Nb;
            if (n > 8192u) atomicAdd(&gstat->nbig, 1u); else if (n > 4096u) atomicAdd(&gstat->nmid, 1u);
        }
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        st_distinct += __shfl_xor(st_distinct, d, 64); st_sumsq += __shfl_xor(st_sumsq, d, 64);
        const uint32_t o2 = __shfl_xor(st_maxcol, d, 64); st_maxcol = o2 > st_maxcol ? o2 : st_maxcol;
    }
    if (lane == 0) { if (st_distinct) atomicAdd(&gstat->distinct, st_distinct); if (st_sumsq) atomicAdd(&gstat->sumsq, st_sumsq); if (st_maxcol) atomicMax(&gstat->maxcol, st_maxcol); }
}

// A matrix handed over as device triples (elba_set_kmer_matrix_device, INTEGRATION.md option B) through the same bucket kernels: the "value" is the
// column id, a bucket 2^vb consecutive columns.  Every column counts as reliable; a matrix with an EMPTY column (the bucket kernels number the
// columns they find), or with a bucket beyond what the LDS sort takes, keeps the sort of matrix.hip.
struct MsdTriples { int64_t M, N; uint64_t maxpos; const int64_t *rows, *cols; const uint32_t *vals; unsigned long long bad; };

// The k-mer stage of one GPU for 9 <= k <= 17 (see the file header).  Leaves behind exactly what runs_to_columns (kmer.hip) leaves: rel_kmers,
// rel_counts, a_colptr, a_csc, the CSR sort keys (csr_words, hint bits included) or kid_of_entry — plus the padded column store.
static bool msd_run(Ctx &c, uint64_t I, elba_kmer_stats *stp, MsdTriples *tri)
{
    const int k = tri ? 17 : c.cfg.k, k2 = 2 * k;
    if (c.opt.kmer_no_msd || c.opt.kmer_pairs || c.opt.kmer_unfused || c.opt.emit_plain || c.opt.kmer_drop) return false;
    if (k > 31 || (!tri && c.cfg.upper > 255) || I == 0) return false;
    // k <= 17: the partition takes all but 16 value bits.  19 <= k <= 31 ("wide", 16-byte records): as many bits as make a bucket of ~1500 instances
    const bool wide = !tri && k2 - VBITS > 2 * MT_MAXBITS;
    int T = k2 - VBITS, vb = VBITS;      // partitioned bits; value bits below them
    if (wide) {
        T = 12;
        while (T < 2 * W2_MAXBITS && (I >> T) > 512) ++T;      // (a bucket may hold W2_CAP: six times this average — k-mers are not spread evenly over real genomes)
        if (c.opt.msd_wide_bits > 0) T = std::min(std::max(c.opt.msd_wide_bits, 2), 2 * W2_MAXBITS);      // (tests: other splits)
        if (T > k2 - 2) return false;
    }
    if (tri) {
        // buckets of ~2048 entries: 2^vb columns of Z / N entries each; at most 2 x 9 partitioned bits
        if (tri->N < 8 || c.opt.csr_pairs) return false;
        const int nbc = bits_needed_u((uint64_t)tri->N - 1);
        const uint64_t avg = (I + (uint64_t)tri->N - 1) / (uint64_t)tri->N;
        vb = 1;
        while (vb < 13 && (avg << (vb + 1)) <= 2048) ++vb;
        if (nbc - vb > 2 * MT_MAXBITS) vb = nbc - 2 * MT_MAXBITS;
        // (a bucket holds at most 1024 columns: the emit kernels keep a column table of half their capacity + 1 — enough for k-mers, which come with LOWER >= 2
        //  entries each, and for 1024 one-entry columns in the smallest class; a bucket beyond the largest class sends the whole matrix to the sort: the
        //  average must stay clear of it)
        if (vb > 10 || (avg << vb) > 6144) return false;
        T = nbc - vb;
    }
    if (T < 2) return false;
    // worth it from ~512 instances per bucket on (the bucket kernels pay a few us per bucket whatever it holds: BASELINE config 2 — 530 per bucket — 6.9 ms
    // here, 7.1 ms through the sort); smaller inputs keep the sort
    if (!c.opt.kmer_msd && (wide ? I < (1ull << 22) : I < ((uint64_t)512 << T))) return false;
    const uint32_t maxlen = c.max_read_len;      // (stage_count_kmers' walk over the read lengths)
    const uint64_t maxpos = tri ? tri->maxpos : maxlen >= (uint32_t)k ? maxlen - (uint32_t)k : 0;
    const int64_t nrows = tri ? tri->M : c.nreads;
    const uint32_t lower = tri ? 1u : (uint32_t)c.cfg.lower, upper = tri ? 0xFFFFu : (uint32_t)c.cfg.upper;
    MsdParams m{};
    m.k2 = k2; m.b1 = (T + 1) / 2; m.b2 = T - m.b1; m.I = I;
    m.pbits = bits_needed_u(maxpos);
    const int mb = bits_needed_u((uint64_t)(nrows > 0 ? nrows - 1 : 0));
    m.PB = mb + m.pbits;
    if ((wide ? VBITS : tri ? T + vb : m.b2 + VBITS) + m.PB > 62) return false;      // (the two top bits of a staged entry carry its hint)
    // an entry's column rank inside its bucket (< 8192: the emit kernels take no more entries) above the 16 value bits, where there is room for it
    // ... and where columns grow long enough for one value to fill a sort range of the emit kernels (UPPER beyond HINT_MAX_COL; the wide path ranks its columns anyway)
    m.rk = (m.PB + VBITS + 13 <= 64 && !c.opt.msd_no_rank && (wide || tri || upper > HINT_MAX_COL || c.opt.msd_rank)) ? m.PB + VBITS : 0;
    m.rkmask = 0xFFFFFFFFu;
    if (tri) { m.rk = m.PB; m.rkmask = (1u << vb) - 1u; m.dup = 1u; }      // (the rank of a column inside its bucket = the low bits of its id: every column holds entries, or the matrix is refused below)
    hipStream_t s = c.stream;
    const uint32_t nb1 = 1u << m.b1, nb2 = 1u << m.b2, nbuckets = nb1 * nb2;
    const uint32_t tile = wide ? (uint32_t)W2_TILE : (uint32_t)MT_TILE;
    // more instances than a 32-bit place holds (or than "kmer_batch_instances": tests): passes over value ranges (reads, k <= 17 only)
    const uint64_t batch_cap = c.opt.kmer_batch_instances > 0 ? (uint64_t)c.opt.kmer_batch_instances : 0xE0000000ull;
    const bool batched = !wide && !tri && I > batch_cap;
    if (!batched && I >= 0xFFFFFFF0ull) return false;      // (the caller refuses: the sort, the wide partition and the triples hold 32-bit places)
    ELBA_REQUIRE((I + tile - 1) / tile < 0xFFFFFFF0ull, ELBA_ERR_UNSUPPORTED, "count_kmers: more than 2^45 k-mer instances");
    const uint64_t Ibuf = batched ? std::min<uint64_t>(I, batch_cap + (I >> (m.b1 - 1)) + (1u << 20)) : I;      // (a pass: the cap, or one digit beyond it — a digit holds ~2 I / 2^b1 at most on canonical k-mers; checked per pass below)
    const uint32_t ntiles1 = (uint32_t)((I + tile - 1) / tile), ntiles2 = ntiles1 + nb1;

    c.ws_a.reserve((size_t)(Ibuf + 2) * (wide ? 16 : 8)); c.ws_c.reserve((size_t)(Ibuf + 2) * (wide ? 16 : 8));
    c.ws_sort.reserve(((size_t)ntiles2 << (wide ? W2_MAXBITS : MT_MAXBITS)) * 4 + 4096);
    c.ws_e.reserve((size_t)(nbuckets + 2) * 4 * 6 + (size_t)(2 * nb1 + 8) * 4 + 256 + 64 + (size_t)(ntiles2 + 4) * 8);
    uint32_t *hist = c.ws_sort.as<uint32_t>();
    BucketStats *gstat = c.ws_e.as<BucketStats>();
    uint32_t *b2start = c.ws_e.as<uint32_t>() + 16, *bN = b2start + (nbuckets + 2), *bZ = bN + (nbuckets + 2), *kidbase = bZ + (nbuckets + 2), *entbase = kidbase + (nbuckets + 2);
    uint32_t *crowded = entbase + (nbuckets + 2), *b1start = crowded + (nbuckets + 2), *tile0 = b1start + (nb1 + 2);
    uint32_t *one_seg = tile0 + (nb1 + 2);      // (triples: the whole input as ONE segment of tiles, for the first digit's pass)
    uint2 *tinfo = reinterpret_cast<uint2 *>(c.ws_e.as<char>() + (((size_t)((char *)(one_seg + 16) - c.ws_e.as<char>()) + 15) & ~(size_t)15));      // (k <= 17: the second pass's tiles)
    uint64_t *wa = c.ws_a.as<uint64_t>(), *wb = c.ws_c.as<uint64_t>();

    c.t_total.start(s);
    c.t_a.start(s);
    EnumParams e{};
    e.packed = c.d_packed; e.byte_off = c.d_byte_off; e.len = c.d_len; e.inst_off = c.inst_off.as<uint64_t>();
    e.nreads = (uint32_t)c.nreads; e.I = I; e.k = k;
    const uint64_t nib = (I >> IB_SHIFT) + 1;
    if (!tri) c.ws_b.reserve((size_t)(nib + 1) * sizeof(BlockInfo));
    const BlockInfo *bi = c.ws_b.as<BlockInfo>();
    if (!tri) hipLaunchKernelGGL(k_block_reads, dim3((unsigned)((nib + 255) / 256)), dim3(256), 0, s, e.inst_off, e.byte_off, e.nreads, nib, c.ws_b.as<BlockInfo>());
    SegTiles sg{b1start, tile0, nb1};
    BucketOut o{};
    const uint32_t small_cap = c.opt.msd_small_cap > 0 && (uint32_t)c.opt.msd_small_cap < ES_CAP_MAX ? (uint32_t)c.opt.msd_small_cap : ES_CAP_MAX;
    static DeviceOnce attr_once;
    attr_once.run(c.device, [&] {
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_msd_count<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_msd_count<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_msd_bucket<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k31_count), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    });
    // pseudo-buckets of the wide partition's crowded buckets (the section above k31_gather_crowded)
    struct { bool on = false; uint32_t nc = 0, np = 0; const uint64_t *words = nullptr; uint64_t *wrel = nullptr; uint32_t *b2s = nullptr, *bN = nullptr, *bZ = nullptr, *kidbase = nullptr, *entbase = nullptr, *crowded = nullptr;
             BucketStats *gstat = nullptr; const uint32_t *clist = nullptr, *pbase = nullptr; const uint64_t *cdist = nullptr; const uint32_t *dist_base = nullptr; BucketStats hs{}; } ps;
    // value-range batching (the section "VALUE-RANGE BATCHING" below): the passes, and what phase A learnt of each
    struct Pass { uint32_t dlo = 0, dhi = 0; uint64_t I = 0, N = 0, Z = 0; BucketStats hs{}; };
    std::vector<Pass> passes;
    std::function<void(uint32_t, uint32_t, uint64_t, bool)> partition_count;
    // k-mers and entries in front of every bucket (+ the totals), the count kernels' statistics: one host round trip
    auto scan_buckets = [&](BucketStats *hs_out, uint64_t *N_out, uint64_t *Z_out) {
        ELBA_HIP(hipMemsetAsync(bN + nbuckets, 0, 4, s)); ELBA_HIP(hipMemsetAsync(bZ + nbuckets, 0, 4, s));
        exclusive_scan_u32(s, bN, kidbase, (int64_t)nbuckets + 1, c.ws_scan);
        exclusive_scan_u32(s, bZ, entbase, (int64_t)nbuckets + 1, c.ws_scan);      // (a pass holds fewer than 2^32 instances)
        if (!hs_out) return;
        uint32_t h2[2] = {0, 0};
        ELBA_HIP(hipMemcpyAsync(&h2[0], kidbase + nbuckets, 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipMemcpyAsync(&h2[1], entbase + nbuckets, 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipMemcpyAsync(hs_out, gstat, sizeof(BucketStats), hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        *N_out = h2[0]; *Z_out = h2[1];
    };
    if (!wide) {
    const int shift2 = m.PB + vb;
#ifndef ELBA_SCATTER_ONE_TILE
    const uint32_t sgrid = c.opt.tune[0] == 1 ? 0xFFFFFFFFu : (uint32_t)c.num_cus * (MT_TILE <= 8192 ? 2u : 1u);      // (tune0 = 1: a workgroup per tile — A/B)
#else
    const uint32_t sgrid = 0xFFFFFFFFu;      // a workgroup per tile
#endif
    if (tri) {
        // first digit, from the packed triples: one segment of ntiles1 tiles
        const uint32_t seg[4] = {0u, (uint32_t)I, 0u, ntiles1};
        ELBA_HIP(hipMemcpyAsync(one_seg, seg, sizeof(seg), hipMemcpyHostToDevice, s));
        SegTiles sg1{one_seg, one_seg + 2, 1u};
        unsigned long long *badctr = reinterpret_cast<unsigned long long *>(one_seg + 4);
        ELBA_HIP(hipMemsetAsync(badctr, 0, 8, s));
        hipLaunchKernelGGL(k_pack_triple_msd, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, s, tri->rows, tri->cols, tri->vals, I, tri->M, tri->N, m.pbits, m.PB, wb, badctr);
        hipLaunchKernelGGL(k_msd_hist2, dim3(ntiles1), dim3(MT_THREADS), 0, s, (const uint64_t *)wb, sg1, shift2 + m.b2, m.b1, hist);
        radix_column_scan(s, hist, (int64_t)ntiles1, nb1, c.ws_scan);
        hipLaunchKernelGGL(k_msd_tiles, dim3(1), dim3(MT_MAXBINS), 0, s, (const uint32_t *)hist, nb1, I, b1start, tile0);
        hipLaunchKernelGGL((k_msd_scatter<false>), dim3(std::min<uint32_t>(ntiles1, sgrid)), dim3(MT_THREADS), 0, s, e, bi, m, (const uint64_t *)wb, sg1, shift2 + m.b2, m.b1, (const uint32_t *)hist, wa, ntiles1, 0u, nb1);
    // second digit, inside every first-digit bucket
    hipLaunchKernelGGL(k_msd_hist2, dim3(ntiles2), dim3(MT_THREADS), 0, s, (const uint64_t *)wa, sg, shift2, m.b2, hist);
    hipLaunchKernelGGL(k_msd_segscan, dim3(nb1), dim3(MT_MAXBINS), 0, s, hist, sg, nb2, b2start, I);
    hipLaunchKernelGGL((k_msd_scatter<false>), dim3(std::min<uint32_t>(ntiles2, sgrid)), dim3(MT_THREADS), 0, s, e, bi, m, (const uint64_t *)wa, sg, shift2, m.b2, (const uint32_t *)hist, wb, ntiles2, 0u, nb1);
    c.t_a.stop(s);
    c.t_b.start(s);
    ELBA_HIP(hipMemsetAsync(gstat, 0, sizeof(BucketStats), s));
    hipLaunchKernelGGL(k_tri_stats, dim3((unsigned)std::min<uint32_t>(nbuckets, (uint32_t)c.num_cus * 8u)), dim3(256), 0, s, (const uint64_t *)wb, (const uint32_t *)b2start, nbuckets, m.PB, m.rkmask, small_cap, bN, bZ, gstat);
    } else {
    // The reads' instances whose FIRST digit lies in [dlo, dhi) — Iv of them; one pass: every instance —: two-level partition, then one workgroup per bucket
    // counts.  hist1 and the first scatter enumerate ALL reads (tiles of the whole instance range) and keep their pass's instances.
    partition_count = [&, shift2, sgrid](uint32_t dlo, uint32_t dhi, uint64_t Iv, bool timed) {
        const uint32_t nt2 = (uint32_t)((Iv + tile - 1) / tile) + nb1;
        if (timed) c.t_a.start(s);
        hipLaunchKernelGGL(k_msd_hist1, dim3(ntiles1), dim3(MT_THREADS), 0, s, e, bi, m, hist, dlo, dhi);
        radix_column_scan(s, hist, (int64_t)ntiles1, nb1, c.ws_scan);
        hipLaunchKernelGGL(k_msd_tiles, dim3(1), dim3(MT_MAXBINS), 0, s, (const uint32_t *)hist, nb1, Iv, b1start, tile0);
        if (batched) hipLaunchKernelGGL((k_msd_scatter<true, true>), dim3(std::min<uint32_t>(ntiles1, sgrid)), dim3(MT_THREADS), 0, s, e, bi, m, (const uint64_t *)nullptr, sg, 0, m.b1, (const uint32_t *)hist, wa, ntiles1, dlo, dhi);
        else hipLaunchKernelGGL((k_msd_scatter<true, false>), dim3(std::min<uint32_t>(ntiles1, sgrid)), dim3(MT_THREADS), 0, s, e, bi, m, (const uint64_t *)nullptr, sg, 0, m.b1, (const uint32_t *)hist, wa, ntiles1, dlo, dhi);
        // second digit, inside every first-digit bucket
        SegTiles sgi = sg; sgi.tinfo = tinfo;
        hipLaunchKernelGGL(k_msd_tile_info, dim3((nt2 + 255u) / 256u), dim3(256), 0, s, sg, tinfo);
        hipLaunchKernelGGL(k_msd_hist2, dim3(nt2), dim3(MT_THREADS), 0, s, (const uint64_t *)wa, sgi, shift2, m.b2, hist);
        hipLaunchKernelGGL(k_msd_segscan, dim3(nb1), dim3(MT_MAXBINS), 0, s, hist, sg, nb2, b2start, Iv);
        hipLaunchKernelGGL((k_msd_scatter<false>), dim3(std::min<uint32_t>(nt2, sgrid)), dim3(MT_THREADS), 0, s, e, bi, m, (const uint64_t *)wa, sgi, shift2, m.b2, (const uint32_t *)hist, wb, nt2, 0u, nb1);
#ifdef ELBA_SCATTER_CLOCK
        {
            ELBA_HIP(hipStreamSynchronize(s));
            unsigned long long h[2][2][12];
            ELBA_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sc_phase), sizeof(h)));
            static const char *nm[12] = {"top", "zero+bar", "rank", "bar", "scan+bar", "place", "bitmap+delta", "writeout", "fetch next", "end bar", "", ""};
            for (int en = 1; en >= 0; --en) for (int wv = 0; wv < 2; ++wv) {
                unsigned long long tot = 0; for (int q = 0; q < 12; ++q) tot += h[en][wv][q];
                fprintf(stderr, "k_msd_scatter<%s> wavefront %d:", en ? "ENUM" : "MEM", wv);
                for (int q = 0; q < 10; ++q) fprintf(stderr, " %s %.1f%%", nm[q], tot ? 100.0 * (double)h[en][wv][q] / (double)tot : 0.0);
                fprintf(stderr, "  (total %.0f cycles of the 100 MHz clock per workgroup)\n", (double)tot / 256.0);
            }
            unsigned long long z[2][2][12] = {};
            ELBA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sc_phase), z, sizeof(z)));
        }
#endif
        if (timed) { c.t_a.stop(s); c.t_b.start(s); }
        // buckets: count
        ELBA_HIP(hipMemsetAsync(gstat, 0, sizeof(BucketStats), s));
        const unsigned bgrid = (unsigned)std::min<uint32_t>(nbuckets, (uint32_t)c.num_cus);
        // (columns of up to HINT_MAX_COL entries: the value bits spread a bucket's entries evenly over the emit kernels' sort ranges already)
        // entries per bucket: up to 2048 / 4096 / 8192 -> k_msd_emit_small on 256 / 512 / 1024 lanes, beyond -> k_msd_bucket (option "msd_small_cap": tests lower the last bound)
        // (the first pass's words are dead: their buffer takes the entries — the instances of reliable k-mers —, bucket by bucket)
        if (m.rk != 0) hipLaunchKernelGGL(k_msd_count<true>, dim3(bgrid), dim3(BK_THREADS), CT_LDS, s, (const uint64_t *)wb, (const uint32_t *)b2start, nbuckets, m, lower, upper, small_cap, bN, bZ, gstat, crowded, wa);
        else hipLaunchKernelGGL(k_msd_count<false>, dim3(bgrid), dim3(BK_THREADS), CT_LDS, s, (const uint64_t *)wb, (const uint32_t *)b2start, nbuckets, m, lower, upper, small_cap, bN, bZ, gstat, crowded, wa);
    };
    if (batched) {
        // VALUE-RANGE BATCHING (round 5; include/elba_amd.h, "limits"): more instances than a 32-bit place holds (or than the option "kmer_batch_instances"
        // allows) are counted in passes over RANGES OF FIRST DIGITS — the reference batches its exchange so that size is no limit (include/KmerOps.hpp:33-56)
        // —: the buckets are value ranges, so pass after pass yields consecutive k-mer ids and consecutive stretches of the columns.  Planning: the
        // per-digit totals of one unfiltered histogram; phase A (here): partition + count of every pass for N, Z and the longest column — what the layout
        // of A depends on —; phase B (below): partition + count again, then the emit with the pass's id / entry bases.
        hipLaunchKernelGGL(k_msd_hist1, dim3(ntiles1), dim3(MT_THREADS), 0, s, e, bi, m, hist, 0u, nb1);
        c.ws_scan.reserve((size_t)nb1 * 8 + 64);
        ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, (size_t)nb1 * 8, s));
        hipLaunchKernelGGL(k_msd_digit_totals, dim3((unsigned)((ntiles1 + 255) / 256)), dim3(nb1), 0, s, (const uint32_t *)hist, (uint64_t)ntiles1, nb1, c.ws_scan.as<unsigned long long>());
        std::vector<unsigned long long> dt(nb1);
        ELBA_HIP(hipMemcpyAsync(dt.data(), c.ws_scan.p, (size_t)nb1 * 8, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        for (uint32_t d = 0; d < nb1;) {
            Pass ps1{}; ps1.dlo = d;
            do { ps1.I += dt[d]; ++d; } while (d < nb1 && ps1.I + dt[d] <= batch_cap);
            ps1.dhi = d;
            ELBA_REQUIRE(ps1.I < 0xFFFFFFF0ull, ELBA_ERR_UNSUPPORTED, "count_kmers: one first-digit bucket alone holds more than 2^32 k-mer instances");
            if (ps1.I) passes.push_back(ps1);
        }
        if (c.opt.trace) fprintf(stderr, "[elba] count_kmers: %llu instances in %zu value-range passes\n", (unsigned long long)I, passes.size());
        for (Pass &ps1 : passes) {
            partition_count(ps1.dlo, ps1.dhi, ps1.I, false);
            scan_buckets(&ps1.hs, &ps1.N, &ps1.Z);
        }
        c.t_a.stop(s); c.t_b.start(s);      // (the stage's two labels: phase A | phase B)
    } else partition_count(0u, nb1, I, true);
    }
    } else {
        // 19 <= k <= 31: 16-byte records (the section above k31_hist1)
        Rec2 *ra = c.ws_a.as<Rec2>(), *rb = c.ws_c.as<Rec2>();
        hipLaunchKernelGGL(k31_hist1, dim3(ntiles1), dim3(W2_THREADS), 0, s, e, bi, m.b1, hist);
        radix_column_scan(s, hist, (int64_t)ntiles1, nb1, c.ws_scan);
        hipLaunchKernelGGL(k31_tiles, dim3(1), dim3(W2_MAXBINS), 0, s, (const uint32_t *)hist, nb1, I, b1start, tile0);
        hipLaunchKernelGGL((k31_scatter<true>), dim3(ntiles1), dim3(W2_THREADS), 0, s, e, bi, k2, m.pbits, (const Rec2 *)nullptr, sg, 32 - m.b1, m.b1, (const uint32_t *)hist, ra);
        hipLaunchKernelGGL(k31_hist2, dim3(ntiles2), dim3(W2_THREADS), 0, s, (const Rec2 *)ra, sg, k2, 32 - T, m.b2, hist);
        hipLaunchKernelGGL(k31_segscan, dim3(nb1), dim3(W2_MAXBINS), 0, s, hist, sg, nb2, b2start, I);
        hipLaunchKernelGGL((k31_scatter<false>), dim3(ntiles2), dim3(W2_THREADS), 0, s, e, bi, k2, m.pbits, (const Rec2 *)ra, sg, 32 - T, m.b2, (const uint32_t *)hist, rb);
        c.t_a.stop(s);
        c.t_b.start(s);
        ELBA_HIP(hipMemsetAsync(gstat, 0, sizeof(BucketStats), s));
        // (the first pass's records are dead: the front half of their buffer takes the kept entries, one word each, the back half the buckets' reliable k-mers)
        hipLaunchKernelGGL(k31_count, dim3((unsigned)std::min<uint32_t>(nbuckets, (uint32_t)c.num_cus * (c.opt.tune[6] == 1 ? 1u : 2u))), dim3(W2C_THREADS), W2C_LDS, s, (const Rec2 *)rb, (const uint32_t *)b2start, nbuckets, k2, T, m.PB, m.rk,
                           (uint32_t)c.cfg.lower, (uint32_t)c.cfg.upper, bN, bZ, gstat, wa, wa + (I + 2), crowded);
        o.kmer_src = wa + (I + 2);
        // buckets k31_count gave up (the section above k31_gather_crowded): taken out, sorted by k-mer, cut into pseudo-buckets of 2^16 distinct k-mers and
        // counted by the k <= 17 kernel — their counts join their bucket's before the scan over the buckets
        BucketStats hs0{};
        ELBA_HIP(hipMemcpyAsync(&hs0, gstat, sizeof(hs0), hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        if (hs0.ncrowded) {
            const uint32_t nc = hs0.ncrowded;
            std::vector<uint32_t> clist(nc), hb2((size_t)nbuckets + 1);
            ELBA_HIP(hipMemcpyAsync(clist.data(), crowded, (size_t)nc * 4, hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipMemcpyAsync(hb2.data(), b2start, ((size_t)nbuckets + 1) * 4, hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipStreamSynchronize(s));
            std::sort(clist.begin(), clist.end());      // (bucket order = value order: what the sorted records follow)
            std::vector<uint64_t> coff((size_t)nc + 1);
            uint64_t Rc = 0;
            for (uint32_t q = 0; q < nc; ++q) { coff[q] = Rc; Rc += hb2[clist[q] + 1] - hb2[clist[q]]; }
            coff[nc] = Rc;
            if (c.opt.trace) fprintf(stderr, "[elba] count_kmers: %u crowded buckets of the wide partition (%llu records) take the pseudo-bucket path\n", nc, (unsigned long long)Rc);
            // workspace: two (key, value) buffer pairs for the sort, head flags + their scan, the distinct k-mers, the small per-bucket arrays
            const size_t R8 = ((size_t)Rc + 8) * 8;
            c.ws_g.reserve(4 * R8 + 2 * (((size_t)Rc + 8) * 4) + R8 + ((size_t)nc + 2) * 24 + 4096);
            char *g0 = c.ws_g.as<char>();
            uint64_t *ck0 = reinterpret_cast<uint64_t *>(g0), *cv0 = reinterpret_cast<uint64_t *>(g0 + R8), *ck1 = reinterpret_cast<uint64_t *>(g0 + 2 * R8), *cv1 = reinterpret_cast<uint64_t *>(g0 + 3 * R8);
            uint32_t *head = reinterpret_cast<uint32_t *>(g0 + 4 * R8), *dpos = head + (Rc + 8);
            uint64_t *cdist = reinterpret_cast<uint64_t *>(g0 + 4 * R8 + 2 * (((size_t)Rc + 8) * 4));
            uint64_t *coff_d = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(cdist) + R8);
            uint32_t *clist_d = reinterpret_cast<uint32_t *>(coff_d + (nc + 2)), *pbase_d = clist_d + (nc + 2), *dfirst_d = pbase_d + (nc + 2);
            ELBA_HIP(hipMemcpyAsync(coff_d, coff.data(), ((size_t)nc + 1) * 8, hipMemcpyHostToDevice, s));
            ELBA_HIP(hipMemcpyAsync(clist_d, clist.data(), (size_t)nc * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k31_gather_crowded, dim3(64, (unsigned)std::min<uint32_t>(nc, 1024u)), dim3(256), 0, s, (const Rec2 *)rb, (const uint32_t *)b2start, (const uint32_t *)clist_d, (const uint64_t *)coff_d, nc, ck0, cv0);
            const int where = radix_sort_pairs(s, ck0, cv0, ck1, cv1, (int64_t)Rc, 0, k2, c.ws_sort);      // (stable; the order inside a k-mer is the emit kernels' business)
            uint64_t *sk = where ? ck1 : ck0, *sv = where ? cv1 : cv0, *words2 = where ? ck0 : ck1, *wrel2 = where ? cv0 : cv1;
            hipLaunchKernelGGL(k31_crowded_heads, dim3((unsigned)((Rc + 255) / 256)), dim3(256), 0, s, (const uint64_t *)sk, Rc, head);
            exclusive_scan_u32(s, head, dpos, (int64_t)Rc, c.ws_scan);
            hipLaunchKernelGGL(k_gather_u32_at, dim3((nc + 255) / 256), dim3(256), 0, s, (const uint32_t *)dpos, (const uint64_t *)coff_d, nc, dfirst_d);
            std::vector<uint32_t> dfirst((size_t)nc + 1);
            uint32_t lasth = 0, lastd = 0;
            ELBA_HIP(hipMemcpyAsync(dfirst.data(), dfirst_d, (size_t)nc * 4, hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipMemcpyAsync(&lasth, head + (Rc - 1), 4, hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipMemcpyAsync(&lastd, dpos + (Rc - 1), 4, hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipStreamSynchronize(s));
            dfirst[nc] = lastd + lasth;      // distinct k-mers of all crowded buckets
            // a pseudo-bucket holds 2^pshift distinct k-mers: as many as make ~4096 records at the crowded buckets' average multiplicity (a satellite at 3-6
            // copies: 1024 k-mers; HiFi coverage, every genomic k-mer ~34 times: 64) — one workgroup counts a pseudo-bucket, the emit kernels sort <= 12288 entries in LDS
            uint32_t pshift = 16;
            { const uint64_t D = std::max<uint32_t>(dfirst[nc], 1u); while (pshift > 4 && ((Rc << pshift) / D) > 4096) --pshift; }
            std::vector<uint32_t> pbase((size_t)nc + 1);
            uint32_t np = 0;
            for (uint32_t q = 0; q < nc; ++q) { pbase[q] = np; np += (uint32_t)(((uint64_t)(dfirst[q + 1] - dfirst[q]) + (1u << pshift) - 1u) >> pshift); }
            pbase[nc] = np;
            ELBA_HIP(hipMemcpyAsync(pbase_d, pbase.data(), ((size_t)nc + 1) * 4, hipMemcpyHostToDevice, s));
            c.ws_h.reserve(((size_t)np + 4) * 4 * 9 + sizeof(BucketStats) + 256);
            uint32_t *b2s = c.ws_h.as<uint32_t>() + 64, *parent_of = b2s + (np + 4), *dist_base = parent_of + (np + 4), *bN2 = dist_base + (np + 4), *bZ2 = bN2 + (np + 4),
                     *kidbase2 = bZ2 + (np + 4), *entbase2 = kidbase2 + (np + 4), *crowded2 = entbase2 + (np + 4);
            BucketStats *gstat2 = c.ws_h.as<BucketStats>();
            static_assert(sizeof(BucketStats) <= 256, "the pseudo-buckets' statistics sit in front of their arrays");
            hipLaunchKernelGGL(k31_crowded_words, dim3((unsigned)((Rc + 255) / 256)), dim3(256), 0, s, (const uint64_t *)sk, (const uint64_t *)sv, Rc, (const uint32_t *)head, (const uint32_t *)dpos, (const uint64_t *)coff_d,
                               (const uint32_t *)pbase_d, nc, k2, m.PB, pshift, words2, cdist, b2s, parent_of, dist_base);
            const uint32_t rc32 = (uint32_t)Rc;
            ELBA_HIP(hipMemcpyAsync(b2s + np, &rc32, 4, hipMemcpyHostToDevice, s));
            ELBA_HIP(hipMemsetAsync(gstat2, 0, sizeof(BucketStats), s));
            const unsigned pgrid = (unsigned)std::min<uint32_t>(np, (uint32_t)c.num_cus);
            if (m.rk) hipLaunchKernelGGL(k_msd_count<true>, dim3(pgrid), dim3(BK_THREADS), CT_LDS, s, (const uint64_t *)words2, (const uint32_t *)b2s, np, m, (uint32_t)c.cfg.lower, (uint32_t)c.cfg.upper, small_cap, bN2, bZ2, gstat2, crowded2, wrel2);
            else hipLaunchKernelGGL(k_msd_count<false>, dim3(pgrid), dim3(BK_THREADS), CT_LDS, s, (const uint64_t *)words2, (const uint32_t *)b2s, np, m, (uint32_t)c.cfg.lower, (uint32_t)c.cfg.upper, small_cap, bN2, bZ2, gstat2, crowded2, wrel2);
            hipLaunchKernelGGL(k31_fold_pseudo, dim3((np + 255) / 256), dim3(256), 0, s, (const uint32_t *)bN2, (const uint32_t *)bZ2, (const uint32_t *)parent_of, (const uint32_t *)clist_d, np, bN, bZ);
            ps.on = true; ps.nc = nc; ps.np = np; ps.words = words2; ps.wrel = wrel2; ps.b2s = b2s; ps.bN = bN2; ps.bZ = bZ2; ps.kidbase = kidbase2; ps.entbase = entbase2; ps.crowded = crowded2;
            ps.gstat = gstat2; ps.clist = clist_d; ps.pbase = pbase_d; ps.cdist = cdist; ps.dist_base = dist_base;
        }
    }
    BucketStats hs{};
    uint64_t N = 0, Z = 0;
    unsigned long long nbad = 0;
    if (batched) {      // (phase A has scanned every pass: the totals, and what the layout of A depends on)
        for (const Pass &ps1 : passes) { N += ps1.N; Z += ps1.Z; hs.distinct += ps1.hs.distinct; hs.sumsq += ps1.hs.sumsq; hs.maxcol = std::max(hs.maxcol, ps1.hs.maxcol); }
    } else {
        if (tri) ELBA_HIP(hipMemcpyAsync(&nbad, one_seg + 4, 8, hipMemcpyDeviceToHost, s));
        if (ps.on) ELBA_HIP(hipMemcpyAsync(&ps.hs, ps.gstat, sizeof(BucketStats), hipMemcpyDeviceToHost, s));
        scan_buckets(&hs, &N, &Z);
        if (ps.on) {
            hipLaunchKernelGGL(k31_pseudo_bases, dim3((ps.nc + 255) / 256), dim3(256), 0, s, (const uint32_t *)kidbase, (const uint32_t *)entbase, ps.clist, ps.pbase, ps.nc, (const uint32_t *)ps.bN, (const uint32_t *)ps.bZ, ps.kidbase, ps.entbase);
            hs.distinct += ps.hs.distinct; hs.sumsq += ps.hs.sumsq; hs.maxcol = std::max(hs.maxcol, ps.hs.maxcol);
        }
        Pass whole{}; whole.dlo = 0; whole.dhi = nb1; whole.I = I; whole.N = N; whole.Z = Z; whole.hs = hs;
        passes.assign(1, whole);
    }
    ELBA_REQUIRE(nbad == 0, ELBA_ERR_INVALID_ARG, "triple index out of range");
    if (tri && (hs.ncrowded || (int64_t)N != tri->N || Z != I)) {      // an empty column (the buckets number the columns they find), a bucket beyond the LDS sort: matrix.hip sorts
        if (c.opt.trace) fprintf(stderr, "[elba] set_kmer_matrix_device: %llu of %lld columns hold entries, %u crowded buckets: sorting instead\n", (unsigned long long)N, (long long)tri->N, hs.ncrowded);
        c.t_b.stop(s); c.t_total.stop(s);
        return false;
    }
    ELBA_REQUIRE(Z < 0xFFFFFFF0ull, ELBA_ERR_UNSUPPORTED, "count_kmers: nnz(A) beyond 32-bit device offsets");
    // buckets: emit
    if (!tri) c.rel_kmers.reserve((size_t)(N + 1) * 8);
    c.rel_counts.reserve((size_t)(N + 2) * 4);
    c.a_colptr.reserve((size_t)(N + 2) * 4);
    c.a_csc.reserve((size_t)(Z + 8) * 8);
    const int nb = bits_needed_u((uint64_t)(N > 0 ? N - 1 : 0)), pb = m.pbits;
    const bool words = mb + nb + pb + 2 <= 64 && !c.opt.csr_pairs;
    const bool hints = pb <= 30 && !c.opt.no_hints;
    o.rel_kmers = tri ? nullptr : c.rel_kmers.as<uint64_t>(); o.rel_counts = c.rel_counts.as<uint32_t>(); o.colptr = c.a_colptr.as<uint32_t>();
    c.max_col_nnz = (int64_t)hs.maxcol;
    choose_column_store(c, (int64_t)N, c.max_col_nnz);
    // inline partners (Ctx::csr_inline): whole matrix, general (not dense) SpGEMM path with position-carrying accumulators, and a sort word wide
    // enough for flag | read | partner >> 1 | posQ | posT
    const bool dense = c.use_ell && maxpos < 65536 && c.max_col_nnz > 16 && !c.opt.no_pay && !c.opt.no_suffix;
    // The inline key is flag | read << rs | (partner >> 1) << 2 pbi | posQ << pbi | posT with the read as high as it goes (rs = 63 - mb): pbi position
    // bits are what is left, and an entry is written inline only if both positions fit them (200 100 reads of up to 16.6 kb: 14 bits, all but
    // the last bases of a handful of reads)
    int rs = nb + pb + 2, pbi = 0;
    bool inl = words && hints && c.use_ell && !dense && maxpos < 65536 && !c.opt.no_pay && !c.opt.no_inline && !c.opt.no_symmetry && N < (1ull << 31) && mb >= 2;
    if (inl) {
        const int rs2 = 63 - mb;
        pbi = std::min(pb, (rs2 - (mb - 1)) / 2);
        if (rs2 >= rs && pbi >= 10) rs = rs2; else inl = false;
    }
    o.csc = c.a_csc.as<uint64_t>(); o.nb = nb; o.pb = pb; o.rs = rs; o.mb = mb; o.inl = inl ? (uint32_t)pbi : 0u; o.hints = hints && words && !dense ? 1u : 0u;
    // a dense matrix's CSR build sorts (read, entry) pairs (matrix.hip, csr_suffix: the entry carries its column's length and its place in it — known
    // here, where the column lies sorted in LDS): they are written instead of sort words, the values where a sort that ends in a_csr starts
    const bool pairs = dense && N < (1ull << 32) && c.max_col_nnz < 128 && !c.opt.csr_pairs_late;
    c.pre_pairs = pairs;
    if (pairs) {
        c.ws_b.reserve((size_t)(Z + 1) * 8); c.ws_d.reserve((size_t)(Z + 1) * 8); c.a_csr.reserve((size_t)(Z + 1) * 8);      // (ws_b: the enumeration's block table is dead)
        o.pair_key = c.ws_b.as<uint32_t>();
        o.pair_val = radix_sort_where((int64_t)Z, 0, mb) == 0 ? c.a_csr.as<uint64_t>() : c.ws_d.as<uint64_t>();
    }
    else if (words) { c.csr_words.reserve((size_t)(Z + 8) * 8); o.csr_words = c.csr_words.as<uint64_t>(); }
    else { c.kid_of_entry.reserve((size_t)(Z + 8) * 8); o.kid_of_entry = c.kid_of_entry.as<uint64_t>(); }
    o.ell = c.use_ell ? c.a_ell.as<uint64_t>() : nullptr; o.ell_stride = c.use_ell ? c.s_stride : 0u;
    // gather slots: with inline partners the padded store holds the columns that are still fetched, not all of them (BucketOut)
    const bool compact = inl && words && c.use_ell && !c.opt.no_ell_compact;
    c.ell_compact = compact; c.ell_nslots = compact ? 0 : (int64_t)N;
    const uint32_t grid16 = std::min<uint32_t>(nbuckets, (uint32_t)c.num_cus * 12u), grid32 = std::min<uint32_t>(nbuckets, (uint32_t)c.num_cus * 4u), grid8 = std::min<uint32_t>(nbuckets, (uint32_t)c.num_cus * 24u);
    if (compact) {
        // Slots are drawn a chunk at a time per workgroup and a chunk's tail may stay unused (a bucket that needs more than what is left takes a new
        // chunk, or exactly what it needs when that is more than a chunk): the store is sized for every column + one chunk per workgroup + the
        // largest single draw a workgroup can leave behind — the slots can never run past it.
        const uint64_t nwg = (uint64_t)grid8 + grid16 + grid32 + (ps.on ? 3ull * std::min<uint32_t>(ps.np, (uint32_t)c.num_cus * 24u) : 0ull);
        const uint32_t chunk = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(4096, N / (4 * nwg)));
        // (a chunk's tail stays unused when the next bucket needs more than what is left: a quarter more than the columns covers every read set seen —
        //  ~4 % are wasted on BASELINE config 3 —; a draw past the store is refused on the device and the emit repeated without slots)
        uint64_t cap_cols = N + N / 4 + nwg * chunk + 4096;
        c.a_ell.reserve((size_t)cap_cols * c.s_stride * 8 + 64);
        if (c.opt.ell_slot_cap > 0 && (uint64_t)c.opt.ell_slot_cap < cap_cols) cap_cols = (uint64_t)c.opt.ell_slot_cap;      // (test hook: a store too small for the slots)
        const int idbits = std::min(32, rs - pb - 2);      // the id field of a sort key: from the hint bits up to the read
        if (idbits < 32 && cap_cols > (1ull << idbits)) cap_cols = 1ull << idbits;      // (slots beyond it are refused like slots beyond the store)
        ELBA_HIP(hipMemsetAsync(c.a_ell.as<char>() + (size_t)cap_cols * c.s_stride * 8, 0xFF, 64, s));
        c.ell_cap_cols = (int64_t)cap_cols;
        o.ell = c.a_ell.as<uint64_t>(); o.slot_chunk = chunk; o.slot_cap = cap_cols;
        c.ell_slot_kid.reserve((size_t)(cap_cols + 1) * 4);
        c.ws_cursor.reserve(64);      // (a buffer of its own: the scans of the later value-range passes use ws_scan)
        ELBA_HIP(hipMemsetAsync(c.ws_cursor.p, 0, 16, s));
        o.slot_cursor = c.ws_cursor.as<unsigned long long>(); o.slot_kid = c.ell_slot_kid.as<uint32_t>(); o.compact = 1u;
    }
    const uint64_t *wrel = tri ? wb : wa;      // (triples: the partitioned words are the entries)
    auto launch_emit = [&](const BucketStats &hs) {      // (hs: the statistics of the pass whose buckets are emitted — which classes hold buckets)
        if (Z == 0) return;
        const uint32_t cap16 = std::min<uint32_t>(small_cap, 4096u), cap8 = std::min<uint32_t>(cap16, c.opt.msd_no_emit8 ? 0u : 2048u);
        // (buckets of up to 2048 entries — more than half of them on BASELINE config 3, where a bucket holds 2040 on average — through an instantiation
        //  with 8 entries per lane: half the predicated-off work of the 16-entry one, 26 KB of LDS instead of 49: six workgroups per CU)
        // (512 lanes x 4 for them was measured too: 17.6 against 17.5 ms on config 3, 48.4 against 47.6 ms on the k = 31 workload)
        const bool spec = m.rk == 0 && m.dup == 0u && o.compact && o.hints && o.csr_words && !o.pair_val && !o.kmer_src && !o.kmer_dist && !o.ncols && o.rel_kmers && c.opt.tune[6] != 3;      // ("tune6" = 3: the general instantiation — A/B)
        if (cap8 && spec) hipLaunchKernelGGL((k_msd_emit_small<8, 256, true>), dim3(grid8), dim3(256), 0, s, wrel, (const uint32_t *)b2start, (const uint32_t *)bZ, nbuckets, m, 0u, cap8,
                                     (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        else
        if (cap8) hipLaunchKernelGGL((k_msd_emit_small<8>), dim3(grid8), dim3(256), 0, s, wrel, (const uint32_t *)b2start, (const uint32_t *)bZ, nbuckets, m, 0u, cap8,
                                     (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        // 2049..4096 entries: 512 lanes x 8 (three workgroups of eight wavefronts per CU, not of four: 19.0 -> 17.5 ms for the bucket kernels on config 3)
        if (cap16 > cap8 && spec) hipLaunchKernelGGL((k_msd_emit_small<8, 512, true>), dim3(grid16), dim3(512), 0, s, wrel, (const uint32_t *)b2start, (const uint32_t *)bZ, nbuckets, m, cap8, cap16,
                           (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        else
        if (cap16 > cap8) hipLaunchKernelGGL((k_msd_emit_small<8, 512>), dim3(grid16), dim3(512), 0, s, wrel, (const uint32_t *)b2start, (const uint32_t *)bZ, nbuckets, m, cap8, cap16,
                           (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        if (small_cap > 4096u && hs.nmid)
            // (8192 entries on 1024 lanes x 8: the 99 KB of LDS allow ONE workgroup per CU — sixteen wavefronts hide the barriers better than eight:
            //  39.7 against 46.8 ms for the bucket kernels on BASELINE config 5 at one GPU's share)
            hipLaunchKernelGGL((k_msd_emit_small<8, 1024>), dim3(grid32), dim3(1024), 0, s, wrel, (const uint32_t *)b2start, (const uint32_t *)bZ, nbuckets, m, 4096u, std::min(small_cap, 8192u),
                               (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        // 8193..12288 entries (the lowest values of a deep read set: the canonical k-mer is the smaller of two, the first buckets hold twice the average):
        // twelve entries per lane, 144 KB of LDS — one kernel launch over a few percent of the buckets instead of the windowed kernel's ~45 us per bucket
        if (small_cap > 8192u && hs.nbig)
            hipLaunchKernelGGL((k_msd_emit_small<12, 1024>), dim3(grid32), dim3(1024), 0, s, wrel, (const uint32_t *)b2start, (const uint32_t *)bZ, nbuckets, m, 8192u, small_cap,
                               (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        if (hs.ncrowded && !wide)
            hipLaunchKernelGGL((k_msd_bucket<true>), dim3((unsigned)std::min<uint32_t>(hs.ncrowded, (uint32_t)c.num_cus)), dim3(BK_THREADS), BK_LDS_EMIT, s, (const uint64_t *)wb, (const uint32_t *)b2start, nbuckets, m,
                               (uint32_t)c.cfg.lower, (uint32_t)c.cfg.upper, (const uint32_t *)crowded, (const BucketStats *)gstat, (const uint32_t *)kidbase, (const uint32_t *)entbase, o);
        if (ps.on) {
            // the pseudo-buckets of the wide partition's crowded buckets: the same kernels on their own bucket table; the k-mer of a column is looked up by its rank among the distinct ones
            BucketOut o2 = o;
            o2.kmer_src = nullptr; o2.kmer_dist = ps.cdist; o2.dist_base = ps.dist_base; o2.ncols = ps.bN;
            const uint32_t np = ps.np, g8 = std::min<uint32_t>(np, (uint32_t)c.num_cus * 24u), g16 = std::min<uint32_t>(np, (uint32_t)c.num_cus * 12u), g32 = std::min<uint32_t>(np, (uint32_t)c.num_cus * 4u);
            if (cap8) hipLaunchKernelGGL((k_msd_emit_small<8>), dim3(g8), dim3(256), 0, s, (const uint64_t *)ps.wrel, (const uint32_t *)ps.b2s, (const uint32_t *)ps.bZ, np, m, 0u, cap8, (const uint32_t *)ps.kidbase, (const uint32_t *)ps.entbase, o2);
            if (cap16 > cap8) hipLaunchKernelGGL((k_msd_emit_small<8, 512>), dim3(g16), dim3(512), 0, s, (const uint64_t *)ps.wrel, (const uint32_t *)ps.b2s, (const uint32_t *)ps.bZ, np, m, cap8, cap16, (const uint32_t *)ps.kidbase, (const uint32_t *)ps.entbase, o2);
            if (small_cap > 4096u && ps.hs.nmid)
                hipLaunchKernelGGL((k_msd_emit_small<8, 1024>), dim3(g32), dim3(1024), 0, s, (const uint64_t *)ps.wrel, (const uint32_t *)ps.b2s, (const uint32_t *)ps.bZ, np, m, 4096u, std::min(small_cap, 8192u), (const uint32_t *)ps.kidbase, (const uint32_t *)ps.entbase, o2);
            if (small_cap > 8192u && ps.hs.nbig)
                hipLaunchKernelGGL((k_msd_emit_small<12, 1024>), dim3(g32), dim3(1024), 0, s, (const uint64_t *)ps.wrel, (const uint32_t *)ps.b2s, (const uint32_t *)ps.bZ, np, m, 8192u, small_cap, (const uint32_t *)ps.kidbase, (const uint32_t *)ps.entbase, o2);
            if (ps.hs.ncrowded)
                hipLaunchKernelGGL((k_msd_bucket<true>), dim3((unsigned)std::min<uint32_t>(ps.hs.ncrowded, (uint32_t)c.num_cus)), dim3(BK_THREADS), BK_LDS_EMIT, s, ps.words, (const uint32_t *)ps.b2s, np, m,
                                   (uint32_t)c.cfg.lower, (uint32_t)c.cfg.upper, (const uint32_t *)ps.crowded, (const BucketStats *)ps.gstat, (const uint32_t *)ps.kidbase, (const uint32_t *)ps.entbase, o2);
        }
    };
    const bool mprep = c.opt.measure_prep && !pairs && words && !batched && !(hs.ncrowded && !wide);      // (the crowded-bucket kernel still reads the partition's buffer)
    c.prep_us = -1; c.emit_us = -1;
    if (mprep) c.t_emit.start(s);
    auto emit_passes = [&]() {
        // one pass: the buckets are counted, emit them.  Value-range batching, phase B: every pass is partitioned and counted again (phase A kept its
        // figures only), its k-mer ids and entries start behind those of the passes before it
        uint64_t Nprev = 0, Zprev = 0;
        for (const Pass &pp : passes) {
            if (batched) {
                partition_count(pp.dlo, pp.dhi, pp.I, false);
                scan_buckets(nullptr, nullptr, nullptr);
                if (Nprev) hipLaunchKernelGGL(k_add_u32, dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, s, kidbase, nbuckets + 1, (uint32_t)Nprev);
                if (Zprev) hipLaunchKernelGGL(k_add_u32, dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, s, entbase, nbuckets + 1, (uint32_t)Zprev);
            }
            launch_emit(pp.hs);
            Nprev += pp.N; Zprev += pp.Z;
        }
    };
    emit_passes();
    if (mprep) {
        // (diagnostic) the same kernels once more, without what they write for the SpGEMM's sake alone: no hint bits, no inline partners, no padded column
        // store / gather slots — the columns, k-mers and counts are rewritten with what they hold, the sort keys go to the partition's dead buffer
        c.t_emit.stop(s);
        const BucketOut keep = o;
        o.hints = 0; o.inl = 0; o.ell = nullptr; o.ell_stride = 0; o.compact = 0; o.csr_words = tri ? wa : wb;
        c.t_emit_plain.start(s);
        launch_emit(passes[0].hs);
        c.t_emit_plain.stop(s);
        o = keep;
    }
    const uint32_t Zz = (uint32_t)Z;
    ELBA_HIP(hipMemcpyAsync(c.a_colptr.as<uint32_t>() + N, &Zz, 4, hipMemcpyHostToDevice, s));
    c.prod_ctr.reserve(64 * 128);
    ELBA_HIP(hipMemsetAsync(c.prod_ctr.p, 0, 64 * 128, s));
    const unsigned long long sq = hs.sumsq;
    ELBA_HIP(hipMemcpyAsync(c.prod_ctr.p, &sq, 8, hipMemcpyHostToDevice, s));
    unsigned long long slots[2] = {0, 0};
    if (compact) ELBA_HIP(hipMemcpyAsync(slots, c.ws_cursor.p, 16, hipMemcpyDeviceToHost, s));
    c.t_b.stop(s);
    c.t_total.stop(s);
    ELBA_HIP(hipStreamSynchronize(s));
    if (mprep) { const float a = c.t_emit.ms(), b = c.t_emit_plain.ms(); c.emit_us = (int64_t)(a * 1000.0f); c.prep_us = (int64_t)((a - b) * 1000.0f); }
    if (compact && slots[1] != 0) {
        // more chunk tails were left unused than the store has room for (never seen; a draw past it writes nothing): every column gets the place
        // of its k-mer id after all — the emit once more, without slots
        if (c.opt.trace) fprintf(stderr, "[elba] gather slots ran past the padded column store (%llu of %llu): emitting again without them\n", slots[0], (unsigned long long)c.ell_cap_cols);
        o.compact = 0; c.ell_compact = false; c.ell_nslots = (int64_t)N; c.ell_cap_cols = (int64_t)N;
        // (the guard words behind column N: choose_column_store's were lost when the store was sized for the slots, and the re-emit writes right up to them)
        ELBA_HIP(hipMemsetAsync(c.a_ell.as<char>() + (size_t)N * c.s_stride * 8, 0xFF, 64, s));
        emit_passes();
        ELBA_HIP(hipStreamSynchronize(s));
    } else if (compact) c.ell_nslots = (int64_t)slots[0];      // (an upper bound of the slots in use: chunks are drawn whole)
    c.pre_ready = true; c.pre_consumed = false; c.pre_words = words; c.pre_hints = hints; c.pre_hints_done = hints && words; c.pre_ell_done = true; c.pre_inline_pending = false;
    c.pre_nb = nb; c.pre_pb = pb; c.pre_maxpos = maxpos; c.pre_rs = rs; c.pre_inline = inl; c.pre_pbi = pbi;
    if (tri) return true;
    elba_kmer_stats &st = *stp;
    st.instances = (int64_t)I; st.distinct = (int64_t)hs.distinct; st.reliable = (int64_t)N; st.entries = (int64_t)Z;
    st.ms_total = c.t_total.ms(); st.ms_count = c.t_a.ms(); st.ms_sort = c.t_b.ms(); st.ms_lookup = 0;
    c.ndistinct = (int64_t)hs.distinct;
    c.N = (int64_t)N; c.Z = (int64_t)Z;
    c.kmer_path = wide ? 2 : 1;
    c.kmer_passes = (int)passes.size();
    return true;
}

bool msd_count_kmers(Ctx &c, uint64_t I, elba_kmer_stats &st) { return msd_run(c, I, &st, nullptr); }

// (matrix.hip, stage_set_kmer_matrix_device: the triples were checked — indices in range, maxpos = the largest position)
bool msd_matrix_from_triples(Ctx &c, int64_t M, int64_t N, int64_t Z, const int64_t *d_rows, const int64_t *d_cols, const uint32_t *d_vals)
{
    if (M <= 0 || N < 8 || Z <= 0 || c.opt.kmer_no_msd || c.opt.csr_pairs || (!c.opt.kmer_msd && Z < (1ll << 20))) return false;
    // the largest position decides the layout of a partition word; the indices are checked by the pass that packs them
    hipStream_t s = c.stream;
    c.ws_scan.reserve(64);
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 8, s));
    hipLaunchKernelGGL(k_tri_maxpos, dim3((unsigned)std::min<int64_t>((Z + 1023) / 1024, (int64_t)c.num_cus * 16)), dim3(256), 0, s, d_vals, (uint64_t)Z, c.ws_scan.as<unsigned long long>());
    unsigned long long maxpos = 0;
    ELBA_HIP(hipMemcpyAsync(&maxpos, c.ws_scan.p, 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    MsdTriples t{M, N, maxpos, d_rows, d_cols, d_vals, 0ull};
    return msd_run(c, (uint64_t)Z, nullptr, &t);
}

}  // namespace elba
