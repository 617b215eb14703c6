// kmer_enum.hpp — enumeration of the canonical k-mers of packed 2-bit reads (included inside an anonymous namespace by kmer.hip and kmer_msd.hip).
// Packed k-mer layout (src/Kmer.cpp:67-87): base i at bits 2*(31-i), low 64-2k bits zero; canonical = min(kmer, twin)
// (src/Kmer.cpp:200-205); position = forward start index (include/KmerOps.hpp:91-103).
#pragma once

constexpr int EN_ITEMS = 8;                    // instances per lane
constexpr int EN_THREADS = 256;
constexpr int EN_PER_WAVE = 64 * EN_ITEMS;
constexpr int EN_PER_BLOCK = EN_THREADS * EN_ITEMS;

__device__ __forceinline__ uint64_t twin64(uint64_t w, int k)
{
    uint64_t x = ~w;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    return x << (2 * (32 - k));
}

struct EnumParams {
    const uint8_t *packed; const uint64_t *byte_off; const uint32_t *len; const uint64_t *inst_off;
    uint32_t nreads; uint64_t I; int k;
};

// canonical k-mer at position `p` of the read that starts at byte `boff` (two aligned 8-byte loads; the buffer carries 16 guard bytes)
__device__ __forceinline__ uint64_t canonical_at_off(const EnumParams &e, uint64_t boff, uint32_t p)
{
    const uint64_t b = boff + (p >> 2);
    const uint64_t a = b & ~7ull;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(e.packed + a);
    const uint64_t hi = __builtin_bswap64(w[0]), lo = __builtin_bswap64(w[1]);
    const uint32_t sh = (uint32_t)(b - a) * 8 + 2 * (p & 3);              // 0..62
    const uint64_t win = sh ? ((hi << sh) | (lo >> (64 - sh))) : hi;
    const uint64_t fwd = win & (~0ull << (64 - 2 * e.k));
    const uint64_t tw = twin64(fwd, e.k);
    return tw < fwd ? tw : fwd;
}
__device__ __forceinline__ uint64_t canonical_at(const EnumParams &e, uint32_t r, uint32_t p) { return canonical_at_off(e, e.byte_off[r], p); }

// Two-word k-mers, 32 < k <= 63 (NLONGS == 2, include/Kmer.hpp:95-97): bases 0..31 in the first word, the rest left-aligned in the second;
// twin = reverse complement over 128 bits; canonical = the smaller of the two, first word compared first (src/Kmer.cpp:118-131, :200-205).
// Three aligned 8-byte loads cover the window (the k-mer itself spans >= 9 bytes, so the 16 guard bytes behind the reads suffice).
__device__ __forceinline__ uint64_t rev2bit64(uint64_t x)
{
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
}
__device__ __forceinline__ void canonical2_at(const EnumParams &e, uint32_t r, uint32_t p, uint64_t &hi, uint64_t &lo)
{
    const uint64_t b = e.byte_off[r] + (p >> 2);
    const uint64_t a = b & ~7ull;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(e.packed + a);
    const uint64_t w0 = __builtin_bswap64(w[0]), w1 = __builtin_bswap64(w[1]), w2 = __builtin_bswap64(w[2]);
    const uint32_t sh = (uint32_t)(b - a) * 8 + 2 * (p & 3);              // 0..62
    const uint64_t fh = sh ? ((w0 << sh) | (w1 >> (64 - sh))) : w0;
    const uint64_t fl = (sh ? ((w1 << sh) | (w2 >> (64 - sh))) : w1) & (~0ull << (2 * (64 - e.k)));
    // reverse complement of the 128-bit left-aligned value: complement, reverse the 64 two-bit groups, shift the k real ones to the top
    const uint64_t rh = rev2bit64(~fl), rl = rev2bit64(~fh);
    const uint32_t s2 = 2 * (64 - (uint32_t)e.k);                          // 2..62
    const uint64_t th = (rh << s2) | (rl >> (64 - s2)), tl = rl << s2;
    const bool twin = th < fh || (th == fh && tl < fl);
    hi = twin ? th : fh; lo = twin ? tl : fl;
}

// Three-word k-mers, 64 < k <= 95 (NLONGS == 3): the same over 192 bits, four aligned loads (the k-mer spans >= 17 bytes).
__device__ __forceinline__ void canonical3_at(const EnumParams &e, uint32_t r, uint32_t p, uint64_t &k0, uint64_t &k1, uint64_t &k2)
{
    const uint64_t b = e.byte_off[r] + (p >> 2);
    const uint64_t a = b & ~7ull;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(e.packed + a);
    const uint64_t w0 = __builtin_bswap64(w[0]), w1 = __builtin_bswap64(w[1]), w2 = __builtin_bswap64(w[2]), w3 = __builtin_bswap64(w[3]);
    const uint32_t sh = (uint32_t)(b - a) * 8 + 2 * (p & 3);              // 0..62
    const uint64_t f0 = sh ? ((w0 << sh) | (w1 >> (64 - sh))) : w0;
    const uint64_t f1 = sh ? ((w1 << sh) | (w2 >> (64 - sh))) : w1;
    const uint64_t f2 = (sh ? ((w2 << sh) | (w3 >> (64 - sh))) : w2) & (~0ull << (2 * (96 - e.k)));
    const uint64_t r0 = rev2bit64(~f2), r1 = rev2bit64(~f1), r2 = rev2bit64(~f0);     // reversed order of the words
    const uint32_t s2 = 2 * (96 - (uint32_t)e.k);                                      // 2..62
    const uint64_t t0 = (r0 << s2) | (r1 >> (64 - s2)), t1 = (r1 << s2) | (r2 >> (64 - s2)), t2 = r2 << s2;
    const bool twin = t0 != f0 ? t0 < f0 : (t1 != f1 ? t1 < f1 : t2 < f2);
    k0 = twin ? t0 : f0; k1 = twin ? t1 : f1; k2 = twin ? t2 : f2;
}

// Calls f(instance index g, read r, pos p) for the EN_ITEMS instances of this lane; instances of a wave are consecutive, so the read
// is found by ONE binary search per wave plus a short forward walk.
template <class F>
__device__ __forceinline__ void for_each_position(const EnumParams &e, F &&f)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * EN_THREADS + threadIdx.x) >> 6;
    const uint64_t g0 = wave * EN_PER_WAVE;
    if (g0 >= e.I) return;
    uint32_t lo = 0, hi = e.nreads;                 // last r with inst_off[r] <= g0
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (e.inst_off[mid] <= g0) lo = mid; else hi = mid;
    }
    uint32_t r = lo;
#pragma unroll
    for (int it = 0; it < EN_ITEMS; ++it) {
        const uint64_t g = g0 + (uint64_t)it * 64 + lane;
        if (g >= e.I) break;
        while (g >= e.inst_off[r + 1]) ++r;         // reads shorter than k have empty ranges and are skipped here
        f(g, r, (uint32_t)(g - e.inst_off[r]));
    }
}

// Calls f(instance index g, read r, pos p, canonical k-mer) for the EN_ITEMS instances of this lane among the EN_PER_WAVE that start at g0 (k <= 31)
template <class F>
__device__ __forceinline__ void for_each_instance_from(const EnumParams &e, uint64_t g0, F &&f)
{
    const uint32_t lane = threadIdx.x & 63;
    if (g0 >= e.I) return;
    uint32_t lo = 0, hi = e.nreads;                 // last r with inst_off[r] <= g0
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (e.inst_off[mid] <= g0) lo = mid; else hi = mid;
    }
    uint32_t r = lo;
    uint64_t off_lo = e.inst_off[r], off_hi = e.inst_off[r + 1], boff = e.byte_off[r];      // the read's bounds stay in registers: reloaded only where a lane crosses into the next read
#pragma unroll
    for (int it = 0; it < EN_ITEMS; ++it) {
        const uint64_t g = g0 + (uint64_t)it * 64 + lane;
        if (g >= e.I) break;
        while (g >= off_hi) { ++r; off_lo = off_hi; off_hi = e.inst_off[r + 1]; boff = e.byte_off[r]; }         // reads shorter than k have empty ranges and are skipped here
        const uint32_t p = (uint32_t)(g - off_lo);
        f(g, r, p, canonical_at_off(e, boff, p));
    }
}
template <class F>
__device__ __forceinline__ void for_each_instance(const EnumParams &e, F &&f)
{
    for_each_instance_from(e, (((uint64_t)blockIdx.x * EN_THREADS + threadIdx.x) >> 6) * EN_PER_WAVE, f);
}

// read of the first instance of every block of 2^IB_SHIFT instances: an entry then finds its read with ONE table load and a step or two
// along the instance offsets instead of a binary search over all reads (18 dependent L2 round trips per entry on 200 k reads: the
// search was 54 ms of the 219 ms k-mer stage of the 200 k-read set)
constexpr int IB_SHIFT = 11;
// the read holding the block's first instance, in ONE 16-byte load: its index, the block's first position in it, the instances it still
// holds from there, and its byte offset (0xFFFFFFFF: does not fit 32 bits — fetched from the reads' offsets instead)
struct alignas(16) BlockInfo { uint32_t read, pos0, remain, byte_off; };
struct ReadCursor { uint32_t lo; uint64_t off_lo, off_hi, boff; };
__device__ __forceinline__ ReadCursor cursor_at(const EnumParams &e, const BlockInfo *block_read, uint64_t g)
{
    const uint64_t g0 = g & ~((1ull << IB_SHIFT) - 1);
    const BlockInfo bi = block_read[g >> IB_SHIFT];
    ReadCursor c;
    c.lo = bi.read; c.off_lo = g0 - bi.pos0; c.off_hi = g0 + bi.remain;
    c.boff = bi.byte_off != 0xFFFFFFFFu ? (uint64_t)bi.byte_off : e.byte_off[bi.read];
    return c;
}
__global__ void k_block_reads(const uint64_t *inst_off, const uint64_t *byte_off, uint32_t nreads, uint64_t nblocks, BlockInfo *block_read)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint64_t g = b << IB_SHIFT;
    uint32_t lo = 0, hi = nreads;                                     // last read with inst_off[read] <= g
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (inst_off[mid] <= g) lo = mid; else hi = mid; }
    const uint64_t rem = inst_off[lo + 1] - g, bo = byte_off[lo];      // (g beyond the last instance: the table's closing entry, never dereferenced past)
    block_read[b] = BlockInfo{lo, (uint32_t)(g - inst_off[lo]), (uint32_t)(inst_off[lo + 1] > g ? rem : 0u), bo < 0xFFFFFFFFull ? (uint32_t)bo : 0xFFFFFFFFu};
}

