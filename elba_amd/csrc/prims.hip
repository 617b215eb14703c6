// prims.hip — device-wide plumbing primitives written for 64-wide wavefronts: exclusive scan, stable LSD radix
// sort of u64 keys and of (u64,u64) pairs (tile-major histograms + column scan), fills, group offsets.  Used by the k-mer stage and the matrix builders; the SpGEMM
// hot loop does not call into here except for the row-pointer scan.
#include "common.hpp"

namespace elba {

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *total, uint64_t *lds /*[5]*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = wave_inclusive_scan(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint64_t base = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; ++i) {
        uint64_t t = lds[i];
        if (i < w) base += t;
    }
    if (threadIdx.x == SCAN_THREADS - 1) lds[4] = base + inc;
    __syncthreads();
    *total = lds[4];
    __syncthreads();
    return base + inc - v;
}

template <class Tin>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const Tin *in, uint64_t *sums, int64_t n)
{
    __shared__ uint64_t lds[5];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += (uint64_t)in[base + i];
    uint64_t total;
    (void)block_exclusive_scan(s, &total, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <class Tin, class Tout>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const Tin *in, Tout *out, const uint64_t *block_off, int64_t n)
{
    __shared__ uint64_t lds[5];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? (uint64_t)in[base + i] : 0;
        s += v[i];
    }
    uint64_t total;
    uint64_t ex = block_exclusive_scan(s, &total, lds) + (block_off ? block_off[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = (Tout)ex;
        ex += v[i];
    }
}

template <class Tin, class Tout>
void scan_rec(hipStream_t s, const Tin *in, Tout *out, int64_t n, uint64_t *tmp)
{
    if (n <= 0) return;
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb == 1) {
        hipLaunchKernelGGL((k_scan_apply<Tin, Tout>), dim3(1), dim3(SCAN_THREADS), 0, s, in, out, (const uint64_t *)nullptr, n);
        return;
    }
    uint64_t *sums = tmp;
    hipLaunchKernelGGL((k_scan_block_sums<Tin>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, sums, n);
    scan_rec<uint64_t, uint64_t>(s, sums, sums, nb, tmp + nb);
    hipLaunchKernelGGL((k_scan_apply<Tin, Tout>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, (const uint64_t *)sums, n);
}

size_t scan_tmp_elems(int64_t n)
{
    size_t tot = 0;
    while (n > SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; tot += (size_t)n; }
    return tot + 8;
}

template <class T>
__global__ void k_fill(T *p, T v, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

template <class K>
__global__ void k_group_offsets(const K *keys, int shift, int64_t n, uint32_t *ptr, int64_t nkeys)
{
    int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z > n) return;
    int64_t prev = (z == 0) ? -1 : (int64_t)(keys[z - 1] >> shift);
    int64_t cur = (z == n) ? nkeys : (int64_t)(keys[z] >> shift);
    for (int64_t k = prev + 1; k <= cur; ++k) ptr[k] = (uint32_t)z;
}

__global__ void k_reduce_max(const uint64_t *p, int64_t n, unsigned long long *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long m = 0;
    for (; i < n; i += stride) m = p[i] > m ? p[i] : m;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ---------------------------------------------------------------------------------------------------------------
// Stable LSD radix sort, digits of up to 9 bits (34 key bits = 4 passes of 9/9/8/8, 18 bits = 2 passes).  One pass:
//   k_rs_hist      a workgroup counts the digits of its tile of 4096 keys in LDS and stores the row  hist[tile][digit]   (one coalesced row)
//   column scan    exclusive prefix down every digit's column (+ the totals of all smaller digits): hist[tile][digit] becomes the place in
//                  the output of the tile's first key with that digit.  k_cs_sums / k_cs_top / k_cs_apply: thread d walks rows of
//                  column d, so a wavefront always touches 64 consecutive counters.
//   k_rs_scatter   ranks the tile's keys per digit (wave ballots), orders the tile by digit in LDS and writes it out by consecutive lanes.
// The histogram is tile-major: the digit-major layout of round 1 cost a scattered 4-byte store per (tile, digit) in the histogram kernel and
// the same scattered load again in the scatter — 5*10^8 extra memory requests per pass over 2*10^9 keys, as many as the keys themselves make.
// (A one-sweep variant — one kernel per pass, decoupled look-back over per-tile status words — was measured here and was 1.7x slower per
// pass: with ~770 tiles in flight the look-back reads dozens of predecessors' status rows, each an agent-scope 8-byte access.)
#ifndef ELBA_RS_SCATTER_THREADS
#define ELBA_RS_SCATTER_THREADS 512      // (round 5: 512 lanes x 16 keys held to 128 VGPRs — two workgroups of eight wavefronts per CU; 256 x 32 needs 247 VGPRs: eight wavefronts per CU.  Round 4 measured 512 x 16 without the bound — 130 VGPRs, ONE workgroup per CU — and found it slower)
#endif
#ifndef ELBA_RS_PAIR_ITEMS
#define ELBA_RS_PAIR_ITEMS 32
#endif
#ifndef ELBA_RS_KEY_ITEMS
#define ELBA_RS_KEY_ITEMS 32
#endif
constexpr int RS_THREADS = 256;
constexpr int RS_MAXBITS = 9;
constexpr int RS_MAXBINS = 1 << RS_MAXBITS;
constexpr int CS_ROWS = 128;           // rows of the histogram one workgroup of the column scan folds

template <int ITEMS, class K = uint64_t>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const K *keys, int64_t n, int shift, int bits, uint32_t *hist)
{
    __shared__ uint32_t h[RS_MAXBINS];
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    for (uint32_t i = threadIdx.x; i < nbins; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (RS_THREADS * ITEMS);
    K k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t idx = base + r * RS_THREADS + threadIdx.x;
        k[r] = idx < n ? keys[idx] : 0;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t idx = base + r * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(k[r] >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += RS_THREADS) hist[(size_t)blockIdx.x * nbins + i] = h[i];
}

// column sums of CS_ROWS rows: out[chunk][d] = sum over the chunk's rows of in[row][d]; blockDim.x = nbins
__global__ void k_cs_sums(const uint32_t *in, int64_t nrows, uint32_t nbins, uint32_t *out)
{
    const uint32_t d = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS, r1 = r0 + CS_ROWS < nrows ? r0 + CS_ROWS : nrows;
    uint32_t s = 0;
#pragma unroll 16
    for (int64_t r = r0; r < r1; ++r) s += in[(size_t)r * nbins + d];
    out[(size_t)blockIdx.x * nbins + d] = s;
}

// the last level (any number of rows, one workgroup of nbins threads): exclusive prefix down every column, plus the totals of all smaller digits
__global__ void k_cs_top(uint32_t *rows, int64_t nrows, uint32_t nbins)
{
    __shared__ uint32_t wsum[RS_MAXBINS / 64];
    const uint32_t d = threadIdx.x, lane = d & 63, w = d >> 6;
    uint32_t tot = 0;
#pragma unroll 8
    for (int64_t r = 0; r < nrows; ++r) tot += rows[(size_t)r * nbins + d];
    uint32_t inc = tot;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= (uint32_t)s2) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t run = inc - tot;
    for (uint32_t ww = 0; ww < w; ++ww) run += wsum[ww];
    for (int64_t r = 0; r < nrows; ++r) { const uint32_t x = rows[(size_t)r * nbins + d]; rows[(size_t)r * nbins + d] = run; run += x; }
}

// rows[chunk's rows][d] -> exclusive prefix inside the chunk + base[chunk][d]
__global__ void k_cs_apply(uint32_t *rows, int64_t nrows, uint32_t nbins, const uint32_t *base)
{
    const uint32_t d = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS, r1 = r0 + CS_ROWS < nrows ? r0 + CS_ROWS : nrows;
    uint32_t run = base[(size_t)blockIdx.x * nbins + d];
    for (int64_t r = r0; r < r1; r += 8) {
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = r + u < r1 ? rows[(size_t)(r + u) * nbins + d] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) { if (r + u < r1) rows[(size_t)(r + u) * nbins + d] = run; run += x[u]; }
    }
}

void column_scan(hipStream_t s, uint32_t *rows, int64_t nrows, uint32_t nbins, uint32_t *tmp)
{
    if (nrows <= 2 * CS_ROWS) {
        hipLaunchKernelGGL(k_cs_top, dim3(1), dim3(nbins), 0, s, rows, nrows, nbins);
        return;
    }
    const int64_t nchunks = (nrows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(k_cs_sums, dim3((unsigned)nchunks), dim3(nbins), 0, s, (const uint32_t *)rows, nrows, nbins, tmp);
    column_scan(s, tmp, nchunks, nbins, tmp + (size_t)nchunks * nbins);
    hipLaunchKernelGGL(k_cs_apply, dim3((unsigned)nchunks), dim3(nbins), 0, s, rows, nrows, nbins, (const uint32_t *)tmp);
}

size_t column_scan_tmp_elems(int64_t nrows, uint32_t nbins)
{
    size_t tot = 0;
    while (nrows > 2 * CS_ROWS) { nrows = (nrows + CS_ROWS - 1) / CS_ROWS; tot += (size_t)nrows * nbins; }
    return tot + 64;
}

// FIN (the last pass of the CSR build's sort, radix_sort_keys_to_csr): the keys leave UNPACKED — the rows of A — and the row pointers with them.  The
// tile lies ordered by this pass's digit (the read's high bits) in LDS, and inside a digit by the earlier passes' (its low bits): the keys of one read
// are contiguous there.  A key whose left neighbour in the digit's run belongs to another read is the FIRST entry of its read in the whole
// output (an earlier tile holds smaller low bits only) and writes the row pointer; the first key of a run cannot know — the run may continue a
// read of the tile before — and takes the minimum with what is there (rowptr starts as all ones; empty rows are closed by k_rowptr_close_*).
template <bool HAS_VAL, int THREADS, int ITEMS, class K = uint64_t, bool FIN = false>
__global__ __launch_bounds__(THREADS, (THREADS == 512 ? 4 : 0)) void k_rs_scatter(const K *keys_in, const uint64_t *vals_in, K *keys_out, uint64_t *vals_out,
                                                        int64_t n, int shift, int bits, const uint32_t *hist_scanned, CsrFin fin = CsrFin{})
{
    constexpr int TILE = THREADS * ITEMS, WAVES = THREADS / 64, DPT = RS_MAXBINS / THREADS > 0 ? RS_MAXBINS / THREADS : 1;      // digits per thread in the per-digit step (more threads than digits: the others idle there)
    static_assert(ITEMS * 64 < 65536, "a wave's count of a digit fits 16 bits");
    __shared__ uint16_t whist[WAVES][RS_MAXBINS];
    __shared__ uint32_t lstart[RS_MAXBINS], gbase[RS_MAXBINS], wsum[WAVES];
    __shared__ K lkey[TILE];
    __shared__ uint64_t lval[HAS_VAL ? TILE : 1];
    volatile uint16_t(*vh)[RS_MAXBINS] = whist;
    const uint32_t nbins = 1u << bits, dmask = nbins - 1u;
    for (int i = threadIdx.x; i < WAVES * RS_MAXBINS / 2; i += THREADS) reinterpret_cast<uint32_t *>(&whist[0][0])[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1;
    // a wave owns ITEMS*64 CONSECUTIVE items of the tile so that tile order == (wave, round, lane) order: stability
    const int64_t tbase = (int64_t)blockIdx.x * TILE;
    const int64_t wbase = tbase + (int64_t)w * (ITEMS * 64);
    K key[ITEMS];
    uint32_t rank[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t idx = wbase + r * 64 + lane;
        key[r] = idx < n ? keys_in[idx] : 0;
    }
    // (the tile's row of output places: one coalesced load, in flight while the ranks are computed)
    uint32_t gb[DPT];
#pragma unroll
    for (int u = 0; u < DPT; ++u) { const uint32_t d = threadIdx.x + u * THREADS; gb[u] = d < nbins ? hist_scanned[(size_t)blockIdx.x * nbins + d] : 0u; }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t idx = wbase + r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
        uint64_t mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RS_MAXBITS; ++b) {
            if (b < bits) {
                const uint64_t bal = __ballot((d >> b) & 1u);
                mask &= ((d >> b) & 1u) ? bal : ~bal;
            }
        }
        const int leader = valid ? (__ffsll((unsigned long long)mask) - 1) : lane;
        const uint32_t cnt = (uint32_t)__popcll(mask);
        uint32_t pre = 0;
        if (valid && lane == leader) {
            pre = vh[w][d];
            vh[w][d] = (uint16_t)(pre + cnt);
        }
        pre = __shfl(pre, leader, 64);
        rank[r] = pre + (uint32_t)__popcll(mask & lt);
    }
    __syncthreads();
    {
        // per digit (thread t: digits DPT * t ...): this tile's count, the waves' offsets inside the digit's chunk, the chunk's place in the tile
        uint32_t tot[DPT], both = 0;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const uint32_t d = DPT * threadIdx.x + u;
            uint32_t t = 0;
            if (d < nbins) {
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) { const uint32_t x = whist[ww][d]; whist[ww][d] = (uint16_t)t; t += x; }
            }
            tot[u] = t; both += t;
        }
        uint32_t inc = both;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= s2) inc += o; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t run = inc - both;
        for (int ww = 0; ww < w; ++ww) run += wsum[ww];
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = DPT * threadIdx.x + u; if (d < nbins) lstart[d] = run; run += tot[u]; }
#pragma unroll
        for (int u = 0; u < DPT; ++u) { const uint32_t d = threadIdx.x + u * THREADS; if (d < nbins) gbase[d] = gb[u]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t idx = wbase + r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
            const uint32_t lp = lstart[d] + whist[w][d] + rank[r];
            lkey[lp] = key[r];
            if (HAS_VAL) lval[lp] = vals_in[idx];
        }
    }
    __syncthreads();
    const int64_t left = n - tbase;
    const uint32_t nvalid = left < (int64_t)TILE ? (uint32_t)left : (uint32_t)TILE;
#pragma unroll 2      // (two places per trip: their LDS chains — key, then the digit's two offsets — overlap: a plain pass of the CSR sort 3.04 -> 2.92 ms)
    for (uint32_t t = threadIdx.x; t < nvalid; t += THREADS) {
        const K k = lkey[t];
        const uint32_t d = (uint32_t)(k >> shift) & dmask;
        const uint32_t dst = gbase[d] + (t - lstart[d]);
        if (FIN) {
            const uint64_t w = (uint64_t)k, rmask = (1ull << fin.mb) - 1;
            const uint32_t read = (uint32_t)((w >> fin.rs) & rmask);
            if (t == lstart[d]) atomicMin(&fin.rowptr[read], dst);
            else if ((uint32_t)(((uint64_t)lkey[t - 1] >> fin.rs) & rmask) != read) fin.rowptr[read] = dst;
            uint64_t e;
            if (w >> 63) {      // inline partner (Ctx::csr_inline): flag | partner >> 1 | posQ | posT << 16
                const uint64_t pm = (1ull << fin.pbi) - 1;
                e = (1ull << 63) | (((w >> (2 * fin.pbi)) & ((1ull << (fin.mb - 1)) - 1)) << 32) | ((w >> fin.pbi) & pm) | ((w & pm) << 16);
            } else e = (((w >> (fin.pb + 2)) & ((1ull << fin.idbits) - 1)) << 32) | (((w >> fin.pb) & 3ull) << 30) | (w & ((1ull << fin.pb) - 1));
            fin.csr[dst] = e;
        } else {
        keys_out[dst] = k;
        if (HAS_VAL) vals_out[dst] = lval[t];
        }
    }
}

// rowptr[r] = all ones for a read without entries: closed to the next read's first entry (rowptr[M] = n), i.e. a suffix minimum.  Three small
// kernels: per block of 1024 rows the block's minimum, a suffix minimum over the blocks (one workgroup), the rows themselves.
__global__ __launch_bounds__(256) void k_rowptr_close_a(const uint32_t *rowptr, int64_t M1, uint32_t *bmin)
{
    __shared__ uint32_t ws[4];
    const int64_t r0 = (int64_t)blockIdx.x * 1024;
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int64_t r = r0 + u * 256 + threadIdx.x; if (r < M1) { const uint32_t x = rowptr[r]; m = x < m ? x : m; } }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(m, d, 64); m = o < m ? o : m; }
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { for (int q = 1; q < 4; ++q) m = ws[q] < m ? ws[q] : m; bmin[blockIdx.x] = m; }
}
__global__ __launch_bounds__(1024) void k_rowptr_close_b(uint32_t *bmin, int64_t nb)      // bmin[b] becomes the minimum over the blocks BEHIND b
{
    __shared__ uint32_t ws[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t hi = nb; hi > 0; hi -= 1024) {      // chunks of 1024 blocks, from the last to the first
        const int64_t b = hi - 1 - (int64_t)threadIdx.x;      // thread 0 takes the chunk's last block
        const uint32_t mine = b >= 0 ? bmin[b] : 0xFFFFFFFFu;
        uint32_t inc = mine;      // inclusive minimum over the threads 0 .. own (the blocks from the chunk's end down to this one)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d && o < inc) inc = o; }
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        uint32_t before = carry_s;      // everything behind this thread's wavefront
        for (uint32_t q = 0; q < w; ++q) before = ws[q] < before ? ws[q] : before;
        const uint32_t excl_in_wave = __shfl_up(inc, 1, 64);
        uint32_t behind = before;
        if (lane > 0 && excl_in_wave < behind) behind = excl_in_wave;
        if (b >= 0) bmin[b] = behind;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = inc < before ? inc : before;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_rowptr_close_c(uint32_t *rowptr, int64_t M1, const uint32_t *bmin)
{
    // one wavefront per 256 rows would do; kept simple: thread t of the block walks 4 consecutive rows from the back, a suffix minimum over the threads between
    __shared__ uint32_t ws[4];
    const int64_t r0 = (int64_t)blockIdx.x * 1024 + (int64_t)(255 - threadIdx.x) * 4;      // thread 0 holds the block's LAST four rows
    uint32_t x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = r0 + u < M1 ? rowptr[r0 + u] : 0xFFFFFFFFu;
    uint32_t m = x[3];
    m = x[2] < m ? x[2] : m; m = x[1] < m ? x[1] : m; m = x[0] < m ? x[0] : m;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d && o < inc) inc = o; }
    if (lane == 63) ws[w] = inc;
    __syncthreads();
    uint32_t behind = bmin[blockIdx.x];
    for (uint32_t q = 0; q < w; ++q) behind = ws[q] < behind ? ws[q] : behind;
    const uint32_t ex = __shfl_up(inc, 1, 64);
    if (lane > 0 && ex < behind) behind = ex;
    // rows r0+3 .. r0: each takes the minimum of itself and everything behind it
    uint32_t run = behind;
#pragma unroll
    for (int u = 3; u >= 0; --u) { run = x[u] < run ? x[u] : run; if (r0 + u < M1) rowptr[r0 + u] = run; }
}

}  // namespace

void exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, int64_t n, DevBuf &tmp)
{
    tmp.reserve(scan_tmp_elems(n) * sizeof(uint64_t));
    scan_rec<uint32_t, uint32_t>(s, in, out, n, tmp.as<uint64_t>());
}

void exclusive_scan_u32_to_i64(hipStream_t s, const uint32_t *in, int64_t *out, int64_t n, DevBuf &tmp)
{
    tmp.reserve(scan_tmp_elems(n) * sizeof(uint64_t));
    scan_rec<uint32_t, int64_t>(s, in, out, n, tmp.as<uint64_t>());
}

void fill_u32(hipStream_t s, uint32_t *p, uint32_t v, int64_t n)
{
    if (n <= 0) return;
    int64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_fill<uint32_t>, dim3((unsigned)nb), dim3(256), 0, s, p, v, n);
}

void fill_u64(hipStream_t s, uint64_t *p, uint64_t v, int64_t n)
{
    if (n <= 0) return;
    int64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_fill<uint64_t>, dim3((unsigned)nb), dim3(256), 0, s, p, v, n);
}

void group_offsets_u32(hipStream_t s, const uint64_t *sorted_keys, int key_shift, int64_t n, uint32_t *ptr, int64_t nkeys)
{
    int64_t nb = (n + 1 + 255) / 256;
    hipLaunchKernelGGL(k_group_offsets<uint64_t>, dim3((unsigned)nb), dim3(256), 0, s, sorted_keys, key_shift, n, ptr, nkeys);
}

void group_offsets_k32(hipStream_t s, const uint32_t *sorted_keys, int64_t n, uint32_t *ptr, int64_t nkeys)
{
    int64_t nb = (n + 1 + 255) / 256;
    hipLaunchKernelGGL(k_group_offsets<uint32_t>, dim3((unsigned)nb), dim3(256), 0, s, sorted_keys, 0, n, ptr, nkeys);
}

uint64_t reduce_max_u64(hipStream_t s, const uint64_t *p, int64_t n, DevBuf &tmp)
{
    if (n <= 0) return 0;
    tmp.reserve(64);
    ELBA_HIP(hipMemsetAsync(tmp.p, 0, 8, s));
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_reduce_max, dim3((unsigned)nb), dim3(256), 0, s, p, n, tmp.as<unsigned long long>());
    uint64_t h = 0;
    ELBA_HIP(hipMemcpyAsync(&h, tmp.p, 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    return h;
}


namespace {

// digits of a sort of the bits [bit_lo, bit_hi): as even as they come, at most `maxbits` wide (34 bits = 9 + 9 + 8 + 8)
static int radix_digits(int bit_lo, int bit_hi, int *shift, int *bits)
{
    const int B = bit_hi - bit_lo, mb = RS_MAXBITS;
    const int npass = (B + mb - 1) / mb;
    for (int q = 0, at = bit_lo; q < npass; ++q) { bits[q] = B / npass + (q < B % npass ? 1 : 0); shift[q] = at; at += bits[q]; }
    return npass;
}

template <bool HAS_VAL, class K = uint64_t>
static int radix_sort_impl(hipStream_t s, K *k0, uint64_t *v0, K *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp, bool first_hist_done, const CsrFin *fin = nullptr)
{
    if (!fin && (n <= 1 || bit_hi <= bit_lo)) return 0;
    ELBA_REQUIRE(n < (int64_t)0xFFFFFFFFLL, ELBA_ERR_UNSUPPORTED, "radix sort of >= 2^32 items");
    constexpr int ITEMS = HAS_VAL ? ELBA_RS_PAIR_ITEMS : ELBA_RS_KEY_ITEMS, TILE = RS_THREADS * ITEMS;
    // (pairs: 8192-pair tiles on 1024 lanes x 8 — the CSR build of the k = 31 workload 34.1 -> 27.0 ms against 4096-pair tiles on 256 x 16; keys alone
    //  stay on 256 x 32: 10.0 against 10.6 ms on 1024 x 8)
#ifndef ELBA_RS_PAIR_THREADS
#define ELBA_RS_PAIR_THREADS 1024
#endif
    constexpr int STHREADS = HAS_VAL ? ELBA_RS_PAIR_THREADS : ELBA_RS_SCATTER_THREADS;      // the scatter's workgroup (512 threads on the same 8192-key tile were measured: 14.1-14.9 ms per pass against 10.8-13.7)
    int shifts[64], widths[64];
    const int npass = radix_digits(bit_lo, bit_hi, shifts, widths);
    const uint32_t nblocks = (uint32_t)((n + TILE - 1) / TILE);
    const size_t hist_elems = (size_t)nblocks << RS_MAXBITS;
    tmp.reserve((hist_elems + column_scan_tmp_elems((int64_t)nblocks, RS_MAXBINS)) * sizeof(uint32_t));      // (never reallocates behind radix_first_histogram: same size)
    uint32_t *hist = tmp.as<uint32_t>(), *scan_tmp = hist + hist_elems;
    int cur = 0;
    K *ki = k0, *ko = k1;
    uint64_t *vi = v0, *vo = v1;
    for (int q = 0; q < npass; ++q) {
        const int shift = shifts[q], bits = widths[q];
        const uint32_t nbins = 1u << bits;
        if (q > 0 || !first_hist_done)
            hipLaunchKernelGGL((k_rs_hist<ITEMS, K>), dim3(nblocks), dim3(RS_THREADS), 0, s, (const K *)ki, n, shift, bits, hist);
        column_scan(s, hist, (int64_t)nblocks, nbins, scan_tmp);
        if (!HAS_VAL && sizeof(K) == 8 && fin && q == npass - 1)
            hipLaunchKernelGGL((k_rs_scatter<false, STHREADS, TILE / STHREADS, uint64_t, true>), dim3(nblocks), dim3(STHREADS), 0, s, (const uint64_t *)ki, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, n, shift, bits, (const uint32_t *)hist, *fin);
        else
        hipLaunchKernelGGL((k_rs_scatter<HAS_VAL, STHREADS, TILE / STHREADS, K>), dim3(nblocks), dim3(STHREADS), 0, s, (const K *)ki, (const uint64_t *)vi, ko, vo, n, shift, bits, (const uint32_t *)hist, CsrFin{});
        { K *t = ki; ki = ko; ko = t; }
        { uint64_t *t = vi; vi = vo; vo = t; }
        cur ^= 1;
    }
    return cur;
}

}  // namespace

// Exclusive prefix down every column of a tile-major histogram rows[nrows][nbins] plus the totals of all smaller digits (the sort's own
// column scan, for the two-level partition of kmer_msd.hip)
void radix_column_scan(hipStream_t s, uint32_t *rows, int64_t nrows, uint32_t nbins, DevBuf &tmp)
{
    tmp.reserve(column_scan_tmp_elems(nrows, nbins) * sizeof(uint32_t));
    column_scan(s, rows, nrows, nbins, tmp.as<uint32_t>());
}

// For a producer that writes the keys of a radix_sort_keys call itself: where the first pass expects its histogram — row t = the digit
// counts (1 << *bits of them, digit = key >> *shift) of the keys [t * tile, (t + 1) * tile) — so that the producer can count while it writes
// and the sort skips its first histogram pass (radix_sort_keys(..., first_hist_done = true) with the same n, bits and workspace).
uint32_t *radix_first_histogram(int64_t n, int bit_lo, int bit_hi, DevBuf &tmp, int *shift, int *bits, int *tile)
{
    constexpr int TILE = RS_THREADS * ELBA_RS_KEY_ITEMS;
    int shifts[64], widths[64];
    radix_digits(bit_lo, bit_hi, shifts, widths);
    const uint32_t nblocks = (uint32_t)((n + TILE - 1) / TILE);
    const size_t hist_elems = (size_t)nblocks << RS_MAXBITS;
    tmp.reserve((hist_elems + column_scan_tmp_elems((int64_t)nblocks, RS_MAXBINS)) * sizeof(uint32_t));
    *shift = shifts[0]; *bits = widths[0]; *tile = TILE;
    return tmp.as<uint32_t>();
}

// which of its two buffers a sort of n items on the bits [bit_lo, bit_hi) ends in (what radix_sort_pairs / _keys return): a caller that wants the
// result in a buffer of its own hands that buffer over as the one the sort ends in
int radix_sort_where(int64_t n, int bit_lo, int bit_hi)
{
    int shifts[64], widths[64];
    return (n <= 1 || bit_hi <= bit_lo) ? 0 : (radix_digits(bit_lo, bit_hi, shifts, widths) & 1);
}

int radix_sort_pairs(hipStream_t s, uint64_t *k0, uint64_t *v0, uint64_t *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp)
{
    return radix_sort_impl<true>(s, k0, v0, k1, v1, n, bit_lo, bit_hi, tmp, false);
}

// (32-bit keys — row ids — with 64-bit values: 12 bytes per pair and pass instead of 16: the dense matrices' CSR build)
int radix_sort_pairs_k32(hipStream_t s, uint32_t *k0, uint64_t *v0, uint32_t *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp)
{
    return radix_sort_impl<true, uint32_t>(s, k0, v0, k1, v1, n, bit_lo, bit_hi, tmp, false);
}

int radix_sort_keys(hipStream_t s, uint64_t *k0, uint64_t *k1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp, bool first_hist_done)
{
    return radix_sort_impl<false>(s, k0, nullptr, k1, nullptr, n, bit_lo, bit_hi, tmp, first_hist_done);
}

void radix_sort_keys_to_csr(hipStream_t s, uint64_t *k0, uint64_t *k1, int64_t n, const CsrFin &f, DevBuf &tmp)
{
    ELBA_REQUIRE(f.mb >= 1 && n >= 1 && f.M >= 1, ELBA_ERR_INTERNAL, "radix_sort_keys_to_csr: empty matrix");
    const int64_t M1 = f.M + 1;
    ELBA_HIP(hipMemsetAsync(f.rowptr, 0xFF, (size_t)M1 * 4, s));
    const uint32_t nn = (uint32_t)n;
    ELBA_HIP(hipMemcpyAsync(f.rowptr + f.M, &nn, 4, hipMemcpyHostToDevice, s));
    radix_sort_impl<false>(s, k0, nullptr, k1, nullptr, n, f.rs, f.rs + f.mb, tmp, false, &f);
    // (the sort's workspace is free again: the block minima of the row pointers go there)
    const int64_t nb = (M1 + 1023) / 1024;
    tmp.reserve((size_t)(nb + 1) * 4);
    uint32_t *bmin = tmp.as<uint32_t>();
    hipLaunchKernelGGL(k_rowptr_close_a, dim3((unsigned)nb), dim3(256), 0, s, (const uint32_t *)f.rowptr, M1, bmin);
    hipLaunchKernelGGL(k_rowptr_close_b, dim3(1), dim3(1024), 0, s, bmin, nb);
    hipLaunchKernelGGL(k_rowptr_close_c, dim3((unsigned)nb), dim3(256), 0, s, f.rowptr, M1, (const uint32_t *)bmin);
}

}  // namespace elba
