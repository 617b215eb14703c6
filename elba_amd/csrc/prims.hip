// prims.hip — device-wide plumbing primitives written for 64-wide wavefronts: exclusive scan, stable LSD radix
// sort of (u64,u64) pairs, fills, group offsets.  Used by the k-mer stage and the matrix builders; the SpGEMM
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

__global__ void k_group_offsets(const uint64_t *keys, int shift, int64_t n, uint32_t *ptr, int64_t nkeys)
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
// radix sort, 8 bits per pass
constexpr int RS_THREADS = 256;
#ifndef ELBA_RS_ITEMS
#define ELBA_RS_ITEMS 8
#endif
constexpr int RS_ITEMS = ELBA_RS_ITEMS;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint64_t *keys, int64_t n, int shift, uint32_t *hist, uint32_t nblocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        int64_t idx = base + r * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Scatter of one radix pass.  The tile is first ordered by digit in LDS, then written out by consecutive lanes: a store instruction
// of a wavefront then covers a few digits' chunks (a few pages) instead of up to 64 — on 10^8 items and more, where the 256 output streams of
// a pass lie megabytes apart, the direct per-item scatter ran at a third of the bandwidth it reaches on 10^7 items (address translation).
template <bool HAS_VAL>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint64_t *keys_in, const uint64_t *vals_in, uint64_t *keys_out, uint64_t *vals_out,
                                                           int64_t n, int shift, const uint32_t *hist_scanned, uint32_t nblocks)
{
    __shared__ uint32_t whist[RS_WAVES][256];
    __shared__ uint32_t lstart[256], gbase[256], wsum[RS_WAVES];
    __shared__ uint64_t lkey[RS_TILE], lval[HAS_VAL ? RS_TILE : 1];
    volatile uint32_t(*vh)[256] = whist;
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1;
    // a wave owns RS_ITEMS*64 CONSECUTIVE items of the tile so that tile order == (wave, round, lane) order: stability
    const int64_t tbase = (int64_t)blockIdx.x * RS_TILE;
    const int64_t wbase = tbase + (int64_t)w * (RS_ITEMS * 64);
    uint64_t key[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        int64_t idx = wbase + r * 64 + lane;
        bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0;
        uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
        uint64_t mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            mask &= ((d >> b) & 1u) ? bal : ~bal;
        }
        int leader = valid ? (__ffsll((unsigned long long)mask) - 1) : lane;
        uint32_t cnt = (uint32_t)__popcll(mask);
        uint32_t pre = 0;
        if (valid && lane == leader) {
            pre = vh[w][d];
            vh[w][d] = pre + cnt;
        }
        pre = __shfl(pre, leader, 64);
        rank[r] = pre + (uint32_t)__popcll(mask & lt);
    }
    __syncthreads();
    {
        // per digit: this tile's count, the waves' offsets inside the digit's chunk, the chunk's place in the tile and in the output
        const int d = threadIdx.x;
        uint32_t tot = 0;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) { const uint32_t t = whist[ww][d]; whist[ww][d] = tot; tot += t; }
        uint32_t inc = tot;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t o = __shfl_up(inc, s2, 64); if (lane >= s2) inc += o; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (int ww = 0; ww < w; ++ww) before += wsum[ww];
        lstart[d] = before + inc - tot;
        gbase[d] = hist_scanned[(size_t)d * nblocks + blockIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        int64_t idx = wbase + r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t lp = lstart[d] + whist[w][d] + rank[r];
            lkey[lp] = key[r];
            if (HAS_VAL) lval[lp] = vals_in[idx];
        }
    }
    __syncthreads();
    const int64_t left = n - tbase;
    const uint32_t nvalid = left < (int64_t)RS_TILE ? (uint32_t)left : (uint32_t)RS_TILE;
    for (uint32_t t = threadIdx.x; t < nvalid; t += RS_THREADS) {
        const uint64_t k = lkey[t];
        const uint32_t d = (uint32_t)(k >> shift) & 255u;
        const uint32_t dst = gbase[d] + (t - lstart[d]);
        keys_out[dst] = k;
        if (HAS_VAL) vals_out[dst] = lval[t];
    }
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
    hipLaunchKernelGGL(k_group_offsets, dim3((unsigned)nb), dim3(256), 0, s, sorted_keys, key_shift, n, ptr, nkeys);
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

template <bool HAS_VAL>
static int radix_sort_impl(hipStream_t s, uint64_t *k0, uint64_t *v0, uint64_t *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp)
{
    if (n <= 1 || bit_hi <= bit_lo) return 0;
    ELBA_REQUIRE(n < (int64_t)0xFFFFFFFFLL, ELBA_ERR_UNSUPPORTED, "radix sort of >= 2^32 items");
    const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t hist_elems = (size_t)256 * nblocks;
    const size_t hist_bytes = (hist_elems * sizeof(uint32_t) + 255) & ~(size_t)255;
    tmp.reserve(hist_bytes + scan_tmp_elems((int64_t)hist_elems) * sizeof(uint64_t));
    uint32_t *hist = tmp.as<uint32_t>();
    uint64_t *scan_tmp = reinterpret_cast<uint64_t *>(tmp.as<char>() + hist_bytes);
    int cur = 0;
    uint64_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(RS_THREADS), 0, s, ki, n, shift, hist, nblocks);
        scan_rec<uint32_t, uint32_t>(s, hist, hist, (int64_t)hist_elems, scan_tmp);
        hipLaunchKernelGGL(k_rs_scatter<HAS_VAL>, dim3(nblocks), dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, shift, hist, nblocks);
        uint64_t *t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
        cur ^= 1;
    }
    return cur;
}

int radix_sort_pairs(hipStream_t s, uint64_t *k0, uint64_t *v0, uint64_t *k1, uint64_t *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp)
{
    return radix_sort_impl<true>(s, k0, v0, k1, v1, n, bit_lo, bit_hi, tmp);
}

int radix_sort_keys(hipStream_t s, uint64_t *k0, uint64_t *k1, int64_t n, int bit_lo, int bit_hi, DevBuf &tmp)
{
    return radix_sort_impl<false>(s, k0, nullptr, k1, nullptr, n, bit_lo, bit_hi, tmp);
}

}  // namespace elba
