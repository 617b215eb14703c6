// spgemm.hip — B = A·Aᵀ over ELBA's SharedSeeds semiring, then Prune(numshared <= 1).
//
// Replaces create_seed_matrix (src/SharedSeeds.cpp:4-10: CombBLAS Mult_AnXBn_DoubleBuff<SharedSeeds::Semiring> + Prune)
// with a row-wise hash SpGEMM written for CDNA4:
//
//   * one workgroup owns one read-row i of CSR(A); its lanes walk the row's entries (kid, posQ) with coalesced 8-byte
//     loads and, for each, the k-mer's column of CSC(A) — (read j, posT) entries — gathered from HBM/L2;
//   * every product (i,k)x(j,k) updates an open-addressed accumulator keyed by the partner read j that lives in LDS
//     (16 B per slot, SoA: key | count | smin | smax); the semiring's non-commutative add (include/SharedSeeds.hpp:41-46:
//     keep the FIRST seed of the left operand and the FIRST seed of the right operand) is made order-free by the
//     canonical rule of SURVEY.md §8c-2: the row's products carry a sequence number s = (entry index in row << fbits) |
//     (entry index in column) that is monotone in (kid, posQ, posT); ds_min_u32/ds_max_u32 of s give exactly the first
//     and last operand of an ascending-k left fold, ds_add_u32 gives numshared;
//   * survivors (numshared >= 2) are compacted with wavefront ballots + popcount prefix, their two seed positions are
//     decoded from smin/smax, and the row is appended to an HBM staging area; a scan over per-row counts gives the CSR
//     row pointers and a last pass sorts each row's columns and moves it to its final place;
//   * rows whose partner bound exceeds the largest LDS table take the same code path with the table in HBM (spill),
//     so no input can overflow: LDS capacity is a performance tier, not a correctness limit.
//
// No MFMA anywhere: the contraction is index matching plus integer min/max/add.
#include "common.hpp"

namespace elba {

namespace {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int NUM_LDS_BINS = 4;                 // table bits 10, 11, 12, 13
constexpr int NUM_BINS = NUM_LDS_BINS + 1;      // + HBM spill
constexpr int LDS_TBITS0 = 10;

struct OvCounters {              // device-side counters, zeroed per call
    unsigned long long cursor;   // next free staging slot
    unsigned long long products; // P
    unsigned long long yraw;     // nnz before prune
    unsigned long long ndiag, nupper;
    unsigned long long cap_need; // sum_i min(ub_i, M): staging capacity that can never overflow
    unsigned int maxshared;
    unsigned int overflow;       // staging area too small: rerun after growing
    unsigned int bin_count[NUM_BINS];
    unsigned int pad[1];
};

struct OvParams {
    const uint32_t *a_rowptr; const uint64_t *a_csr; const uint32_t *a_colptr; const uint64_t *a_csc;
    uint32_t M;
    uint32_t fbits;
    uint32_t *row_ub;        // [M]
    uint32_t *row_cnt;       // [M+1]
    unsigned long long *row_off;   // [M]
    uint32_t *lists;         // [NUM_BINS][M]
    OvCounters *ctr;
    uint32_t *tmp_col; elba_seed_t *tmp_val; unsigned long long tmp_cap;
    uint32_t *gtable; unsigned long long gstride;   // HBM spill tables: per block 4*gstride u32
};

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- symbolic: products per row, table-size bin ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_bounds(OvParams p)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < p.M; i += nwaves) {
        const uint32_t rs = p.a_rowptr[i], re = p.a_rowptr[i + 1];
        uint32_t ub = 0;
        for (uint32_t e = rs + lane; e < re; e += 64) {
            const uint32_t kid = (uint32_t)(p.a_csr[e] >> 32);
            ub += p.a_colptr[kid + 1] - p.a_colptr[kid];
        }
        ub = wave_sum_u32(ub);
        if (lane == 0) {
            p.row_ub[i] = ub;
            if (ub == 0) { p.row_cnt[i] = 0; p.row_off[i] = 0; continue; }
            const uint32_t need = ub < p.M ? ub : p.M;          // distinct partners <= min(products, reads)
            uint32_t tbits = 32 - __clz(2 * need - 1);          // ceil(log2(2*need)): load factor <= 1/2
            if (need <= 1) tbits = 1;
            int bin = tbits <= LDS_TBITS0 ? 0 : (int)tbits - LDS_TBITS0;
            if (bin > NUM_LDS_BINS) bin = NUM_LDS_BINS;
            const uint32_t at = atomicAdd(&p.ctr->bin_count[bin], 1u);
            p.lists[(size_t)bin * p.M + at] = i;
            atomicAdd(&p.ctr->products, (unsigned long long)ub);
            atomicAdd(&p.ctr->cap_need, (unsigned long long)need);
        }
    }
}

// ---- numeric -----------------------------------------------------------------------------------------------------
template <bool GLOBAL>
struct Table {
    uint32_t *keys, *cnt, *smin, *smax;
    uint32_t tbits;
    __device__ __forceinline__ uint32_t size() const { return 1u << tbits; }
    __device__ __forceinline__ uint32_t home(uint32_t j) const { return (j * 0x9E3779B1u) >> (32 - tbits); }
    __device__ __forceinline__ void insert(uint32_t j, uint32_t s) const
    {
        const uint32_t mask = size() - 1;
        uint32_t slot = home(j);
        for (;;) {
            const uint32_t old = atomicCAS(&keys[slot], EMPTY, j);
            if (old == EMPTY || old == j) break;
            slot = (slot + 1) & mask;
        }
        atomicAdd(&cnt[slot], 1u);
        atomicMin(&smin[slot], s);
        atomicMax(&smax[slot], s);
    }
    __device__ __forceinline__ uint32_t ld(const uint32_t *a, uint32_t slot) const
    {
        if (GLOBAL) return __hip_atomic_load(&a[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2, never a stale L1 line
        return a[slot];
    }
};

template <int BLOCK, bool GLOBAL>
__global__ __launch_bounds__(BLOCK) void k_spgemm_rows(OvParams p, int bin, uint32_t lds_tbits)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // misc words: 0 diag n, 1 diag smin, 2 diag smax, 3 compaction cursor, 4 y, 5 yraw, 6/7 staging offset lo/hi, 8 fits
    uint32_t *misc = GLOBAL ? smem : smem + (size_t)4 * (1u << lds_tbits);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint64_t lt = (1ull << lane) - 1;
    const uint32_t nrows = p.ctr->bin_count[bin];
    const uint32_t fmask = (1u << p.fbits) - 1;

    for (uint32_t it = blockIdx.x; it < nrows; it += gridDim.x) {
        const uint32_t i = p.lists[(size_t)bin * p.M + it];
        Table<GLOBAL> tab;
        if (GLOBAL) {
            const uint32_t ub = p.row_ub[i];
            const uint32_t need = ub < p.M ? ub : p.M;
            tab.tbits = 32 - __clz(2 * need - 1);
            uint32_t *base = p.gtable + (size_t)blockIdx.x * 4 * p.gstride;
            tab.keys = base; tab.cnt = base + p.gstride; tab.smin = base + 2 * p.gstride; tab.smax = base + 3 * p.gstride;
        } else {
            tab.tbits = lds_tbits;
            const uint32_t T = 1u << lds_tbits;
            tab.keys = smem; tab.cnt = smem + T; tab.smin = smem + 2 * T; tab.smax = smem + 3 * T;
        }
        const uint32_t T = tab.size();
        for (uint32_t s = tid; s < T; s += BLOCK) { tab.keys[s] = EMPTY; tab.cnt[s] = 0; tab.smin[s] = 0xFFFFFFFFu; tab.smax[s] = 0; }
        if (tid < 16) misc[tid] = (tid == 1) ? 0xFFFFFFFFu : 0u;
        __syncthreads();

        // ---- expand + accumulate ----
        const uint32_t rs = p.a_rowptr[i], re = p.a_rowptr[i + 1];
        uint32_t dn = 0, dmin = 0xFFFFFFFFu, dmax = 0;
        for (uint32_t e = rs + tid; e < re; e += BLOCK) {
            const uint32_t kid = (uint32_t)(p.a_csr[e] >> 32);
            const uint32_t c0 = p.a_colptr[kid], c1 = p.a_colptr[kid + 1];
            const uint32_t sbase = (e - rs) << p.fbits;
            for (uint32_t f = c0; f < c1; ++f) {
                const uint32_t j = (uint32_t)(p.a_csc[f] >> 32);
                const uint32_t s = sbase | (f - c0);
                if (j == i) { ++dn; dmin = s < dmin ? s : dmin; dmax = s > dmax ? s : dmax; }   // diagonal: registers, not 1 hot slot
                else tab.insert(j, s);
            }
        }
        if (dn) { atomicAdd(&misc[0], dn); atomicMin(&misc[1], dmin); atomicMax(&misc[2], dmax); }
        __syncthreads();

        // ---- count survivors, reserve staging space ----
        uint32_t y = 0, yraw = 0;
        for (uint32_t s = tid; s < T; s += BLOCK) {
            const uint32_t k = tab.ld(tab.keys, s);
            if (k != EMPTY) { ++yraw; if (tab.ld(tab.cnt, s) >= 2) ++y; }
        }
        y = wave_sum_u32(y); yraw = wave_sum_u32(yraw);
        if (lane == 0) { atomicAdd(&misc[4], y); atomicAdd(&misc[5], yraw); }
        __syncthreads();
        if (tid == 0) {
            const uint32_t dcount = misc[0];
            const uint32_t ytot = misc[4] + (dcount >= 2 ? 1u : 0u);
            const unsigned long long off = atomicAdd(&p.ctr->cursor, (unsigned long long)ytot);
            const bool fits = off + ytot <= p.tmp_cap;
            if (!fits) atomicOr(&p.ctr->overflow, 1u);
            p.row_cnt[i] = ytot;
            p.row_off[i] = off;
            misc[4] = ytot; misc[6] = (uint32_t)off; misc[7] = (uint32_t)(off >> 32); misc[8] = fits ? 1u : 0u;
            atomicAdd(&p.ctr->yraw, (unsigned long long)(misc[5] + (dcount >= 1 ? 1u : 0u)));
            if (dcount >= 2) atomicAdd(&p.ctr->ndiag, 1ull);
        }
        __syncthreads();
        const unsigned long long off = ((unsigned long long)misc[7] << 32) | misc[6];
        if (misc[8]) {
            // ---- ballot compaction + seed decode ----
            uint32_t nup = 0, mx = 0;
            for (uint32_t b0 = 0; b0 < T; b0 += BLOCK) {                       // wave-uniform trip count: ballots are safe
                const uint32_t s0 = b0 + tid;
                const bool in = s0 < T;
                uint32_t j = EMPTY, n = 0;
                if (in) { j = tab.ld(tab.keys, s0); if (j != EMPTY) n = tab.ld(tab.cnt, s0); }
                const bool keep = n >= 2;
                const uint64_t bal = __ballot(keep);
                if (bal == 0) continue;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&misc[3], (uint32_t)__popcll(bal));
                base = __shfl(base, 0, 64);
                if (keep) {
                    const uint32_t at = base + (uint32_t)__popcll(bal & lt);
                    const uint32_t a = tab.ld(tab.smin, s0), b = tab.ld(tab.smax, s0);
                    const uint64_t ea = p.a_csr[rs + (a >> p.fbits)], eb = p.a_csr[rs + (b >> p.fbits)];
                    elba_seed_t v;
                    v.q0 = (uint32_t)ea; v.t0 = (uint32_t)p.a_csc[p.a_colptr[(uint32_t)(ea >> 32)] + (a & fmask)];
                    v.q1 = (uint32_t)eb; v.t1 = (uint32_t)p.a_csc[p.a_colptr[(uint32_t)(eb >> 32)] + (b & fmask)];
                    v.numshared = (int32_t)n;
                    p.tmp_col[off + at] = j;
                    p.tmp_val[off + at] = v;
                    if (j > i) ++nup;
                    mx = n > mx ? n : mx;
                }
            }
            __syncthreads();
            if (tid == 0 && misc[0] >= 2) {
                const uint32_t at = misc[3];
                const uint32_t a = misc[1], b = misc[2];
                const uint64_t ea = p.a_csr[rs + (a >> p.fbits)], eb = p.a_csr[rs + (b >> p.fbits)];
                elba_seed_t v;
                v.q0 = (uint32_t)ea; v.t0 = (uint32_t)p.a_csc[p.a_colptr[(uint32_t)(ea >> 32)] + (a & fmask)];
                v.q1 = (uint32_t)eb; v.t1 = (uint32_t)p.a_csc[p.a_colptr[(uint32_t)(eb >> 32)] + (b & fmask)];
                v.numshared = (int32_t)misc[0];
                p.tmp_col[off + at] = i;
                p.tmp_val[off + at] = v;
                mx = misc[0] > mx ? misc[0] : mx;
            }
            nup = wave_sum_u32(nup);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
            if (lane == 0) {
                if (nup) atomicAdd(&p.ctr->nupper, (unsigned long long)nup);
                if (mx) atomicMax(&p.ctr->maxshared, mx);
            }
        }
        __syncthreads();    // table and misc are re-initialised by the next row
    }
}

// ---- finalize: per-row column sort + move to final CSR ------------------------------------------------------------
struct FinParams {
    const uint32_t *row_cnt; const unsigned long long *row_off; const int64_t *b_rowptr;
    const uint32_t *tmp_col; const elba_seed_t *tmp_val;
    uint32_t *b_col; elba_seed_t *b_val;
    uint32_t M;
    uint64_t *sortkeys; unsigned long long sort_stride;
};

constexpr uint32_t FIN_WAVE_MAX = 64;
constexpr uint32_t FIN_LDS_MAX = 4096;

// rows with <= 64 entries: one wavefront per row, rank by 64-wide shuffle compare
__global__ __launch_bounds__(256) void k_finalize_wave(FinParams p)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < p.M; i += nwaves) {
        const uint32_t y = p.row_cnt[i];
        if (y == 0 || y > FIN_WAVE_MAX) continue;
        const unsigned long long off = p.row_off[i];
        const int64_t dst = p.b_rowptr[i];
        uint32_t col = 0xFFFFFFFFu;
        elba_seed_t v{};
        if ((uint32_t)lane < y) { col = p.tmp_col[off + lane]; v = p.tmp_val[off + lane]; }
        uint32_t rank = 0;
        for (uint32_t l = 0; l < y; ++l) rank += (__shfl(col, (int)l, 64) < col) ? 1u : 0u;
        if ((uint32_t)lane < y) { p.b_col[dst + rank] = col; p.b_val[dst + rank] = v; }
    }
}

// bitonic sort of n2 (power of two) u64 keys held in `keys` (LDS or HBM), one workgroup
template <int BLOCK>
__device__ __forceinline__ void bitonic_sort(uint64_t *keys, uint32_t n2)
{
    for (uint32_t k = 2; k <= n2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < n2; t += BLOCK) {
                const uint32_t ixj = t ^ j;
                if (ixj > t) {
                    const uint64_t a = keys[t], b = keys[ixj];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { keys[t] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_finalize_block(FinParams p)
{
    __shared__ uint64_t lkeys[GLOBAL ? 1 : FIN_LDS_MAX];
    for (uint32_t i = blockIdx.x; i < p.M; i += gridDim.x) {
        const uint32_t y = p.row_cnt[i];
        if (GLOBAL ? (y <= FIN_LDS_MAX) : (y <= FIN_WAVE_MAX || y > FIN_LDS_MAX)) continue;
        uint64_t *keys = GLOBAL ? p.sortkeys + (size_t)blockIdx.x * p.sort_stride : lkeys;
        const unsigned long long off = p.row_off[i];
        const int64_t dst = p.b_rowptr[i];
        uint32_t n2 = 1;
        while (n2 < y) n2 <<= 1;
        for (uint32_t t = threadIdx.x; t < n2; t += 256)
            keys[t] = t < y ? (((uint64_t)p.tmp_col[off + t] << 32) | t) : ~0ull;
        __syncthreads();
        bitonic_sort<256>(keys, n2);
        for (uint32_t t = threadIdx.x; t < y; t += 256) {
            const uint64_t k = keys[t];
            p.b_col[dst + t] = (uint32_t)(k >> 32);
            p.b_val[dst + t] = p.tmp_val[off + (uint32_t)k];
        }
        __syncthreads();
    }
}

int bits_for_u(uint64_t v)
{
    int b = 1;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

}  // namespace

void stage_create_seed_matrix(Ctx &c)
{
    ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "create_seed_matrix: no k-mer matrix (call elba_create_kmer_matrix or elba_set_kmer_matrix)");
    hipStream_t s = c.stream;
    const int64_t M = c.M, N = c.N, Z = c.Z;
    elba_overlap_stats st{};
    st.nrows = M;
    c.have_B = false;

    const uint32_t fbits = (uint32_t)bits_for_u((uint64_t)(c.max_col_nnz > 1 ? c.max_col_nnz - 1 : 1));
    ELBA_REQUIRE(fbits < 31 && (uint64_t)c.max_row_nnz <= (1ull << (32 - fbits)), ELBA_ERR_UNSUPPORTED,
                 "row nnz x column nnz exceeds the 32-bit product sequence number");
    ELBA_REQUIRE((uint64_t)c.max_row_nnz * (uint64_t)(c.max_col_nnz > 0 ? c.max_col_nnz : 1) < 0xFFFFFFFFull, ELBA_ERR_UNSUPPORTED,
                 "products per row exceed 32 bits");

    c.ov_rowub.reserve((size_t)(M + 1) * 4);
    c.ov_rowcnt.reserve((size_t)(M + 2) * 4);
    c.ov_rowoff.reserve((size_t)(M + 1) * 8);
    c.ov_lists.reserve((size_t)NUM_BINS * (size_t)(M + 1) * 4);
    c.ov_counters.reserve(sizeof(OvCounters));
    c.b_rowptr.reserve((size_t)(M + 2) * 8);

    // HBM spill tables: one per resident workgroup of the spill kernel
    const int spill_blocks = 64;
    uint64_t gstride = 2;
    while (gstride < 2ull * (uint64_t)(M > 1 ? M : 1)) gstride <<= 1;
    c.ov_gtable.reserve((size_t)spill_blocks * 4 * gstride * 4);

    if (c.ov_tmp_cap == 0) {
        int64_t guess = c.cfg.workspace_hint_bytes > 0 ? c.cfg.workspace_hint_bytes / 24 : 0;
        c.ov_tmp_cap = guess;   // grown below once cap_need is known
    }

    OvParams p{};
    p.a_rowptr = c.a_rowptr.as<uint32_t>(); p.a_csr = c.a_csr.as<uint64_t>();
    p.a_colptr = c.a_colptr.as<uint32_t>(); p.a_csc = c.a_csc.as<uint64_t>();
    p.M = (uint32_t)M; p.fbits = fbits;
    p.row_ub = c.ov_rowub.as<uint32_t>(); p.row_cnt = c.ov_rowcnt.as<uint32_t>();
    p.row_off = c.ov_rowoff.as<unsigned long long>(); p.lists = c.ov_lists.as<uint32_t>();
    p.ctr = c.ov_counters.as<OvCounters>();
    p.gtable = c.ov_gtable.as<uint32_t>(); p.gstride = gstride;

    static bool attr_done = false;
    if (!attr_done) {
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spgemm_rows<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }

    const int cus = c.num_cus;
    OvCounters hc{};
    int passes = 0;
    float ms_sym = 0, ms_num = 0;
    c.t_total.start(s);
    for (;;) {
        ++passes;
        ELBA_HIP(hipMemsetAsync(c.ov_counters.p, 0, sizeof(OvCounters), s));
        ELBA_HIP(hipMemsetAsync(c.ov_rowcnt.p, 0, (size_t)(M + 2) * 4, s));
        c.t_a.start(s);
        if (M > 0) {
            int nb = (int)((M + 3) / 4);
            if (nb > cus * 8) nb = cus * 8;
            hipLaunchKernelGGL(k_row_bounds, dim3(nb), dim3(256), 0, s, p);
        }
        c.t_a.stop(s);
        if (passes == 1 && c.ov_tmp_cap == 0) {
            // first call on this context: size the staging area from the bound that can never overflow (one sync)
            ELBA_HIP(hipMemcpyAsync(&hc, c.ov_counters.p, sizeof(OvCounters), hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipStreamSynchronize(s));
            size_t free_b = 0, total_b = 0;
            ELBA_HIP(hipMemGetInfo(&free_b, &total_b));
            int64_t want = (int64_t)hc.cap_need + 64;
            int64_t budget = (int64_t)(free_b / 2 / 24);
            c.ov_tmp_cap = want < budget ? want : budget;
            if (c.ov_tmp_cap < 1024) c.ov_tmp_cap = 1024;
        }
        c.ov_tmp_col.reserve((size_t)c.ov_tmp_cap * 4);
        c.ov_tmp_val.reserve((size_t)c.ov_tmp_cap * sizeof(elba_seed_t));
        p.tmp_col = c.ov_tmp_col.as<uint32_t>(); p.tmp_val = c.ov_tmp_val.as<elba_seed_t>(); p.tmp_cap = (unsigned long long)c.ov_tmp_cap;

        c.t_b.start(s);
        if (M > 0) {
            // LDS tiers: 1024/2048/4096/8192 slots of 16 B; block size scales with the table so LDS per wave stays 16 KiB
            const int gridcap = cus * 16;
            hipLaunchKernelGGL((k_spgemm_rows<64, false>), dim3(gridcap), dim3(64), (size_t)16 * 1024 + 64, s, p, 0, 10u);
            hipLaunchKernelGGL((k_spgemm_rows<128, false>), dim3(cus * 8), dim3(128), (size_t)16 * 2048 + 64, s, p, 1, 11u);
            hipLaunchKernelGGL((k_spgemm_rows<256, false>), dim3(cus * 4), dim3(256), (size_t)16 * 4096 + 64, s, p, 2, 12u);
            hipLaunchKernelGGL((k_spgemm_rows<256, false>), dim3(cus * 2), dim3(256), (size_t)16 * 8192 + 64, s, p, 3, 13u);
            hipLaunchKernelGGL((k_spgemm_rows<256, true>), dim3(spill_blocks), dim3(256), (size_t)64, s, p, 4, 0u);
        }
        c.t_b.stop(s);
        ELBA_HIP(hipMemcpyAsync(&hc, c.ov_counters.p, sizeof(OvCounters), hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        ms_sym += c.t_a.ms(); ms_num += c.t_b.ms();
        if (!hc.overflow) break;
        ELBA_REQUIRE(passes < 3, ELBA_ERR_INTERNAL, "overlap staging area overflowed twice");
        c.ov_tmp_cap = (int64_t)hc.cursor + 64;      // exact need is now known
    }

    // row pointers, final arrays
    const int64_t Y = (int64_t)hc.cursor;
    c.t_c.start(s);
    exclusive_scan_u32_to_i64(s, c.ov_rowcnt.as<uint32_t>(), c.b_rowptr.as<int64_t>(), M + 1, c.ws_scan);
    c.b_col.reserve((size_t)(Y + 1) * 4);
    c.b_val.reserve((size_t)(Y + 1) * sizeof(elba_seed_t));
    if (M > 0 && Y > 0) {
        FinParams f{};
        f.row_cnt = c.ov_rowcnt.as<uint32_t>(); f.row_off = c.ov_rowoff.as<unsigned long long>(); f.b_rowptr = c.b_rowptr.as<int64_t>();
        f.tmp_col = p.tmp_col; f.tmp_val = p.tmp_val; f.b_col = c.b_col.as<uint32_t>(); f.b_val = c.b_val.as<elba_seed_t>();
        f.M = (uint32_t)M;
        const int gblocks = 32;
        uint64_t sstride = 2;
        while (sstride < (uint64_t)M) sstride <<= 1;
        c.ov_sortkeys.reserve((size_t)gblocks * sstride * 8);
        f.sortkeys = c.ov_sortkeys.as<uint64_t>(); f.sort_stride = sstride;
        int nb = (int)((M + 3) / 4);
        if (nb > cus * 8) nb = cus * 8;
        hipLaunchKernelGGL(k_finalize_wave, dim3(nb), dim3(256), 0, s, f);
        int nb2 = (int)(M < (int64_t)cus * 8 ? M : (int64_t)cus * 8);
        hipLaunchKernelGGL((k_finalize_block<false>), dim3(nb2), dim3(256), 0, s, f);
        hipLaunchKernelGGL((k_finalize_block<true>), dim3(gblocks), dim3(256), 0, s, f);
    }
    c.t_c.stop(s);
    c.t_total.stop(s);
    ELBA_HIP(hipStreamSynchronize(s));

    st.products = (int64_t)hc.products;
    st.nnz = Y;
    st.nnz_diag = (int64_t)hc.ndiag;
    st.nnz_upper = (int64_t)hc.nupper;
    st.max_numshared = (int64_t)hc.maxshared;
    st.rows_lds = (int64_t)hc.bin_count[0] + hc.bin_count[1] + hc.bin_count[2] + hc.bin_count[3];
    st.rows_global = (int64_t)hc.bin_count[4];
    st.algorithmic_bytes = 16 * Z + 8 * (2 * M + N + 3) + 24 * Y;
    st.passes = passes;
    st.nnz_before_prune = (int64_t)hc.yraw;
    st.ms_total = c.t_total.ms();
    st.ms_symbolic = ms_sym;
    st.ms_numeric = ms_num;
    st.ms_finalize = c.t_c.ms();
    c.Y = Y;
    c.ostats = st;
    c.have_B = true;
}

}  // namespace elba
