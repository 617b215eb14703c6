// spgemm.hip — B = A·Aᵀ over ELBA's SharedSeeds semiring, then Prune(numshared <= 1).
//
// Replaces create_seed_matrix (src/SharedSeeds.cpp:4-10: CombBLAS Mult_AnXBn_DoubleBuff<SharedSeeds::Semiring> + Prune)
// with a row-wise hash SpGEMM written for CDNA4:
//
//   * one workgroup owns one read-row i of A (spgemm_direct.hpp, the plan-free kernel: it walks CSR(A) and fetches the k-mer columns as the
//     k-mer stage leaves them);
//   * every product (i,k)x(j,k) updates an open-addressed accumulator keyed by the partner read j that lives in LDS
//     (SoA: key | count | first | last); the semiring's non-commutative add (include/SharedSeeds.hpp:41-46: keep the FIRST seed of the left
//     operand and the FIRST seed of the right operand) is made order-free by the canonical rule of SURVEY.md §8c-2: the row's products carry
//     a sequence number s = (entry index in row << fbits) | (entry index in column) that is monotone in (kid, posQ, posT); ds_min / ds_max
//     of s give exactly the first and last operand of an ascending-k left fold, ds_add_u32 gives numshared;
//   * tables are sized OPTIMISTICALLY (512 ... 8192 slots) from an estimate of the row's distinct partners (row entries x a ratio that a
//     cold call measures on a sample of rows first); a row that fills its table beyond 3/4 is abandoned and re-queued on the next tier; the
//     last tier keeps the table in HBM, sized by the bound that cannot overflow (min(products, reads)).  LDS capacity is therefore a
//     performance tier, never a correctness limit;
//   * a pair of rows is accumulated on ONE of them (owns_pair); survivors (numshared >= 2) are compacted with wavefront ballots + popcount
//     prefix and appended to an HBM staging area; a scan over per-row counts gives the CSR row pointers, k_mirror hands every survivor's
//     transposed image to the partner's row, and a last pass sorts each row's columns and moves it to its final place.
//
// No MFMA anywhere: the contraction is index matching plus integer min/max/add.
#include "common.hpp"
#include <algorithm>
#include <type_traits>

namespace elba {

namespace {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int NUM_LDS_TIERS = 5;                // 512, 1024, 2048, 4096, 8192 slots (18 B per slot incl. the 16-bit survivor list)
constexpr int NUM_TIERS = NUM_LDS_TIERS + 1;    // + HBM spill
constexpr int LDS_TBITS0 = 9;
constexpr uint32_t STAGE_CHUNK = 1024;          // staging entries a workgroup draws from the global cursor at a time
constexpr uint32_t FIN_WAVE_MAX = 256;          // widest row the one-wave rank sort takes
constexpr uint32_t FIN_WAVE2_MAX = 1024;        // widest row the one-wave bucket sort takes (wider rows: one workgroup each)
constexpr uint32_t FIN_LDS_MAX = 4096;          // widest row the LDS bitonic sort takes

// End-of-kernel statistics are flushed into one of 64 shards (each on its own 128-B line): thousands of workgroups adding to a
// single line serialise at ~15 ns per atomic, which showed up as 0.1-0.2 ms on a 0.6 ms kernel.
struct alignas(128) OvShard {
    unsigned long long yraw, nnz, ndiag, nupper, fb_claims, fb_ub, products;
    unsigned int maxshared;
    unsigned int tier_done[NUM_TIERS];
};
constexpr int NUM_SHARDS = 64;

struct OvCounters {              // device-side counters, zeroed per call
    unsigned long long cursor;   // next free staging slot
    unsigned long long products; // P
    unsigned long long yraw;     // nnz before prune
    unsigned long long nnz;      // nnz after prune (the cursor also counts unused chunk tails)
    unsigned long long ndiag, nupper;
    unsigned long long cap_need; // sum_i min(ub_i, M): staging capacity that can never overflow
    unsigned int maxshared;
    unsigned int overflow;       // staging area too small: rerun after growing
    unsigned int tier_count[NUM_TIERS];   // rows queued per tier (grows while lower tiers escalate rows)
    unsigned int tier_done[NUM_TIERS];    // rows completed per tier
    unsigned int fin_count[2];            // rows needing the workgroup bucket sort / the HBM-bitonic sort of their columns
    unsigned int slab_q16;                // mirror slabs: slab entries per row entry of A (x 65536) of this call, 0 = no slabs (k_classify_direct writes it: the sample's rows run before it is known)
    unsigned int ntick;                   // staged entries whose image took a ticket: the list k_mirror walks (OvParams::tick, 16-byte records)
    unsigned long long mir_placed;        // diagnostic: mirrored entries k_mirror placed (those that found no room in their row's slab, or had none)
    unsigned long long pad2[13];          // keep the feedback sums on a cache line of their own
    alignas(128) unsigned int sample_next[8][32];          // the same for the sample queue (below)
    unsigned int sample_count;                             // rows of the sample queue: a few hundred rows computed FIRST on a cold call — their distinct-partner ratio then picks the other rows' tiers
    alignas(128) unsigned int tier_next[NUM_TIERS][8][32]; // plan-free kernel: heads ([..][..][0]) of every tier's 8 interleaved sub-queues, a 128-byte line each
    alignas(128)
    unsigned long long fb_claims, fb_ub;  // feedback: distinct partners found / products, summed over rows done so far in this call
    unsigned long long fb_surv;           // entries the sample's rows staged (what sizes the mirror slabs)
    OvShard shard[NUM_SHARDS];
    unsigned long long phase[12];          // diagnostic (cfg.flags & 16): shader-clock cycles per kernel phase, summed over workgroups
};

// One staged or mirrored entry of B: 32 bytes, 32-byte aligned, moved as two 16-byte words — a scattered record is one full
// sector (20-byte seeds at a 20-byte stride straddle sectors and make every scattered store a partial write).
//   a = (partner read, mirror ticket or ~0, q0, t0)   b = (q1, t1, numshared, 0)
struct alignas(32) StageRec { uint4 a, b; };

struct OvParams {
    // plan-free path (spgemm_direct.hpp): the two orientations of A and nothing else
    const uint32_t *a_rowptr; const uint64_t *a_csr; const uint64_t *a_ell; const uint32_t *a_colptr; const uint64_t *a_csc;
    const uint32_t *a_ellj;         // dense matrices: the partner reads of every column, right-aligned in an aligned block of 1 << j_shift four-byte slots (what the dense path gathers)
    uint32_t j_shift, dense_up;
    uint32_t hint_mask, pos_mask;   // which hint bit of a row entry lets this call skip its column (0: none) / the position bits (Ctx::csr_hints)
    uint32_t inl;                   // Ctx::csr_inline: row entries with bit 63 set carry their (only) partner: 1 << 63 | (partner >> 1) << 32 | posQ | posT << 16
    uint32_t *tick_rows;            // bit i: row i staged at least one entry whose image took a TICKET (k_mirror places those; rows without one are not walked)
    uint32_t min_tier;              // rows start on this tier at least (k_classify_direct)
    uint32_t qblk_log2;             // the tier queues' sub-queues take blocks of this many (log2) consecutive places (spgemm_direct.hpp: qplace)
    uint32_t pay16;                 // the LDS tiers' 32-bit accumulators carry posT (sequence number << 16 | posT) and the FIFO entries are 8 bytes: positions and every row's
                                    // sequence numbers fit 16 bits (spgemm_direct.hpp) — 18 bytes per table slot and 2 KB of rings per wavefront: three rows per CU on the 2048-slot tier
    uint32_t suffix;                // dense matrices (Ctx::csr_suffix): row entries carry column length and own place, the smaller row owns a pair
    uint32_t s_stride, lpc_log2, max_col;    // padded column stride in entries (a_ell); lanes per row entry 2^lpc_log2; longest column
    unsigned long long fb_enough;            // row entries behind the in-call partner / entry ratio at which it counts as settled (nobody touches the hot sums any more)
    uint32_t M;              // number of rows of A held here
    uint32_t Mcols;          // number of reads overall (partner id range)
    uint32_t row_lo, row_hi; // rows of B computed by this context
    uint32_t fbits;
    uint32_t half;           // 1: the schedule lists an in-window pair on its smaller row only; survivors are mirrored into the partner's row
    uint32_t tier_limit[NUM_LDS_TIERS];   // claimed slots at which a row abandons the tier: min(3T/4, T - BLOCK) - 1 (every lane can overshoot by one claim)
    uint32_t use_feedback;   // 1 on the first call for a matrix (no measured prior yet): in-call self-correction through the hot fb_* sums
    uint32_t prior_q16;      // distinct-partners / products estimate in 1/65536 units (1/16 before anything is known; measured by the previous call afterwards)
    uint32_t *row_cnt;       // [M+1] entries the row staged itself (partners it was scheduled with + diagonal)
    uint32_t *low_cnt;       // [M+1] zero at entry: mirrored entries per row; its returning atomic hands every mirrored entry its slot
    unsigned long long *row_off;   // [M]
    uint32_t *lists;         // [NUM_TIERS][M]
    const uint32_t *row_order, *row_label;      // dense path with partners named by label (Ctx::row_order: label -> row, Ctx::row_label: row -> label); or null
    uint32_t *sample_list; uint32_t nsample, sstep;      // cold calls: rows row_lo + q * sstep, q < nsample, are the sample (k_classify_direct, mode 1)
    uint32_t *fin_lists;     // [2][M]
    OvCounters *ctr;
    StageRec *tmp; unsigned long long tmp_cap;
    uint32_t rec16; uint4 *rec; uint32_t *tick;     // rec16: positions fit 16 bits — a staged entry is ONE 16-byte word (partner, q0 | t0 << 16, q1 | t1 << 16, numshared) in `rec` (the staging area itself) + its mirror ticket in `tick` (behind the tmp_cap words)
    uint32_t *gtable; unsigned long long gstride;   // HBM spill tables: per block 4*gstride u32
    // mirror slabs (below): null = none.  slab_prior_q16: mirrored entries per row entry measured by an earlier call on this matrix (0: take the sample's);
    // slab_pct: margin in percent
    uint4 *slab; unsigned long long slab_cap; uint32_t slab_prior_q16, slab_pct;
    unsigned long long *slab_pos;      // [M] per row: slab end << 32 | next free slab entry (k_classify_direct sets it to end << 32 | start)
};

// several small buffers zeroed by one launch (the counters of a call)
struct ZeroList { uint32_t *p[6]; size_t words[6]; int n; };
__global__ __launch_bounds__(256) void k_zero_regions(ZeroList z)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (int q = 0; q < z.n; ++q)
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < z.words[q]; i += stride) z.p[q][i] = 0u;
}

// ---- mirror slabs ---------------------------------------------------------------------------------------------------------------------
// The transposed image of a staged entry (i, j) belongs to row j.  Rounds 1-3 left it in the staging area with a ticket (its place among row
// j's mirrored entries, drawn from low_cnt[j]) and k_mirror placed it once the row pointers of B were known: 49 M random 16-byte stores behind
// a dependent random read on BASELINE config 3, 1.9 ms = a fifth of the call, at the memory system's rate for such stores.  The numeric kernel
// itself is bound by latency, not by memory: it now writes the image STRAIGHT into row j's slab — room for the row's mirrored entries reserved
// before they are counted, sized from what the call already measures on its sample of rows: survivors per row entry of A (x slab_pct / 100, +
// SLAB_PAD).  The slab of row j starts at floor((a_rowptr[j] - a_rowptr[row_lo]) x ratio) + SLAB_PAD x (j - row_lo), and ONE returning 64-bit atomic
// on slab_pos[j] = (slab end << 32 | next free entry) — written by the classification pass, which walks every row anyway — hands an image its
// place and tells it whether the place is still inside the slab.  An image that finds its row's slab full (or is staged while no ratio is known:
// the sample's own rows, small matrices) draws a ticket from low_cnt[j] and takes the old way (k_mirror, the mirror area); the finalize reads a
// row's mirrored entries from the slab first (slab_n[j] of them: k_slab_fold), then from the mirror area.  Capacity is a performance matter only,
// never a correctness limit.
constexpr uint32_t SLAB_PAD = 16;
struct __attribute__((packed, aligned(4))) RowPair { uint32_t a, b; };      // a_rowptr[j], a_rowptr[j + 1]
__device__ __forceinline__ uint32_t slab_base(uint32_t rp, uint32_t rp0, uint32_t row_rel, uint32_t q16)
{
    return (uint32_t)(((unsigned long long)(rp - rp0) * q16) >> 16) + SLAB_PAD * row_rel;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ uint32_t guaranteed_tbits(uint32_t ub, uint32_t mcols)
{
    const uint32_t need = ub < mcols ? ub : mcols;        // distinct partners <= min(products, reads)
    return need <= 1 ? 1u : (uint32_t)(32 - __clz(2 * need - 1));   // ceil(log2(2*need)): load factor <= 1/2
}

// ---- numeric -----------------------------------------------------------------------------------------------------
#include "spgemm_table.hpp"
#include "spgemm_direct.hpp"

// ---- finalize: row pointers, mirror, per-row column sort + move to final CSR --------------------------------------------
// A row's entries come from two places: mirrored ones scattered by k_mirror (mir, laid out like B: the first low_cnt
// positions of the row's extent) and the ones it staged itself (tmp at row_off, row_cnt of them).
struct FinParams {
    const uint32_t *row_cnt; uint32_t *low_cnt; const unsigned long long *row_off; int64_t *b_rowptr;
    const StageRec *tmp; StageRec *mir;
    uint32_t rec16; const uint4 *rec; const uint32_t *tick;      // (OvParams::rec16)
    uint32_t *b_col; elba_seed_t *b_val;
    uint32_t M, row_lo, row_hi, half;
    uint32_t *fin_lists; OvCounters *ctr;
    uint64_t *sortkeys; unsigned long long sort_stride;
    uint32_t *sum_tmp;
    long long b_cap;         // capacity of b_col / b_val in entries: rows that would not fit are left out (the host regrows and reruns)
    uint32_t mir16;          // positions fit 16 bits: a mirrored entry is ONE 16-byte word (i, q0 | t0 << 16, q1 | t1 << 16, numshared) — one store
                             // request per scattered entry instead of two, half the bytes read back
    const uint32_t *tick_rows;      // OvParams::tick_rows
    const uint32_t *a_rowptr; uint4 *slab;      // mirror slabs (above; mir16 records): null = none
    const unsigned long long *slab_pos; uint32_t *slab_n;      // [M] the slabs' fill words as the numeric kernels left them / entries in every row's slab (k_slab_fold; null with slab)
};

// this call's slabs: ratio (0 = none) and the first row entry of the window
struct SlabCall { uint32_t q16, rp0; };
__device__ __forceinline__ SlabCall slab_call(const FinParams &p)
{
    SlabCall s{0u, 0u};
    if (p.slab) { s.q16 = p.ctr->slab_q16; if (s.q16) s.rp0 = p.a_rowptr[p.row_lo]; }
    return s;
}
// row i's slab: first entry and entries in it
__device__ __forceinline__ uint2 slab_row(const FinParams &p, const SlabCall &s, uint32_t i)
{
    if (!s.q16) return make_uint2(0u, 0u);
    return make_uint2(slab_base(p.a_rowptr[i], s.rp0, i - p.row_lo, s.q16), p.slab_n[i]);
}
// entries in every row's slab, from the fill words (a full slab's word has run past its end: one add per image that found no room)
__global__ void k_slab_fold(FinParams p)
{
    const uint32_t i = p.row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.row_hi) return;
    const SlabCall s = slab_call(p);
    uint32_t n = 0;
    if (s.q16) {
        const unsigned long long w = p.slab_pos[i];
        const uint32_t base = slab_base(p.a_rowptr[i], s.rp0, i - p.row_lo, s.q16), cur = (uint32_t)w, lim = (uint32_t)(w >> 32);
        n = (cur < lim ? cur : lim) - base;
    }
    p.slab_n[i] = n;
}

// entry t of a row's extent: mirrored (the first `low`) or staged by the row itself; as the two halves of a staged record
__device__ __forceinline__ void fin_load(const FinParams &p, uint32_t low, unsigned long long off, int64_t dst, uint32_t t, uint4 &a, uint4 &b, uint2 sl = make_uint2(0u, 0u))
{
    if (t >= low) {
        if (p.rec16) {
            const uint4 m = p.rec[off + (t - low)];
            a = make_uint4(m.x, 0xFFFFFFFFu, m.y & 0xFFFFu, m.y >> 16); b = make_uint4(m.z & 0xFFFFu, m.z >> 16, m.w, 0u);
        } else { const StageRec *r = &p.tmp[off + (t - low)]; a = r->a; b = r->b; }
    }
    else if (p.mir16) {
        const uint4 m = t < sl.y ? p.slab[sl.x + t] : reinterpret_cast<const uint4 *>(p.mir)[dst + t - sl.y];
        a = make_uint4(m.x, 0xFFFFFFFFu, m.y & 0xFFFFu, m.y >> 16); b = make_uint4(m.z & 0xFFFFu, m.z >> 16, m.w, 0u);
    } else { const StageRec *r = &p.mir[dst + t]; a = r->a; b = r->b; }
}
__device__ __forceinline__ uint32_t fin_col(const FinParams &p, uint32_t low, unsigned long long off, int64_t dst, uint32_t t, uint2 sl = make_uint2(0u, 0u))
{
    if (t >= low) return p.rec16 ? p.rec[off + (t - low)].x : p.tmp[off + (t - low)].a.x;
    return p.mir16 ? (t < sl.y ? p.slab[sl.x + t].x : reinterpret_cast<const uint4 *>(p.mir)[dst + t - sl.y].x) : p.mir[dst + t].a.x;
}
__device__ __forceinline__ elba_seed_t rec_seed(const uint4 a, const uint4 b)
{
    elba_seed_t v;
    v.q0 = a.z; v.t0 = a.w; v.q1 = b.x; v.t1 = b.y; v.numshared = (int32_t)b.z;
    return v;
}

// Row pointers of B = exclusive scan of (staged + mirrored) counts, M+1 outputs, in ONE launch for up to 2^17 rows: every
// workgroup sums the counts before its tile itself (L2-resident, a few hundred loads per lane at most) instead of waiting for
// a second and third launch.
constexpr int RP_TILE = 1024;
__global__ __launch_bounds__(256) void k_row_pointers(FinParams p)
{
    __shared__ unsigned long long wsum[4], bsum;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = p.M + 1, t0 = blockIdx.x * RP_TILE;
    unsigned long long pre = 0;
    {   // t0 is a multiple of 1024: 16-byte loads, 8 in flight per lane
        const uint4 *ca = reinterpret_cast<const uint4 *>(p.row_cnt), *cb = reinterpret_cast<const uint4 *>(p.low_cnt), *cc = reinterpret_cast<const uint4 *>(p.slab_n);
#pragma unroll 4
        for (uint32_t q = tid; q < t0 / 4; q += 256) {
            const uint4 x = ca[q], z = cb[q];
            pre += (unsigned long long)x.x + x.y + x.z + x.w + z.x + z.y + z.z + z.w;
            if (cc) { const uint4 y2 = cc[q]; pre += (unsigned long long)y2.x + y2.y + y2.z + y2.w; }
        }
    }
    uint32_t v[4];
    unsigned long long mine = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t e = t0 + tid * 4 + u;
        v[u] = e < p.M ? p.row_cnt[e] + p.low_cnt[e] + (p.slab_n ? p.slab_n[e] : 0u) : 0u;
        mine += v[u];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d, 64);
    unsigned long long inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
    if (lane == 63) wsum[w] = inc;
    if (tid == 0) bsum = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&bsum, pre);
    __syncthreads();
    unsigned long long run = bsum + inc - mine;
    for (uint32_t k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t e = t0 + tid * 4 + u;
        if (e < n) p.b_rowptr[e] = (int64_t)run;
        run += v[u];
    }
}

__global__ void k_sum_counts(FinParams p)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= p.M) p.sum_tmp[e] = e < p.M ? p.row_cnt[e] + p.low_cnt[e] + (p.slab_n ? p.slab_n[e] : 0u) : 0u;
}

// Mirror pass: every staged entry (i,j) whose partner row j is computed here as well is the transpose's entry (j,i) with the two
// positions of each seed exchanged (see matrix.hip, above k_products).  Its slot inside row j's extent is the ticket the numeric
// kernel drew from low_cnt[j]; the extent is known now (b_rowptr).  One wavefront per row; rows too wide for the one-wave sort
// are queued for the block sorts here (their final length is only known after the scan).
__global__ __launch_bounds__(256) void k_mirror(FinParams p)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t placed = 0;
    for (uint32_t i = p.row_lo + wave; i < p.row_hi; i += nwaves) {
        const uint32_t own = p.row_cnt[i];
        const unsigned long long off = p.row_off[i];
        const uint32_t y = own + p.low_cnt[i] + (p.slab_n ? p.slab_n[i] : 0u);
        if (lane == 0 && y > FIN_WAVE2_MAX) {
            const int which = y > FIN_LDS_MAX ? 1 : 0;
            const uint32_t at = atomicAdd(&p.ctr->fin_count[which], 1u);
            p.fin_lists[(size_t)which * p.M + at] = i;
        }
        if (!p.half || p.rec16) continue;      // (16-byte records: the ticketed entries are a list, below)
        if (p.tick_rows && !((p.tick_rows[i >> 5] >> (i & 31u)) & 1u)) continue;      // (none of the row's images took a ticket: nothing to place)
        for (uint32_t t = lane; t < own; t += 64) {
            const uint4 a = p.tmp[off + t].a;
            if (a.y == 0xFFFFFFFFu) continue;
            const uint4 b = p.tmp[off + t].b;
            const int64_t at = p.b_rowptr[a.x] + (int64_t)a.y;
            if (at >= p.b_cap) continue;
            if (p.mir16) reinterpret_cast<uint4 *>(p.mir)[at] = make_uint4(i, a.w | a.z << 16, b.y | b.x << 16, b.z);
            else {
                p.mir[at].a = make_uint4(i, 0xFFFFFFFFu, a.w, a.z);
                p.mir[at].b = make_uint4(b.y, b.x, b.z, 0u);
            }
        }
    }
    if (p.half && p.rec16) {
        // 16-byte records: one lane per listed entry (row, place among the row's staged entries, ticket) — one word in, one out, the two positions of each seed change places
        const uint32_t n = p.ctr->ntick, stride = gridDim.x * blockDim.x;
        for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
            const uint32_t i = p.tick[3ull * e], t = p.tick[3ull * e + 1u], tk = p.tick[3ull * e + 2u];
            const uint4 r = p.rec[p.row_off[i] + t];
            const uint4 img = make_uint4(i, (r.y >> 16) | (r.y << 16), (r.z >> 16) | (r.z << 16), r.w);
            ++placed;
            const int64_t at = p.b_rowptr[r.x] + (int64_t)tk;
            if (at >= p.b_cap) continue;
            if (p.mir16) reinterpret_cast<uint4 *>(p.mir)[at] = img;
            else { p.mir[at].a = make_uint4(i, 0xFFFFFFFFu, img.y & 0xFFFFu, img.y >> 16); p.mir[at].b = make_uint4(img.z & 0xFFFFu, img.z >> 16, img.w, 0u); }
        }
    }
    placed = wave_sum_u32(placed);
    if (lane == 0 && placed) atomicAdd(&p.ctr->mir_placed, (unsigned long long)placed);
}

// Rows of up to FIN_WAVE_MAX entries: one wavefront per row; the row's columns are staged in LDS and every lane ranks its (up to 4)
// elements against all of them — columns are distinct, so ranks are a permutation.  (Every row passes here once: the ticket counters are
// handed back clean.)
__global__ __launch_bounds__(256) void k_finalize_wave(FinParams p)
{
    __shared__ uint32_t colsm[4][FIN_WAVE_MAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    uint32_t *cols = colsm[w];
    const SlabCall sc = slab_call(p);
    for (uint32_t i = p.row_lo + wave; i < p.row_hi; i += nwaves) {
        const int64_t dst = p.b_rowptr[i];
        const uint32_t y = (uint32_t)(p.b_rowptr[i + 1] - dst);
        const uint32_t lowt = p.low_cnt[i];
        if (lane == 0 && lowt) p.low_cnt[i] = 0;          // the ticket counters are handed back clean (k_mirror, the only other reader, has finished)
        if (y == 0 || y > FIN_WAVE_MAX) continue;
        if (dst + (int64_t)y > p.b_cap) continue;
        const unsigned long long off = p.row_off[i];
        const uint2 sl = slab_row(p, sc, i);
        const uint32_t low = lowt + sl.y;                  // mirrored entries: the slab's, then the ticketed ones
        uint32_t mine[4];
        uint4 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = (uint32_t)lane + 64u * u;
            mine[u] = 0xFFFFFFFFu;
            if (t < y) {
                fin_load(p, low, off, dst, t, ra[u], rb[u], sl);
                mine[u] = ra[u].x;
                cols[t] = mine[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t rank[4] = {0, 0, 0, 0};
        for (uint32_t l = 0; l < y; ++l) {
            const uint32_t c = cols[l];
#pragma unroll
            for (int u = 0; u < 4; ++u) rank[u] += c < mine[u] ? 1u : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = (uint32_t)lane + 64u * u;
            if (t < y) { p.b_col[dst + rank[u]] = mine[u]; p.b_val[dst + rank[u]] = rec_seed(ra[u], rb[u]); }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Rows of FIN_WAVE_MAX + 1 .. FIN_WAVE2_MAX entries: still ONE WAVEFRONT per row — 16 rows in flight per CU, no workgroup
// barriers — with a bucket + rank sort as in k_finalize_bucket below: ~y/8 equal column ranges, count, scan, scatter the keys into their
// buckets, rank inside the bucket (~8 compares).  The 200 k-read set's rows average 490 entries.
__global__ __launch_bounds__(256) void k_finalize_mid(FinParams p)
{
    __shared__ uint64_t lkeys[4][FIN_WAVE2_MAX];
    __shared__ uint32_t bst[4][128], bfl[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const SlabCall sc = slab_call(p);
    for (uint32_t i = p.row_lo + wave; i < p.row_hi; i += nwaves) {      // (no queue: 200 k rows drawing slots from one counter cost more than looking at every row's length)
        const int64_t dst = p.b_rowptr[i];
        const uint32_t y = (uint32_t)(p.b_rowptr[i + 1] - dst);
        if (y <= FIN_WAVE_MAX || y > FIN_WAVE2_MAX) continue;
        if (dst + (int64_t)y > p.b_cap) continue;
        const uint32_t low = y - p.row_cnt[i];          // (low_cnt itself has been handed back by k_finalize_wave)
        const unsigned long long off = p.row_off[i];
        const uint2 sl = slab_row(p, sc, i);
        uint32_t nb = y / 8;
        nb = nb > 128u ? 128u : nb;
        const unsigned long long scale = ((unsigned long long)nb << 32) / (p.M > 0 ? p.M : 1u);       // bucket(col) = col * nb / M, monotone in col, < nb
        bst[w][lane] = 0; bst[w][lane + 64] = 0; bfl[w][lane] = 0; bfl[w][lane + 64] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t col[FIN_WAVE2_MAX / 64];
#pragma unroll
        for (int u = 0; u < (int)(FIN_WAVE2_MAX / 64); ++u) {
            const uint32_t t = (uint32_t)lane + 64u * u;
            col[u] = t < y ? fin_col(p, low, off, dst, t, sl) : 0u;
        }
#pragma unroll
        for (int u = 0; u < (int)(FIN_WAVE2_MAX / 64); ++u)
            if ((uint32_t)lane + 64u * u < y) atomicAdd(&bst[w][(uint32_t)(((unsigned long long)col[u] * scale) >> 32)], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // exclusive scan of the <= 128 bucket counts: 2 per lane
            const uint32_t c0 = bst[w][2 * lane], c1 = bst[w][2 * lane + 1];
            const uint32_t inc = wave_add_scan(c0 + c1);
            const uint32_t ex = inc - c0 - c1;
            __builtin_amdgcn_wave_barrier();
            bst[w][2 * lane] = ex; bst[w][2 * lane + 1] = ex + c0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < (int)(FIN_WAVE2_MAX / 64); ++u) {
            const uint32_t t = (uint32_t)lane + 64u * u;
            if (t < y) {
                const uint32_t b = (uint32_t)(((unsigned long long)col[u] * scale) >> 32);
                lkeys[w][bst[w][b] + atomicAdd(&bfl[w][b], 1u)] = ((uint64_t)col[u] << 32) | t;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e = (uint32_t)lane; e < y; e += 64u) {
            const uint64_t k = lkeys[w][e];
            const uint32_t c = (uint32_t)(k >> 32);
            const uint32_t b = (uint32_t)(((unsigned long long)c * scale) >> 32);
            const uint32_t lo = bst[w][b], hi = lo + bfl[w][b];
            uint4 ra, rb;
            fin_load(p, low, off, dst, (uint32_t)k, ra, rb, sl);      // (requested before the ranking: an L2 round trip that overlaps with it)
            // rank inside the bucket (~8 keys): the first eight requested at once — a loop with a per-lane trip count is a round trip per key
            uint32_t rank = 0;
            uint64_t kk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) kk[q] = lkeys[w][lo + (uint32_t)q < hi ? lo + (uint32_t)q : lo];
#pragma unroll
            for (int q = 0; q < 8; ++q) rank += (lo + (uint32_t)q < hi && kk[q] < k) ? 1u : 0u;
            for (uint32_t x = lo + 8u; x < hi; ++x) rank += lkeys[w][x] < k ? 1u : 0u;
            p.b_col[dst + lo + rank] = c;
            p.b_val[dst + lo + rank] = rec_seed(ra, rb);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The same rows when a record is ONE 16-byte word (FinParams::rec16 && mir16: positions below 2^16): the lane that fetched a record KEEPS it
// in registers (up to 16 of them), only the 4-byte columns go through LDS, and the lane itself ranks and writes its records — k_finalize_mid
// above fetches every record a second time (by the lane that finds it in its bucket: a dependent HBM / L2 round trip per element of the last loop,
// eight in a row on the 200 k-read set).
// (162 VGPRs, three workgroups per CU; a budget of 128 or fewer makes it spill: finalize 3.46 -> 3.61 / 4.7 / 5.9 ms at 4 / 5 / 6 wavefronts per SIMD)
__global__ __launch_bounds__(256, 3) void k_finalize_mid16(FinParams p)
{
    constexpr int NU = (int)(FIN_WAVE2_MAX / 64);
    __shared__ uint32_t lkeys[4][FIN_WAVE2_MAX];
    __shared__ uint32_t bst[4][128], bfl[4][128];
    __shared__ uint4 stage[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint4 *mir = reinterpret_cast<const uint4 *>(p.mir);
    const SlabCall sc = slab_call(p);
    for (uint32_t i = p.row_lo + wave; i < p.row_hi; i += nwaves) {
        const int64_t dst = p.b_rowptr[i];
        const uint32_t y = (uint32_t)(p.b_rowptr[i + 1] - dst);
        if (y <= FIN_WAVE_MAX || y > FIN_WAVE2_MAX) continue;
        if (dst + (int64_t)y > p.b_cap) continue;
        const uint32_t low = y - p.row_cnt[i];          // (low_cnt itself has been handed back by k_finalize_wave)
        const unsigned long long off = p.row_off[i];
        uint2 sl = slab_row(p, sc, i);
        sl = make_uint2(sfirst(sl.x), sfirst(sl.y));      // (the row is the wavefront's: scalar registers — the kernel sits two VGPRs below its budget)
        uint32_t nb = y / 8;
        nb = nb > 128u ? 128u : nb;
        const unsigned long long scale = ((unsigned long long)nb << 32) / (p.M > 0 ? p.M : 1u);       // bucket(col) = col * nb / M, monotone in col, < nb
        bst[w][lane] = 0; bst[w][lane + 64] = 0; bfl[w][lane] = 0; bfl[w][lane + 64] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint4 m[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const uint32_t t = (uint32_t)lane + 64u * u;
            m[u] = t < y ? (t >= low ? p.rec[off + (t - low)] : (t < sl.y ? p.slab[sl.x + t] : mir[dst + t - sl.y])) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if ((uint32_t)lane + 64u * u < y) atomicAdd(&bst[w][(uint32_t)(((unsigned long long)m[u].x * scale) >> 32)], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // exclusive scan of the <= 128 bucket counts: 2 per lane
            const uint32_t c0 = bst[w][2 * lane], c1 = bst[w][2 * lane + 1];
            const uint32_t inc = wave_add_scan(c0 + c1);
            const uint32_t ex = inc - c0 - c1;
            __builtin_amdgcn_wave_barrier();
            bst[w][2 * lane] = ex; bst[w][2 * lane + 1] = ex + c0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if ((uint32_t)lane + 64u * u < y) {
                const uint32_t b = (uint32_t)(((unsigned long long)m[u].x * scale) >> 32);
                lkeys[w][bst[w][b] + atomicAdd(&bfl[w][b], 1u)] = m[u].x;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if ((uint32_t)lane + 64u * u < y) {
                const uint32_t c = m[u].x;
                const uint32_t b = (uint32_t)(((unsigned long long)c * scale) >> 32);
                const uint32_t lo = bst[w][b], hi = lo + bfl[w][b];
                // rank inside the bucket (~8 columns, all distinct): the first eight requested at once
                uint32_t rank = 0, kk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) kk[q] = lkeys[w][lo + (uint32_t)q < hi ? lo + (uint32_t)q : lo];
#pragma unroll
                for (int q = 0; q < 8; ++q) rank += (lo + (uint32_t)q < hi && kk[q] < c) ? 1u : 0u;
                for (uint32_t x = lo + 8u; x < hi; ++x) rank += lkeys[w][x] < c ? 1u : 0u;
                m[u].w |= (lo + rank) << 22;      // the record's place in the row, beside its count (positions below 2^16: a pair shares fewer than 2^16 k-mers)
            }
        }
        // The records leave through LDS, 256 places of the row at a time, so that consecutive lanes write consecutive entries: a lane that wrote
        // its own record to its place issued three partial-sector stores per entry (PMC: 219 M write requests and 3.7 GB written for 2.4 GB of B).
        for (uint32_t w0 = 0; w0 < y; w0 += 256u) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const uint32_t pos = (m[u].w >> 22) - w0;
                if ((uint32_t)lane + 64u * u < y && pos < 256u) stage[w][pos] = make_uint4(m[u].x, m[u].y, m[u].z, m[u].w & 0x3FFFFFu);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t e = w0 + 64u * q + (uint32_t)lane;
                if (e < y) {
                    const uint4 r = stage[w][64u * q + (uint32_t)lane];
                    p.b_col[dst + e] = r.x;
                    p.b_val[dst + e] = rec_seed(make_uint4(r.x, 0xFFFFFFFFu, r.y & 0xFFFFu, r.y >> 16), make_uint4(r.z & 0xFFFFu, r.z >> 16, r.w, 0u));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
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

// Rows of 1025..4096 entries: one workgroup per row, bucket + rank sort in LDS.  Partner ids are spread evenly over [0, M), so the row
// is cut into ~y/8 equal column ranges: count per bucket, scan, scatter the keys into their buckets, then every element ranks itself
// inside its bucket (~8 compares).  ~100 instructions per element and 4 barriers per row, against ~45 barrier-separated compare-exchange
// stages of a bitonic network (this pass was 44 % of the step on the 200 k-read workload, whose rows average 490 entries).  A skewed row
// (all partners in one range) degrades to a rank sort, never to a wrong result.
constexpr uint32_t FIN_BUCKETS = 512;
__global__ __launch_bounds__(256) void k_finalize_bucket(FinParams p)
{
    __shared__ uint64_t lkeys[FIN_LDS_MAX];
    __shared__ uint32_t bstart[FIN_BUCKETS], bfill[FIN_BUCKETS];
    __shared__ unsigned long long scale_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t n = p.ctr->fin_count[0];
    for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
        const uint32_t i = p.fin_lists[it];
        const int64_t dst = p.b_rowptr[i];
        const uint32_t y = (uint32_t)(p.b_rowptr[i + 1] - dst);
        if (dst + (int64_t)y > p.b_cap) continue;
        const uint32_t low = y - p.row_cnt[i];          // (low_cnt itself may already have been handed back by k_finalize_wave)
        const unsigned long long off = p.row_off[i];
        const uint2 sl = slab_row(p, slab_call(p), i);
        uint32_t nb = y / 8;
        nb = nb < 1 ? 1u : (nb > FIN_BUCKETS ? FIN_BUCKETS : nb);
        // bucket(col) = col * nb / M without a division per element: col * floor(nb * 2^32 / M) >> 32  (monotone in col, < nb)
        if (tid == 0) scale_s = ((unsigned long long)nb << 32) / (p.M > 0 ? p.M : 1u);
        for (uint32_t b = tid; b < FIN_BUCKETS; b += 256) { bstart[b] = 0; bfill[b] = 0; }
        __syncthreads();
        const unsigned long long scale = scale_s;
        for (uint32_t t = tid; t < y; t += 256) {
            const uint32_t col = fin_col(p, low, off, dst, t, sl);
            atomicAdd(&bstart[(uint32_t)(((unsigned long long)col * scale) >> 32)], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                    // exclusive scan of <= 512 bucket counts by one wavefront: 8 buckets per lane
            uint32_t c[8], sum = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) { c[u] = bstart[lane * 8 + u]; sum += c[u]; }
            uint32_t inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
            uint32_t run = inc - sum;
#pragma unroll
            for (int u = 0; u < 8; ++u) { bstart[lane * 8 + u] = run; run += c[u]; }
        }
        __syncthreads();
        for (uint32_t t = tid; t < y; t += 256) {
            const uint32_t col = fin_col(p, low, off, dst, t, sl);
            const uint32_t b = (uint32_t)(((unsigned long long)col * scale) >> 32);
            lkeys[bstart[b] + atomicAdd(&bfill[b], 1u)] = ((uint64_t)col << 32) | t;
        }
        __syncthreads();
        for (uint32_t e = tid; e < y; e += 256) {
            const uint64_t k = lkeys[e];
            const uint32_t col = (uint32_t)(k >> 32);
            const uint32_t b = (uint32_t)(((unsigned long long)col * scale) >> 32);
            const uint32_t lo = bstart[b], hi = lo + bfill[b];
            uint32_t rank = 0;
            for (uint32_t x = lo; x < hi; ++x) rank += lkeys[x] < k ? 1u : 0u;
            uint4 ra, rb;
            fin_load(p, low, off, dst, (uint32_t)k, ra, rb, sl);
            p.b_col[dst + lo + rank] = col;
            p.b_val[dst + lo + rank] = rec_seed(ra, rb);
        }
        __syncthreads();
    }
}

// Rows beyond 4096 entries: bitonic sort of (column << 32 | source index) keys in an HBM scratch slab, one workgroup per row.
__global__ __launch_bounds__(256) void k_finalize_huge(FinParams p)
{
    const uint32_t n = p.ctr->fin_count[1];
    for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
        const uint32_t i = p.fin_lists[(size_t)p.M + it];
        uint64_t *keys = p.sortkeys + (size_t)blockIdx.x * p.sort_stride;
        const int64_t dst = p.b_rowptr[i];
        const uint32_t y = (uint32_t)(p.b_rowptr[i + 1] - dst);
        if (dst + (int64_t)y > p.b_cap) continue;
        const uint32_t low = y - p.row_cnt[i];
        const unsigned long long off = p.row_off[i];
        const uint2 sl = slab_row(p, slab_call(p), i);
        uint32_t n2 = 1;
        while (n2 < y) n2 <<= 1;
        for (uint32_t t = threadIdx.x; t < n2; t += 256)
            keys[t] = t < y ? (((uint64_t)fin_col(p, low, off, dst, t, sl) << 32) | t) : ~0ull;
        __syncthreads();
        bitonic_sort<256>(keys, n2);
        for (uint32_t t = threadIdx.x; t < y; t += 256) {
            const uint64_t k = keys[t];
            p.b_col[dst + t] = (uint32_t)(k >> 32);
            uint4 ra, rb;
            fin_load(p, low, off, dst, (uint32_t)k, ra, rb, sl);
            p.b_val[dst + t] = rec_seed(ra, rb);
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

// ---- the plan-free call: CSR + columns of A -> pruned CSR B, everything in between computed here ---------------------------------
// Launch sequence (one stream, ONE host synchronisation at the end when nothing overflows):
//   counters = 0, row_cnt = 0 | k_classify_direct | k_spgemm_direct on every table tier, ascending (+ the HBM-table tier) |
//   k_row_pointers (scan of the rows' counts) | k_mirror (only queues the wide rows: nothing is mirrored, both triangles were computed) |
//   k_finalize_wave / _bucket / _huge (per-row column sort + move to b_col / b_val) | counter read-back.
// What a call may remember from earlier calls on the same matrix is a HINT only — the distinct-partner / row-entry ratio that picks the
// starting tiers, which tiers and sorts got rows — and `cold_calls` (elba_set_option) or a new matrix forgets it: then the ratio starts at
// 1/4 and the kernel corrects itself from the rows already done, every tier is launched.  Capacities (staging, output) start from nnz(A)
// and only ever grow; a call that overflows them is repeated with what it measured.
// ---- mirror exchange between ranks (sharded call with global pair ownership) ---------------------------------------------------------
// A pair {i, j} whose rows live on two ranks is accumulated by ONE of them (owns_pair's parity rule, whatever the window); the other rank
// receives the mirrored entry.  k_remote_mirror walks the staged records of this rank's rows, picks those whose partner row is another
// rank's and counts (FILL = false) or writes (FILL = true) their mirror images grouped by destination rank: a block takes 64 rows, counts
// per destination in LDS, reserves its share of every destination's segment with one atomic each, then writes.
// Record on the wire (32 bytes): a = (destination row j, partner i, position in j, position in i), b = (the same of seeds[1], numshared, 0).
constexpr int REMOTE_MAX_RANKS = 64, REMOTE_ROWS_PER_BLOCK = 64;
struct RemoteParams {
    const uint32_t *row_cnt; const unsigned long long *row_off; const StageRec *tmp;
    uint32_t rec16; const uint4 *rec;
    uint32_t row_lo, row_hi, nranks;
    uint64_t bounds[REMOTE_MAX_RANKS + 1];
};
__device__ __forceinline__ uint32_t remote_rank_of(const RemoteParams &p, uint64_t read)
{
    uint32_t lo = 0, hi = p.nranks;             // last r with bounds[r] <= read
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (p.bounds[mid] <= read) lo = mid; else hi = mid; }
    return lo;
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_remote_mirror(RemoteParams p, unsigned long long *cnt_or_cursor, StageRec *send)
{
    __shared__ uint32_t cnt[REMOTE_MAX_RANKS], fill[REMOTE_MAX_RANKS];
    __shared__ unsigned long long base[REMOTE_MAX_RANKS];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t r0 = p.row_lo + blockIdx.x * REMOTE_ROWS_PER_BLOCK, r1 = min(r0 + (uint32_t)REMOTE_ROWS_PER_BLOCK, p.row_hi);
    if (tid < REMOTE_MAX_RANKS) { cnt[tid] = 0; fill[tid] = 0; }
    __syncthreads();
    for (uint32_t i = r0 + w; i < r1; i += 4) {
        const uint32_t n = p.row_cnt[i];
        const unsigned long long off = p.row_off[i];
        for (uint32_t t = lane; t < n; t += 64) {
            const uint32_t j = p.rec16 ? p.rec[off + t].x : p.tmp[off + t].a.x;
            if (j != i && (j < p.row_lo || j >= p.row_hi)) atomicAdd(&cnt[remote_rank_of(p, j)], 1u);
        }
    }
    __syncthreads();
    if (tid < p.nranks && cnt[tid]) {
        const unsigned long long b = atomicAdd(&cnt_or_cursor[tid], (unsigned long long)cnt[tid]);
        if (FILL) base[tid] = b;
    }
    if (!FILL) return;
    __syncthreads();
    for (uint32_t i = r0 + w; i < r1; i += 4) {
        const uint32_t n = p.row_cnt[i];
        const unsigned long long off = p.row_off[i];
        for (uint32_t t = lane; t < n; t += 64) {
            uint4 a, b = make_uint4(0u, 0u, 0u, 0u);
            if (p.rec16) { const uint4 m = p.rec[off + t]; a = make_uint4(m.x, 0u, m.y & 0xFFFFu, m.y >> 16); b = make_uint4(m.z & 0xFFFFu, m.z >> 16, m.w, 0u); }
            else a = p.tmp[off + t].a;
            const uint32_t j = a.x;
            if (j != i && (j < p.row_lo || j >= p.row_hi)) {
                if (!p.rec16) b = p.tmp[off + t].b;
                const uint32_t d = remote_rank_of(p, j);
                const unsigned long long at = base[d] + atomicAdd(&fill[d], 1u);
                send[at].a = make_uint4(j, i, a.w, a.z);
                send[at].b = make_uint4(b.y, b.x, b.z, 0u);
            }
        }
    }
}

// received mirror images: every record draws its slot in its row's mirror extent (the ticket lands in b.w) before the row pointers are summed
__global__ void k_ingest_remote(StageRec *rem, unsigned long long n, uint32_t *low_cnt, uint32_t row_lo, uint32_t row_hi, unsigned long long *bad, unsigned long long *nupper)
{
    const unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long up = 0;
    if (r < n) {
        const uint32_t j = rem[r].a.x;
        if (j < row_lo || j >= row_hi) atomicAdd(bad, 1ull);      // not one of this rank's rows: the driver mixed up its buffers
        else { rem[r].b.w = atomicAdd(&low_cnt[j], 1u); up = rem[r].a.y > j ? 1u : 0u; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) up += __shfl_xor(up, d, 64);
    if ((threadIdx.x & 63) == 0 && up) atomicAdd(nupper, up);
}

// ... and are placed like the local mirror images (k_mirror), once the row pointers are known
// (slot != 0: rem is gridDim.y slots of `slot` records each, record 0 of a slot its header — the count of records behind it in a.x;
//  blockIdx.y = slot, and the blocks of a slot stride over its records only)
__global__ void k_place_remote(FinParams p, const StageRec *rem, unsigned long long n, unsigned long long slot)
{
    if (slot) { rem += (unsigned long long)blockIdx.y * slot; n = (unsigned long long)rem->a.x + 1; if (n > slot) n = slot; }      // (a header that claims more than the slot holds — a failed collective, a slot mismatch — is flagged by k_ingest_remote_slots; never read past the slot)
    for (unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + (slot ? 1 : 0); r < n; r += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 a = rem[r].a, b = rem[r].b;
        if (a.x < p.row_lo || a.x >= p.row_hi) continue;
        const int64_t at = p.b_rowptr[a.x] + (int64_t)b.w;
        if (at >= p.b_cap) continue;
        if (p.mir16) reinterpret_cast<uint4 *>(p.mir)[at] = make_uint4(a.y, a.z | a.w << 16, b.x | b.y << 16, b.z);
        else { p.mir[at].a = make_uint4(a.y, 0xFFFFFFFFu, a.z, a.w); p.mir[at].b = make_uint4(b.x, b.y, b.z, 0u); }
    }
}

// row pointers, mirror pass (local pairs, and the mirrored entries other ranks sent: `remote`), per-row column sort + move to b_col / b_val
static void ov_launch_finalize(Ctx &c, uint32_t half, bool all_sorts, uint32_t &skipped_sorts, const StageRec *remote, int64_t nremote, int64_t slot = 0)
{
    hipStream_t s = c.stream;
    const int64_t M = c.M;
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? M : c.row_hi, nrows = row_hi - row_lo;
    const int cus = c.num_cus;
    const bool mir16 = c.pos16 && !c.opt.mir32;
    FinParams f{};
    f.row_cnt = c.ov_rowcnt.as<uint32_t>(); f.low_cnt = c.ov_totcnt.as<uint32_t>(); f.tmp = c.ov_tmp.as<StageRec>(); f.mir = half ? c.ov_mir.as<StageRec>() : nullptr; f.half = half; f.row_off = c.ov_rowoff.as<unsigned long long>(); f.b_rowptr = c.b_rowptr.as<int64_t>();
    f.b_col = c.b_col.as<uint32_t>(); f.b_val = c.b_val.as<elba_seed_t>();
    f.M = (uint32_t)M; f.row_lo = (uint32_t)row_lo; f.row_hi = (uint32_t)row_hi; f.fin_lists = c.ov_lists.as<uint32_t>() + (size_t)NUM_TIERS * (size_t)(M + 1); f.ctr = c.ov_counters.as<OvCounters>();
    f.b_cap = c.b_cap_entries;
    f.mir16 = mir16 ? 1u : 0u;
    f.tick_rows = c.ov_tickrows.as<uint32_t>();
    f.rec16 = c.ov_rec16 ? 1u : 0u; f.rec = c.ov_tmp.as<uint4>(); f.tick = reinterpret_cast<const uint32_t *>(c.ov_tmp.as<char>() + (size_t)c.ov_tmp_cap * 16);
    f.a_rowptr = c.a_rowptr.as<uint32_t>(); f.slab = (half == 1u && mir16 && c.ov_slab_on) ? c.ov_slab.as<uint4>() : nullptr;
    f.slab_pos = f.slab ? c.ov_slabpos.as<unsigned long long>() : nullptr; f.slab_n = f.slab ? c.ov_slabn.as<uint32_t>() : nullptr;
    if (f.slab && nrows > 0) hipLaunchKernelGGL(k_slab_fold, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, f);
    const int gblocks = 32;
    uint64_t sstride = 2;
    while (sstride < (uint64_t)M) sstride <<= 1;
    c.ov_sortkeys.reserve((size_t)gblocks * sstride * 8);
    f.sortkeys = c.ov_sortkeys.as<uint64_t>(); f.sort_stride = sstride;
    if (M + 1 <= (1 << 17)) {
        hipLaunchKernelGGL(k_row_pointers, dim3((unsigned)((M + 1 + RP_TILE - 1) / RP_TILE)), dim3(256), 0, s, f);
    } else {
        c.ov_sum_tmp.reserve((size_t)(M + 2) * 4);
        f.sum_tmp = c.ov_sum_tmp.as<uint32_t>();
        hipLaunchKernelGGL(k_sum_counts, dim3((unsigned)((M + 1 + 255) / 256)), dim3(256), 0, s, f);
        exclusive_scan_u32_to_i64(s, f.sum_tmp, c.b_rowptr.as<int64_t>(), M + 1, c.ws_scan);
    }
    if (nrows > 0) {
        int nb = (int)((nrows + 3) / 4);
        if (nb > cus * 32) nb = cus * 32;
        hipLaunchKernelGGL(k_mirror, dim3(nb), dim3(256), 0, s, f);
        if (nremote > 0 && slot == 0) hipLaunchKernelGGL(k_place_remote, dim3((unsigned)std::min<int64_t>((nremote + 255) / 256, (int64_t)cus * 32)), dim3(256), 0, s, f, remote, (unsigned long long)nremote, 0ull);
        if (nremote > 0 && slot != 0) hipLaunchKernelGGL(k_place_remote, dim3((unsigned)std::min<int64_t>((slot + 255) / 256, (int64_t)cus * 4), (unsigned)(nremote / slot)), dim3(256), 0, s, f, remote, (unsigned long long)nremote, (unsigned long long)slot);
        hipLaunchKernelGGL(k_finalize_wave, dim3(nb), dim3(256), 0, s, f);
        if (f.rec16 && f.mir16) hipLaunchKernelGGL(k_finalize_mid16, dim3(nb), dim3(256), 0, s, f);
        else hipLaunchKernelGGL(k_finalize_mid, dim3(nb), dim3(256), 0, s, f);
        skipped_sorts = 0;
        bool narrow = false;
        // (a row of B holds at most min(longest row of A x longest column, reads) entries: the sorts for wider rows are not launched for a matrix that cannot have them)
        if (c.opt.tune[4] != 2 && std::min<uint64_t>((uint64_t)std::max<int64_t>(c.max_row_nnz, 1) * (uint64_t)std::max<int64_t>(c.max_col_nnz, 1), (uint64_t)M) <= (uint64_t)FIN_WAVE2_MAX && nremote == 0) narrow = true;
        if (!narrow && (all_sorts || c.ov_sort_used[0])) hipLaunchKernelGGL(k_finalize_bucket, dim3((unsigned)(nrows < (int64_t)cus * 4 ? nrows : (int64_t)cus * 4)), dim3(256), 0, s, f);
        else skipped_sorts |= 1u;
        if (!narrow && (all_sorts || c.ov_sort_used[1])) hipLaunchKernelGGL(k_finalize_huge, dim3(gblocks), dim3(256), 0, s, f);
        else skipped_sorts |= 2u;
    }
}

// fold the counter read-back into elba_overlap_stats and remember the hints for the next call; `extra_*`: entries / strict-upper entries that
// arrived from other ranks (mirror exchange)
static void ov_finish_stats(Ctx &c, OvCounters &hc, elba_overlap_stats &st, int passes, bool was_timed, float ms_tot, float ms_sym, float ms_num, float ms_fin, int64_t extra_nnz, int64_t extra_upper)
{
    const int64_t M = c.M, N = c.N, Z = c.Z;
#ifdef ELBA_PHASE_CLOCK
    fprintf(stderr, "[elba phase] wave-0 cycles summed over %llu workgroups: header=%llu init=%llu accumulate=%llu handoff=%llu sweep=%llu reserve=%llu store=%llu | rows %u,%u,%u,%u,%u,%u\n",
            hc.phase[10], hc.phase[0], hc.phase[1], hc.phase[2], hc.phase[3], hc.phase[4], hc.phase[5], hc.phase[6],
            hc.tier_count[0], hc.tier_count[1], hc.tier_count[2], hc.tier_count[3], hc.tier_count[4], hc.tier_count[5]);
#endif
    for (int sh = 0; sh < NUM_SHARDS; ++sh) {       // fold the statistics shards
        const OvShard &x = hc.shard[sh];
        hc.yraw += x.yraw; hc.nnz += x.nnz; hc.ndiag += x.ndiag; hc.nupper += x.nupper; hc.products += x.products;
        if (x.maxshared > hc.maxshared) hc.maxshared = x.maxshared;
        for (int t = 0; t < NUM_TIERS; ++t) hc.tier_done[t] += x.tier_done[t];
    }
    unsigned long long fbc = 0, fbu = 0;
    for (int sh = 0; sh < NUM_SHARDS; ++sh) { fbc += hc.shard[sh].fb_claims; fbu += hc.shard[sh].fb_ub; }
    const int64_t Y = (int64_t)hc.nnz + extra_nnz;
    for (int t = 0; t < NUM_TIERS; ++t) c.ov_tier_used[t] = hc.tier_count[t] > 0;
    c.ov_tiers_known = true;
    c.ov_sort_used[0] = hc.fin_count[0] > 0; c.ov_sort_used[1] = hc.fin_count[1] > 0;
    c.ov_mir_placed = (int64_t)hc.mir_placed;
    c.ov_slab_q16_used = hc.slab_q16;
    if (c.row_lo == 0 && (c.row_hi < 0 || c.row_hi == M) && Z > 0 && extra_nnz == 0) {      // mirrored entries per row entry of A, for the next call's slabs (a hint, like the ratio below)
        const double r = 0.5 * (double)((int64_t)hc.nnz - (int64_t)hc.ndiag) / (double)Z * 65536.0;
        c.ov_slab_q16 = r < 1.0 ? 1u : (r > 4.0e9 ? 4000000000u : (uint32_t)r);
    }
    if (fbu > 0) {   // the measured distinct-partner / row-entry ratio (+25 %) picks the next call's starting tiers
        double r = 1.25 * (double)fbc / (double)fbu * 65536.0;
        const uint32_t q = r < 64.0 ? 64u : (r > 4.0e9 ? 4000000000u : (uint32_t)r);
        const uint32_t old = c.ov_prior_q16;
        if (old == 0 || q > old + old / 10 || q + old / 10 < old) c.ov_prior_q16 = q;
    }
    st.products = c.ov_hints_used ? c.A_products : (int64_t)hc.products;      // (entries that skip their column do not see its length: counted when A was built)
    st.nnz_before_prune = (int64_t)hc.yraw;
    st.nnz = Y;
    st.nnz_diag = (int64_t)hc.ndiag;
    st.nnz_upper = (int64_t)hc.nupper + extra_upper;
    st.max_numshared = (int64_t)hc.maxshared;
    st.rows_lds = 0;
    for (int t = 0; t < NUM_LDS_TIERS; ++t) st.rows_lds += hc.tier_done[t];
    st.rows_global = (int64_t)hc.tier_done[NUM_LDS_TIERS];
    int64_t queued = 0;
    for (int t = 0; t < NUM_TIERS; ++t) queued += hc.tier_count[t];
    queued += hc.sample_count;
    st.rows_escalated = queued - st.rows_lds - st.rows_global;
    st.algorithmic_bytes = 16 * Z + 8 * (2 * M + N + 3) + 24 * Y;
    st.passes = passes;
    st.timed = was_timed ? 1 : 0;
    st.ms_total = ms_tot; st.ms_symbolic = ms_sym; st.ms_numeric = ms_num; st.ms_finalize = ms_fin;
    c.Y = Y;
    c.ostats = st;
    c.have_B = true;
}

// phase 0: the whole call.  phase 1: the first half of a sharded call with mirror exchange (stage_seed_matrix_begin): classify + numeric with
// GLOBAL pair ownership, stops before the finalize pass.  phase 2: the same, only QUEUED — every tier is launched, nothing is read back and the
// host does not wait (stage_seed_matrix_send; what phase 1 checks after its synchronisation, stage_seed_matrix_recv checks at the end of the step).
static void create_seed_matrix_direct(Ctx &c, int phase)
{
    hipStream_t s = c.stream;
    const int64_t M = c.M, Z = c.Z;
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? M : c.row_hi;
    elba_overlap_stats st{};
    st.nrows = row_hi - row_lo;
    c.have_B = false;
    c.ov_phase = 0;      // (a begun sharded call that was never ended is abandoned here: its staged records are about to be overwritten)
    ELBA_REQUIRE(M < 0xFFFFFF00ll, ELBA_ERR_UNSUPPORTED, "read ids beyond 2^32 - 256 (the top of the id range marks empty slots and idle lanes)");
    if (c.cold_calls) { c.ov_prior_q16 = 0; c.ov_slab_q16 = 0; c.ov_tiers_known = false; c.ov_sort_used[0] = c.ov_sort_used[1] = true; }

    c.ov_rowcnt.reserve((size_t)(M + 2) * 4);
    c.ov_rowoff.reserve((size_t)(M + 1) * 8);
    c.ov_lists.reserve((size_t)(NUM_TIERS + 2) * (size_t)(M + 1) * 4);
    c.ov_counters.reserve(sizeof(OvCounters));
    c.b_rowptr.reserve((size_t)(M + 2) * 8);
    if (c.ov_totcnt.cap < (size_t)(M + 2) * 4) c.ov_low_clean = false;
    c.ov_totcnt.reserve((size_t)(M + 2) * 4);

    uint64_t gstride = 2;
    while (gstride < 2ull * (uint64_t)(M > 1 ? M : 1)) gstride <<= 1;
    int spill_blocks = (int)((4ull << 30) / (20ull * gstride));
    spill_blocks = spill_blocks < 64 ? 64 : (spill_blocks > c.num_cus * 2 ? c.num_cus * 2 : spill_blocks);
    c.ov_gtable.reserve((size_t)spill_blocks * 5 * gstride * 4);

    const int cus = c.num_cus;
    const int64_t nrows = row_hi - row_lo;
    // B is symmetric up to exchanging the two positions of every seed (exactly: the canonical seeds are min / max over a cross product of
    // positions per shared k-mer): a pair of rows of this context's window is accumulated on its smaller row only and the surviving
    // entries are mirrored into the partner's row afterwards (k_mirror) — half the accumulator updates, tables half as full.
    const bool half = phase >= 1 || !c.opt.no_symmetry || c.csr_inline;      // (rows with inline partners hold one triangle's pairs only: "no_symmetry" counts when A is built)
    const int64_t slack = (int64_t)cus * 32 * STAGE_CHUNK + 64;      // one open chunk per resident workgroup
    if (c.ov_tmp_cap == 0) {
        if (c.cfg.workspace_hint_bytes > 0) c.ov_tmp_cap = c.cfg.workspace_hint_bytes / (int64_t)sizeof(StageRec);
        else {
            // nnz(B) <= products / 2 and, on every read set seen so far, < nnz(A) / 4: start from nnz(A) (bounded by half the free memory)
            size_t free_b = 0, total_b = 0;
            ELBA_HIP(hipMemGetInfo(&free_b, &total_b));
            const int64_t budget = (int64_t)(free_b / 2 / (sizeof(StageRec) + (half ? 2 * (24 + 32) : 24)));
            c.ov_tmp_cap = std::min<int64_t>(std::max<int64_t>(half ? Z / 2 : Z, 1 << 16) + slack, std::max<int64_t>(budget, 1024));
        }
        if (c.ov_tmp_cap < 1024) c.ov_tmp_cap = 1024;
    }

    OvParams p{};
    p.a_rowptr = c.a_rowptr.as<uint32_t>(); p.a_csr = c.a_csr.as<uint64_t>();
    p.a_ell = c.use_ell ? c.a_ell.as<uint64_t>() : nullptr; p.a_ellj = c.csr_suffix ? c.a_ellj.as<uint32_t>() : nullptr; p.j_shift = c.j_shift; p.dense_up = (uint32_t)c.opt.dense_up; p.a_colptr = c.a_colptr.as<uint32_t>(); p.a_csc = c.a_csc.as<uint64_t>();
    p.s_stride = c.s_stride; p.lpc_log2 = c.lpc_log2; p.max_col = (uint32_t)(c.max_col_nnz > 0 ? c.max_col_nnz : 1);
    p.M = (uint32_t)M; p.Mcols = (uint32_t)M; p.row_lo = (uint32_t)row_lo; p.row_hi = (uint32_t)row_hi; p.fbits = c.fbits;
    p.half = phase >= 1 ? 2u : (half ? 1u : 0u);      // 2: a pair is accumulated on ONE of its two rows wherever the other row lives (its rank gets the mirrored entry by exchange)
    p.pos_mask = c.csr_suffix ? 0xFFFFu : (c.csr_hints ? 0x3FFFFFFFu : 0xFFFFFFFFu);
    p.hint_mask = !c.csr_hints ? 0u : (p.half == 2u ? 1u << 30 : (p.half == 1u ? 1u << 31 : 0u));
    // (the dense path: one triangle per window, partners outside the window kept — its candidate hand-out knows no other rule.  Both triangles
    //  ("no_symmetry") and the mirror exchange between ranks (half == 2: the parity rule over all ranks) take the general path, which reads the
    //  same entries through pos_mask)
    p.suffix = c.csr_suffix && p.half == 1u ? 1u : 0u;
    p.inl = c.csr_inline ? 1u : 0u;
    p.row_order = p.suffix && c.have_row_order ? c.row_order.as<uint32_t>() : nullptr; p.row_label = p.row_order ? c.row_label.as<uint32_t>() : nullptr;
    c.ov_hints_used = c.ov_hints_used || c.csr_inline;
    // (inline partners follow the parity rule over ALL rows: a whole matrix in one call, or a shard's rows with the mirror exchange; a windowed matrix
    //  multiplied alone keeps every partner outside its window — another rule)
    ELBA_REQUIRE(!c.csr_inline || (c.use_ell && (phase >= 1 || (row_lo == 0 && row_hi == M))), ELBA_ERR_STATE,
                 "this windowed matrix carries inline partners (option panel_inline): multiply it through elba_seed_matrix_send / _begin, or rebuild it without the option");
    c.ov_hints_used = p.hint_mask != 0u || p.suffix != 0u || c.csr_inline;      // (entries that fetch no column do not see its length: the product count comes from the build of A)
    p.prior_q16 = c.ov_prior_q16 ? c.ov_prior_q16 : 16384u;      // distinct partners per row entry: 1/4 until measured
    p.use_feedback = c.ov_prior_q16 ? 0u : 1u;
    p.fb_enough = (unsigned long long)std::min<int64_t>(std::max<int64_t>(Z / 32, 1 << 16), 1 << 23);
    bool pay = c.pos16 && !c.opt.no_pay;
    // Round 5: where the positions AND every row's product sequence numbers (rank in the row << fbits | place in the column) fit 16 bits — every read set of
    // ~10 kb reads — the extremes live in 32-bit words that carry posT (posQ is looked up in the row entry the sequence number names): the 2048-slot tier
    // then needs 37 KB of table + 16 KB of rings per 512-lane workgroup instead of 53 + 24.5: THREE rows per CU in flight instead of two (the kernel waits for
    // memory 69 % of its time: profiles/r04_summary.json), at the same 72-78 VGPRs.  Option "tune3" = 1 keeps the 64-bit accumulators (A/B).
    const bool pay16 = pay && c.use_ell && !c.csr_suffix && c.opt.tune[3] != 1 && ((uint64_t)c.max_row_nnz << c.fbits) <= 65536ull;
    p.pay16 = pay16 ? 1u : 0u;
    if (pay16) pay = false;
    // workgroup sizes grow with the table so that the largest tiers still bring enough waves to a CU (one or two workgroups fit its LDS)
    const uint32_t blk[NUM_LDS_TIERS] = {p.suffix ? 256u : 128u, p.suffix && p.dense_up >= 1u ? 512u : 256u, p.suffix && p.dense_up >= 2u ? 1024u : 512u, 1024u, 512u};      // (dense path: four wavefronts share a 512-slot table — 32 per CU)
    for (int t = 0; t < NUM_LDS_TIERS; ++t) {
        const uint32_t T = 1u << (LDS_TBITS0 + t);
        p.tier_limit[t] = std::min((T >> 2) * 3, T - blk[t]) - 1;      // a lane overshoots by at most one claim (Table::insert_lds)
    }
    c.ov_tickrows.reserve((size_t)(M / 32 + 2) * 4);
    p.tick_rows = c.ov_tickrows.as<uint32_t>();
    // a large matrix's rows start on the 2048-slot tier at least (three rows per CU with pay16): the two smaller tiers would receive a percent of the rows and cost a
    // ~60 us launch each — 6.48 -> 6.3x ms on config 3; small matrices keep them (their rows ARE small); option "tune4" = 1: every tier (A/B)
    p.qblk_log2 = c.opt.tune[5] > 0 ? (uint32_t)std::min<int64_t>(c.opt.tune[5] - 1, 12) : 0u;      // ("tune5" = log2 + 1.  Measured on config 5 at 1/25 — label-ordered queue, blocks of 32 / 128 / 512 places per XCD: 8.66-8.74 against 8.74-8.77 ms: nothing; single places stay)
    p.min_tier = (pay16 && nrows >= 65536 && Z / nrows >= 1024 && c.opt.tune[4] != 1) ? 2u : 0u;      // (long rows only: a 512-lane workgroup on a row of 75 entries would idle)
    if (c.opt.tune[7] >= 1 && c.opt.tune[7] <= NUM_TIERS) p.min_tier = (uint32_t)(c.opt.tune[7] - 1);      // ("tune7" = tier + 1: every row starts there at least; 6 = the HBM-table tier for all of them — what the spill tier costs when forced, bench.py)
    p.row_cnt = c.ov_rowcnt.as<uint32_t>(); p.low_cnt = c.ov_totcnt.as<uint32_t>();
    p.row_off = c.ov_rowoff.as<unsigned long long>(); p.lists = c.ov_lists.as<uint32_t>();
    p.fin_lists = c.ov_lists.as<uint32_t>() + (size_t)NUM_TIERS * (size_t)(M + 1);
    p.ctr = c.ov_counters.as<OvCounters>();
    p.gtable = c.ov_gtable.as<uint32_t>(); p.gstride = gstride;

    static DeviceOnce attr_once;
    attr_once.run(c.device, [&] {
        const int lds = 160 * 1024;
#define ELBA_ATTR(B, P, D) ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spgemm_direct<B, false, P, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spgemm_direct<512, false, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spgemm_direct<1024, false, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ELBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spgemm_direct<1024, false, false, 2, true, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ELBA_ATTR(512, true, 0); ELBA_ATTR(1024, true, 0); ELBA_ATTR(256, false, 0); ELBA_ATTR(512, false, 0); ELBA_ATTR(1024, false, 0);
        ELBA_ATTR(512, true, 1); ELBA_ATTR(512, true, 2); ELBA_ATTR(512, true, 4); ELBA_ATTR(1024, true, 1); ELBA_ATTR(1024, true, 2); ELBA_ATTR(1024, true, 4);
        ELBA_ATTR(256, false, 1); ELBA_ATTR(256, false, 2); ELBA_ATTR(256, false, 4);
        ELBA_ATTR(512, false, 1); ELBA_ATTR(512, false, 2); ELBA_ATTR(512, false, 4); ELBA_ATTR(1024, false, 1); ELBA_ATTR(1024, false, 2); ELBA_ATTR(1024, false, 4);
#undef ELBA_ATTR
    });

    c.ov_host.reserve(sizeof(OvCounters));
    OvCounters &hc = *static_cast<OvCounters *>(c.ov_host.p);
    hc = OvCounters{};
    uint32_t skipped_tiers = 0, skipped_sorts = 0;
    int passes = 0;
    float ms_sym = 0, ms_num = 0, ms_fin = 0, ms_tot = 0;
    bool was_timed = true;
    for (;;) {
        ++passes;
        const int stride = c.cfg.timing_stride > 1 ? c.cfg.timing_stride : 1;
        const bool timed = passes > 1 || (c.ov_calls++ % (uint64_t)stride) == 0;
        c.ov_tmp.reserve((size_t)c.ov_tmp_cap * sizeof(StageRec));
        p.tmp = c.ov_tmp.as<StageRec>(); p.tmp_cap = (unsigned long long)c.ov_tmp_cap;
        const bool mir16 = c.pos16 && !c.opt.mir32;
        c.ov_rec16 = mir16;
        p.rec16 = mir16 ? 1u : 0u; p.rec = c.ov_tmp.as<uint4>(); p.tick = reinterpret_cast<uint32_t *>(c.ov_tmp.as<char>() + (size_t)c.ov_tmp_cap * 16);
        c.b_cap_entries = half ? 2 * c.ov_tmp_cap : c.ov_tmp_cap;      // the output cannot be larger than what was staged (and mirrored)
        c.b_col.reserve((size_t)(c.b_cap_entries + 1) * 4);
        c.b_val.reserve((size_t)(c.b_cap_entries + 1) * sizeof(elba_seed_t));
        if (half) c.ov_mir.reserve((size_t)(c.b_cap_entries + 1) * (mir16 ? 16 : sizeof(StageRec)));
        // mirror slabs (above): one call on the window, 16-byte records, a ratio to size them by — a sample of this call's rows, an earlier call's
        // measurement, or the test hook
        p.use_feedback = c.ov_prior_q16 ? 0u : 1u;      // (a repeated pass starts like the first)
        const bool sampling = p.use_feedback && nrows >= 8192 && !c.opt.no_sample;
        p.slab_prior_q16 = c.opt.slab_q16 > 0 ? (uint32_t)c.opt.slab_q16 : c.ov_slab_q16;
        p.slab_pct = (uint32_t)c.opt.slab_pct;
        bool zero_slabn = false;
        c.ov_slab_on = phase == 0 && half && mir16 && !c.opt.no_slab && nrows > 0 && (sampling || p.slab_prior_q16 != 0u);
        p.slab = nullptr; p.slab_cap = 0;
        if (c.ov_slab_on) {
            c.ov_slab_cap = std::min<int64_t>(c.ov_tmp_cap + (int64_t)SLAB_PAD * nrows, 0xFFFF0000ll);
            // (the fill word of a row is slab end << 32 | next free entry, bumped once per image — also by those that find the slab full: a row receives
            //  at most M images, so the low half cannot carry into the end as long as the slab area + M stays below 2^32)
            if (c.ov_slab_cap + M >= 0xFFFFFFFFll) c.ov_slab_cap = std::max<int64_t>(0xFFFFFFFFll - M - 1, 0);
            if (c.ov_slab_cap <= (int64_t)SLAB_PAD * nrows + 1) c.ov_slab_on = false;      // (no room left for slabs under that bound: tickets + k_mirror)
        }
        if (c.ov_slab_on) {
            c.ov_slab.reserve((size_t)c.ov_slab_cap * 16);
            p.slab = c.ov_slab.as<uint4>(); p.slab_cap = (unsigned long long)c.ov_slab_cap;
            c.ov_slabpos.reserve((size_t)(M + 4) * 8); c.ov_slabn.reserve((size_t)(M + 8) * 4);
            p.slab_pos = c.ov_slabpos.as<unsigned long long>();
            zero_slabn = true;      // (rows outside the window hold no slab entries; zeroed with the call's other counters, below)
        }

        if (timed) c.ov_marks.mark(0, s);
        // the ticket counters come back clean from a call that ran to its end (k_finalize_wave); otherwise zero them here
        {   // the call's counters, in ONE launch (five memsets were five ~8 us gaps in front of a 6 ms call)
            ZeroList z{};
            auto add = [&](void *ptr, size_t bytes) { z.p[z.n] = static_cast<uint32_t *>(ptr); z.words[z.n] = (bytes + 3) / 4; ++z.n; };
            if (!c.ov_low_clean) add(c.ov_totcnt.p, (size_t)(M + 2) * 4);
            add(c.ov_counters.p, sizeof(OvCounters));
            add(c.ov_rowcnt.p, (size_t)(M + 2) * 4);
            add(c.ov_tickrows.p, (size_t)(M / 32 + 2) * 4);
            if (zero_slabn) add(c.ov_slabn.p, (size_t)(M + 8) * 4);
            size_t most = 0;
            for (int q = 0; q < z.n; ++q) most = std::max(most, z.words[q]);
            hipLaunchKernelGGL(k_zero_regions, dim3((unsigned)std::min<size_t>((most + 255) / 256, (size_t)cus * 8)), dim3(256), 0, s, z);
        }
        c.ov_low_clean = false;
        // A cold call on a matrix of some size computes a SAMPLE of its rows first (every sstep-th row, on the 4096-slot tier): what they find
        // — distinct partners per row entry — picks the starting tier of all the others, instead of a guess that sends most rows of a
        // 15 %-error read set to a tier too small (an abandoned attempt or a forwarding each: 0.9 ms of a 14.7 ms call on the 200 k-read set).
        p.nsample = sampling ? 256u : 0u; p.sstep = sampling ? (uint32_t)(nrows / 256) : 1u;
        c.ov_sample.reserve(256 * 4);
        p.sample_list = c.ov_sample.as<uint32_t>();
        if (sampling) hipLaunchKernelGGL(k_classify_direct, dim3(1), dim3(256), 0, s, p, 1);
        if (timed) c.ov_marks.mark(1, s);
        if (nrows > 0) {
            // bytes behind the table: misc words + per wavefront one product ring (128 entries of 12 / 8 bytes) and one row-entry FIFO (128 x 12 bytes)
            auto X = [&](int B, bool P) { return (size_t)256 + (size_t)(B / 64) * (P ? 3072 : (pay16 ? 2048 : 2560)); };
            const bool all_tiers = !c.ov_tiers_known || phase == 2;
            skipped_tiers = 0;
            // the highest tier ANY row of this matrix can reach: a row's distinct partners <= min(its entries x the longest column, reads), the tier that holds
            // twice that is guaranteed to fit it and k_classify_direct never starts a row above it.  A cold call on a small matrix launched five tiers
            // nobody could queue on (~5 us each, dependent: hifi-half 0.59 -> 0.53 ms with the finalize's counterpart).  (Still under the `missed` check below.)
            int tmax = NUM_TIERS;
            if (c.opt.tune[4] != 2) {
                const uint64_t ubm = std::min<uint64_t>((uint64_t)std::max<int64_t>(c.max_row_nnz, 1) * (uint64_t)p.max_col, (uint64_t)p.Mcols);
                const int gmax = ubm <= 1 ? 1 : 64 - __builtin_clzll(2 * ubm - 1);
                tmax = gmax <= LDS_TBITS0 ? 0 : gmax - LDS_TBITS0;
                tmax = std::max(tmax, (int)p.min_tier);
                if (p.suffix) tmax = std::max(tmax, (int)p.dense_up);
            }
#define ELBA_DTIER(t, stmt) do { if ((all_tiers || c.ov_tier_used[t]) && (t) <= tmax) { stmt; } else skipped_tiers |= 1u << (t); } while (0)
#define ELBA_LAUNCH_D(B, G, P, grid, lds, tier, tb, smp)                                                                                  \
    do {                                                                                                                                  \
        if (dk == 0) hipLaunchKernelGGL((k_spgemm_direct<B, G, P, 0>), dim3(grid), dim3(B), (lds), s, p, (tier), (tb), (smp));            \
        else if (dk == 1) hipLaunchKernelGGL((k_spgemm_direct<B, G, P, 1>), dim3(grid), dim3(B), (lds), s, p, (tier), (tb), (smp));       \
        else if (dk == 4) hipLaunchKernelGGL((k_spgemm_direct<B, G, P, 4>), dim3(grid), dim3(B), (lds), s, p, (tier), (tb), (smp));       \
        else hipLaunchKernelGGL((k_spgemm_direct<B, G, P, 2>), dim3(grid), dim3(B), (lds), s, p, (tier), (tb), (smp));                    \
    } while (0)
            // gather trips per iteration of the padded-column loop: 1 (DK = 0) where the rows mostly carry their products inline — columns of 2-3
            // reads, 15 %-error reads: 6.49 -> 6.27 ms on config 3 —, 2 (DK = 1) otherwise (columns of ~7 reads at 5 % error lose 4 % with one trip);
            // the option "dk" (0, 1, 2, 4) overrides
            const int dk = c.opt.dk >= 0 ? c.opt.dk : ((c.csr_inline && c.N > 0 && c.Z < 3 * c.N) ? 0 : 1);
// (dense path: 32-bit accumulators + seed look-ups for the few survivors; the first tier's grid is a tuning knob: the path waits for memory)
#define ELBA_LAUNCH_S(B, TBC, grid, lds, tier, tb, smp) hipLaunchKernelGGL((k_spgemm_direct<B, false, false, 2, true, TBC>), dim3((tier) == 0 ? cus * c.opt.dense_wgs : (grid)), dim3(B), (size_t)18 * (1u << (tb)) + 256 + (size_t)((B) / 64) * 2368, s, p, (tier), (tb), (smp))
            if (sampling) {
                if (p.suffix) ELBA_LAUNCH_S(1024, 0, cus, (size_t)26 * 4096 + X(1024, true), 3, 12u, 1u);
                else if (pay) ELBA_LAUNCH_D(1024, false, true, cus, (size_t)26 * 4096 + X(1024, true), 3, 12u, 1u);
                else ELBA_LAUNCH_D(1024, false, false, cus, (size_t)18 * 4096 + X(1024, false), 3, 12u, 1u);
                p.use_feedback = 0;      // the ratio is measured: nothing is forwarded on a prediction any more, nobody touches the hot sums
            }
            {
                int nb = (int)((nrows + 255) / 256);
                if (nb > cus * 4) nb = cus * 4;
                hipLaunchKernelGGL(k_classify_direct, dim3(nb), dim3(256), 0, s, p, 0);
            }
            if (p.suffix) {      // (dense matrices: the LDS tiers with 64-bit accumulators run the dense path; p.suffix implies pay)
                ELBA_DTIER(0, ELBA_LAUNCH_S(256, 9, cus * 9, (size_t)26 * 512 + X(128, true), 0, 9u, 0u));
                if (p.dense_up >= 1u) ELBA_DTIER(1, ELBA_LAUNCH_S(512, 10, cus * 4, (size_t)26 * 1024 + X(256, true), 1, 10u, 0u));      // (eight wavefronts share a 1024-slot table: 32 per CU again, half the load)
                else ELBA_DTIER(1, ELBA_LAUNCH_S(256, 0, cus * 4, (size_t)26 * 1024 + X(256, true), 1, 10u, 0u));
                if (p.dense_up >= 2u) ELBA_DTIER(2, ELBA_LAUNCH_S(1024, 11, cus * 2, (size_t)26 * 2048 + X(512, true), 2, 11u, 0u));
                else ELBA_DTIER(2, ELBA_LAUNCH_S(512, 0, cus * 2, (size_t)26 * 2048 + X(512, true), 2, 11u, 0u));
                ELBA_DTIER(3, ELBA_LAUNCH_S(1024, 0, cus, (size_t)26 * 4096 + X(1024, true), 3, 12u, 0u));
            } else if (pay) {
                ELBA_DTIER(0, ELBA_LAUNCH_D(128, false, true, cus * 9, (size_t)26 * 512 + X(128, true), 0, 9u, 0u));
                ELBA_DTIER(1, ELBA_LAUNCH_D(256, false, true, cus * 4, (size_t)26 * 1024 + X(256, true), 1, 10u, 0u));
                ELBA_DTIER(2, ELBA_LAUNCH_D(512, false, true, cus * 2, (size_t)26 * 2048 + X(512, true), 2, 11u, 0u));
                ELBA_DTIER(3, ELBA_LAUNCH_D(1024, false, true, cus, (size_t)26 * 4096 + X(1024, true), 3, 12u, 0u));
            } else {
                ELBA_DTIER(0, ELBA_LAUNCH_D(128, false, false, cus * 12, (size_t)18 * 512 + X(128, false), 0, 9u, 0u));
                ELBA_DTIER(1, ELBA_LAUNCH_D(256, false, false, cus * 7, (size_t)18 * 1024 + X(256, false), 1, 10u, 0u));
                ELBA_DTIER(2, ELBA_LAUNCH_D(512, false, false, cus * 3, (size_t)18 * 2048 + X(512, false), 2, 11u, 0u));
                ELBA_DTIER(3, ELBA_LAUNCH_D(1024, false, false, cus, (size_t)18 * 4096 + X(1024, false), 3, 12u, 0u));
            }
            ELBA_DTIER(4, ELBA_LAUNCH_D(256, false, false, cus, (size_t)18 * 8192 + X(256, false), 4, 13u, 0u));      // (4 wavefronts: 8192 slots + their rings fill the 160 KB)
            ELBA_DTIER(5, ELBA_LAUNCH_D(256, true, false, spill_blocks, (size_t)256 + 4 * 2560, NUM_LDS_TIERS, 0u, 0u));      // (the HBM tier never packs its FIFO entries: 2560 bytes of rings per wavefront)
#undef ELBA_LAUNCH_S
#undef ELBA_LAUNCH_D
#undef ELBA_DTIER
            ELBA_HIP(hipGetLastError());
        }
        if (timed) c.ov_marks.mark(2, s);
        if (phase == 2) { c.ov_pend_passes = 1; c.ov_pend_timed = timed; c.ov_phase = 2; return; }      // (queued: no read-back, no wait)
        if (phase == 0) ov_launch_finalize(c, p.half, !c.ov_tiers_known, skipped_sorts, nullptr, 0);
        if (timed) c.ov_marks.mark(3, s);
        ELBA_HIP(hipMemcpyAsync(&hc, c.ov_counters.p, sizeof(OvCounters), hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        if (timed) { ms_sym += c.ov_marks.ms(0, 1); ms_num += c.ov_marks.ms(1, 2); ms_fin = c.ov_marks.ms(2, 3); ms_tot += c.ov_marks.ms(0, 3); }
        was_timed = timed;
        bool missed = false;
        for (int t = 0; t < NUM_TIERS; ++t) missed |= ((skipped_tiers >> t) & 1u) && hc.tier_count[t] > 0;
        missed |= ((skipped_sorts & 1u) && hc.fin_count[0] > 0) || ((skipped_sorts & 2u) && hc.fin_count[1] > 0);
        if (hc.overflow || missed) {      // staging too small, or a row reached a tier / sort that was not launched: repeat with what is known now
            if (c.opt.trace) fprintf(stderr, "[elba] overlap call repeated: overflow=%u missed=%d cursor=%llu tmp_cap=%lld\n", hc.overflow, (int)missed, hc.cursor, (long long)c.ov_tmp_cap);
            ELBA_REQUIRE(passes < 4, ELBA_ERR_INTERNAL, "overlap output did not settle");
            if (hc.overflow) c.ov_tmp_cap = (int64_t)hc.cursor + slack;      // every row drew its space even when it did not fit: the cursor is the need
            c.ov_tiers_known = false;
            continue;
        }
        c.ov_low_clean = phase == 0;
        break;
    }
    if (phase == 1) { c.ov_pend_passes = passes; c.ov_pend_timed = was_timed; c.ov_pend_ms[0] = ms_tot; c.ov_pend_ms[1] = ms_sym; c.ov_pend_ms[2] = ms_num; c.ov_phase = 1; return; }

    ov_finish_stats(c, hc, st, passes, was_timed, ms_tot, ms_sym, ms_num, ms_fin, 0, 0);
}



// ---- sharded call with mirror exchange: begin -> (counts, fill: the driver's all-to-all) -> end -----------------------------------------
static RemoteParams remote_params(Ctx &c)
{
    RemoteParams r{};
    r.row_cnt = c.ov_rowcnt.as<uint32_t>(); r.row_off = c.ov_rowoff.as<unsigned long long>(); r.tmp = c.ov_tmp.as<StageRec>(); r.rec16 = c.ov_rec16 ? 1u : 0u; r.rec = c.ov_tmp.as<uint4>();
    r.row_lo = (uint32_t)c.row_lo; r.row_hi = (uint32_t)(c.row_hi < 0 ? c.M : c.row_hi); r.nranks = (uint32_t)c.ov_remote_bounds.size() - 1u;
    for (size_t k = 0; k < c.ov_remote_bounds.size(); ++k) r.bounds[k] = c.ov_remote_bounds[k];
    return r;
}

void stage_seed_matrix_begin(Ctx &c, int nranks, const uint64_t *bounds_host, uint64_t *send_counts_host)
{
    ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "seed_matrix_begin: no k-mer matrix");
    ELBA_REQUIRE(nranks >= 1 && nranks <= REMOTE_MAX_RANKS && bounds_host && send_counts_host, ELBA_ERR_INVALID_ARG, "seed_matrix_begin: 1..64 ranks, read bounds and a count array");
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? c.M : c.row_hi;
    bool found = false;
    for (int r = 0; r < nranks; ++r) found |= (int64_t)bounds_host[r] == row_lo && (int64_t)bounds_host[r + 1] == row_hi;
    ELBA_REQUIRE(found && (int64_t)bounds_host[nranks] == c.M, ELBA_ERR_INVALID_ARG, "seed_matrix_begin: this context's row window is not one of the ranks' row ranges");
    c.ov_remote_bounds.assign(bounds_host, bounds_host + nranks + 1);
    create_seed_matrix_direct(c, 1);
    hipStream_t s = c.stream;
    c.ws_scan.reserve(REMOTE_MAX_RANKS * 8);
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, REMOTE_MAX_RANKS * 8, s));
    const int64_t nrows = row_hi - row_lo;
    if (nrows > 0 && nranks > 1)
        hipLaunchKernelGGL((k_remote_mirror<false>), dim3((unsigned)((nrows + REMOTE_ROWS_PER_BLOCK - 1) / REMOTE_ROWS_PER_BLOCK)), dim3(256), 0, s, remote_params(c), c.ws_scan.as<unsigned long long>(), (StageRec *)nullptr);
    ELBA_HIP(hipMemcpyAsync(send_counts_host, c.ws_scan.p, (size_t)nranks * 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
}

void stage_seed_matrix_fill(Ctx &c, void *d_send, const uint64_t *offsets_host)
{
    ELBA_REQUIRE(c.ov_phase == 1, ELBA_ERR_STATE, "seed_matrix_fill: call seed_matrix_begin first");
    hipStream_t s = c.stream;
    const int nranks = (int)c.ov_remote_bounds.size() - 1;
    c.ws_scan.reserve(REMOTE_MAX_RANKS * 8);
    ELBA_HIP(hipMemcpyAsync(c.ws_scan.p, offsets_host, (size_t)nranks * 8, hipMemcpyHostToDevice, s));
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? c.M : c.row_hi, nrows = row_hi - row_lo;
    if (nrows > 0 && nranks > 1)
        hipLaunchKernelGGL((k_remote_mirror<true>), dim3((unsigned)((nrows + REMOTE_ROWS_PER_BLOCK - 1) / REMOTE_ROWS_PER_BLOCK)), dim3(256), 0, s, remote_params(c), c.ws_scan.as<unsigned long long>(), static_cast<StageRec *>(d_send));
    ELBA_HIP(hipStreamSynchronize(s));
}

void stage_seed_matrix_end(Ctx &c, const void *d_recv, int64_t nrecv)
{
    ELBA_REQUIRE(c.ov_phase == 1, ELBA_ERR_STATE, "seed_matrix_end: call seed_matrix_begin first");
    ELBA_REQUIRE(nrecv >= 0 && (nrecv == 0 || d_recv), ELBA_ERR_INVALID_ARG, "seed_matrix_end: null records");
    hipStream_t s = c.stream;
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? c.M : c.row_hi;
    OvCounters &hc = *static_cast<OvCounters *>(c.ov_host.p);          // the numeric phase's counters (read back by begin)
    int64_t staged = 0;
    for (int sh = 0; sh < NUM_SHARDS; ++sh) staged += (int64_t)hc.shard[sh].nnz;
    // the output holds what was staged, mirrored locally and received
    if (c.b_cap_entries < staged + nrecv + 1) c.b_cap_entries = staged + nrecv + 1;
    c.b_col.reserve((size_t)(c.b_cap_entries + 1) * 4);
    c.b_val.reserve((size_t)(c.b_cap_entries + 1) * sizeof(elba_seed_t));
    const bool mir16 = c.pos16 && !c.opt.mir32;
    c.ov_mir.reserve((size_t)(c.b_cap_entries + 1) * (mir16 ? 16 : sizeof(StageRec)));
    c.ov_marks.mark(2, s);
    c.ov_remote.reserve((size_t)(nrecv + 1) * sizeof(StageRec));
    c.ws_scan.reserve(64);
    ELBA_HIP(hipMemsetAsync(c.ws_scan.p, 0, 16, s));
    if (nrecv > 0) {
        ELBA_HIP(hipMemcpyAsync(c.ov_remote.p, d_recv, (size_t)nrecv * sizeof(StageRec), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_ingest_remote, dim3((unsigned)((nrecv + 255) / 256)), dim3(256), 0, s, c.ov_remote.as<StageRec>(), (unsigned long long)nrecv, c.ov_totcnt.as<uint32_t>(),
                           (uint32_t)row_lo, (uint32_t)row_hi, c.ws_scan.as<unsigned long long>(), c.ws_scan.as<unsigned long long>() + 1);
    }
    unsigned long long chk[2] = {0, 0};
    ELBA_HIP(hipMemcpyAsync(chk, c.ws_scan.p, 16, hipMemcpyDeviceToHost, s));
    uint32_t skipped_sorts = 0;
    ov_launch_finalize(c, 2u, true, skipped_sorts, c.ov_remote.as<StageRec>(), nrecv);
    c.ov_marks.mark(3, s);
    ELBA_HIP(hipMemcpyAsync(&hc, c.ov_counters.p, sizeof(OvCounters), hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    ELBA_REQUIRE(chk[0] == 0, ELBA_ERR_INVALID_ARG, "seed_matrix_end: received records for rows outside this context's window");
    c.ov_low_clean = true;
    c.ov_phase = 0;
    elba_overlap_stats st{};
    st.nrows = row_hi - row_lo;
    const float ms_fin = c.ov_marks.ms(2, 3);
    ov_finish_stats(c, hc, st, c.ov_pend_passes, c.ov_pend_timed, c.ov_pend_ms[0] + ms_fin, c.ov_pend_ms[1], c.ov_pend_ms[2], ms_fin, nrecv, (int64_t)chk[1]);
}

// ---- the same step with ONE host synchronisation: fixed-size slots, nothing about the exchange is known on the host ------------------------
// d_send = nranks slots of `slot` 32-byte records: record 0 of slot r is a header — a = (count, flags, need lo, need hi): records for rank r that
// follow, 1 = "this rank ran out of room: every rank repeats the step", the slot size this rank would have needed — written on the device.
// The all-to-all has equal splits; every rank receives a header from every rank, so all of them reach the same verdict.
__global__ __launch_bounds__(256) void k_remote_mirror_slots(RemoteParams p, unsigned long long *cursors, StageRec *send, unsigned long long slot)
{
    __shared__ uint32_t cnt[REMOTE_MAX_RANKS], fill[REMOTE_MAX_RANKS];
    __shared__ unsigned long long base[REMOTE_MAX_RANKS];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t r0 = p.row_lo + blockIdx.x * REMOTE_ROWS_PER_BLOCK, r1 = min(r0 + (uint32_t)REMOTE_ROWS_PER_BLOCK, p.row_hi);
    if (tid < REMOTE_MAX_RANKS) { cnt[tid] = 0; fill[tid] = 0; }
    __syncthreads();
    for (uint32_t i = r0 + w; i < r1; i += 4) {
        const uint32_t n = p.row_cnt[i];
        const unsigned long long off = p.row_off[i];
        for (uint32_t t = lane; t < n; t += 64) {
            const uint32_t j = p.rec16 ? p.rec[off + t].x : p.tmp[off + t].a.x;
            if (j != i && (j < p.row_lo || j >= p.row_hi)) atomicAdd(&cnt[remote_rank_of(p, j)], 1u);
        }
    }
    __syncthreads();
    if (tid < p.nranks && cnt[tid]) base[tid] = atomicAdd(&cursors[tid], (unsigned long long)cnt[tid]);
    __syncthreads();
    for (uint32_t i = r0 + w; i < r1; i += 4) {
        const uint32_t n = p.row_cnt[i];
        const unsigned long long off = p.row_off[i];
        for (uint32_t t = lane; t < n; t += 64) {
            uint4 a, b = make_uint4(0u, 0u, 0u, 0u);
            if (p.rec16) { const uint4 m = p.rec[off + t]; a = make_uint4(m.x, 0u, m.y & 0xFFFFu, m.y >> 16); b = make_uint4(m.z & 0xFFFFu, m.z >> 16, m.w, 0u); }
            else a = p.tmp[off + t].a;
            const uint32_t j = a.x;
            if (j != i && (j < p.row_lo || j >= p.row_hi)) {
                if (!p.rec16) b = p.tmp[off + t].b;
                const uint32_t d = remote_rank_of(p, j);
                const unsigned long long k = base[d] + atomicAdd(&fill[d], 1u);
                if (k + 1 < slot) {                                     // (beyond the slot: dropped — the header says so and the step is repeated)
                    StageRec *dst = send + (unsigned long long)d * slot + 1 + k;
                    dst->a = make_uint4(j, i, a.w, a.z);
                    dst->b = make_uint4(b.y, b.x, b.z, 0u);
                }
            }
        }
    }
}
__global__ void k_slot_headers(const unsigned long long *cursors, const OvCounters *ctr, StageRec *send, unsigned long long slot, uint32_t nranks)
{
    __shared__ unsigned long long need;
    if (threadIdx.x == 0) need = 0;
    __syncthreads();
    const uint32_t d = threadIdx.x;
    if (d < nranks) atomicMax(&need, cursors[d] + 1);
    __syncthreads();
    if (d < nranks) {
        const bool over = need > slot || ctr->overflow != 0;
        const unsigned long long cntd = cursors[d] + 1 <= slot ? cursors[d] : slot - 1;
        send[(unsigned long long)d * slot].a = make_uint4((uint32_t)cntd, over ? 1u : 0u, (uint32_t)need, (uint32_t)(need >> 32));
        send[(unsigned long long)d * slot].b = make_uint4(0u, 0u, 0u, 0u);
    }
}
// received slots: every record draws its place in its row's mirror extent; chk = {records for rows outside the window, strict-upper entries, retry
// flags, slot need, records merged}.  blockIdx.y = slot; a block beyond the slot's count leaves after one load (slots are sized by a bound).
__global__ void k_ingest_remote_slots(StageRec *rem, unsigned long long slot, uint32_t *low_cnt, uint32_t row_lo, uint32_t row_hi, unsigned long long *chk)
{
    StageRec *base = rem + (unsigned long long)blockIdx.y * slot;
    uint4 h = base->a;
    const bool garbage = (unsigned long long)h.x + 1 > slot;      // more records than the slot holds: not a header this library wrote for this slot size
    if (garbage) h.x = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (garbage) atomicAdd(&chk[0], 1ull);
        if (h.y) atomicOr(&chk[2], 1ull);
        atomicMax(&chk[3], ((unsigned long long)h.w << 32) | h.z);
    }
    const unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1;      // record 0 is the header
    if ((unsigned long long)blockIdx.x * blockDim.x >= h.x) return;
    unsigned long long up = 0, got = 0;
    if (idx <= h.x) {
        const uint32_t j = base[idx].a.x;
        if (j < row_lo || j >= row_hi) atomicAdd(&chk[0], 1ull);
        else { base[idx].b.w = atomicAdd(&low_cnt[j], 1u); up = base[idx].a.y > j ? 1u : 0u; got = 1; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { up += __shfl_xor(up, d, 64); got += __shfl_xor(got, d, 64); }
    if ((threadIdx.x & 63) == 0) { if (up) atomicAdd(&chk[1], up); if (got) atomicAdd(&chk[4], got); }
}

void stage_seed_matrix_send(Ctx &c, int nranks, const uint64_t *bounds_host, void *d_send, int64_t slot)
{
    ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "seed_matrix_send: no k-mer matrix");
    ELBA_REQUIRE(nranks >= 1 && nranks <= REMOTE_MAX_RANKS && bounds_host && d_send && slot >= 2, ELBA_ERR_INVALID_ARG, "seed_matrix_send: 1..64 ranks, read bounds, a send buffer of nranks slots of >= 2 records");
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? c.M : c.row_hi;
    bool found = false;
    for (int r = 0; r < nranks; ++r) found |= (int64_t)bounds_host[r] == row_lo && (int64_t)bounds_host[r + 1] == row_hi;
    ELBA_REQUIRE(found && (int64_t)bounds_host[nranks] == c.M, ELBA_ERR_INVALID_ARG, "seed_matrix_send: this context's row window is not one of the ranks' row ranges");
    c.ov_remote_bounds.assign(bounds_host, bounds_host + nranks + 1);
    create_seed_matrix_direct(c, 2);
    c.ov_send_slot = slot;
    hipStream_t s = c.stream;
    c.ov_cursors.reserve(REMOTE_MAX_RANKS * 8 + 64);
    ELBA_HIP(hipMemsetAsync(c.ov_cursors.p, 0, REMOTE_MAX_RANKS * 8 + 64, s));
    const int64_t nrows = row_hi - row_lo;
    if (nrows > 0 && nranks > 1)
        hipLaunchKernelGGL(k_remote_mirror_slots, dim3((unsigned)((nrows + REMOTE_ROWS_PER_BLOCK - 1) / REMOTE_ROWS_PER_BLOCK)), dim3(256), 0, s, remote_params(c), c.ov_cursors.as<unsigned long long>(),
                           static_cast<StageRec *>(d_send), (unsigned long long)slot);
    hipLaunchKernelGGL(k_slot_headers, dim3(1), dim3(64), 0, s, (const unsigned long long *)c.ov_cursors.as<unsigned long long>(), (const OvCounters *)c.ov_counters.as<OvCounters>(),
                       static_cast<StageRec *>(d_send), (unsigned long long)slot, (uint32_t)nranks);
    ELBA_HIP(hipGetLastError());
}

// returns false when the step has to be repeated (this rank or any sender ran out of room; *slot_needed = the slot size that would have done)
bool stage_seed_matrix_recv(Ctx &c, void *d_recv, int64_t slot, int64_t *slot_needed)
{
    ELBA_REQUIRE(c.ov_phase == 2, ELBA_ERR_STATE, "seed_matrix_recv: call seed_matrix_send first");
    ELBA_REQUIRE(d_recv && slot >= 2, ELBA_ERR_INVALID_ARG, "seed_matrix_recv: null records");
    ELBA_REQUIRE(slot == c.ov_send_slot, ELBA_ERR_INVALID_ARG, "seed_matrix_recv: slot_records differs from the value given to seed_matrix_send");
    hipStream_t s = c.stream;
    const int64_t row_lo = c.row_lo, row_hi = c.row_hi < 0 ? c.M : c.row_hi;
    const int nranks = (int)c.ov_remote_bounds.size() - 1;
    const int64_t nslotrec = (int64_t)nranks * slot;
    // the output holds what was staged, mirrored locally (both bounded by the staging capacity) and received (bounded by the slots)
    c.b_cap_entries = 2 * c.ov_tmp_cap + nslotrec + 1;
    c.b_col.reserve((size_t)(c.b_cap_entries + 1) * 4);
    c.b_val.reserve((size_t)(c.b_cap_entries + 1) * sizeof(elba_seed_t));
    const bool mir16 = c.pos16 && !c.opt.mir32;
    c.ov_mir.reserve((size_t)(c.b_cap_entries + 1) * (mir16 ? 16 : sizeof(StageRec)));
    c.ov_marks.mark(4, s);
    unsigned long long *chk = c.ov_cursors.as<unsigned long long>() + REMOTE_MAX_RANKS;      // (zeroed by send)
    hipLaunchKernelGGL(k_ingest_remote_slots, dim3((unsigned)((slot + 255) / 256), (unsigned)nranks), dim3(256), 0, s, static_cast<StageRec *>(d_recv), (unsigned long long)slot,
                       c.ov_totcnt.as<uint32_t>(), (uint32_t)row_lo, (uint32_t)row_hi, chk);
    uint32_t skipped_sorts = 0;
    ov_launch_finalize(c, 2u, true, skipped_sorts, static_cast<const StageRec *>(d_recv), nslotrec, slot);
    c.ov_marks.mark(3, s);
    OvCounters &hc = *static_cast<OvCounters *>(c.ov_host.p);
    unsigned long long hchk[5] = {0, 0, 0, 0, 0};
    ELBA_HIP(hipMemcpyAsync(&hc, c.ov_counters.p, sizeof(OvCounters), hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipMemcpyAsync(hchk, chk, sizeof(hchk), hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));      // the step's one synchronisation
    c.ov_phase = 0;
    ELBA_REQUIRE(hchk[0] == 0, ELBA_ERR_INVALID_ARG, "seed_matrix_recv: received records for rows outside this context's window, or a slot header that claims more records than a slot holds");
    if (hc.overflow || hchk[2]) {
        // some rank's staging area or slot was too small: every rank saw the flag (it travels in every header) and repeats the step
        const int64_t slack = (int64_t)c.num_cus * 32 * STAGE_CHUNK + 64;
        if (hc.overflow) c.ov_tmp_cap = (int64_t)hc.cursor + slack;
        if (slot_needed) *slot_needed = std::max<int64_t>((int64_t)hchk[3] + (int64_t)hchk[3] / 8 + 16, slot);
        c.ov_low_clean = false;
        return false;
    }
    // (what would have sufficed — the same number on every rank: the largest count any sender put into any slot, + 1/8 — lets the caller SHRINK a
    //  first guess that was several times too generous: the all-to-all moves whole slots)
    if (slot_needed) *slot_needed = std::min<int64_t>(slot, (int64_t)hchk[3] + (int64_t)hchk[3] / 8 + 16);
    c.ov_low_clean = true;
    elba_overlap_stats st{};
    st.nrows = row_hi - row_lo;
    const float ms_sym = c.ov_pend_timed ? c.ov_marks.ms(0, 1) : 0.f, ms_num = c.ov_pend_timed ? c.ov_marks.ms(1, 2) : 0.f, ms_fin = c.ov_marks.ms(4, 3);
    ov_finish_stats(c, hc, st, 1, c.ov_pend_timed, c.ov_pend_timed ? c.ov_marks.ms(0, 3) : ms_fin, ms_sym, ms_num, ms_fin, (int64_t)hchk[4], (int64_t)hchk[1]);
    return true;
}

void stage_create_seed_matrix(Ctx &c)
{
    ELBA_REQUIRE(c.have_A, ELBA_ERR_STATE, "create_seed_matrix: no k-mer matrix (call elba_create_kmer_matrix or elba_set_kmer_matrix)");
    create_seed_matrix_direct(c, 0);
}

}  // namespace elba
