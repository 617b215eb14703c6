// tr.hip — from the aligned pairs to the string graph: what src/main.cpp:305-312 does to the upper-triangular R of
// PairwiseAlignment —
//   find_bad_reads (src/main.cpp:553-571)       reads v with (passed entries of row v and column v + 1) / (entries + 1) <= cutoff
//   R->Prune(!passed); PruneFull(bad, bad)      (:306-307)
//   find_contained_reads (:573-583)             containedQ marks the row's read, containedT the column's; PruneFull (:311)
//   TransitiveReduction (src/TransitiveReduction.cpp:3-90)   R += transposed R; N = R (x) R over MinPlusSR; I = (R.suffix + FUZZ >=
//                                               N.suffix_paths[R.direction]); I |= I^T; S = R \ I, entries without a direction dropped.
//
// The reference computes N in full — every pair (i,j) joined by some k, four path slots each — with a distributed SpGEMM, prunes it, and
// then reads ONE slot of it at the positions where R has an entry (EWiseApply on the intersection, GreaterThanSR, include/
// TransitiveReduction.hpp:53-65).  Here only that is computed: for every entry R(i,j) with direction d, the minimum over the common
// neighbours k of R(i,k).suffix + R(k,j).suffix among the products MinPlusSR::multiply (:88-104) puts in slot d — a masked product,
// rows of the symmetrised R intersected pairwise.  Same lookups as the SpGEMM has products (sum over reads of degree^2), no
// intermediate matrix, and min / + over int are order-free, so the result is the reference's for any fold order.
//
// The reference's do-while runs a second pass when the first removed something: its P is then N, whose entries are built by Overlap()
// inside multiply and so have direction -1 (include/Overlap.hpp:10 via src/Overlap.cpp:4-10) — arrows() fails for every product, the
// pass finds nothing and the loop ends.  One masked pass therefore IS the reference's result; stats.iterations reports the pass count
// the reference would have logged (the oracle runs the loop literally and agrees).
//
// Layout: the symmetrised R as CSR over the reads — u32 row pointers, 16-byte entries {col, suffix, suffixT, direction | directionT << 3
// | below-diagonal << 6}, columns ascending — built by one radix sort of the 2 x kept (row << b | col) keys.  The mark kernel stages row
// i (columns, suffixes, directions) in LDS; a wavefront takes one entry (i,j) at a time, its lanes the entries (j,k) of row j, each lane
// a binary search for k in the staged row.
#include "common.hpp"

namespace elba {

namespace {

struct alignas(16) SymEntry { uint32_t col; int32_t suffix, suffixT; uint32_t meta; };

struct TrParams {
    const int64_t *rows, *cols; const elba_overlap_t *vals; int64_t n;
    uint32_t M; int mb; double cutoff; int fuzz;
    uint32_t *deg, *pas; uint8_t *flags;
    uint64_t *keys, *kv;
    unsigned long long *ctr;       // 0 kept entries (upper), 1 bad reads, 2 contained reads, 3 passed entries, 4 products, 5 marked, 6 removed (directed)
};

__global__ void k_tr_degrees(TrParams p)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.n) return;
    const uint32_t i = (uint32_t)p.rows[a], j = (uint32_t)p.cols[a];
    atomicAdd(&p.deg[i], 1u); atomicAdd(&p.deg[j], 1u);
    if (p.vals[a].passed) { atomicAdd(&p.pas[i], 1u); atomicAdd(&p.pas[j], 1u); }
}

__global__ void k_tr_bad(TrParams p)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= p.M) return;
    const double r = ((double)p.pas[v] + 1.0) / ((double)p.deg[v] + 1.0);          // src/main.cpp:568
    const bool bad = r <= p.cutoff;
    p.flags[v] = bad ? 1 : 0;
    const unsigned long long b = __ballot(bad);
    if (b && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(b)) atomicAdd(&p.ctr[1], (unsigned long long)__builtin_popcountll(b));
}

// contained reads, found on the entries that survive the passed / bad-read prune (every writer stores the same 1: no atomics needed)
__global__ void k_tr_contained(TrParams p, uint32_t *cont)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.n) return;
    const elba_overlap_t o = p.vals[a];
    const uint32_t i = (uint32_t)p.rows[a], j = (uint32_t)p.cols[a];
    if (!o.passed || (p.flags[i] & 1) || (p.flags[j] & 1)) return;
    if (o.containedQ) cont[i] = 1u;
    if (o.containedT) cont[j] = 1u;
}

__global__ void k_tr_merge_flags(TrParams p, const uint32_t *cont)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= p.M) return;
    const bool c = cont[v] != 0;
    if (c) p.flags[v] |= 2;
    const unsigned long long b = __ballot(c);
    if (b && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(b)) atomicAdd(&p.ctr[2], (unsigned long long)__builtin_popcountll(b));
}

// the entries handed to TransitiveReduction, each emitted twice: (i, j) as it is and (j, i) to be transposed.  Slots come from one
// atomic per wavefront; the sort that follows restores an order.
__global__ void k_tr_emit(TrParams p)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool passed = false, keep = false;
    uint32_t i = 0, j = 0;
    if (a < p.n) {
        i = (uint32_t)p.rows[a]; j = (uint32_t)p.cols[a];
        passed = p.vals[a].passed && !(p.flags[i] & 1) && !(p.flags[j] & 1);
        keep = passed && !(p.flags[i] & 2) && !(p.flags[j] & 2);
    }
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long bp = __ballot(passed), bk = __ballot(keep);
    if (bp && lane == (unsigned)__builtin_ctzll(bp)) atomicAdd(&p.ctr[3], (unsigned long long)__builtin_popcountll(bp));
    if (!bk) return;
    const int leader = __builtin_ctzll(bk);
    unsigned long long base = 0;
    if ((int)lane == leader) base = atomicAdd(&p.ctr[0], (unsigned long long)__builtin_popcountll(bk));
    base = __shfl(base, leader);
    if (keep) {
        const unsigned long long slot = 2 * (base + __builtin_popcountll(bk & ((1ull << lane) - 1)));
        p.keys[slot] = ((uint64_t)i << p.mb) | j;     p.kv[slot] = (uint64_t)a << 1;
        p.keys[slot + 1] = ((uint64_t)j << p.mb) | i; p.kv[slot + 1] = ((uint64_t)a << 1) | 1;
    }
}

__device__ __forceinline__ uint32_t dir3(int8_t d) { return d < 0 ? 7u : (uint32_t)d & 3u; }

__global__ void k_tr_gather(const uint64_t *keys, const uint64_t *kv, int64_t n2, int mb, const elba_overlap_t *vals, SymEntry *sym, uint32_t *src)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= n2) return;
    const uint64_t v = kv[z];
    const elba_overlap_t o = vals[v >> 1];
    SymEntry e;
    e.col = (uint32_t)(keys[z] & ((1ull << mb) - 1));
    if (v & 1) { e.suffix = o.suffixT; e.suffixT = o.suffix; e.meta = dir3(o.directionT) | dir3(o.direction) << 3 | 1u << 6; }     // Overlap::Transpose
    else { e.suffix = o.suffix; e.suffixT = o.suffixT; e.meta = dir3(o.direction) | dir3(o.directionT) << 3; }
    sym[z] = e; src[z] = (uint32_t)(v >> 1);
}

__device__ __forceinline__ int wave_min_i32(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false); v = t < v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}

// The masked min-plus product and the comparison.  For the entry (i,j), direction d = 2 t + h, the products that land in slot d of
// N(i,j) are R(i,k) (x) R(k,j) with tail bit of R(i,k) = t, head bit of R(k,j) = h, and tail bit of R(k,j) != head bit of R(i,k)
// (MinPlusSR::multiply, include/TransitiveReduction.hpp:88-104: the walk must leave k by the end it did not enter).  R(k,j) is the
// transposed image of the stored (j,k): its direction is that entry's directionT, its suffix that entry's suffixT.
constexpr int TR_THREADS = 256, TR_CAP = 2048;
__global__ __launch_bounds__(TR_THREADS) void k_tr_mark(const uint32_t *ptr, const SymEntry *sym, const uint32_t *src, uint32_t *mark, int fuzz, uint32_t M,
                                                        unsigned long long *ctr)
{
    __shared__ uint32_t lcol[TR_CAP];
    __shared__ int32_t lsfx[TR_CAP];
    __shared__ uint8_t ldir[TR_CAP];
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long products = 0, marked = 0;
    for (uint32_t i = blockIdx.x; i < M; i += gridDim.x) {
        const uint32_t rs = ptr[i], d = ptr[i + 1] - rs;
        if (threadIdx.x == 0) products += (unsigned long long)d * d;    // sum over reads of degree^2 = the products of R (x) R
        if (d < 2) continue;                                        // a two-edge path through k needs R(i,k) next to R(i,j)
        const bool staged = d <= (uint32_t)TR_CAP;
        __syncthreads();                                            // the previous row's searches are over
        if (staged)
            for (uint32_t t = threadIdx.x; t < d; t += TR_THREADS) { const SymEntry a = sym[rs + t]; lcol[t] = a.col; lsfx[t] = a.suffix; ldir[t] = (uint8_t)(a.meta & 7u); }
        __syncthreads();
        for (uint32_t x = w; x < d; x += TR_THREADS / 64) {
            const SymEntry eij = sym[rs + x];
            const uint32_t dij = eij.meta & 7u;
            const uint32_t js = ptr[eij.col], dj = ptr[eij.col + 1] - js;
            if (dij == 7u) continue;                                // GreaterThanSR: no direction, never transitive
            const uint32_t want_t1 = (dij >> 1) & 1u, want_h2 = dij & 1u;
            int best = 0x7fffffff;
            for (uint32_t y = lane; y < dj; y += 64) {
                const SymEntry b = sym[js + y];
                const uint32_t bdir = (b.meta >> 3) & 7u;
                if (bdir == 7u || (bdir & 1u) != want_h2) continue;
                const uint32_t k = b.col;
                uint32_t lo = 0, hi = d;                            // first position with column >= k
                if (staged) { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lcol[mid] < k) lo = mid + 1; else hi = mid; } }
                else { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sym[rs + mid].col < k) lo = mid + 1; else hi = mid; } }
                if (lo >= d) continue;
                uint32_t acol, adir; int32_t asfx;
                if (staged) { acol = lcol[lo]; adir = ldir[lo]; asfx = lsfx[lo]; }
                else { const SymEntry a = sym[rs + lo]; acol = a.col; adir = a.meta & 7u; asfx = a.suffix; }
                if (acol != k || adir == 7u) continue;
                if (((adir >> 1) & 1u) != want_t1 || ((bdir >> 1) & 1u) == (adir & 1u)) continue;
                const int val = asfx + b.suffixT;
                best = val < best ? val : best;
            }
            best = wave_min_i32(best);
            if (lane == 0 && best != 0x7fffffff && eij.suffix + fuzz >= best) { atomicOr(&mark[src[rs + x]], 1u << ((eij.meta >> 6) & 1u)); ++marked; }
        }
    }
    if (lane == 0) { if (products) atomicAdd(&ctr[4], products); if (marked) atomicAdd(&ctr[5], marked); }
}

// S = R without the marked entries (either orientation marked removes both, I += I^T) and without entries that have no direction
// (InvalidSRing, include/TransitiveReduction.hpp:20-23)
__global__ void k_tr_select(const SymEntry *sym, const uint32_t *src, const uint32_t *mark, int64_t n2, uint32_t *sel, unsigned long long *ctr)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool removed = false;
    if (z < n2) {
        removed = mark[src[z]] != 0;
        sel[z] = (!removed && ((sym[z].meta >> 3) & 7u) != 7u) ? 1u : 0u;      // the entry written for z is the mirror image (k_tr_scatter): its direction is this one's directionT
    } else if (z == n2) sel[z] = 0u;
    const unsigned long long b = __ballot(removed);
    if (b && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(b)) atomicAdd(&ctr[6], (unsigned long long)__builtin_popcountll(b));
}

// The CSR walk (row r, columns c ascending) is the reference's DCSC walk of the mirror entries: entry number z of the output is
// S(c, r), the transposed image of the stored (r, c).
__global__ void k_tr_scatter(const uint64_t *keys, const uint64_t *kv, const uint32_t *sel, const uint32_t *pos, int64_t n2, int mb, const elba_overlap_t *vals,
                             int64_t *orow, int64_t *ocol, elba_overlap_t *oval)
{
    const int64_t z = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= n2 || !sel[z]) return;
    const uint64_t key = keys[z], v = kv[z];
    const uint32_t r = (uint32_t)(key >> mb), c = (uint32_t)(key & ((1ull << mb) - 1));
    elba_overlap_t o = vals[v >> 1];
    if (!(v & 1)) {                                                 // stored (r,c) is the original entry: S(c,r) is its transpose
        elba_overlap_t t = o;
        t.begQ = o.begT; t.begT = o.begQ; t.endQ = o.endT; t.endT = o.endQ;
        t.suffix = o.suffixT; t.suffixT = o.suffix;
        t.direction = o.directionT; t.directionT = o.direction;
        t.containedQ = o.containedT; t.containedT = o.containedQ;
        o = t;
    }
    const uint32_t at = pos[z];
    orow[at] = (int64_t)c; ocol[at] = (int64_t)r; oval[at] = o;
}

}  // namespace

void stage_set_overlaps(Ctx &c, int64_t nreads, const int64_t *rows, const int64_t *cols, const elba_overlap_t *vals, int64_t n)
{
    ELBA_REQUIRE(nreads >= 0 && n >= 0 && (n == 0 || (rows && cols && vals)), ELBA_ERR_INVALID_ARG, "set_overlaps: null array");
    ELBA_REQUIRE(nreads < 0x7fffffff && n < 0x7fffffff, ELBA_ERR_UNSUPPORTED, "set_overlaps: more than 2^31 reads or pairs");
    for (int64_t a = 0; a < n; ++a) {
        ELBA_REQUIRE(rows[a] >= 0 && rows[a] < cols[a] && cols[a] < nreads, ELBA_ERR_INVALID_ARG, "set_overlaps: need 0 <= row < col < nreads");
        if (a) ELBA_REQUIRE(rows[a - 1] < rows[a] || (rows[a - 1] == rows[a] && cols[a - 1] < cols[a]), ELBA_ERR_INVALID_ARG, "set_overlaps: pairs must be strictly ascending in (row, col)");
    }
    hipStream_t s = c.stream;
    c.have_edges = false; c.have_S = false;
    c.tr_in_rows.reserve((size_t)(n + 1) * 8); c.tr_in_cols.reserve((size_t)(n + 1) * 8); c.tr_in_vals.reserve((size_t)(n + 1) * sizeof(elba_overlap_t));
    if (n) {
        ELBA_HIP(hipMemcpyAsync(c.tr_in_rows.p, rows, (size_t)n * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.tr_in_cols.p, cols, (size_t)n * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.tr_in_vals.p, vals, (size_t)n * sizeof(elba_overlap_t), hipMemcpyHostToDevice, s));
    }
    ELBA_HIP(hipStreamSynchronize(s));
    c.tr_in_M = nreads; c.tr_in_n = n; c.have_edges = true;
}

void stage_transitive_reduction(Ctx &c, double bad_read_cutoff, int fuzz)
{
    TrParams p{};
    int64_t M, n;
    if (c.have_edges) {
        M = c.tr_in_M; n = c.tr_in_n; c.tr_id_base = 0;
        p.rows = c.tr_in_rows.as<int64_t>(); p.cols = c.tr_in_cols.as<int64_t>(); p.vals = c.tr_in_vals.as<elba_overlap_t>();
    } else {
        ELBA_REQUIRE(c.have_aln, ELBA_ERR_STATE, "transitive_reduction: no overlaps (call elba_align_seeds or elba_set_overlaps)");
        ELBA_REQUIRE(c.row_hi < 0 || (c.row_lo == 0 && c.row_hi == c.M), ELBA_ERR_STATE,
                     "transitive_reduction: this context aligned a row shard; gather the ranks' overlaps and load them with elba_set_overlaps");
        M = c.M; n = c.naln; c.tr_id_base = c.first_global_id_rows();
        p.rows = c.aln_rows.as<int64_t>(); p.cols = c.aln_cols.as<int64_t>(); p.vals = c.aln_out.as<elba_overlap_t>();
    }
    ELBA_REQUIRE(fuzz >= 0, ELBA_ERR_INVALID_ARG, "transitive_reduction: negative fuzz");
    ELBA_REQUIRE(M < 0x7fffffff && n < 0x7fffffff, ELBA_ERR_UNSUPPORTED, "transitive_reduction: more than 2^31 reads or pairs");
    hipStream_t s = c.stream;
    c.have_S = false;
    elba_string_stats st{};
    st.nreads = M; st.nedges = n;
    int mb = 1;
    while ((1ll << mb) < M + 1) ++mb;
    p.n = n; p.M = (uint32_t)M; p.mb = mb; p.cutoff = bad_read_cutoff; p.fuzz = fuzz;
    c.tr_deg.reserve((size_t)(M + 1) * 4); c.tr_pas.reserve((size_t)(M + 1) * 4); c.tr_flags.reserve((size_t)M + 4); c.tr_ctr.reserve(64);
    c.tr_k0.reserve((size_t)(2 * n + 2) * 8); c.tr_v0.reserve((size_t)(2 * n + 2) * 8); c.tr_k1.reserve((size_t)(2 * n + 2) * 8); c.tr_v1.reserve((size_t)(2 * n + 2) * 8);
    c.tr_mark.reserve((size_t)(n + 1) * 4);
    p.deg = c.tr_deg.as<uint32_t>(); p.pas = c.tr_pas.as<uint32_t>(); p.flags = c.tr_flags.as<uint8_t>();
    p.keys = c.tr_k0.as<uint64_t>(); p.kv = c.tr_v0.as<uint64_t>(); p.ctr = c.tr_ctr.as<unsigned long long>();
    c.t_total.start(s);
    ELBA_HIP(hipMemsetAsync(c.tr_deg.p, 0, (size_t)(M + 1) * 4, s));
    ELBA_HIP(hipMemsetAsync(c.tr_pas.p, 0, (size_t)(M + 1) * 4, s));
    ELBA_HIP(hipMemsetAsync(c.tr_flags.p, 0, (size_t)M + 4, s));
    ELBA_HIP(hipMemsetAsync(c.tr_ctr.p, 0, 64, s));
    ELBA_HIP(hipMemsetAsync(c.tr_mark.p, 0, (size_t)(n + 1) * 4, s));
    const unsigned nbn = (unsigned)((n + 255) / 256), nbM = (unsigned)((M + 255) / 256);
    if (n > 0) hipLaunchKernelGGL(k_tr_degrees, dim3(nbn), dim3(256), 0, s, p);
    if (M > 0) hipLaunchKernelGGL(k_tr_bad, dim3(nbM), dim3(256), 0, s, p);
    uint32_t *cont = p.deg;                                         // the degree counters are free again: reuse them as the contained marks
    if (M > 0) ELBA_HIP(hipMemsetAsync(c.tr_deg.p, 0, (size_t)(M + 1) * 4, s));
    if (n > 0) hipLaunchKernelGGL(k_tr_contained, dim3(nbn), dim3(256), 0, s, p, cont);
    if (M > 0) hipLaunchKernelGGL(k_tr_merge_flags, dim3(nbM), dim3(256), 0, s, p, cont);
    if (n > 0) hipLaunchKernelGGL(k_tr_emit, dim3(nbn), dim3(256), 0, s, p);
    ELBA_HIP(hipGetLastError());
    unsigned long long h[8] = {0};
    ELBA_HIP(hipMemcpyAsync(h, c.tr_ctr.p, 64, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    const int64_t kept = (int64_t)h[0], n2 = 2 * kept;
    st.bad_reads = (int64_t)h[1]; st.contained_reads = (int64_t)h[2]; st.edges_passed = (int64_t)h[3]; st.edges_kept = kept;
    c.tr_ptr.reserve((size_t)(M + 2) * 4); c.tr_sym.reserve((size_t)(n2 + 1) * sizeof(SymEntry)); c.tr_src.reserve((size_t)(n2 + 1) * 4);
    c.tr_sel.reserve((size_t)(n2 + 2) * 8);
    c.tr_out_rows.reserve((size_t)(n2 + 1) * 8); c.tr_out_cols.reserve((size_t)(n2 + 1) * 8); c.tr_out_vals.reserve((size_t)(n2 + 1) * sizeof(elba_overlap_t));
    int64_t nnz = 0;
    float ms_mark = 0.f;
    if (n2 > 0) {
        const int which = radix_sort_pairs(s, c.tr_k0.as<uint64_t>(), c.tr_v0.as<uint64_t>(), c.tr_k1.as<uint64_t>(), c.tr_v1.as<uint64_t>(), n2, 0, 2 * mb, c.ws_sort);
        const uint64_t *keys = which ? c.tr_k1.as<uint64_t>() : c.tr_k0.as<uint64_t>(), *kv = which ? c.tr_v1.as<uint64_t>() : c.tr_v0.as<uint64_t>();
        group_offsets_u32(s, keys, mb, n2, c.tr_ptr.as<uint32_t>(), M);
        const unsigned nb2 = (unsigned)((n2 + 255) / 256);
        hipLaunchKernelGGL(k_tr_gather, dim3(nb2), dim3(256), 0, s, keys, kv, n2, mb, p.vals, c.tr_sym.as<SymEntry>(), c.tr_src.as<uint32_t>());
        c.t_a.start(s);
        const unsigned grid = (unsigned)(M < (int64_t)c.num_cus * 32 ? M : (int64_t)c.num_cus * 32);
        hipLaunchKernelGGL(k_tr_mark, dim3(grid), dim3(TR_THREADS), 0, s, c.tr_ptr.as<uint32_t>(), c.tr_sym.as<SymEntry>(), c.tr_src.as<uint32_t>(), c.tr_mark.as<uint32_t>(), fuzz,
                           (uint32_t)M, p.ctr);
        c.t_a.stop(s);
        uint32_t *sel = c.tr_sel.as<uint32_t>(), *pos = sel + (n2 + 2);
        hipLaunchKernelGGL(k_tr_select, dim3((unsigned)((n2 + 1 + 255) / 256)), dim3(256), 0, s, c.tr_sym.as<SymEntry>(), c.tr_src.as<uint32_t>(), c.tr_mark.as<uint32_t>(), n2, sel, p.ctr);
        exclusive_scan_u32(s, sel, pos, n2 + 1, c.ws_scan);
        hipLaunchKernelGGL(k_tr_scatter, dim3(nb2), dim3(256), 0, s, keys, kv, sel, pos, n2, mb, p.vals, c.tr_out_rows.as<int64_t>(), c.tr_out_cols.as<int64_t>(),
                           c.tr_out_vals.as<elba_overlap_t>());
        ELBA_HIP(hipGetLastError());
        c.t_total.stop(s);
        uint32_t total = 0;
        ELBA_HIP(hipMemcpyAsync(&total, pos + n2, 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipMemcpyAsync(h, c.tr_ctr.p, 64, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipStreamSynchronize(s));
        nnz = (int64_t)total;
        ms_mark = c.t_a.ms();
    } else {
        c.t_total.stop(s);
        ELBA_HIP(hipStreamSynchronize(s));
    }
    st.products = (int64_t)h[4]; st.marked = (int64_t)h[5]; st.removed = (int64_t)h[6]; st.nnz = nnz;
    st.iterations = st.removed > 0 ? 2 : 1;
    st.ms_total = c.t_total.ms(); st.ms_minplus = ms_mark;
    c.tr_M = M; c.tr_nnz = nnz; c.sstats = st; c.have_S = true;
}

}  // namespace elba
