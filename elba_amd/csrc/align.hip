// align.hip — x-drop seed-and-extend of every candidate pair of B, on the GPU (the step right after the path).
//
// Replaces PairwiseAlignment's loop (src/PairwiseAlignment.cpp:28-95) on one rank: every stored B(i,j) with i < j is aligned from
// seeds[0] by xdrop_aligner (src/XDropAligner.cpp:224-282), classified (classify_alignment, :7-44) and turned into the fields of
// Overlap that extend_overlap fills (src/Overlap.cpp:24-73).  Results are bit-identical to the reference's: the antidiagonal recurrence,
// its undef sentinel, the band trimming rules and the "last column of the antidiagonal that beats the best of the antidiagonals before
// it" choice of the extension's end are those of _extend_seed_one_direction (src/XDropAligner.cpp:46-208).
//
// Mapping to CDNA4: one WAVEFRONT per extension (left and right extensions of a pair are independent tasks), one LANE per column of
// the current antidiagonal.  The three antidiagonals of the recurrence live in registers (two carried, one computed); the neighbour
// column arrives through a DPP wave shift (one cycle, no LDS); the band's bounds, the best score and the antidiagonal number are
// wave-uniform scalars; the band trimming loops of the reference become two ballots and a count-leading/trailing-zeros each.  The
// bases of both reads are unpacked from the 2-bit DnaBuffer layout into two 256-entry LDS rings per wavefront, 64 bases per refill, so
// the recurrence itself touches no global memory.  Lane l holds column cbase + l; when the band reaches lane 63 the window slides to
// the band's lower edge.  Register tiers hold 1, 2, 4 or 8 columns per lane (bands up to 64 / 128 / 256 / 512 columns): an extension whose
// band outgrows its tier is redone from the seed on the next one, and beyond 511 columns by a strided kernel that keeps the
// antidiagonals in HBM — capacity is a performance tier, never a correctness limit.
//
// Integer work throughout (scores are int32, like the reference's); no MFMA.
#include "common.hpp"

namespace elba {

namespace {

struct AlnTask {               // one candidate pair, prepared by k_aln_prepare
    uint32_t i, j;             // rows of B: query read i, target read j (i < j)
    int32_t begQ, endQ;        // seed in the query
    int32_t begT, endT;        // seed in the target's oriented coordinates (reverse-complemented when rc)
    int32_t valid, rc;         // xdrop_aligner's prologue: seed accepted / orientation
    int32_t numshared;         // of B(i,j): only a hint for the starting tier (few shared k-mers = probably unrelated reads = wide band)
};
struct AlnExt { int32_t score, col, row, overflow; };       // best_ext_score / best_ext_col / best_ext_row of one direction

struct AlnParams {
    const uint8_t *packed; const uint64_t *byte_off; const uint32_t *len;
    const int64_t *b_rowptr; const uint32_t *b_col; const elba_seed_t *b_val;
    uint32_t M, row_lo, row_hi;
    uint32_t share;            // 0: one rank, every stored B(i,j) with i < j.  1: rows [row_lo, row_hi) of a row-sharded B, reads replicated: a
                               // pair {i,j} is stored on both of its rows' ranks; the rank of the smaller row takes it when i + j is even, the
                               // rank of the larger row when it is odd — every pair exactly once, the shares balanced without any exchange
    int32_t k, mat, mis, gap, dropoff;
    int32_t wide_hint;         // pairs with numshared <= wide_hint skip the 64-column tier (performance only: any tier gives the same result)
    int32_t long_hint;         // ... and the 128-column tier too when the shorter side of the extension has at least this many bases
    uint32_t *cnt; const int64_t *taskptr;
    AlnTask *tasks; AlnExt *ext; int64_t ntasks;
    int *scratch; unsigned long long scratch_stride;  // strided kernel: three antidiagonals per wavefront
    unsigned long long *cells;      // DP cells computed (statistics)
    int64_t *out_rows, *out_cols; elba_overlap_t *out;
};

__device__ __forceinline__ int base_at(const uint8_t *mem, uint32_t i) { return (mem[i >> 2] >> (6 - 2 * (i & 3))) & 3; }    // src/DnaSeq.cpp:48-54

// ---- tasks: the strict upper triangle of B in CSR order (src/PairwiseAlignment.cpp:52 on one rank) --------------------------------
__global__ void k_aln_count(AlnParams p)
{
    const uint32_t i = p.row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.row_hi) return;
    uint32_t n = 0;
    for (int64_t e = p.b_rowptr[i]; e < p.b_rowptr[i + 1]; ++e) {
        const uint32_t j = p.b_col[e];
        n += !p.share ? (j > i ? 1u : 0u) : ((j != i && (((i + j) & 1u) == (j > i ? 0u : 1u))) ? 1u : 0u);
    }
    p.cnt[i - p.row_lo] = n;
}

// xdrop_aligner's prologue (src/XDropAligner.cpp:228-256): bounds, the (0,0) rejection, orientation from the middle base, seed check
__global__ void k_aln_prepare(AlnParams p)
{
    const uint32_t i = p.row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.row_hi) return;
    int64_t at = p.taskptr[i - p.row_lo];
    for (int64_t e = p.b_rowptr[i]; e < p.b_rowptr[i + 1]; ++e) {
        const uint32_t jj = p.b_col[e];
        if (!p.share ? jj <= i : (jj == i || (((i + jj) & 1u) != (jj > i ? 0u : 1u)))) continue;
        // the pair is always aligned as (query = smaller read id, target = larger): the entry of the larger read's row is B(i,j) with the
        // two positions of each seed exchanged, so its seeds[0] is swapped back
        const bool upper = jj > i;
        const uint32_t qi = upper ? i : jj, tj = upper ? jj : i;
        const uint8_t *q = p.packed + p.byte_off[qi];
        const int lenQ = (int)p.len[qi];
        const uint8_t *t = p.packed + p.byte_off[tj];
        const int lenT = (int)p.len[tj];
        const int begQ = (int)(upper ? p.b_val[e].q0 : p.b_val[e].t0), begT = (int)(upper ? p.b_val[e].t0 : p.b_val[e].q0), k = p.k;
        AlnTask tk{};
        tk.i = qi; tk.j = tj; tk.numshared = p.b_val[e].numshared;
        bool ok = !(begQ < 0 || begQ + k > lenQ) && !(begT < 0 || begT + k > lenT) && !(begQ == 0 && begT == 0);
        if (ok) {
            const bool rc = base_at(q, (uint32_t)(begQ + (k >> 1))) != base_at(t, (uint32_t)(begT + (k >> 1)));
            for (int x = 0; x < k && ok; ++x) {
                // rc: revcomp_at(lenT - begT - k + x) = 3 - T[lenT - 1 - (lenT - begT - k + x)] = 3 - T[begT + k - 1 - x]   (include/DnaSeq.hpp:119)
                const int tb = rc ? 3 - base_at(t, (uint32_t)(begT + k - 1 - x)) : base_at(t, (uint32_t)(begT + x));
                ok = base_at(q, (uint32_t)(begQ + x)) == tb;
            }
            tk.rc = rc ? 1 : 0;
            tk.begQ = begQ; tk.endQ = begQ + k;
            tk.begT = rc ? lenT - begT - k : begT; tk.endT = tk.begT + k;
        }
        tk.valid = ok ? 1 : 0;
        p.tasks[at++] = tk;
    }
}

// ---- one direction of one pair -------------------------------------------------------------------------------------------------
struct ExtGeom {                 // wave-uniform description of an extension
    const uint8_t *q, *t;
    int lenT, rc, left;
    int cols, rows;              // lenQ_ext + 1, lenT_ext + 1
    int offQ, offT;              // right: endQ / endT;  left: begQ / begT
};
// logical column c in [1, cols) -> base of the query; logical row r in [1, rows) -> base of the (oriented) target   (src/XDropAligner.cpp:113-117)
__device__ __forceinline__ int q_base(const ExtGeom &g, int c) { return base_at(g.q, (uint32_t)(g.left ? g.offQ - c : g.offQ + c - 1)); }
__device__ __forceinline__ int t_base(const ExtGeom &g, int r)
{
    const int posT = g.left ? g.offT - r : g.offT + r - 1;
    return g.rc ? 3 - base_at(g.t, (uint32_t)(g.lenT - 1 - posT)) : base_at(g.t, (uint32_t)posT);
}

__device__ __forceinline__ bool ext_geometry(const AlnParams &p, const AlnTask &tk, int left, ExtGeom &g)
{
    g.q = p.packed + p.byte_off[tk.i]; g.t = p.packed + p.byte_off[tk.j];
    const int lenQ = (int)p.len[tk.i];
    g.lenT = (int)p.len[tk.j]; g.rc = tk.rc; g.left = left;
    const int lenQ_ext = left ? tk.begQ : lenQ - tk.endQ, lenT_ext = left ? tk.begT : g.lenT - tk.endT;
    g.cols = lenQ_ext + 1; g.rows = lenT_ext + 1;
    g.offQ = left ? tk.begQ : tk.endQ; g.offT = left ? tk.begT : tk.endT;
    return !(g.rows == 1 || g.cols == 1);
}

__device__ __forceinline__ int wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
// maximum over the wavefront, in registers: four DPP steps make every lane of a 16-lane row hold its row's maximum, two row broadcasts
// carry it into the last row; the result is read from lane 63.  (A shuffle-based butterfly costs six LDS round trips per antidiagonal.)
__device__ __forceinline__ int wave_max_i32(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xf, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false); v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false); v = t > v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}

constexpr int ALN_WAVES = 4;     // independent wavefronts per workgroup

// The register kernel: a band of up to 64*KC - 1 columns.  Lane l holds the KC consecutive columns cbase + l*KC ... + KC - 1 of the
// antidiagonals n-2 and n-1.  Every scalar below is wave-uniform.  Work comes from a queue: all 2*ntasks extensions (in_list == nullptr)
// or the extensions a narrower instantiation gave up on.
template <int KC>
__global__ __launch_bounds__(64 * ALN_WAVES) void k_xdrop_wave(AlnParams p, const uint32_t *in_list, const unsigned int *in_count, unsigned int *next,
                                                               uint32_t *out_list, unsigned int *out_count)
{
    constexpr int W = 64 * KC;              // window of columns the wavefront holds
    constexpr int RING = 2 * W < 256 ? 256 : 2 * W;   // bases per LDS ring (power of two): the band's span (<= W) + one refill (64) fit with room to spare
    __shared__ uint8_t ring[ALN_WAVES][2][RING];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint8_t *rq = ring[w][0], *rt = ring[w][1];
    unsigned long long cells = 0;
    const unsigned int nwork = in_list ? *in_count : (unsigned int)(2 * p.ntasks);
    for (;;) {
        unsigned int wk = 0;
        if (lane == 0) wk = atomicAdd(next, 1u);
        wk = (unsigned int)__builtin_amdgcn_readfirstlane((int)wk);
        if (wk >= nwork) break;
        const unsigned int tk2 = in_list ? in_list[wk] : wk;
        const AlnTask tk = p.tasks[tk2 >> 1];
        const int left = (int)(tk2 & 1u);
        AlnExt res{0, 0, 0, 0};
        ExtGeom g;
        if (!tk.valid || !ext_geometry(p, tk, left, g)) { if (lane == 0) p.ext[tk2] = res; continue; }
        // probably a wide band (few shared k-mers: unrelated reads, which align in the linear regime under +1/-1/-1): straight to the next
        // tier; and a LONG such extension (the band of a linear-regime alignment keeps growing with its length) skips the 128-column tier too
        if ((KC == 1 && in_list == nullptr && tk.numshared <= p.wide_hint) ||
            (KC == 2 && tk.numshared <= p.wide_hint && (g.cols < g.rows ? g.cols : g.rows) >= p.long_hint)) {
            if (lane == 0) { const unsigned int at = atomicAdd(out_count, 1u); out_list[at] = tk2; }
            continue;
        }
        const int cols = g.cols, rows = g.rows;
        const int int_min = (int)0x80000000;
        const int len2 = 2 * (cols > rows ? cols : rows);
        const int min_err = int_min / len2;
        const int gap = p.gap > min_err ? p.gap : min_err, mis = p.mis > min_err ? p.mis : min_err, mat = p.mat, dropoff = p.dropoff;
        const int undef = int_min - gap - mis;
        // antidiagonal 0 = {col 0: 0}; antidiagonal 1 = {col 0, col 1: gap, or undef when one gap already drops off}   (:70-77)
        int cbase = 0;
        const int g1 = (-gap > dropoff) ? undef : gap;
        int A1[KC], A2[KC], A3[KC];                      // antidiagonals n-2, n-1, n at columns cbase + lane*KC + k
#pragma unroll
        for (int k = 0; k < KC; ++k) { const int c = lane * KC + k; A1[k] = c == 0 ? 0 : undef; A2[k] = c <= 1 ? g1 : undef; }
        int min_col = 1, max_col = 2, hi2 = 1;           // hi2: last stored column of antidiagonal n-1
        int best = 0, n = 1;
        int best_col = 0, best_row = 0, best_score = 0;
        int qfill = 0, tfill = 0;                        // logical columns / rows unpacked into the rings so far
        bool overflow = false;
        unsigned long long mycells = 0;                  // (an extension that leaves for a wider kernel is counted there)
        // The bases a lane needs: its KC query bases change only when the window slides or a refill reaches its columns; its target
        // bases move one row per antidiagonal and are read one antidiagonal AHEAD (the ring is filled one row ahead as well), so the
        // recurrence never waits for LDS.
        int qb[KC], tb[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) { qb[k] = 0; tb[k] = 0; }
        bool reload = true;
        while (min_col < max_col) {
            ++n;
            const int off3 = min_col - 1, top_max = max_col;      // this antidiagonal is stored for columns [off3, top_max]   (:93-96)
            if (top_max > cbase + W - 1) {               // slide the window down to the band's lower edge (whole lanes)
                const int sl = (off3 - cbase) / KC;
                if (sl <= 0) { overflow = true; break; }
                const int src = lane + sl;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const int a1 = __shfl(A1[k], src & 63, 64), a2 = __shfl(A2[k], src & 63, 64);
                    A1[k] = src < 64 ? a1 : undef; A2[k] = src < 64 ? a2 : undef;
                }
                cbase += sl * KC;
                if (top_max > cbase + W - 1) { overflow = true; break; }
                reload = true;
            }
            // bases: columns up to top_max (one ahead of this antidiagonal's last computed column), rows up to n + 1 - min_col
            while (qfill < top_max) {
                const int c = qfill + 1 + lane;
                if (c < cols) rq[c & (RING - 1)] = (uint8_t)q_base(g, c);
                qfill += 64; reload = true;
            }
            while (tfill < n + 1 - min_col) {
                const int r = tfill + 1 + lane;
                if (r < rows) rt[r & (RING - 1)] = (uint8_t)t_base(g, r);
                tfill += 64; reload = true;
            }
            if (reload) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < KC; ++k) { const int c = cbase + lane * KC + k; qb[k] = rq[c & (RING - 1)]; tb[k] = rt[(n - c) & (RING - 1)]; }
                reload = false;
            }
            mycells += (unsigned long long)(top_max - min_col);
            const int c0 = cbase + lane * KC;
            const bool gi = n * gap > best - dropoff;          // the border cells may still be reached by gaps alone (:98-102); |n * gap| < 2^31 by the clamp of gap above
            const int up0 = wave_shr1(A2[KC - 1], undef), dg0 = wave_shr1(A1[KC - 1], undef);     // column c0 - 1 of antidiagonals n-1, n-2
            if constexpr (KC == 1) {
                // One column per lane: the bookkeeping is a handful of lane masks.  (The kernel is bound by instruction issue — scalar
                // and vector alike, ~26 of 64 lanes carry a cell — so this common instantiation is written out instead of going
                // through the per-lane first / last / beat indices of the general form below.)
                const int c = c0;
                const bool inr = (unsigned)(c - min_col) < (unsigned)(top_max - min_col);
                int temp = (up0 > A2[0] ? up0 : A2[0]) + gap;
                const int t2 = dg0 + (qb[0] == tb[0] ? mat : mis);
                temp = t2 > temp ? t2 : temp;
                const bool keep = inr && temp >= best - dropoff;
                int v = keep ? temp : undef;
                if (gi) {                                   // only while gaps alone stay within the drop-off: the first few antidiagonals
                    if (c == off3) v = off3 == 0 ? n * gap : undef;
                    if (c == top_max) v = n == top_max ? n * gap : undef;
                }
                A3[0] = v;
                const unsigned long long beat = __ballot(inr && temp > best);
                if (beat != 0) {
                    const int l = 63 - __builtin_clzll(beat);
                    best_col = cbase + l; best_row = n - best_col;
                    best_score = __builtin_amdgcn_readlane(v, l);
                    const int m = wave_max_i32(keep ? temp : int_min);
                    best = m > best ? m : best;
                }
                // trimming (:147-156).  First column >= min_col that is not undef on this antidiagonal or, one column to the left, on the
                // previous one; the reference's size guards stop the scan at top_max + 1 at the latest (hi2 >= top_max - 1 always).
                const unsigned long long live3 = __ballot(v != undef);
                const unsigned long long stop = (live3 | __ballot(up0 != undef)) & (~0ull << (min_col - cbase));
                int nmin = stop ? cbase + (int)__builtin_ctzll(stop) : top_max + 1;
                min_col = nmin < top_max + 1 ? nmin : top_max + 1;
                // one past the last column in [off3, top_max) that is not undef on this or the previous antidiagonal
                const unsigned long long span = (~0ull << (off3 - cbase)) & ~(~0ull << (top_max - cbase));       // top_max - cbase <= 63 (window check above)
                const unsigned long long alive = (live3 | __ballot(A2[0] != undef)) & span;
                max_col = alive ? cbase + 64 - (int)__builtin_clzll(alive) : off3;
            } else {
            int lane_max = int_min, lane_beat = -1, lane_first = KC, lane_last = -1, beat_score = 0;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int c = c0 + k;
                const bool inr = c >= min_col && c < top_max;
                const int up = k == 0 ? up0 : A2[k - 1], dg = k == 0 ? dg0 : A1[k - 1];
                int temp = (up > A2[k] ? up : A2[k]) + gap;
                const int t2 = dg + (qb[k] == tb[k] ? mat : mis);
                temp = t2 > temp ? t2 : temp;
                const bool keep = temp >= best - dropoff;
                int v = (inr && keep) ? temp : undef;
                if (c == off3) v = (gi && off3 == 0) ? n * gap : undef;
                if (c == top_max) v = (gi && n == top_max) ? n * gap : undef;
                A3[k] = v;
                if (inr && keep && temp > lane_max) lane_max = temp;
                if (inr && temp > best) { lane_beat = k; beat_score = v; }                 // ascending k: the last one stays
                // band trimming (:147-156), per column: dead = undef on this antidiagonal and (left neighbour) on the previous one
                const bool dead = (c <= top_max) && v == undef && (c - 1 <= hi2) && up == undef;
                if (c >= min_col && !dead && lane_first == KC) lane_first = k;
                if (c >= off3 && c < top_max && !(v == undef && A2[k] == undef)) lane_last = k;
            }
            // the extension ends at the LAST column of this antidiagonal that beats the best of the antidiagonals before it (:136-142)
            const unsigned long long beat = __ballot(lane_beat >= 0);
            if (beat != 0) {
                const int l = 63 - __builtin_clzll(beat);
                best_col = cbase + l * KC + __builtin_amdgcn_readlane(lane_beat, l); best_row = n - best_col;
                best_score = __builtin_amdgcn_readlane(beat_score, l);
                const int m = wave_max_i32(lane_max);
                best = m > best ? m : best;
            }
            {   // the first column that is alive on this or the previous antidiagonal ...
                const unsigned long long stop = __ballot(lane_first < KC);
                if (stop) { const int l = (int)__builtin_ctzll(stop); min_col = cbase + l * KC + __builtin_amdgcn_readlane(lane_first, l); }
                else min_col = cbase + W;
            }
            {   // ... and one past the last
                const unsigned long long alive = __ballot(lane_last >= 0);
                if (alive) { const int l = 63 - __builtin_clzll(alive); max_col = cbase + l * KC + __builtin_amdgcn_readlane(lane_last, l) + 1; }
                else max_col = off3;
            }
            }
            ++max_col;
            if (min_col < n + 2 - rows) min_col = n + 2 - rows;
            if (max_col > cols) max_col = cols;
            hi2 = top_max;
#pragma unroll
            for (int k = 0; k < KC; ++k) { A1[k] = A2[k]; A2[k] = A3[k]; tb[k] = rt[(n + 1 - (c0 + k)) & (RING - 1)]; }     // next antidiagonal's target bases
        }
        if (!overflow) cells += mycells;
        else if (lane == 0) atomicAdd(p.cells + 1, mycells);          // abandoned work (diagnostic: ELBA_TRACE)
        res.score = best_score; res.col = best_col; res.row = best_row; res.overflow = overflow ? 1 : 0;
        if (lane == 0) {
            p.ext[tk2] = res;
            if (overflow) { const unsigned int at = atomicAdd(out_count, 1u); out_list[at] = tk2; }
        }
    }
    if (lane == 0 && cells) atomicAdd(p.cells, cells);
}

// The strided kernel: any band width.  One wavefront per extension that left the fast kernel; the three antidiagonals live in HBM
// (indexed by absolute column, read and written with agent-scope accesses: lanes exchange cells through L2), lanes stride over the
// band.  Same recurrence, same trimming, statement for statement (src/XDropAligner.cpp:79-160).
__device__ __forceinline__ int ld_cell(const int *a, int c) { return __hip_atomic_load(&a[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_cell(int *a, int c, int v) { __hip_atomic_store(&a[c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void k_xdrop_strided(AlnParams p, const uint32_t *in_list, const unsigned int *in_count)
{
    const int lane = threadIdx.x & 63;
    const unsigned int nofl = *in_count;
    unsigned long long cells = 0;
    for (unsigned int it = blockIdx.x; it < nofl; it += gridDim.x) {
        const unsigned int tk2 = in_list[it];
        const AlnTask tk = p.tasks[tk2 >> 1];
        const int left = (int)(tk2 & 1u);
        ExtGeom g;
        ext_geometry(p, tk, left, g);
        const int cols = g.cols, rows = g.rows;
        const int int_min = (int)0x80000000;
        const int len2 = 2 * (cols > rows ? cols : rows);
        const int min_err = int_min / len2;
        const int gap = p.gap > min_err ? p.gap : min_err, mis = p.mis > min_err ? p.mis : min_err, mat = p.mat, dropoff = p.dropoff;
        const int undef = int_min - gap - mis;
        int *ad1 = p.scratch + (size_t)(3 * blockIdx.x + 0) * p.scratch_stride;
        int *ad2 = p.scratch + (size_t)(3 * blockIdx.x + 1) * p.scratch_stride;
        int *ad3 = p.scratch + (size_t)(3 * blockIdx.x + 2) * p.scratch_stride;
        // (cells outside an antidiagonal's stored range [off, top_max] are never read — by the reference's own index bounds — so the
        //  arrays are neither cleared nor resized when they are recycled)
        const int g1 = (-gap > dropoff) ? undef : gap;
        if (lane == 0) { st_cell(ad2, 0, 0); st_cell(ad3, 0, g1); st_cell(ad3, 1, g1); }
        __builtin_amdgcn_s_waitcnt(0);
        int min_col = 1, max_col = 2, hi2 = 0, hi3 = 1;
        int best = 0, n = 1, best_col = 0, best_row = 0, best_score = 0;
        while (min_col < max_col) {
            ++n;
            { int *tb = ad1; ad1 = ad2; ad2 = ad3; ad3 = tb; }
            hi2 = hi3;
            const int off3 = min_col - 1, top_max = max_col;
            hi3 = top_max;
            cells += (unsigned long long)(top_max - min_col);
            const bool gi = (long long)n * gap > (long long)best - dropoff;
            if (lane == 0) {
                st_cell(ad3, off3, (gi && off3 == 0) ? n * gap : undef);
                st_cell(ad3, top_max, (gi && n == top_max) ? n * gap : undef);
            }
            int my_col = -1, my_score = 0, my_max = int_min;
            for (int c = min_col + lane; c < top_max; c += 64) {
                const int up = ld_cell(ad2, c - 1), lf = ld_cell(ad2, c), dg = ld_cell(ad1, c - 1);
                int temp = (up > lf ? up : lf) + gap;
                const int t2 = dg + (q_base(g, c) == t_base(g, n - c) ? mat : mis);
                temp = t2 > temp ? t2 : temp;
                const bool keep = temp >= best - dropoff;
                st_cell(ad3, c, keep ? temp : undef);
                if (keep && temp > my_max) my_max = temp;
                if (temp > best) { my_col = c; my_score = temp; }          // ascending c per lane: the last one stays
            }
            // last column over the whole band that beat `best`: maximum of the lanes' candidates
            int bc = my_col;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(bc, d, 64); bc = o > bc ? o : bc; }
            if (bc >= 0) {
                const unsigned long long who = __ballot(my_col == bc);
                best_col = bc; best_row = n - bc;
                best_score = __builtin_amdgcn_readlane(my_score, (int)__builtin_ctzll(who));
                const int m = wave_max_i32(my_max);
                best = m > best ? m : best;
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            // trimming, as written in the reference (every lane runs the same scalar loops)
            while (min_col <= top_max && ld_cell(ad3, min_col) == undef && min_col - 1 <= hi2 && ld_cell(ad2, min_col - 1) == undef) ++min_col;
            while (max_col - off3 > 0 && ld_cell(ad3, max_col - 1) == undef && ld_cell(ad2, max_col - 1) == undef) --max_col;
            ++max_col;
            if (min_col < n + 2 - rows) min_col = n + 2 - rows;
            if (max_col > cols) max_col = cols;
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        }
        if (lane == 0) p.ext[tk2] = AlnExt{best_score, best_col, best_row, 0};
    }
    if (lane == 0 && cells) atomicAdd(p.cells, cells);
}

// ---- xdrop_aligner's epilogue (:258-281), classify_alignment (:7-44), Overlap::extend_overlap (src/Overlap.cpp:24-73) ---------------
__global__ void k_aln_combine(AlnParams p)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.ntasks) return;
    const AlnTask tk = p.tasks[a];
    const int lenQ = (int)p.len[tk.i], lenT = (int)p.len[tk.j];
    elba_overlap_t o{};
    o.direction = -1; o.directionT = -1;
    int begQ = 0, endQ = 0, begT = 0, endT = 0, score = -1, rc = 0;          // XSeed's defaults when the seed is rejected (include/XDropAligner.hpp:22)
    if (tk.valid) {
        const AlnExt l = p.ext[2 * a + 1], r = p.ext[2 * a];
        const int begQ_ext = tk.begQ - l.col, begT_ext = tk.begT - l.row;
        const int endQ_ext = tk.endQ + r.col, endT_ext = tk.endT + r.row;
        rc = tk.rc;
        score = l.score + r.score + p.mat * p.k;
        begQ = begQ_ext; endQ = endQ_ext;
        begT = rc ? lenT - endT_ext : begT_ext;
        endT = rc ? lenT - begT_ext : endT_ext;
    }
    int kind = 0;
    {
        if (score > 0) {
            const int begTr = rc ? lenT - endT : begT, endTr = rc ? lenT - begT : endT;
            const int maplen = ((endT - begT) + (endQ - begQ)) / 2;
            const int overhang = (begQ < begTr ? begQ : begTr) + ((lenQ - endQ) < (lenT - endTr) ? (lenQ - endQ) : (lenT - endTr));
            const int overlap = maplen + overhang;
            const float my_thr = (float)((1.0 - 0.1) * (0.99 * overlap));
            if (begQ <= begTr && lenQ - endQ <= lenT - endTr) kind = 1;
            else if (begQ >= begTr && lenQ - endQ >= lenT - endTr) kind = 2;
            else if ((float)score < my_thr || overlap < 500) kind = 0;
            else if (begQ > begTr) kind = 3;
            else kind = 4;
        }
    }
    o.rc = (uint8_t)rc; o.score = score; o.kind = (uint8_t)kind;
    o.begQ = begQ; o.begT = begT; o.endQ = endQ; o.endT = endT;
    const int begTr = rc ? lenT - endT : begT, endTr = rc ? lenT - begT : endT;
    if (kind != 0) {
        o.passed = 1;
        if (kind == 1) o.containedQ = 1;
        else if (kind == 2) o.containedT = 1;
        else if (kind == 3) { o.direction = rc ? 0 : 1; o.directionT = rc ? 0 : 2; o.suffix = (lenT - endTr) - (lenQ - endQ); o.suffixT = begQ - begTr; }
        else { o.direction = rc ? 3 : 2; o.directionT = rc ? 3 : 1; o.suffix = begTr - begQ; o.suffixT = (lenQ - endQ) - (lenT - endTr); }
    }
    p.out_rows[a] = (int64_t)tk.i; p.out_cols[a] = (int64_t)tk.j; p.out[a] = o;
}

}  // namespace

void stage_align_seeds(Ctx &c, int mat, int mis, int gap, int dropoff)
{
    ELBA_REQUIRE(c.have_B, ELBA_ERR_STATE, "align_seeds: needs the seed matrix on this context");
    const bool shard = c.row_hi >= 0 && !(c.row_lo == 0 && c.row_hi == c.M);        // rows of a row-sharded B (multi-GPU)
    if (shard) ELBA_REQUIRE(c.aln_all_n == c.M, ELBA_ERR_STATE, "align_seeds: a row shard of B needs every read resident (elba_dist_set_all_reads)");
    else ELBA_REQUIRE(c.have_reads && c.nreads == c.M, ELBA_ERR_UNSUPPORTED, "align_seeds: every read of B must be resident on this context");
    ELBA_REQUIRE(dropoff >= 0, ELBA_ERR_INVALID_ARG, "align_seeds: negative x-drop");
    hipStream_t s = c.stream;
    const int64_t M = c.M;
    c.have_aln = false;
    c.have_edges = false; c.have_S = false;        // fresh alignments replace a loaded edge list as the string graph's input (tr.hip)
    AlnParams p{};
    if (shard) { p.packed = c.aln_all_packed.as<uint8_t>(); p.byte_off = c.aln_all_off.as<uint64_t>(); p.len = c.aln_all_len.as<uint32_t>(); }
    else { p.packed = c.d_packed; p.byte_off = c.d_byte_off; p.len = c.d_len; }
    p.b_rowptr = c.b_rowptr.as<int64_t>(); p.b_col = c.b_col.as<uint32_t>(); p.b_val = c.b_val.as<elba_seed_t>();
    p.M = (uint32_t)M; p.row_lo = shard ? (uint32_t)c.row_lo : 0u; p.row_hi = shard ? (uint32_t)c.row_hi : (uint32_t)M; p.share = shard ? 1u : 0u;
    p.k = c.cfg.k; p.mat = mat; p.mis = mis; p.gap = gap; p.dropoff = dropoff;
    p.wide_hint = c.opt.aln_wide_hint;
    p.long_hint = c.opt.aln_long_hint;
    c.t_total.start(s);
    const int64_t nrows = (int64_t)p.row_hi - (int64_t)p.row_lo;
    c.aln_cnt.reserve((size_t)(nrows + 2) * 4); c.aln_ptr.reserve((size_t)(nrows + 2) * 8); c.aln_ctr.reserve(256);
    p.cnt = c.aln_cnt.as<uint32_t>();
    ELBA_HIP(hipMemsetAsync(c.aln_cnt.p, 0, (size_t)(nrows + 2) * 4, s));
    ELBA_HIP(hipMemsetAsync(c.aln_ctr.p, 0, 256, s));
    const unsigned nbM = (unsigned)((nrows + 255) / 256);
    if (nrows > 0) hipLaunchKernelGGL(k_aln_count, dim3(nbM), dim3(256), 0, s, p);
    exclusive_scan_u32_to_i64(s, p.cnt, c.aln_ptr.as<int64_t>(), nrows + 1, c.ws_scan);
    int64_t K = 0;
    ELBA_HIP(hipMemcpyAsync(&K, c.aln_ptr.as<int64_t>() + nrows, 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    ELBA_REQUIRE(2 * K < 0xFFFFFFF0ll, ELBA_ERR_UNSUPPORTED, "align_seeds: more than 2^31 candidate pairs");
    c.naln = K;
    c.aln_tasks.reserve((size_t)(K + 1) * sizeof(AlnTask)); c.aln_ext.reserve((size_t)(2 * K + 2) * sizeof(AlnExt)); c.aln_ofl.reserve((size_t)(2 * K + 2) * 4);
    c.aln_rows.reserve((size_t)(K + 1) * 8); c.aln_cols.reserve((size_t)(K + 1) * 8); c.aln_out.reserve((size_t)(K + 1) * sizeof(elba_overlap_t));
    p.taskptr = c.aln_ptr.as<int64_t>(); p.tasks = c.aln_tasks.as<AlnTask>(); p.ext = c.aln_ext.as<AlnExt>(); p.ntasks = K;
    unsigned int *ctr = c.aln_ctr.as<unsigned int>();       // [0..7] queue cursors per tier, [16..23] overflow counts per tier, [32] cells
    p.cells = reinterpret_cast<unsigned long long *>(ctr + 32);
    c.aln_ofl.reserve((size_t)(4 * K + 4) * 4);
    uint32_t *lists[2] = {c.aln_ofl.as<uint32_t>(), c.aln_ofl.as<uint32_t>() + 2 * K + 2};
    p.out_rows = c.aln_rows.as<int64_t>(); p.out_cols = c.aln_cols.as<int64_t>(); p.out = c.aln_out.as<elba_overlap_t>();
    // strided kernel: three antidiagonals per wavefront, as long as the longest read + 2
    uint32_t maxlen = shard ? c.aln_all_maxlen : 0;
    if (!shard) for (uint32_t l : c.h_len) maxlen = l > maxlen ? l : maxlen;
    const int sblocks = c.num_cus * 2;
    p.scratch_stride = (unsigned long long)maxlen + 8;
    elba_align_stats st{};
    st.nalignments = K;
    if (K > 0) {
        hipLaunchKernelGGL(k_aln_prepare, dim3(nbM), dim3(256), 0, s, p);
        c.t_a.start(s);
        // persistent wavefronts pulling extensions from a queue: durations range from a handful of antidiagonals to tens of thousands.
        // Tiers of 64 / 128 / 256 / 512 columns per wavefront: an extension whose band outgrows a tier is redone from its seed on the next;
        // beyond 511 columns the strided kernel takes over.  (option "aln_tiers" selects the instantiations, for A/B runs.)
        int tiers[4] = {1, 2, 4, 8}, ntiers = 4;
        if (c.opt.aln_tiers > 0) {      // decimal digits, first tier first: 1248 = all four, 24 = the 128- and 256-column tiers only
            int digits[8], nd = 0;
            for (int64_t v = c.opt.aln_tiers; v > 0 && nd < 8; v /= 10) digits[nd++] = (int)(v % 10);
            ntiers = 0;
            for (int q = nd - 1; q >= 0 && ntiers < 4; --q) if (digits[q] == 1 || digits[q] == 2 || digits[q] == 4 || digits[q] == 8) tiers[ntiers++] = digits[q];
            if (ntiers == 0) { tiers[0] = 1; ntiers = 1; }
        }
        const int64_t resident = (int64_t)c.num_cus * 8;
        unsigned int nwork = (unsigned int)(2 * K);
        const uint32_t *in_list = nullptr; const unsigned int *in_count = nullptr;
        for (int t = 0; t < ntiers && nwork > 0; ++t) {
            int64_t nb = ((int64_t)nwork + ALN_WAVES - 1) / ALN_WAVES;
            if (nb > resident) nb = resident;
            uint32_t *out_list = lists[t & 1]; unsigned int *out_count = ctr + 16 + t;
            if (tiers[t] == 1) hipLaunchKernelGGL((k_xdrop_wave<1>), dim3((unsigned)nb), dim3(64 * ALN_WAVES), 0, s, p, in_list, in_count, ctr + t, out_list, out_count);
            else if (tiers[t] == 2) hipLaunchKernelGGL((k_xdrop_wave<2>), dim3((unsigned)nb), dim3(64 * ALN_WAVES), 0, s, p, in_list, in_count, ctr + t, out_list, out_count);
            else if (tiers[t] == 4) hipLaunchKernelGGL((k_xdrop_wave<4>), dim3((unsigned)nb), dim3(64 * ALN_WAVES), 0, s, p, in_list, in_count, ctr + t, out_list, out_count);
            else hipLaunchKernelGGL((k_xdrop_wave<8>), dim3((unsigned)nb), dim3(64 * ALN_WAVES), 0, s, p, in_list, in_count, ctr + t, out_list, out_count);
            ELBA_HIP(hipMemcpyAsync(&nwork, out_count, 4, hipMemcpyDeviceToHost, s));
            ELBA_HIP(hipStreamSynchronize(s));
            if (c.opt.trace) {
                unsigned long long wasted = 0;
                ELBA_HIP(hipMemcpy(&wasted, p.cells + 1, 8, hipMemcpyDeviceToHost));
                fprintf(stderr, "[elba] x-drop tier %d columns/lane: %u extensions left, %llu cells abandoned so far\n", tiers[t], nwork, wasted);
            }
            in_list = out_list; in_count = out_count;
        }
        unsigned int nofl = nwork;
        if (nofl > 0) {
            c.aln_scratch.reserve((size_t)sblocks * 3 * p.scratch_stride * sizeof(int));
            p.scratch = c.aln_scratch.as<int>();
            hipLaunchKernelGGL(k_xdrop_strided, dim3((unsigned)(nofl < (unsigned)sblocks ? nofl : (unsigned)sblocks)), dim3(64), 0, s, p, in_list, in_count);
        }
        c.t_a.stop(s);
        hipLaunchKernelGGL(k_aln_combine, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, p);
        ELBA_HIP(hipGetLastError());
        st.extensions_strided = nofl;
    }
    c.t_total.stop(s);
    unsigned long long cells = 0;
    ELBA_HIP(hipMemcpyAsync(&cells, p.cells, 8, hipMemcpyDeviceToHost, s));
    ELBA_HIP(hipStreamSynchronize(s));
    st.cells = (int64_t)cells;
    st.ms_total = c.t_total.ms();
    st.ms_extend = K > 0 ? c.t_a.ms() : 0.f;
    // counts for the statistics come from the results themselves
    if (K > 0) {
        std::vector<elba_overlap_t> h((size_t)K);
        ELBA_HIP(hipMemcpy(h.data(), c.aln_out.p, (size_t)K * sizeof(elba_overlap_t), hipMemcpyDeviceToHost));
        for (const auto &o : h) { st.seeds_rejected += o.score == -1 && o.endQ == 0 && o.endT == 0; st.passed += o.passed; st.contained += (o.containedQ | o.containedT); }
    }
    c.astats = st;
    c.have_aln = true;
}

// Multi-GPU alignment: the reads are small next to HBM (2 bits per base), so every rank keeps ALL of them (one all-gather by the
// driver) and aligns its share of the candidate pairs with no further communication — the reference's DistributedFastaData exchanges
// row and column read blocks of a 2D grid instead (src/DistributedFastaData.cpp).
void stage_dist_set_all_reads(Ctx &c, const void *d_packed, int64_t packed_bytes, const void *d_byte_off, const void *d_len, int64_t nreads_total)
{
    ELBA_REQUIRE(nreads_total >= 0 && packed_bytes >= 0 && (nreads_total == 0 || (d_packed && d_byte_off && d_len)), ELBA_ERR_INVALID_ARG, "dist_set_all_reads: null array");
    hipStream_t s = c.stream;
    c.aln_all_packed.reserve((size_t)packed_bytes + 16); c.aln_all_off.reserve((size_t)(nreads_total + 1) * 8); c.aln_all_len.reserve((size_t)(nreads_total + 1) * 4);
    ELBA_HIP(hipMemsetAsync(c.aln_all_packed.p, 0, (size_t)packed_bytes + 16, s));
    if (packed_bytes) ELBA_HIP(hipMemcpyAsync(c.aln_all_packed.p, d_packed, (size_t)packed_bytes, hipMemcpyDeviceToDevice, s));
    std::vector<uint32_t> hl((size_t)nreads_total);
    std::vector<uint64_t> ho((size_t)nreads_total);
    if (nreads_total) {
        ELBA_HIP(hipMemcpyAsync(c.aln_all_off.p, d_byte_off, (size_t)nreads_total * 8, hipMemcpyDeviceToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.aln_all_len.p, d_len, (size_t)nreads_total * 4, hipMemcpyDeviceToDevice, s));
        ELBA_HIP(hipMemcpyAsync(hl.data(), d_len, (size_t)nreads_total * 4, hipMemcpyDeviceToHost, s));
        ELBA_HIP(hipMemcpyAsync(ho.data(), d_byte_off, (size_t)nreads_total * 8, hipMemcpyDeviceToHost, s));
    }
    ELBA_HIP(hipStreamSynchronize(s));
    uint32_t mx = 0;
    for (uint32_t l : hl) mx = l > mx ? l : mx;
    for (int64_t r = 0; r < nreads_total; ++r)          // the x-drop kernels trust these offsets (as elba_set_reads_device checks its own)
        ELBA_REQUIRE((int64_t)ho[(size_t)r] + ((int64_t)hl[(size_t)r] + 3) / 4 <= packed_bytes, ELBA_ERR_INVALID_ARG, "dist_set_all_reads: read exceeds the packed buffer");
    c.aln_all_maxlen = mx; c.aln_all_n = nreads_total;
    c.have_aln = false;
}

}  // namespace elba
