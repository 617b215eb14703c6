// spgemm_direct.hpp — the numeric kernel of the overlap SpGEMM, PLAN-FREE (included by spgemm.hip inside its anonymous namespace).
//
// Input is A as the k-mer stage leaves it and nothing else: CSR (a_rowptr, a_csr: kid << 32 | hint << 30 | pos, rows in (kid, pos) order) and the
// k-mer columns — padded to a common stride (a_ell: column kid occupies the S = s_stride consecutive 8-byte words from kid * S,
// entries (read << 32 | pos) in (read, pos) order, the rest all ones) when no column is longer than 64 entries, else plain CSC
// (a_colptr, a_csc).  No per-row schedule, no descriptors, no row order, no product counts: whatever the product needs beyond the two
// orientations of A is computed inside the call (reference region: src/SharedSeeds.cpp:4-10 under the timer of src/main.cpp:280-282).
//
// One workgroup per row of B, rows claimed from the tier's queue with one atomic per row (claimed one row ahead).
//   * every wavefront walks the row in chunks of 64 consecutive entries and compacts those that must fetch their column into an LDS FIFO:
//     an entry whose hint bit says "this row accumulates no pair of this column" (Ctx::csr_hints) only counts its diagonal product;
//   * a group of LPC lanes takes one FIFO entry (kid, posQ, rank in the row) and reads the padded column as one contiguous segment, one
//     aligned 16-byte load per lane: a column costs the L1 one request per 64 bytes and the address needs no column pointer: kid * S.
//     (CSC matrices pay the dependent colptr load and walk long columns in chunks of 2 * LPC entries.)
//   * every lane then holds two candidate partners (read j, posT): all ones = padding, j == i = the diagonal (counted, never inserted),
//     anything else is a product; a pair of rows is accumulated on ONE of them (owns_pair below) and mirrored into the other afterwards
//     (spgemm.hip: k_mirror, or the exchange between ranks).  Products are compacted into a per-wavefront ring and update the LDS
//     accumulator of partner j: count, and first / last product of the ascending-k left fold through ds_min / ds_max of the sequence number
//     (rank of the row entry << fbits | index in the column), which is monotone in (kid, posQ, posT).  With PAY the 64-bit
//     extremes carry posQ and posT themselves: no seed look-ups afterwards.
//   * table tiers, optimistic sizing, escalation, HBM spill tier, ballot compaction of the survivors, staging: spgemm.hip.

enum : uint32_t { D_NEXT = 20, D_FB0 = 16, W_ACC_P = 54 /*u64*/, W_END2 = 56 };
constexpr uint32_t RING = 128;      // per-wavefront product ring (entries): < 64 left over + <= 64 new ones per candidate slot
constexpr uint32_t NOROW = 0xFFFFFFFFu, UNRESOLVED = 0xFFFFFFFEu;

// inclusive maximum over the lanes 0 .. own of a wavefront, values >= 0: six DPP steps, no LDS round trip
__device__ __forceinline__ uint32_t wave_max_scan(uint32_t v)
{
    int x = (int)v, t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false); x = t > x ? t : x;
    t = __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false); x = t > x ? t : x;
    t = __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false); x = t > x ? t : x;
    t = __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false); x = t > x ? t : x;
    t = __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false); x = t > x ? t : x;
    t = __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false); x = t > x ? t : x;
    return (uint32_t)x;
}

// sum over the wavefront (uniform result): the same six DPP steps with an add — the __shfl_xor butterfly is six LDS-crossbar round trips
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// inclusive prefix sum over the lanes 0 .. own of a wavefront: six DPP steps (a __shfl_up ladder is six LDS-crossbar round trips)
__device__ __forceinline__ uint32_t wave_add_scan(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)x;
}

// `half`: which of the two rows of a pair {i, j} accumulates it (the other row receives the mirrored entry).  The smaller row when i + j is
// even, the larger when it is odd: every row then owns about half of its partners whatever its place in the matrix (owned by the smaller
// row alone, the first rows would own all of theirs and need tables twice the size).  A partner outside this context's row window is
// always kept: its row lives on another rank.
__device__ __forceinline__ bool owns_pair(uint32_t i, uint32_t j, uint32_t row_lo, uint32_t row_hi, bool upper = false)
{
    // upper (dense matrices, OvParams::suffix): the smaller row owns the pair — the owned candidates of a row entry are then the entries of
    // its column BEHIND its own (columns are in read order); tables are small there (a few hundred partners), balance does not matter
    return j < row_lo || j >= row_hi || (upper ? j > i : ((((i ^ j) & 1u) != 0u) ? j < i : j > i));
}

// SUFFIX: the dense path (OvParams::suffix) instead of the general one — an instantiation of its own, so that the general kernel does not carry its registers
// (second launch bound = wavefronts per SIMD the register allocation must leave room for: 4 keeps two 512-thread workgroups on a CU — at 132
//  VGPRs instead of 128 the kernel loses one of them and runs twice as long (measured) —, 8 is what the dense path's 8 workgroups of 4 wavefronts per CU need)
template <int BLOCK, bool GLOBAL, bool PAY, int DK = 2, bool SUFFIX = false, int TB = 0>
#ifndef ELBA_DENSE_OCC
#define ELBA_DENSE_OCC 8
#endif
__global__ __launch_bounds__(BLOCK, SUFFIX ? ELBA_DENSE_OCC : (DK == 4 ? 1 : 4)) void k_spgemm_direct(OvParams p, int tier, uint32_t lds_tbits, uint32_t sample)
{
    static_assert(!PAY || !GLOBAL, "payload accumulators: LDS tiers only");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *misc = GLOBAL ? smem : smem + (size_t)(PAY ? 6 : 4) * (1u << lds_tbits) + ((size_t)1 << lds_tbits) / 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint64_t lt = (1ull << lane) - 1;
    const uint32_t nrows = sample ? p.ctr->sample_count : p.ctr->tier_count[tier];      // complete: every lower tier has finished (same stream); sample: the rows computed before all others on a cold call
    if (nrows == 0) return;      // (a tier nobody queued on — most tiers of a small matrix: its workgroups leave before they initialise a table; rows only ever move to HIGHER tiers, which start later)
    const uint32_t lb = p.lpc_log2, sub = tid & ((1u << lb) - 1u), grp = tid >> lb, EPT = (uint32_t)BLOCK >> lb;
    const uint32_t fbits = p.fbits, fmask = (1u << fbits) - 1u, stride = p.s_stride;
    const bool ell = p.a_ell != nullptr;
    const uint32_t hmask = p.hint_mask, pmask = p.pos_mask;      // ownership hints in the row entries (Ctx::csr_hints): skip bit of this call's mode, position bits
    const uint2 *csr2 = reinterpret_cast<const uint2 *>(p.a_csr);      // .x = position in the read, .y = k-mer id
    // mirror slabs (spgemm.hip): the ratio k_classify_direct settled for this call (0: none — the sample's rows run before it is known) and the
    // window's first row entry; both scalar
    const uint32_t slab_q = (p.slab != nullptr && !sample) ? sfirst(p.ctr->slab_q16) : 0u;
    unsigned long long chunk_off = 0;
    uint32_t chunk_left = 0;
    auto w64 = [&](uint32_t k) { return reinterpret_cast<unsigned long long *>(&misc[k]); };
    if (tid >= 32 && tid < W_END2) misc[tid] = 0;
    if (tid >= D_FB0 && tid < D_FB0 + 4u) misc[tid] = 0;
    const uint32_t *queue = sample ? p.sample_list : p.lists + (size_t)tier * p.M;
    // Rows are claimed one at a time, one row ahead.  One head word saturates at ~88 claims/us (MI355X_MICROARCH.md, "dequeue"); the queue
    // is therefore cut into 8 interleaved sub-queues (positions congruent modulo 8) with a head each, a workgroup draws from the one of
    // its XCD (workgroups go to the XCDs round-robin) and moves on to the next sub-queue when its own is exhausted.
    uint32_t qshard = blockIdx.x & 7u, qtried = 0;
    // (thread 0 only) draw: one atomic, nothing waits for it; resolve: where the draw is first needed — a draw beyond the end of the
    // sub-queue moves on to the next one (then, and only then, the claim is a synchronous round trip)
    // place of a sub-queue's k-th draw in the queue: the sub-queues take BLOCKS of 2^qb consecutive places in turn (qb = 0, the default: single places).  (A/B
    // hook of round 5: the dense path queues its rows in label order and a sub-queue belongs to an XCD — blocks of 128 places put the rows of one locus on ONE
    // XCD at the same time; it bought nothing, profiles/r05_notes.md)
    const uint32_t qb = p.qblk_log2;
    auto qplace = [&](uint32_t k) -> unsigned long long { return ((((unsigned long long)(k >> qb) * 8u + qshard) << qb) | (k & ((1u << qb) - 1u))); };
    auto draw = [&]() -> uint32_t { return atomicAdd(sample ? &p.ctr->sample_next[qshard][0] : &p.ctr->tier_next[tier][qshard][0], 1u); };
    auto resolve = [&](uint32_t k) -> uint32_t {
        for (;;) {
            const unsigned long long idx = qplace(k);
            if (idx < nrows) return queue[idx];
            if (++qtried >= 8u) return NOROW;
            qshard = (qshard + 1u) & 7u;
            k = draw();
        }
    };
    if (tid == 0) { const uint32_t r0 = resolve(draw()); misc[D_NEXT] = r0; if (r0 != NOROW) { misc[D_NEXT + 1] = p.a_rowptr[r0]; misc[D_NEXT + 2] = p.a_rowptr[r0 + 1]; } }
    // LDS tables are initialised ONCE per workgroup: a row hands its table back clean (the sweep resets the slots it does not list, the staging pass
    // the ones it has read) and zeroes the row's counters while nobody reads them — no init pass and no init barrier per row (4 % of the kernel)
    auto init_table = [&]() {
        if (GLOBAL) return;
        const uint32_t T0 = 1u << lds_tbits;
        for (uint32_t s2 = tid; s2 < T0; s2 += BLOCK) {
            smem[s2] = EMPTY; smem[T0 + s2] = 0;
            if (PAY) { reinterpret_cast<unsigned long long *>(smem + 2 * T0)[s2] = ~0ull; reinterpret_cast<unsigned long long *>(smem + 2 * T0)[T0 + s2] = 0ull; }
            else { smem[2 * T0 + s2] = 0xFFFFFFFFu; smem[3 * T0 + s2] = 0; }
        }
        if (tid < 16) misc[tid] = 0;
    };
    init_table();
    uint32_t dpar = 0;      // which of misc[0] / misc[1] counts this row's diagonal products (the other one is zeroed for the next row meanwhile)
    __syncthreads();
    uint32_t fb_seen = 0;
    bool fb_settled = false;
#ifdef ELBA_PHASE_CLOCK
    // diagnostic build only (make dbg): shader-clock cycles of wave 0 per phase, summed over workgroups into ctr->phase
    //   0 row header  1 table init  2 accumulate  3 next-row hand-off + barrier  4 sweep  5 reserve  6 staging stores
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define ELBA_DSTAMP(k) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += tn - tprev; tprev = tn; } while (0)
#else
#define ELBA_DSTAMP(k) do { } while (0)
#endif

    // The row a workgroup works on is published in one of TWO places, alternately: without a barrier at the start of a row a wavefront may still be
    // about to read its row when the first wavefront — done with a short row — publishes the next one (the hand-off barrier keeps them within one row
    // of each other).  (Found by tests/test_gpu_fuzz.py: with one place, two wavefronts disagreed about the current row once in ~20 calls on short reads.)
    uint32_t npar = 0;
    for (;;) {
        const uint32_t NB = D_NEXT + 4u * npar;      // where this row was published
        npar ^= 1u;
        const uint32_t NBN = D_NEXT + 4u * npar;     // where the next one goes
        const uint32_t i = sfirst(misc[NB]);
        if (i == NOROW) break;
        uint32_t nidx = 0, nrow = UNRESOLVED, nrs = 0, nre = 0;
        if (tid == 0 && qtried < 8u) nidx = draw();      // the row after this one: the round trip hides behind this row
        const uint32_t rs = sfirst(misc[NB + 1]), nnz = sfirst(misc[NB + 2]) - rs;      // (the row's bounds travel with its id: thread 0 fetched them a row ago)
        // distinct partners of the row <= min(products, reads); products <= nnz * longest column
        const unsigned long long prod_ub = (unsigned long long)nnz * p.max_col;
        const uint32_t ub_i = prod_ub < (unsigned long long)p.Mcols ? (uint32_t)prod_ub : p.Mcols;
        // the draw has returned by the time the first round's column words have (loads and returning atomics come back in order): its queue
        // entry is requested there and arrives with the second round's words — the hand-off below then waits for nothing
        auto resolve_early = [&]() {
            if (tid == 0 && qtried < 8u) { const unsigned long long idx = qplace(nidx); if (idx < nrows) nrow = queue[idx]; }
        };
        // ... and the next row's bounds with the second round's (they depend on the queue entry)
        auto bounds_early = [&]() { if (tid == 0 && nrow != UNRESOLVED && nrow != NOROW) { nrs = p.a_rowptr[nrow]; nre = p.a_rowptr[nrow + 1]; } };
        auto publish_next = [&]() {
            if (tid == 0) {
                const bool early = nrow != UNRESOLVED;
                if (!early) nrow = qtried < 8u ? resolve(nidx) : NOROW;
                if (nrow != NOROW && (!early || nre == 0)) { nrs = p.a_rowptr[nrow]; nre = p.a_rowptr[nrow + 1]; }      // (short rows: fetched here, one exposed round trip)
                misc[NBN] = nrow; misc[NBN + 1] = nrs; misc[NBN + 2] = nre;
                if (!GLOBAL) misc[3] = 0;      // (the survivor count of the row before: read in its staging pass, needed clean by this row's sweep — behind the hand-off barrier)
            }
        };

        if (!GLOBAL && p.use_feedback) {
            // Self-correction inside a call (no prior for this matrix): rows already done tell how many distinct partners a row entry
            // brings on THIS data; a row predicted not to fit is forwarded without an attempt.  (lane 0 reads the hot sums once per 8 rows
            // and broadcasts through LDS: the decision must be workgroup-uniform)
            // (read before each of the workgroup's first rows — a wrong cold guess should cost a workgroup one abandoned row, not sixteen — then
            //  every 16th; once the sums cover p.fb_enough row entries the ratio is settled: nobody reads or adds to the hot line any more)
            if (!fb_settled && (fb_seen < 4u || (fb_seen & 15u) == 0)) {      // (fb_settled: a register set from the snapshot BEHIND the barrier: workgroup-uniform)
                if (tid == 0) {
                    const unsigned long long u = __hip_atomic_load(&p.ctr->fb_ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long c = __hip_atomic_load(&p.ctr->fb_claims, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    misc[D_FB0] = (uint32_t)u; misc[D_FB0 + 1] = (uint32_t)(u >> 32); misc[D_FB0 + 2] = (uint32_t)c; misc[D_FB0 + 3] = (uint32_t)(c >> 32);
                }
                __syncthreads();
            }
            ++fb_seen;
            const unsigned long long gu = ((unsigned long long)misc[D_FB0 + 1] << 32) | misc[D_FB0], gc = ((unsigned long long)misc[D_FB0 + 3] << 32) | misc[D_FB0 + 2];
            fb_settled = gu >= p.fb_enough;
            if (gu >= (1ull << 18)) {
                const double pred = 1.25 * (double)nnz * (double)gc / (double)gu;
                if (pred > 1.2 * (double)p.tier_limit[tier] && guaranteed_tbits(ub_i, p.Mcols) > lds_tbits) {
                    int t2 = tier + 1;
                    while (t2 < NUM_LDS_TIERS && pred > (double)p.tier_limit[t2]) ++t2;
                    if (tid == 0) {
                        const uint32_t at = atomicAdd(&p.ctr->tier_count[t2], 1u);
                        p.lists[(size_t)t2 * p.M + at] = i;
                    }
                    __syncthreads();          // every wave has read this row's id before the next one is published
                    publish_next();
                    __syncthreads();
                    continue;
                }
            }
        }

        Table<GLOBAL> tab;
        tab.misc = misc;
        uint32_t *list = nullptr;
        uint16_t *list16 = nullptr;
        if (GLOBAL) {
            tab.tbits = guaranteed_tbits(ub_i, p.Mcols);
            tab.limit = 0xFFFFFFFFu;
            uint32_t *base = p.gtable + (size_t)blockIdx.x * 5 * p.gstride;
            tab.keys = base; tab.cnt = base + p.gstride; tab.smin = base + 2 * p.gstride; tab.smax = base + 3 * p.gstride;
            list = base + 4 * p.gstride;
        } else {
            tab.tbits = lds_tbits;
            const uint32_t T = 1u << lds_tbits;
            tab.limit = p.tier_limit[tier];
            tab.keys = smem; tab.cnt = smem + T; tab.smin = smem + 2 * T; tab.smax = smem + 3 * T;
            tab.vmin = reinterpret_cast<unsigned long long *>(smem + 2 * T); tab.vmax = tab.vmin + T;
            list16 = reinterpret_cast<uint16_t *>(smem + (PAY ? 6 : 4) * T);
        }
        const uint32_t T = tab.size();
        ELBA_DSTAMP(0);
        if (GLOBAL) {
            for (uint32_t s = tid; s < T; s += BLOCK) { tab.keys[s] = EMPTY; tab.cnt[s] = 0; tab.smin[s] = 0xFFFFFFFFu; tab.smax[s] = 0; }
            if (tid < 16) misc[tid] = 0;
            __syncthreads();
        }
        auto reset_slot = [&](uint32_t s2) {      // (LDS tables) back to the state init_table leaves
            tab.keys[s2] = EMPTY; tab.cnt[s2] = 0;
            if (PAY) { tab.vmin[s2] = ~0ull; tab.vmax[s2] = 0ull; } else { tab.smin[s2] = 0xFFFFFFFFu; tab.smax[s2] = 0; }
        };
        const uint32_t DG = GLOBAL ? 0u : dpar;      // the word of this row's diagonal count

        // ---- accumulate -----------------------------------------------------------------------------------------------------
        // Products are COMPACTED before they meet the table.  Of the candidate slots a wavefront looks at, one in five holds a product on
        // 15 %-error reads (columns of 2-3 entries padded to 8, one of them the row's own): inserting slot by slot ran the hand-scheduled
        // probe loop — a chain of LDS round trips, as long as the unluckiest lane's — at a fifth of its width, and that chain, not the
        // gathers, bounded the kernel.  Each wavefront therefore appends its products to a 128-entry ring in LDS (ballot + popcount prefix:
        // partner id and the 64-bit payload, or the 32-bit sequence number) and runs the probe loop once per 64 products, every lane busy.
        bool full = false;
        uint32_t dg = 0, pr = 0;                 // diagonal products / all products (valid column entries) of the row, as seen by this wavefront (uniform)
        uint32_t head = 0, tail = 0;             // ring positions (uniform)
        // per wavefront: product ring (128 x 12 or 8 bytes), then the entry FIFO of the padded-column path (128 x 12 bytes); the dense path keeps
        // the ring (for the products its fast look-up misses) and its 336 words of hand-out tables
        // (pay16, OvParams: 32-bit accumulators that CARRY posT — sequence number << 16 | posT — and 8-byte FIFO entries: 512 words per wavefront, 18 bytes per slot:
        //  three 512-lane workgroups per CU where the 64-bit accumulators admit two)
        const bool pay16 = !PAY && !SUFFIX && !GLOBAL && p.pay16 != 0u;
        uint32_t *qj = misc + 64 + (tid >> 6) * (SUFFIX ? 592u : (PAY ? 768u : (pay16 ? 512u : 640u)));
        uint32_t *qs = qj + RING;
        unsigned long long *qv = reinterpret_cast<unsigned long long *>(qj + RING);
        auto drain = [&](uint32_t n) {           // n <= 64 products leave the ring, one per lane
            const uint32_t at = (head + lane) & (RING - 1u);
            const uint32_t j = qj[at];
            if (PAY) { const unsigned long long v = qv[at]; tab.insert_lds64(j, v, v, 1u, lane < n, full); }
            else { const uint32_t sq = qs[at]; tab.insert_lds(j, sq, sq, 1u, lane < n, full); }
            head += n;
        };
        // one candidate slot per lane: j == EMPTY is padding (or a lane without a row entry), j == i the diagonal; with `half` a pair of rows
        // of the window is accumulated on its smaller row only and mirrored afterwards (partners below the window live on another rank: kept)
        auto slot = [&](uint32_t j, uint32_t posT, uint32_t seq, uint32_t posQ) {
            const uint64_t mv = __ballot(j != EMPTY);
            if (mv == 0) return;
            const uint64_t md = __ballot(j == i);
            pr += (uint32_t)__popcll(mv); dg += (uint32_t)__popcll(md);
            uint64_t mi = mv & ~md;
            if (p.half) mi &= __ballot(owns_pair(i, j, p.half == 2u ? 0u : p.row_lo, p.half == 2u ? 0xFFFFFFFFu : p.row_hi, !PAY && p.suffix != 0u));      // (dense matrices reach this path on the tiers without 64-bit accumulators only)      // (2: the rule holds for every partner, wherever its row lives)
            if (mi == 0) return;
            const bool ins = (mi >> lane) & 1ull;
            if (GLOBAL) { if (ins) tab.insert(j, seq, full); return; }
            const uint32_t at = (tail + (uint32_t)__popcll(mi & lt)) & (RING - 1u);
            if (ins) {
                qj[at] = j;
                if (PAY) qv[at] = ((unsigned long long)seq << 32) | (posQ << 16) | posT; else qs[at] = pay16 ? (seq << 16) | posT : seq;
            }
            tail += (uint32_t)__popcll(mi);
            if (tail - head >= 64u) drain(64u);
        };
        if (SUFFIX) {
            // Dense matrices (matrix.hip: Ctx::csr_suffix).  A row entry carries its column's length L and its own place idx in it; the pairs
            // it owns (smaller row owns) are exactly the column's entries idx + 1 .. L - 1.  The partner ids of a column sit RIGHT-ALIGNED in
            // an aligned block of Sj = 32 or 64 four-byte slots (a_ellj: 128 / 256 bytes), so what an entry owns is the tail of that block:
            // one 128-byte segment for all but the first entries of the longest columns, fetched in 16-byte PIECES (four slots).  The path
            // is bound by memory requests (profiles/r03_notes.md: 555 M requests at 85 % of the random-line ceiling when it fetched 4 bytes
            // per lane from columns at a 160-byte stride), and an aligned 128-byte segment costs the memory system what a 64-byte line does
            // (profiles/microbench/gather128.hip: 41-45 G segments/s against 46-49 G lines/s).
            // A wavefront takes 64 consecutive row entries, prefix-sums their numbers of pieces and hands the pieces out to its lanes, 64 at
            // a time: a lane finds its entry through marks the entries leave at their first piece (one LDS round trip + a DPP maximum scan),
            // fetches ONE 16-byte piece and inserts its four candidates straight into the table — every lane a product but for the slots in
            // front of the first owned one (first piece of an entry only) and a read that holds the k-mer twice (the diagonal); no padding
            // to skip, no ownership test, no ring.
            static_assert(!SUFFIX || !PAY, "dense path: 32-bit accumulators, the seeds of the few survivors are looked up");
#ifndef ELBA_DENSE_UN
#define ELBA_DENSE_UN 2
#endif
            constexpr uint32_t UN = ELBA_DENSE_UN;      // pieces per lane in flight
            static_assert(UN % 2 == 0, "windows of two batches");
            const uint32_t jsh = p.j_shift, Sj = 1u << jsh;
            const uint32_t il = p.row_label ? p.row_label[i] : i;      // (partners are named by label: OvParams::row_label)
            const uint4 *ellq = reinterpret_cast<const uint4 *>(p.a_ellj);
            uint32_t *skid = qj + 2u * RING, *smeta = skid + 64, *spre = skid + 128, *sown = skid + 200;      // per wavefront: k-mer id, first piece | first owned slot << 8 | L << 16, exclusive prefix (65 words), window marks (128)
            auto retry = [&](bool pred, uint32_t j, uint32_t sq1) {      // the lanes with pred queue a product for the general insert
                const uint64_t mm = __ballot(pred);
                if (mm == 0) return;
                const uint32_t at = (tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))) & (RING - 1u);
                if (pred) { qj[at] = j; qs[at] = sq1; }
                tail += (uint32_t)__popcll(mm);
                if (tail - head >= 64u) drain(64u);
            };
#pragma unroll 1
            for (uint32_t cit = 0;; ++cit) {
                // (chunks of 64 row entries are drawn from a counter of the row, as on the general path: the numbers of candidates per chunk differ.
                //  Drawing and requesting the NEXT chunk before this one is walked was measured on config 5 at 1/25: 9.50 against 9.51 ms, two more
                //  spilled registers — the path is bound by its LDS accesses, profiles/r04_notes.md)
                uint32_t cdraw = 0;
                if (lane == 0) { const uint32_t a14 = (uint32_t)(uintptr_t)&misc[14], step = 64u; asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(cdraw) : "v"(a14), "v"(step) : "memory"); }
                const uint32_t cbase = sfirst(cdraw);
                if (cbase >= nnz) break;
                const bool valid = cbase + lane < nnz;
                const uint2 en = valid ? csr2[rs + cbase + lane] : make_uint2(0u, 0u);
                const uint32_t L = (en.x >> 23) & 127u, idx = (en.x >> 16) & 127u;
                const uint32_t w = valid ? L - idx - 1u : 0u;
                const uint32_t fs = Sj - w, p0 = fs >> 2;                      // first owned slot of the block, the piece that holds it
                const uint32_t np = w ? (Sj >> 2) - p0 : 0u;
                const uint32_t inc = wave_add_scan(np);
                const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                dg += (uint32_t)__popcll(__ballot(valid));                     // every entry's product with itself
                __builtin_amdgcn_wave_barrier();                                // (the previous chunk's look-ups are done)
                skid[lane] = en.y; smeta[lane] = p0 | fs << 8 | L << 16; spre[lane] = inc - np;
                if (lane == 63) spre[64] = T;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (cit == 0) { resolve_early(); } else if (cit == 1) bounds_early();
#pragma unroll 1
                for (uint32_t base = 0; base < T; base += 64u * UN) {
                    uint4 x[UN];
                    uint32_t sq[UN], vr[UN];
                    bool ok[UN];
                    // piece -> entry, 128 pieces at a time: every entry that starts inside the window (or covers its first place) marks
                    // its first place with its number, a DPP maximum scan spreads the marks to the right
#pragma unroll
                    for (int hw = 0; hw < (int)UN / 2; ++hw) {
                        const uint32_t B = base + (uint32_t)hw * 128u;
                        sown[lane] = 0; sown[lane + 64u] = 0;
                        __builtin_amdgcn_wave_barrier();
                        if (np != 0u) {
                            const uint32_t pre = inc - np;
                            if (pre >= B && pre - B < 128u) sown[pre - B] = lane + 1u;
                            else if (pre < B && inc > B) sown[0] = lane + 1u;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t m0 = wave_max_scan(sown[lane]);
                        const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)m0, 63);
                        uint32_t m1 = wave_max_scan(sown[lane + 64u]);
                        m1 = m1 > carry ? m1 : carry;
#pragma unroll
                        for (int v2 = 0; v2 < 2; ++v2) {
                            const int u = 2 * hw + v2;
                            const uint32_t c = B + (uint32_t)v2 * 64u + lane;
                            ok[u] = c < T;
                            const uint32_t en1 = v2 ? m1 : m0, lo = ok[u] ? en1 - 1u : 0u;
                            const uint32_t kl = skid[lo];                       // (with the two words below: one LDS round trip, not a second one under the load's predicate)
                            const uint32_t mt = smeta[lo], pc = (mt & 63u) + (c - spre[lo]), s0 = pc << 2, f0 = (mt >> 8) & 127u;
                            vr[u] = !ok[u] ? 4u : (f0 > s0 ? f0 - s0 : 0u);                    // slots of the piece in front of the first owned one (a piece not handed out: all four)
                            sq[u] = ((cbase + lo) << fbits) + (s0 + (mt >> 16) - Sj);        // sequence number of the piece's first slot: rank of the row entry | place in the column
                            x[u] = ok[u] ? ellq[((unsigned long long)kl << (jsh - 2u)) + pc] : make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
#pragma unroll
                    for (int u = 0; u < (int)UN; ++u) {
                        if (u > 0 && base + 64u * (uint32_t)u >= T) break;      // (the chunk's last pieces fill less than the whole batch: ~270 pieces per chunk, handed out 128 at a time)
                        const uint32_t xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                        // the read holds the k-mer again behind this entry (rare): that pair of entries counts twice on the diagonal, and the
                        // other candidates of such a piece take the general insert
                        const bool anyd = (xs[0] == il && vr[u] == 0u) || (xs[1] == il && vr[u] <= 1u) || (xs[2] == il && vr[u] <= 2u) || (xs[3] == il && vr[u] <= 3u);      // (among the OWNED slots: the slot in front of them is the entry itself)
                        if (__ballot(anyd)) {
#pragma unroll 1
                            for (uint32_t r = 0; r < 4u; ++r) {      // (not unrolled: one copy of the ring's drain, not four, beside the hot path)
                                const uint32_t xr = r == 0u ? xs[0] : (r == 1u ? xs[1] : (r == 2u ? xs[2] : xs[3]));
                                const bool own = anyd && r >= vr[u];
                                dg += 2u * (uint32_t)__popcll(__ballot(own && xr == il));
                                if (!full) retry(own && xr != il, xr, sq[u] + r);
                            }
                            if (anyd) vr[u] = 4u;
                        }
                        // four look-ups in flight; what they do not settle (a partner not met before or not in its first slot, a new minimum)
                        // queues for the general insert
                        if (!full) {
                            bool miss[4];
                            tab.template hit4m_lds<TB>(xs, sq[u], vr[u], miss);
#pragma unroll
                            for (int r = 0; r < 4; ++r) retry(miss[r], xs[r], sq[u] + (uint32_t)r);
                        }
                    }
                    if (tab.abandoned()) break;
                }
                if (tab.abandoned()) {
                    if (tid == 0) { const uint32_t done = cbase + 64u; misc[11] = done < nnz ? done : nnz; }
                    break;
                }
            }
        } else if (!SUFFIX && ell) {
            // Every wavefront walks the row in chunks of 64 consecutive entries (chunk c belongs to wave c mod #waves: one coalesced 512-byte
            // load) and COMPACTS them: an entry hinted "this row accumulates no pair of its column" (Ctx::csr_hints: 40 % of the entries of
            // 15 %-error reads) only counts its one diagonal product; the others join a 128-entry FIFO in LDS (ballot + popcount prefix).
            // Groups of LPC lanes then take entries off the FIFO, TR wave-trips per iteration, and fetch their columns — every lane of every
            // gather instruction fetches a column that is needed.  Two iterations are in flight: an iteration (1) consumes the column words
            // the previous one requested, (2) refills the FIFO from the chunks prefetched an iteration ago and requests the next ones,
            // (3) requests the next TR trips' column words, (4) updates the accumulator (LDS only) while the requests land.  What bounds the
            // loop is the number of 64-byte lines a CU has in flight (profiles/r02_gather64_microbench.txt: 46-52 G random lines/s chip-wide).
            const uint4 ones = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
            constexpr int TR = DK == 0 ? 1 : 2 * DK;      // (DK = 0: ONE trip per iteration — matrices whose rows mostly carry their products inline)
            constexpr uint32_t FQ = 128, NONE = 0xFFFFFFFFu;
            const uint32_t gw = lane >> lb, EW = 64u >> lb;               // this lane's entry within a wave-trip, entries per wave-trip
            uint32_t *fq = qj + (PAY ? 384u : 256u);                      // FIFO: 3 words per entry (position, k-mer id, rank in the row), behind the product ring
            uint32_t fh = 0, ft = 0;                                      // FIFO positions (uniform)
            uint2 ea, eb;
            uint32_t ca, cb;                                              // the two chunks on their way (first row entry, or NONE)
#ifndef ELBA_CHUNKS
#define ELBA_CHUNKS 3
#endif
            constexpr int XC = ELBA_CHUNKS - 2;                           // further chunks requested ahead: THREE in flight in all (numeric 6.75 -> 6.65 ms against two; four and six: no better)
            uint2 ex[XC > 0 ? XC : 1]; uint32_t cx[XC > 0 ? XC : 1];
            auto load_chunk = [&](uint2 &en, uint32_t &cbase) {
                // chunks are DRAWN, not dealt: a wavefront whose chunks held few products takes more of them (sequence numbers are the entries' ranks
                // in the row and the accumulators are order-free: who processes a chunk does not matter; misc[14] is zeroed with the row's other words): 7.26 -> 7.13 ms
                uint32_t c = 0;
                if (lane == 0) { const uint32_t a14 = (uint32_t)(uintptr_t)&misc[14], step = 64u; asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(a14), "v"(step) : "memory"); }
                const uint32_t cnext = sfirst(c);
                cbase = cnext < nnz ? cnext : NONE;
                en = cbase != NONE && cbase + lane < nnz ? csr2[rs + cbase + lane] : make_uint2(0u, 0u);
            };
            auto consume = [&]() {                                        // chunk `ea` -> FIFO; the chunk after it moves up, the one after that is requested
                const bool valid = ca + lane < nnz;
                // inline partner (Ctx::csr_inline): an entry whose row accumulates exactly one pair of its column IS that product — flag |
                // partner >> 1 | posQ | posT << 16, the partner's low bit from the ownership rule — no column to fetch
                const bool inl = p.inl != 0u && valid && (ea.y >> 31) != 0u;
                const bool need = valid && !inl && !(ea.x & hmask);
                if (p.inl != 0u) {
                    const uint64_t mi = __ballot(inl);
                    if (mi) {
                        const uint32_t jh = ea.y & 0x7FFFFFFFu, ih = i >> 1;
                        const uint32_t j = jh > ih ? 2u * jh + (i & 1u) : (jh < ih ? 2u * jh + ((i & 1u) ^ 1u) : (i ^ 1u));
                        const uint32_t seq = (ca + lane) << fbits;      // (the entry's ONLY product with this partner: its place in the column orders nothing)
                        if (GLOBAL) { if (inl) tab.insert(j, seq, full); }
                        else {
                            const uint32_t at = (tail + (uint32_t)__popcll(mi & lt)) & (RING - 1u);
                            if (inl) {
                                qj[at] = j;
                                if (PAY) qv[at] = ((unsigned long long)seq << 32) | ((ea.x & 0xFFFFu) << 16) | (ea.x >> 16); else qs[at] = pay16 ? (seq << 16) | (ea.x >> 16) : seq;
                            }
                            tail += (uint32_t)__popcll(mi);
                            if (tail - head >= 64u) drain(64u);
                        }
                    }
                }
                const uint64_t mn = __ballot(need);
                dg += (uint32_t)__popcll(__ballot(valid && !need));       // skipped and inline entries: their one diagonal product
                if (need) {
                    const uint32_t at = ((ft + (uint32_t)__popcll(mn & lt)) & (FQ - 1u)) * (pay16 ? 2u : 3u);
                    if (pay16) { fq[at] = (ea.x & 0xFFFFu) | ((ca + lane) << 16); fq[at + 1u] = ea.y; }      // (positions and the rank in the row fit 16 bits each: the condition of pay16)
                    else { fq[at] = ea.x & pmask; fq[at + 1u] = ea.y; fq[at + 2u] = ca + lane; }
                }
                ft += (uint32_t)__popcll(mn);
                ea = eb; ca = cb;
                if (XC > 0) {
                    eb = ex[0]; cb = cx[0];
#pragma unroll
                    for (int q = 0; q + 1 < XC; ++q) { ex[q] = ex[q + 1]; cx[q] = cx[q + 1]; }
                    load_chunk(ex[XC > 0 ? XC - 1 : 0], cx[XC > 0 ? XC - 1 : 0]);
                } else load_chunk(eb, cb);
            };
            auto issue = [&](uint4 *x, uint32_t *pq, uint32_t *rk) {      // the next TR trips leave the FIFO: their column words are requested
#pragma unroll
                for (int u = 0; u < TR; ++u) {
                    const uint32_t k = (uint32_t)u * EW + gw;
                    x[u] = ones; pq[u] = 0; rk[u] = 0;
                    if (k < ft - fh && 2u * sub < stride) {      // (a stride that is no power of two leaves the group's last lanes without a word)
                        const uint32_t at = ((fh + k) & (FQ - 1u)) * (pay16 ? 2u : 3u);
                        if (pay16) { const uint32_t w0 = fq[at]; pq[u] = w0 & 0xFFFFu; rk[u] = w0 >> 16; }
                        else { pq[u] = fq[at]; rk[u] = fq[at + 2u]; }
                        x[u] = *reinterpret_cast<const uint4 *>(p.a_ell + ((unsigned long long)fq[at + 1u] * stride + 2u * sub));
                    }
                }
                const uint32_t n = ft - fh;
                fh += n < (uint32_t)TR * EW ? n : (uint32_t)TR * EW;
            };
            load_chunk(ea, ca); load_chunk(eb, cb);
#pragma unroll
            for (int q = 0; q < XC; ++q) load_chunk(ex[q], cx[q]);
            if (ca != NONE) consume();
            if (ca != NONE && ft - fh <= 64u) consume();
            uint4 x_cur[TR];
            uint32_t pq_cur[TR], rk_cur[TR];
            issue(x_cur, pq_cur, rk_cur);
#pragma unroll 1
            for (uint32_t it = 0;; ++it) {
#pragma unroll
                for (int u = 0; u < TR; ++u) asm volatile("" : "+v"(x_cur[u].x), "+v"(x_cur[u].y), "+v"(x_cur[u].z), "+v"(x_cur[u].w) : : "memory");                   // (1)
                asm volatile("" : "+v"(ea.x), "+v"(ea.y) : : "memory");
                if (ca != NONE && ft - fh <= 64u) consume();                                                                                          // (2)
                if (ca != NONE && ft - fh <= 64u) consume();
                const bool last = ft == fh && ca == NONE;                 // nothing left to request: this iteration's words are the row's last
                uint4 x_nxt[TR];
                uint32_t pq_nxt[TR], rk_nxt[TR];
                issue(x_nxt, pq_nxt, rk_nxt);                                                                                                         // (3)
                if (it == 0) resolve_early(); else if (it == 1) bounds_early();
#pragma unroll
                for (int u = 0; u < TR; ++u) {                                                                                                        // (4)
                    const uint32_t seq = (rk_cur[u] << fbits) | (2u * sub);
                    slot(x_cur[u].y, x_cur[u].x, seq, pq_cur[u]);
                    slot(x_cur[u].w, x_cur[u].z, seq + 1u, pq_cur[u]);
                }
                if (tab.abandoned()) {
                    if (tid == 0) { const uint32_t done = ca != NONE ? ca : nnz; misc[11] = done < nnz ? done : nnz; }
                    break;
                }
                if (last) break;
#pragma unroll
                for (int u = 0; u < TR; ++u) { x_cur[u] = x_nxt[u]; pq_cur[u] = pq_nxt[u]; rk_cur[u] = rk_nxt[u]; }
            }
        } else if (!SUFFIX) {
#pragma unroll 1
            for (uint32_t t0 = 0; t0 < nnz; t0 += EPT) {
                const uint32_t r = t0 + grp;
                uint2 c = make_uint2(0u, 0u);
                uint32_t c0 = 0, len = 0;
                if (r < nnz) { c = csr2[rs + r]; if (!(c.x & hmask)) { c0 = p.a_colptr[c.y]; len = p.a_colptr[c.y + 1] - c0; } }
                if (hmask) dg += (uint32_t)__popcll(__ballot(sub == 0 && (c.x & hmask) != 0u));
#pragma unroll 1
                for (uint32_t b = 2u * sub; __ballot(b < len) != 0; b += 2u << lb) {      // wave-uniform trip count: chunks of 2 * LPC entries
                    uint2 e0 = make_uint2(0u, EMPTY), e1 = e0;
                    if (b < len) e0 = reinterpret_cast<const uint2 *>(p.a_csc)[c0 + b];
                    if (b + 1u < len) e1 = reinterpret_cast<const uint2 *>(p.a_csc)[c0 + b + 1u];
                    const uint32_t seq = (r << fbits) | b;
                    slot(e0.y, e0.x, seq, c.x & pmask);
                    slot(e1.y, e1.x, seq + 1u, c.x & pmask);
                }
                if (tab.abandoned()) {
                    if (tid == 0) { const uint32_t done = t0 + EPT; misc[11] = done < nnz ? done : nnz; }
                    break;
                }
            }
        }
        if (!GLOBAL && tail != head) drain(tail - head);
        if ((pr | dg) != 0 && lane == 0) { lds_add32(&misc[DG], dg); lds_add32(&misc[12], pr); }
        ELBA_DSTAMP(2);
        publish_next();
        if (GLOBAL) __syncthreads(); else lds_barrier();
        ELBA_DSTAMP(3);
        if (tab.abandoned()) {
            // the optimistic table was too small: hand the row to the next tier (its kernel starts after this one ends)
            if (tid == 0) {
                const uint32_t at = atomicAdd(&p.ctr->tier_count[tier + 1], 1u);
                p.lists[(size_t)(tier + 1) * p.M + at] = i;
                const unsigned long long all = nnz ? nnz : 1u, done = misc[11] ? misc[11] : all;
                lds_add64(w64(W_FB_C), (unsigned long long)misc[9] * all / done); lds_add64(w64(W_FB_U), (unsigned long long)nnz); lds_add32(&misc[W_FB_N], 1u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (p.use_feedback && misc[W_FB_N] >= 2 && ((((unsigned long long)misc[D_FB0 + 1] << 32) | misc[D_FB0]) < p.fb_enough)) {          // abandoned rows are the strongest evidence: publish at once
                    const unsigned long long fc = *w64(W_FB_C), fu = *w64(W_FB_U);
                    atomicAdd(&p.ctr->fb_claims, fc); atomicAdd(&p.ctr->fb_ub, fu);
                    *w64(W_TOT_C) += fc; *w64(W_TOT_U) += fu; *w64(W_FB_C) = 0; *w64(W_FB_U) = 0; misc[W_FB_N] = 0;
                }
            }
            if (!GLOBAL) { __syncthreads(); init_table(); if (tid < 2) misc[tid] = 0; }      // (every wavefront has left the table: an abandoned row leaves it dirty)
            __syncthreads();
            continue;
        }

        // ---- one table sweep: nnz before prune + ballot-compacted survivor list ----
        uint32_t yraw = 0;
        const bool relabel = SUFFIX && p.row_label != nullptr, whole = p.row_lo == 0u && p.row_hi == p.M;
        if (!GLOBAL && T == 4u * (uint32_t)BLOCK) {
            // (the LDS tiers up to 4096 slots: four slots per lane — all eight table words requested at once, the wavefront's survivors take their
            //  places in the list with ONE returning atomic: three LDS round trips per row where the loop below makes a dozen)
            uint32_t jj[4], cc[4], pre[4], tot = 0;
            uint64_t bal[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { jj[u] = tab.keys[(uint32_t)u * BLOCK + tid]; cc[u] = tab.cnt[(uint32_t)u * BLOCK + tid]; }
            if (SUFFIX && relabel && !whole) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (jj[u] != EMPTY) jj[u] = p.row_order[jj[u]];      // (the window test below wants the row)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t j = jj[u];
                if (j != EMPTY) yraw += (p.half == 2u || (p.half && j >= p.row_lo && j < p.row_hi)) ? 2u : 1u;
                bal[u] = __ballot(j != EMPTY && cc[u] >= 2u);
                if (j != EMPTY && cc[u] < 2u) reset_slot((uint32_t)u * BLOCK + tid);      // (not listed: nobody reads this slot again)
                pre[u] = tot; tot += (uint32_t)__popcll(bal[u]);
            }
            if (tot) {
                uint32_t at = 0;
                if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(at) : "v"((uint32_t)(uintptr_t)&misc[3]), "v"(tot) : "memory");
                at = sfirst(at);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((bal[u] >> lane) & 1ull) list16[at + pre[u] + (uint32_t)__popcll(bal[u] & lt)] = (uint16_t)((uint32_t)u * BLOCK + tid);
            }
        } else
        for (uint32_t b0 = 0; b0 < T; b0 += BLOCK) {                       // wave-uniform trip count: ballots are safe
            const uint32_t s0 = b0 + tid;
            bool keep = false;
            if (s0 < T) {
                uint32_t j = tab.ld(tab.keys, s0);
                if (j != EMPTY) {
                    if (SUFFIX && relabel && !whole) j = p.row_order[j];
                    yraw += (p.half == 2u || (p.half && j >= p.row_lo && j < p.row_hi)) ? 2u : 1u; keep = tab.ld(tab.cnt, s0) >= 2;
                    if (!GLOBAL && !keep) reset_slot(s0);
                }
            }
            const uint64_t bal = __ballot(keep);
            if (bal == 0) continue;
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&misc[3], (uint32_t)__popcll(bal));
            at = __shfl(at, 0, 64) + (uint32_t)__popcll(bal & lt);
            if (keep) { if (GLOBAL) list[at] = s0; else list16[at] = (uint16_t)s0; }
        }
        yraw = wave_sum_dpp(yraw);
        if (lane == 0 && yraw) lds_add32(&misc[5], yraw);
        if (GLOBAL) __syncthreads(); else lds_barrier();
        ELBA_DSTAMP(4);
        if (tid == 0) {
            const uint32_t dcount = misc[DG];
            const uint32_t ytot = misc[3] + (dcount >= 2 ? 1u : 0u);
            unsigned long long off;
            if (ytot <= chunk_left) { off = chunk_off; chunk_off += ytot; chunk_left -= ytot; }
            else if (ytot >= STAGE_CHUNK / 2) off = atomicAdd(&p.ctr->cursor, (unsigned long long)ytot);
            else { off = atomicAdd(&p.ctr->cursor, (unsigned long long)STAGE_CHUNK); chunk_off = off + ytot; chunk_left = STAGE_CHUNK - ytot; }
            const bool fits = off + ytot <= p.tmp_cap;
            if (!fits) atomicOr(&p.ctr->overflow, 1u);
            p.row_cnt[i] = fits ? ytot : 0u;      // (a row that found no room stages nothing: the finalize pass must not read behind the area; the call is repeated)
            p.row_off[i] = off;
            misc[6] = (uint32_t)off; misc[7] = (uint32_t)(off >> 32); misc[8] = fits ? 1u : 0u;
            lds_add64(w64(W_ACC_YRAW), (unsigned long long)(misc[5] + (dcount >= 1 ? 1u : 0u)));
            lds_add64(w64(W_ACC_P), (unsigned long long)misc[12]);
            lds_add32(&misc[W_ACC_DONE], 1u);
            lds_add32(&misc[W_ACC_NDIAG], dcount >= 2 ? 1u : 0u);
            lds_add64(w64(W_ACC_Y), (unsigned long long)ytot);
            if (!GLOBAL) { lds_add64(w64(W_FB_C), (unsigned long long)misc[9]); lds_add64(w64(W_FB_U), (unsigned long long)nnz); lds_add32(&misc[W_FB_N], 1u); }
            if (p.use_feedback && misc[W_FB_N] >= 8 && ((((unsigned long long)misc[D_FB0 + 1] << 32) | misc[D_FB0]) < p.fb_enough)) {          // no prior yet: push this workgroup's share to the hot sums
                const unsigned long long fc = *w64(W_FB_C), fu = *w64(W_FB_U);
                atomicAdd(&p.ctr->fb_claims, fc); atomicAdd(&p.ctr->fb_ub, fu);
                *w64(W_TOT_C) += fc; *w64(W_TOT_U) += fu; *w64(W_FB_C) = 0; *w64(W_FB_U) = 0; misc[W_FB_N] = 0;
            }
        }
        lds_barrier();          // row_cnt / row_off stores stay in flight
        ELBA_DSTAMP(5);
        const uint32_t fits_row = misc[8], ysurv_row = misc[3], dcount_row = misc[DG];
        if (!GLOBAL && tid < 16 && tid != 3u && (tid < 6u ? tid == (DG ^ 1u) || tid == 5u : tid > 8u)) misc[tid] = 0;      // the counters nobody reads any more (and the NEXT row's diagonal word); misc[3] goes with the hand-off
        if (fits_row) {
            // ---- all survivors (and the diagonal, by the lane after the last of them) write their staging records ----
            const unsigned long long off = ((unsigned long long)misc[7] << 32) | misc[6];
            const uint32_t ysurv = ysurv_row;
            const uint32_t hasd = dcount_row >= 2 ? 1u : 0u;
            uint32_t nup = 0, mx = 0, nmir = 0;
            bool drew = false;      // an image of this row took a ticket (no slab yet, or a full one): k_mirror has to walk the row's staged entries (OvParams::tick_rows)
            auto seed_at = [&](uint32_t a, uint32_t &q, uint32_t &t) {      // sequence number -> the two positions (32-bit accumulators only)
                const uint2 ce = csr2[rs + (a >> fbits)];
                if (p.inl != 0u && (ce.y >> 31) != 0u) { q = ce.x & 0xFFFFu; t = ce.x >> 16; return; }      // an inline partner carries both positions
                q = ce.x & pmask;
                t = ell ? (uint32_t)p.a_ell[(unsigned long long)ce.y * stride + (a & fmask)] : (uint32_t)p.a_csc[p.a_colptr[ce.y] + (a & fmask)];
            };
            for (uint32_t t = tid; t < ysurv + hasd; t += BLOCK) {
                elba_seed_t v;
                uint32_t j = i;
                if (t < ysurv) {
                    const uint32_t s0 = GLOBAL ? __hip_atomic_load(&list[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)list16[t];
                    j = tab.ld(tab.keys, s0); v.numshared = (int32_t)tab.ld(tab.cnt, s0);
                    if (SUFFIX && relabel) j = p.row_order[j];      // (the dense path's table is keyed by labels)
                    if (PAY) {
                        const uint32_t va = (uint32_t)tab.vmin[s0], vb = (uint32_t)tab.vmax[s0];
                        v.q0 = va >> 16; v.t0 = va & 0xFFFFu; v.q1 = vb >> 16; v.t1 = vb & 0xFFFFu;
                    } else if (pay16) {
                        // the accumulators hold sequence number << 16 | posT; posQ is the row entry's own position — the entry is named by the sequence number
                        const uint32_t va = tab.ld(tab.smin, s0), vb = tab.ld(tab.smax, s0);
                        const uint2 ca2 = csr2[rs + ((va >> 16) >> fbits)], cb2 = csr2[rs + ((vb >> 16) >> fbits)];
                        v.q0 = (p.inl != 0u && (ca2.y >> 31) != 0u) ? (ca2.x & 0xFFFFu) : (ca2.x & pmask); v.t0 = va & 0xFFFFu;
                        v.q1 = (p.inl != 0u && (cb2.y >> 31) != 0u) ? (cb2.x & 0xFFFFu) : (cb2.x & pmask); v.t1 = vb & 0xFFFFu;
                    } else {
                        seed_at(tab.ld(tab.smin, s0), v.q0, v.t0);
                        seed_at(tab.ld(tab.smax, s0), v.q1, v.t1);
                    }
                    if (!GLOBAL) reset_slot(s0);
                } else {
                    // B(i,i): first / last product of the fold = the row's first / last entry paired with itself (rows are in (kid, pos)
                    // order, columns in (read, pos) order: the first entry of the row is the first of read i in its column)
                    v.numshared = (int32_t)dcount_row;
                    const uint2 e0 = csr2[rs], e1 = csr2[rs + nnz - 1u];
                    v.q0 = v.t0 = (p.inl != 0u && (e0.y >> 31) != 0u) ? (e0.x & 0xFFFFu) : (e0.x & pmask);
                    v.q1 = v.t1 = (p.inl != 0u && (e1.y >> 31) != 0u) ? (e1.x & 0xFFFFu) : (e1.x & pmask);
                }
                // the partner's row gets the mirrored entry: draw its slot there now; k_mirror places it once the row pointers are known
                uint32_t tick = 0xFFFFFFFFu;
                if (p.half && j != i && j >= p.row_lo && j < p.row_hi) {
                    ++nmir;
                    if (slab_q) {
                        // the image goes straight to row j's slab: ONE returning atomic on the row's fill word (slab end << 32 | next free entry) hands it its
                        // place and says whether that is still inside the slab (else it draws a ticket and waits here for k_mirror)
                        const unsigned long long w = atomicAdd(&p.slab_pos[j], 1ull);
                        const uint32_t at = (uint32_t)w;
                        if (at < (uint32_t)(w >> 32)) p.slab[at] = make_uint4(i, v.t0 | v.q0 << 16, v.t1 | v.q1 << 16, (uint32_t)v.numshared);
                        else tick = atomicAdd(&p.low_cnt[j], 1u);
                    } else tick = atomicAdd(&p.low_cnt[j], 1u);
                }
                drew |= tick != 0xFFFFFFFFu;
                if (p.rec16) {
                    p.rec[off + t] = make_uint4(j, v.q0 | v.t0 << 16, v.q1 | v.t1 << 16, (uint32_t)v.numshared);
                    // (round 5: an entry whose image took a ticket — rare: the sample's rows, a full slab — joins a LIST for k_mirror; every staged entry used to
                    //  store its ticket, 49 M four-byte store requests of a kernel that is bound by its request rate)
                    if (tick != 0xFFFFFFFFu) { const uint32_t at = atomicAdd(&p.ctr->ntick, 1u); p.tick[3ull * at] = i; p.tick[3ull * at + 1u] = t; p.tick[3ull * at + 2u] = tick; }
                }
                else {
                    p.tmp[off + t].a = make_uint4(j, tick, v.q0, v.t0);
                    p.tmp[off + t].b = make_uint4(v.q1, v.t1, (uint32_t)v.numshared, 0u);
                }
                // strict-upper entries: an in-window pair accumulated here stands for (i,j) and (j,i) — one of the two is upper
                if (j != i && ((p.half && j >= p.row_lo && j < p.row_hi) || j > i)) ++nup;
                mx = (uint32_t)v.numshared > mx ? (uint32_t)v.numshared : mx;
            }
            if (__ballot(drew) != 0 && lane == 0) atomicOr(&p.tick_rows[i >> 5], 1u << (i & 31u));
            if (nup) lds_add64(w64(W_NUP), (unsigned long long)nup);
            if (nmir) lds_add64(w64(W_MIR), (unsigned long long)nmir);
            if (mx) lds_max32(&misc[W_MX], mx);
        } else if (!GLOBAL) {
            for (uint32_t t = tid; t < ysurv_row; t += BLOCK) reset_slot((uint32_t)list16[t]);      // (no room in the staging area: the call is repeated, the table goes back clean all the same)
        }
        dpar ^= 1u;
        if (GLOBAL) __syncthreads(); else lds_barrier();       // the next row finds table and counters clean; staging stores stay in flight
        ELBA_DSTAMP(6);
    }
#ifdef ELBA_PHASE_CLOCK
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&p.ctr->phase[k], ph[k]);
        atomicAdd(&p.ctr->phase[10], 1ull);
    }
#endif
#undef ELBA_DSTAMP
    // flush the workgroup's statistics
    __syncthreads();
    OvShard *sh = &p.ctr->shard[blockIdx.x & (NUM_SHARDS - 1)];
    if (tid == 0) {
        if (*w64(W_NUP)) atomicAdd(&sh->nupper, *w64(W_NUP));
        if (*w64(W_MIR)) atomicAdd(&sh->nnz, *w64(W_MIR));
        if (misc[W_MX]) atomicMax(&sh->maxshared, misc[W_MX]);
        unsigned long long fc = *w64(W_FB_C), fu = *w64(W_FB_U);
        if (p.use_feedback && misc[W_FB_N]) { atomicAdd(&p.ctr->fb_claims, fc); atomicAdd(&p.ctr->fb_ub, fu); }
        if (sample && *w64(W_ACC_Y)) atomicAdd(&p.ctr->fb_surv, *w64(W_ACC_Y));      // (what sizes the mirror slabs)
        fc += *w64(W_TOT_C); fu += *w64(W_TOT_U);
        if (fu) { atomicAdd(&sh->fb_claims, fc); atomicAdd(&sh->fb_ub, fu); }
        if (misc[W_ACC_DONE]) {
            atomicAdd(&sh->yraw, *w64(W_ACC_YRAW));
            atomicAdd(&sh->nnz, *w64(W_ACC_Y));
            atomicAdd(&sh->products, *w64(W_ACC_P));
            atomicAdd(&sh->tier_done[tier], misc[W_ACC_DONE]);
            if (misc[W_ACC_NDIAG]) atomicAdd(&sh->ndiag, (unsigned long long)misc[W_ACC_NDIAG]);
        }
    }
}

// ---- symbolic: queue every non-empty row of the window on its starting tier, in row order ----------------------------------------
// All that is known of a row before the product is its length: distinct partners are estimated as nnz x prior (1/4 before anything is
// known of the matrix — the kernel corrects itself from the rows already done — afterwards the measured ratio).
__global__ __launch_bounds__(256) void k_classify_direct(OvParams p, int mode)
{
    // mode 1: queue the sample — rows row_lo + q * sstep, q < nsample — and nothing else; mode 0: every (other) row on its starting tier.
    // With a sample the ratio comes from what its rows found (fb_claims / fb_ub: flushed by the sample launch), + 25 %.
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt = (1ull << lane) - 1;
    if (mode == 1) {
        for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < p.nsample; q += stride) {
            const uint32_t i = p.row_lo + q * p.sstep;
            if (i < p.row_hi && p.a_rowptr[i + 1] != p.a_rowptr[i]) p.sample_list[atomicAdd(&p.ctr->sample_count, 1u)] = i;
        }
        return;
    }
    uint32_t prior_q16 = p.prior_q16;
    if (p.nsample) {
        const unsigned long long u = p.ctr->fb_ub, cl = p.ctr->fb_claims;
        if (u) { const double r = 1.25 * (double)cl / (double)u * 65536.0; prior_q16 = r < 64.0 ? 64u : (r > 4.0e9 ? 4000000000u : (uint32_t)r); }
    }
    uint32_t slab_q = 0, slab_rp0 = 0;
    if (p.slab != nullptr) {
        // mirror slabs: slab entries per row entry = mirrored entries per row entry (an earlier call's, or what the sample's rows staged — a row
        // stages about as many entries as it receives) x the margin, cut down to what the slab area holds.  (Every lane computes the same figure
        // from the same words; lane 0 of the first workgroup publishes it for the kernels behind this one.)
        double r = 0.0;
        if (p.slab_prior_q16) r = (double)p.slab_prior_q16;
        else if (p.nsample) { const unsigned long long u = p.ctr->fb_ub; if (u) r = (double)p.ctr->fb_surv / (double)u * 65536.0; }
        r *= (double)p.slab_pct / 100.0;
        const unsigned long long nrows = p.row_hi - p.row_lo, zw = p.a_rowptr[p.row_hi] - p.a_rowptr[p.row_lo], padded = (unsigned long long)SLAB_PAD * nrows;
        uint32_t q = 0;
        if (r >= 1.0 && zw > 0 && p.slab_cap > padded + 1ull) {
            const double most = (double)(p.slab_cap - padded - 1ull) * 65536.0 / (double)zw;
            r = r < most ? r : most;
            q = r >= 4.0e9 ? 4000000000u : (uint32_t)r;
        }
        slab_q = q; slab_rp0 = p.a_rowptr[p.row_lo];
        if (blockIdx.x == 0 && threadIdx.x == 0) p.ctr->slab_q16 = q;
    }
    // (dense path with labels: the rows are queued in label order — reads of one locus are multiplied at the same time and find each other's
    //  columns in the caches; the order names every row of the matrix, those outside the window pass)
    const uint32_t qn = p.row_order ? p.M : p.row_hi - p.row_lo;
    for (uint32_t q0 = blockIdx.x * blockDim.x; q0 < qn; q0 += stride) {      // block-uniform trip count
        const uint32_t q = q0 + threadIdx.x;
        const uint32_t i = q >= qn ? p.row_hi : (p.row_order ? p.row_order[q] : p.row_lo + q);
        int mytier = -1;
        const bool inwin = i >= p.row_lo && i < p.row_hi;
        const uint32_t rp_a = inwin ? p.a_rowptr[i] : 0u, rp_b = inwin ? p.a_rowptr[i + 1] : 0u;
        uint32_t nnz = rp_b - rp_a;
        if (slab_q && inwin)      // the row's slab: end << 32 | first entry (the fill word the numeric kernels add to)
            p.slab_pos[i] = ((unsigned long long)slab_base(rp_b, slab_rp0, i + 1u - p.row_lo, slab_q) << 32) | slab_base(rp_a, slab_rp0, i - p.row_lo, slab_q);
        if (p.nsample && i < p.row_hi && (i - p.row_lo) % p.sstep == 0 && (i - p.row_lo) / p.sstep < p.nsample) nnz = 0;      // a row of the sample: done already
        if (nnz != 0) {
            const unsigned long long prod_ub = (unsigned long long)nnz * p.max_col;
            const uint32_t ub = prod_ub < (unsigned long long)p.Mcols ? (uint32_t)prod_ub : p.Mcols;
            const uint32_t gbits = guaranteed_tbits(ub, p.Mcols);
            uint32_t est = (uint32_t)(((unsigned long long)nnz * prior_q16) >> 16);
            if (est < 64) est = 64;
            int tier = 0;
            while (tier < NUM_LDS_TIERS && est > p.tier_limit[tier]) ++tier;
            const int gt = gbits <= LDS_TBITS0 ? 0 : (int)gbits - LDS_TBITS0;      // never start above the tier that is guaranteed to fit
            if (gt < tier) tier = gt;
            if (tier < (int)p.min_tier) tier = (int)p.min_tier;      // (a large matrix skips the small-table tiers: their handful of rows costs a launch of ~60 us each)
            if (p.suffix && tier < (int)p.dense_up) tier = (int)p.dense_up;      // (dense path: a lower load factor means fewer look-ups that miss their first slot)
            if (tier > NUM_LDS_TIERS) tier = NUM_LDS_TIERS;
            mytier = tier;
        }
        // queue places: ONE returning atomic per workgroup and tier (a wavefront each drew its own: 3000 wavefronts on the one counter of the tier that takes
        // nearly every row, ~15 ns apiece — 55 us for 200 k rows)
        __shared__ uint32_t wcnt[4][NUM_TIERS], wbase[NUM_TIERS];
        const uint32_t wv = threadIdx.x >> 6;
        uint64_t mybal = 0;
#pragma unroll
        for (int t = 0; t < NUM_TIERS; ++t) {
            const uint64_t bal = __ballot(mytier == t);
            if (mytier == t) mybal = bal;
            if (lane == 0) wcnt[wv][t] = (uint32_t)__popcll(bal);
        }
        __syncthreads();
        if (threadIdx.x < NUM_TIERS) {
            uint32_t tot = 0;
            for (int w2 = 0; w2 < 4; ++w2) { const uint32_t x = wcnt[w2][threadIdx.x]; wcnt[w2][threadIdx.x] = tot; tot += x; }
            wbase[threadIdx.x] = tot ? atomicAdd(&p.ctr->tier_count[threadIdx.x], tot) : 0u;
        }
        __syncthreads();
        if (mytier >= 0) p.lists[(size_t)mytier * p.M + wbase[mytier] + wcnt[wv][mytier] + (uint32_t)__popcll(mybal & lt)] = i;
        __syncthreads();      // (the counts are reused by the loop's next trip)
    }
}
