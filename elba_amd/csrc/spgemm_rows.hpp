// spgemm_rows.hpp — the numeric kernel of the overlap SpGEMM (included by spgemm.hip inside its anonymous namespace).
//
// One workgroup per row of B.  What bounds it is instruction issue and dependent LDS/memory round trips at the 8-16 waves per
// CU the LDS tables allow — not bytes and not LDS throughput (profiles/r01_notes.md) — so the kernel is organised to run few,
// fully populated instructions per product and to keep few dependent levels per row:
//   level 0   row id of the NEXT row, this row's 32-byte header (bounds of its entries and descriptors): one round trip;
//   level 1   the row's descriptors (matrix.hip: one 16-byte word per row entry and contiguous range of partner entries in the entry's
//             column, longest ranges first), one per lane, coalesced, prefetched one trip ahead;
//   level 2   PK independent 8-byte gathers per lane (the first PK partner entries of the lane's range; longer ranges take further
//             wave-uniform trips), then the LDS accumulator updates;
//   level 3   one sweep of the table builds the survivor list (ballot + popcount compaction) and the counts;
//   level 4   all survivors decode their two seeds in parallel (canonical arrays), then store.

// misc words in LDS: 0 diag n, 1 diag smin, 2 diag smax, 3 survivors, 4 y, 5 yraw, 6/7 staging offset lo/hi, 8 fits,
//                    9 claimed slots, 10 abandon flag, 11 row entries consumed when the row was abandoned   (reset per row)
//                    16..19 feedback snapshot; from 32: the workgroup's statistics across rows (W_* below).  They are only ever ADDED to,
//                    by one lane per row: kept in LDS and updated with no-return ds_add they cost no latency, while as registers they
//                    cost every lane of the kernel ~20 VGPRs
enum : uint32_t {
    W_ACC_YRAW = 32 /*u64*/, W_ACC_Y = 34 /*u64*/, W_ACC_DONE = 36, W_ACC_NDIAG = 37, W_FB_N = 38,
    W_FB_C = 40 /*u64*/, W_FB_U = 42 /*u64*/, W_TOT_C = 44 /*u64*/, W_TOT_U = 46 /*u64*/, W_NUP = 48 /*u64*/, W_MIR = 50 /*u64*/, W_MX = 52, W_END = 54
};
template <bool GLOBAL>
struct Table {
    uint32_t *keys, *cnt, *smin, *smax, *misc;
    unsigned long long *vmin, *vmax;      // payload-carrying accumulators (insert_lds64): sequence number << 32 | payload, instead of smin / smax
    uint32_t tbits, limit;
    __device__ __forceinline__ uint32_t size() const { return 1u << tbits; }
    // relaxed workgroup-scope atomics, NOT volatile: a volatile access defeats address-space inference and becomes a FLAT
    // load/store, which forces s_waitcnt vmcnt(0) lgkmcnt(0) — every insert then drained all gathers in flight
    __device__ __forceinline__ bool abandoned() const { return !GLOBAL && __hip_atomic_load(&misc[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; }
    __device__ __forceinline__ void insert(uint32_t j, uint32_t s, bool &full) const
    {
        // A lane learns that the table is filling up from the claim counter it bumps when it claims a slot; from then on
        // it inserts nothing (`full` lives in a register: no per-insert LDS read).  Every lane can overshoot by one claim,
        // so at most 3T/4 + BLOCK slots are ever claimed (BLOCK <= T/8): the probe loop always meets an empty slot.
        if (full) return;
        const uint32_t mask = size() - 1;
        uint32_t slot = (j * 0x9E3779B1u) >> (32 - tbits);
        // (A/B measured on MI355X: probing with a plain read before the CAS and guarding min/max with reads is SLOWER —
        //  0.84 vs 0.76 ms per step — the extra dependent LDS round trips cost more than the atomics they save.)
        for (;;) {
            const uint32_t k = atomicCAS(&keys[slot], EMPTY, j);
            if (k == j) break;
            if (k == EMPTY) {
                if (!GLOBAL) { if (atomicAdd(&misc[9], 1u) >= limit) { __hip_atomic_store(&misc[10], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); full = true; } }   // abandon the row
                break;
            }
            slot = (slot + 1) & mask;
        }
        atomicAdd(&cnt[slot], 1u);
        atomicMin(&smin[slot], s);
        atomicMax(&smax[slot], s);
    }
    // LDS tables: the probe loop is hand-scheduled.  The kernel is bound by instruction issue (profiles/r01_notes.md) and the compiler's
    // rendering of a divergent compare-and-swap loop spends most of its instructions on exec-mask bookkeeping (~25 per probe round,
    // ~12 here): lanes leave the loop by dropping out of exec, which is restored at the end.
    // `full` is WAVE-UNIFORM here (a scalar): once the claim counter has reached the limit every further claim of any lane returns a
    // value >= limit, so each wave notices within the insert in which it claims next — at most one claim per lane beyond the limit,
    // hence limit <= T - BLOCK — and a uniform flag costs a scalar branch where a per-lane one costs ten mask instructions per insert.
    // (cnt products of the same pair at once: their count, their smallest and their largest sequence number — 1, s, s for a single product)
    __device__ __forceinline__ void insert_lds(uint32_t j, uint32_t s, uint32_t smx, uint32_t cnt, bool valid, bool &full) const
    {
        if (full) return;
        uint32_t slot = (j * 0x9E3779B1u) >> (32 - tbits);
        uint32_t claimed = 0, old, addr = 0;
        unsigned long long save, t;
        const uint32_t base = (uint32_t)(uintptr_t)keys;        // LDS byte offset (a local address is the low half of its flat form)
        const uint32_t mask = size() - 1, empty = EMPTY;
        const uint32_t tb = size() * 4;                         // byte distance between the four arrays
        if (valid) {
            asm volatile(
                "s_mov_b64 %[save], exec\n"
                ".Lprobe%=:\n\t"
                "v_lshl_add_u32 %[addr], %[slot], 2, %[base]\n\t"
                "ds_cmpst_rtn_b32 %[old], %[addr], %[empty], %[j]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_cmp_eq_u32_e64 %[t], %[old], %[empty]\n\t"
                "v_cndmask_b32_e64 %[cl], %[cl], 1, %[t]\n\t"
                "v_cmp_eq_u32_e32 vcc, %[old], %[j]\n\t"
                "s_or_b64 vcc, vcc, %[t]\n\t"
                "s_andn2_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz .Ldone%=\n\t"
                "v_add_u32_e32 %[slot], 1, %[slot]\n\t"
                "v_and_b32_e32 %[slot], %[mask], %[slot]\n\t"
                "s_branch .Lprobe%=\n"
                ".Ldone%=:\n\t"
                "s_mov_b64 exec, %[save]\n\t"
                "v_add_u32_e32 %[old], %[tb], %[addr]\n\t"
                "ds_add_u32 %[old], %[one]\n\t"
                "v_add_u32_e32 %[old], %[tb], %[old]\n\t"
                "ds_min_u32 %[old], %[s]\n\t"
                "v_add_u32_e32 %[old], %[tb], %[old]\n\t"
                "ds_max_u32 %[old], %[smx]\n"
                : [save] "=&s"(save), [addr] "+v"(addr), [old] "=&v"(old), [t] "=&s"(t), [cl] "+v"(claimed), [slot] "+v"(slot)
                : [base] "s"(base), [empty] "v"(empty), [j] "v"(j), [mask] "s"(mask), [tb] "s"(tb), [one] "v"(cnt), [s] "v"(s), [smx] "v"(smx)
                : "vcc", "memory");
        }
        // claims: every claiming lane adds for itself (same-address LDS atomics serialise in the LDS unit, one cycle each)
        if (__ballot(claimed != 0) != 0) {
            uint32_t prev = 0;
            if (claimed) {
                const uint32_t a9 = (uint32_t)(uintptr_t)&misc[9], one = 1u;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(prev) : "v"(a9), "v"(one) : "memory");
            }
            if (__ballot(claimed != 0 && prev >= limit) != 0) {
                misc[10] = 1u;
                full = true;
            }
        }
    }
    // The same with 64-bit extremes: `lo` / `hi` = sequence number << 32 | payload of the smallest / largest product handed in.  ds_min_u64 /
    // ds_max_u64 keep the payload of the extreme sequence number, so the seed positions travel with the extremes and nothing is
    // looked up after the sweep.  Table layout: keys | cnt (u32) | vmin | vmax (u64).
    __device__ __forceinline__ void insert_lds64(uint32_t j, unsigned long long lo, unsigned long long hi, uint32_t cnt, bool valid, bool &full) const
    {
        if (full) return;
        uint32_t slot = (j * 0x9E3779B1u) >> (32 - tbits);
        uint32_t claimed = 0, old, addr = 0;
        unsigned long long save, t;
        const uint32_t base = (uint32_t)(uintptr_t)keys;
        const uint32_t base64 = (uint32_t)(uintptr_t)vmin;
        const uint32_t mask = size() - 1, empty = EMPTY;
        const uint32_t tb = size() * 4, tb8 = size() * 8;
        if (valid) {
            asm volatile(
                "s_mov_b64 %[save], exec\n"
                ".Lprobe%=:\n\t"
                "v_lshl_add_u32 %[addr], %[slot], 2, %[base]\n\t"
                "ds_cmpst_rtn_b32 %[old], %[addr], %[empty], %[j]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_cmp_eq_u32_e64 %[t], %[old], %[empty]\n\t"
                "v_cndmask_b32_e64 %[cl], %[cl], 1, %[t]\n\t"
                "v_cmp_eq_u32_e32 vcc, %[old], %[j]\n\t"
                "s_or_b64 vcc, vcc, %[t]\n\t"
                "s_andn2_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz .Ldone%=\n\t"
                "v_add_u32_e32 %[slot], 1, %[slot]\n\t"
                "v_and_b32_e32 %[slot], %[mask], %[slot]\n\t"
                "s_branch .Lprobe%=\n"
                ".Ldone%=:\n\t"
                "s_mov_b64 exec, %[save]\n\t"
                "v_add_u32_e32 %[old], %[tb], %[addr]\n\t"
                "ds_add_u32 %[old], %[one]\n\t"
                "v_lshl_add_u32 %[old], %[slot], 3, %[base64]\n\t"
                "ds_min_u64 %[old], %[lo]\n\t"
                "v_add_u32_e32 %[old], %[tb8], %[old]\n\t"
                "ds_max_u64 %[old], %[hi]\n"
                : [save] "=&s"(save), [addr] "+v"(addr), [old] "=&v"(old), [t] "=&s"(t), [cl] "+v"(claimed), [slot] "+v"(slot)
                : [base] "s"(base), [base64] "s"(base64), [empty] "v"(empty), [j] "v"(j), [mask] "s"(mask), [tb] "s"(tb), [tb8] "s"(tb8), [one] "v"(cnt), [lo] "v"(lo), [hi] "v"(hi)
                : "vcc", "memory");
        }
        if (__ballot(claimed != 0) != 0) {
            uint32_t prev = 0;
            if (claimed) {
                const uint32_t a9 = (uint32_t)(uintptr_t)&misc[9], one = 1u;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(prev) : "v"(a9), "v"(one) : "memory");
            }
            if (__ballot(claimed != 0 && prev >= limit) != 0) {
                misc[10] = 1u;
                full = true;
            }
        }
    }
    // Dense data (accurate reads, high UPPER): neighbouring lanes hold neighbouring k-mer columns, whose r-th partner is mostly the SAME
    // read, so a wavefront's 64 updates hit a handful of slots and LDS atomics on one address serialise (measured: the whole kernel
    // ran at the pace of 64-way conflicts, 1 300 products per output entry).  Equal partners in adjacent lanes of a 16-lane row are
    // therefore combined first — a segmented scan over (count, min s, max s) in four DPP row shifts — and only the last lane of each run
    // updates the table, with the run's count and extremes.  Results are identical: add / min / max are associative and commutative.
    __device__ __forceinline__ void insert_runs(uint32_t j, uint32_t s, bool valid, bool &full) const
    {
        if (full) return;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t jv = valid ? j : 0xFFFFFF00u + lane;                 // lanes without a product never join a run (partner ids are < 2^32 - 256)
        const uint32_t jprev = (uint32_t)__builtin_amdgcn_update_dpp((int)~jv, (int)jv, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
        const uint32_t jnext = (uint32_t)__builtin_amdgcn_update_dpp((int)~jv, (int)jv, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
        uint32_t f = jv != jprev ? 1u : 0u, cnt = valid ? 1u : 0u, mn = s, mx = s;
#define ELBA_SEG_STEP(CTRL)                                                                                              \
        {                                                                                                                \
            const uint32_t pc = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, CTRL, 0xf, 0xf, false);                \
            const uint32_t pn = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)mn, CTRL, 0xf, 0xf, false);                \
            const uint32_t px = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, CTRL, 0xf, 0xf, false);                 \
            const uint32_t pf = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)f, CTRL, 0xf, 0xf, false);                  \
            if (!f) { cnt += pc; mn = pn < mn ? pn : mn; mx = px > mx ? px : mx; f = pf; }                                \
        }
        ELBA_SEG_STEP(0x111) ELBA_SEG_STEP(0x112) ELBA_SEG_STEP(0x114) ELBA_SEG_STEP(0x118)
#undef ELBA_SEG_STEP
        insert_lds(j, mn, mx, cnt, valid && jv != jnext, full);
    }
    // two independent inserts with their compare-and-swap round trips in flight together
    __device__ __forceinline__ void insert2(uint32_t j0, uint32_t s0, uint32_t j1, uint32_t s1, bool two, bool &full) const
    {
        if (full) return;
        const uint32_t mask = size() - 1;
        uint32_t a = (j0 * 0x9E3779B1u) >> (32 - tbits), b = (j1 * 0x9E3779B1u) >> (32 - tbits);
        bool da = false, db = !two;
        uint32_t claimed = 0;
        while (!(da && db)) {
            uint32_t ka = j0, kb = j1;
            if (!da) ka = atomicCAS(&keys[a], EMPTY, j0);
            if (!db) kb = atomicCAS(&keys[b], EMPTY, j1);
            if (!da) { if (ka == j0 || ka == EMPTY) { da = true; claimed += ka == EMPTY; } else a = (a + 1) & mask; }
            if (!db) { if (kb == j1 || kb == EMPTY) { db = true; claimed += kb == EMPTY; } else b = (b + 1) & mask; }
        }
        if (!GLOBAL) {
            // claims are counted per wavefront: two ballots, one LDS atomic by the first claiming lane (left to the compiler, a
            // uniform-address atomic with per-lane values becomes a serial loop over the active lanes)
            const uint64_t b1 = __ballot(claimed != 0), b2 = __ballot(claimed == 2);
            if (b1 != 0) {
                const uint32_t total = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
                const uint32_t first = (uint32_t)__builtin_ctzll(b1);
                uint32_t old = 0;
                if ((threadIdx.x & 63u) == first) old = atomicAdd(&misc[9], total);
                old = (uint32_t)__builtin_amdgcn_readlane((int)old, (int)first);
                if (old + total > limit) { if ((threadIdx.x & 63u) == first) __hip_atomic_store(&misc[10], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); full = true; }
            }
        }
        atomicAdd(&cnt[a], 1u); atomicMin(&smin[a], s0); atomicMax(&smax[a], s0);
        if (two) { atomicAdd(&cnt[b], 1u); atomicMin(&smin[b], s1); atomicMax(&smax[b], s1); }
    }
    __device__ __forceinline__ uint32_t ldrelaxed(const uint32_t *a) const
    {
        return __hip_atomic_load(a, __ATOMIC_RELAXED, GLOBAL ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ uint32_t ld(const uint32_t *a, uint32_t slot) const
    {
        if (GLOBAL) return __hip_atomic_load(&a[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2, never a stale L1 line
        return a[slot];
    }
};

#ifndef ELBA_PIPE
#define ELBA_PIPE 3    // 1: one trip requested ahead; 2: two (93 VGPRs, 7-10 % slower as written); 3: drain-aware order (see the accumulate loop)
#endif
#ifndef ELBA_PK
#define ELBA_PK 4
#endif
constexpr int PIPE = ELBA_PIPE;   // descriptor trips in flight per lane (2: the trip after next is gathered while this one accumulates)
constexpr int PK = ELBA_PK;
#ifndef ELBA_DENSE_LANES
#define ELBA_DENSE_LANES 24
#endif
constexpr int DENSE_LANES = ELBA_DENSE_LANES;   // lanes (of 64) repeating their left neighbour's partner from which a trip combines runs before the table (65: never)      // partner entries a lane gathers per descriptor and trip (descriptor -> gathers -> accumulator)

// s = canonical rank of the row entry << fbits | index inside the column.  a_dec[rs + rank] holds the entry's position in the read
// and the address of its column in a_cscp (still warm in L2: the numeric loop has just gathered it): two loads on two levels per seed
__device__ __forceinline__ elba_seed_t decode_seed(const OvParams &p, uint32_t rs, uint32_t a, uint32_t b, uint32_t n, uint32_t fmask)
{
    const uint64_t ea = p.a_dec[rs + (a >> p.fbits)], eb = p.a_dec[rs + (b >> p.fbits)];
    elba_seed_t v;
    v.q0 = (uint32_t)ea; v.t0 = (uint32_t)p.a_cscp[(uint32_t)(ea >> 32) + (a & fmask)];
    v.q1 = (uint32_t)eb; v.t1 = (uint32_t)p.a_cscp[(uint32_t)(eb >> 32) + (b & fmask)];
    v.numshared = (int32_t)n;
    return v;
}

// no-return LDS atomics, written out: a C++ atomicAdd on an LDS word whose address is uniform is rewritten by the compiler into a scalar
// loop over the active lanes (its "atomic optimizer"), dozens of instructions where one is meant
__device__ __forceinline__ void lds_add32(uint32_t *w, uint32_t v) { asm volatile("ds_add_u32 %0, %1" : : "v"((uint32_t)(uintptr_t)w), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_max32(uint32_t *w, uint32_t v) { asm volatile("ds_max_u32 %0, %1" : : "v"((uint32_t)(uintptr_t)w), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_add64(unsigned long long *w, unsigned long long v) { asm volatile("ds_add_u64 %0, %1" : : "v"((uint32_t)(uintptr_t)w), "v"(v) : "memory"); }

// workgroup-uniform values belong in scalar registers
__device__ __forceinline__ uint32_t sfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t sfirst64(uint64_t v) { return ((uint64_t)sfirst((uint32_t)(v >> 32)) << 32) | sfirst((uint32_t)v); }

// workgroup barrier that orders LDS only: global stores of this row (staging, row_cnt/row_off) may still be in flight — nobody in
// the workgroup reads them back, and __syncthreads() would wait for their acknowledgement (a full memory round trip per row)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DIAG = true compiles the diagnostic ablations and the phase clock in (cfg.flags != 0); the production instantiation carries none of it
// PAY = true (LDS tiers up to 4096 slots, matrices whose formats carry the positions: p.pay_pb != 0): the accumulators are 64 bits wide and
// hold the seed positions of the extreme products beside their sequence numbers (Table::insert_lds64) — the two dependent rounds of
// seed-decoding loads after the sweep, half of the kernel's read requests, disappear.
template <int BLOCK, bool GLOBAL, bool DIAG, bool PAY = false>
__global__ __launch_bounds__(BLOCK) void k_spgemm_rows(OvParams p, int tier, uint32_t lds_tbits)
{
    static_assert(!PAY || (!GLOBAL && !DIAG), "payload accumulators: production LDS tiers only");
    const uint32_t dbg = DIAG ? p.dbg : 0u;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // LDS tiers: keys | cnt | smin | smax (u32 each, T entries) | survivor list (u16: T <= 8192) | misc  = 18 B per slot, so that an
    // 8192-slot table still fits the 160 KB of a CU;  spill tier: only misc lives in LDS.
    // PAY: keys | cnt (u32) | vmin | vmax (u64) | survivor list | misc = 26 B per slot
    uint32_t *misc = GLOBAL ? smem : smem + (size_t)(PAY ? 6 : 4) * (1u << lds_tbits) + ((size_t)1 << lds_tbits) / 2;
    // per-wave product queue: 64 lanes x SPEC products, partner id and sequence number
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint64_t lt = (1ull << lane) - 1;
    const uint32_t nrows = p.ctr->tier_count[tier];      // complete: every lower tier has finished (same stream)
    const uint32_t fmask = (1u << p.fbits) - 1;
    // the staging chunk cursor is needed by value every row: registers; the statistics live in LDS (see W_*), flushed once at the end
    unsigned long long chunk_off = 0;
    uint32_t chunk_left = 0;
    auto w64 = [&](uint32_t k) { return reinterpret_cast<unsigned long long *>(&misc[k]); };
    if (tid >= 32 && tid < W_END) misc[tid] = 0;
    __syncthreads();
    // diagnostic phase clock (cfg.flags & 16): 0 fetch row, 1 table init, 2 expand+accumulate, 3 sweep, 4 reserve, 5 decode+store
    const bool stamp = DIAG && (dbg & 16u) != 0;
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 6 gather wait, 7 count+scan+queue write, 8 insert loop, 9 loop tail
    unsigned long long tprev = stamp ? __builtin_amdgcn_s_memtime() : 0;
#define ELBA_STAMP(k) do { if (stamp) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += tn - tprev; tprev = tn; } } while (0)

    uint32_t fb_seen = 0;

    // ---- software pipeline ACROSS rows -------------------------------------------------------------------------------------------
    // A row costs ~6 dependent memory round trips (id -> header -> descriptors -> partner entries ... -> seed decoding, 2 levels)
    // and only ~12 rows per CU are in flight.  The next row's header is therefore loaded while this row accumulates, its first
    // descriptors while this row's table is swept, and their partner entries while this row's seeds are decoded: a row with at most
    // BLOCK descriptors meets no memory wait before its table is complete.  `dc` / `ce` are free between the accumulate loop and the
    // end of the row and carry the prefetch.
    struct RowHdr { uint32_t i, rs, nnz, hs, nd, work, own0, ownl; };
    // a header is requested early (two vector loads, nothing waits) and moved to scalar registers where it is first needed
    uint4 hva = make_uint4(0u, 0u, 0u, 0u), hvb = hva;
    auto request_hdr = [&](uint32_t i) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p.a_hdr + sfirst(i));
        hva = q[0]; hvb = q[1];
    };
    auto take_hdr = [&](uint32_t i) {
        RowHdr h;
        h.i = sfirst(i); h.rs = sfirst(hva.x); h.nnz = sfirst(hva.y); h.hs = sfirst(hva.z); h.nd = sfirst(hva.w); h.work = sfirst(hvb.x); h.own0 = sfirst(hvb.y); h.ownl = sfirst(hvb.z);
        return h;
    };
    // One descriptor per lane and trip: x = address of the range's first partner entry, y = sequence number of its first product,
    // z = entries in the range, w = the diagonal's share (run of the row's own read in that column, minus one).  Lanes beyond the row's
    // last descriptor hold an empty range.  A row's descriptors come longest range first, so the lanes of a wavefront walk (nearly)
    // equally long ranges and the trips beyond the first PK entries are wave-uniform and rare.
    // Two trips are kept in flight: (dc, ce) is the trip being accumulated, (d1, ce1) the one after it.
    uint4 dc = make_uint4(0u, 0u, 0u, 0u), d1 = dc;
    uint32_t ce[PK], ce1[PK];              // partner reads of the trip's first PK range entries
    // A descriptor travels raw (d1, d2: in flight) and is unpacked where it is first used (dc): unpacking at the load would put the
    // wait for the load right behind it.  Packed form: 8 bytes, field widths from the matrix (matrix.hip); otherwise the 16-byte form.
    const bool packed = p.hot_xb != 0;
    const uint32_t xb = p.hot_xb, yb = p.hot_yb, zb = p.hot_zb;
    auto load_desc = [&](uint32_t hs, uint32_t at, uint32_t nd) {
        const uint32_t t = at + tid;
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (t < nd) {
            if (packed) { const uint2 q = reinterpret_cast<const uint2 *>(p.a_hot8)[hs + t]; r.x = q.x; r.y = q.y; }
            else r = reinterpret_cast<const uint4 *>(p.a_hot)[hs + t];
        }
        return r;
    };
    // (w: the diagonal's share in the low 16 bits; with position-carrying formats, p.pay_pb != 0, the row entry's position in the read above them)
    const uint32_t pb = p.pay_pb, pmask = (1u << pb) - 1u;
    auto unpack = [&](const uint4 &r) {
        if (!packed) return r;
        const uint64_t q = ((uint64_t)r.y << 32) | r.x;
        uint4 d;
        d.x = (uint32_t)q & (uint32_t)((1ull << xb) - 1);
        d.y = (uint32_t)(q >> xb) & (uint32_t)((1ull << yb) - 1);
        d.z = (uint32_t)(q >> (xb + yb)) & ((1u << zb) - 1u);
        const uint32_t wq = (uint32_t)(q >> (xb + yb + zb));
        d.w = pb ? ((wq & ((1u << zb) - 1u)) | (wq >> zb) << 16) : wq;
        return d;
    };
    // ONE 16-byte load per lane: the partner reads of four consecutive range entries (a_cscj holds nothing else; the load is 4-byte
    // aligned and may run past the range — the surplus is never used, and the array ends in guard entries).  Only the lanes whose
    // range reaches entry r0 load: every divergent lane costs the texture unit a cycle.
    static_assert(PK == 4, "one 16-byte load carries PK = 4 partner reads");
    struct __attribute__((packed, aligned(4))) Quad { uint32_t a, b, c, d; };
    auto gather = [&](uint32_t *out, const uint4 &d, uint32_t r0) {
        Quad q{0u, 0u, 0u, 0u};
        if (r0 < d.z) q = *reinterpret_cast<const Quad *>(p.a_cscj + d.x + r0);
        out[0] = q.a; out[1] = q.b; out[2] = q.c; out[3] = q.d;
        if (DIAG && (dbg & 2u)) {                                   // ablation: synthetic partner ids
#pragma unroll
            for (int k = 0; k < PK; ++k) out[k] = ((d.x + r0 + (uint32_t)k) * 2654435761u) % p.Mcols;
        }
    };
    const uint32_t *queue = p.lists + (size_t)tier * p.M;
    RowHdr cur{}, nxt{};
    uint32_t id_n = 0;
    // The queue is (roughly) in descending-work order and every workgroup takes a static share of it.  With a plain grid stride the
    // first workgroup would get the heaviest row of EVERY round; taking the rounds alternately forwards and backwards (snake order)
    // evens the shares out without any run-time queueing.
#ifndef ELBA_SNAKE
#define ELBA_SNAKE 1
#endif
    const uint32_t G = gridDim.x, bx = blockIdx.x;
    auto qidx = [&](uint32_t r) -> unsigned long long { return (unsigned long long)r * G + ((ELBA_SNAKE && (r & 1u)) ? (G - 1u - bx) : bx); };
    if (qidx(0) < nrows) {
        const uint32_t id0 = queue[qidx(0)];
        request_hdr(id0);
        cur = take_hdr(id0);
        if (qidx(1) < nrows) id_n = queue[qidx(1)];
        dc = unpack(load_desc(cur.hs, 0u, cur.nd)); if (PIPE >= 2) d1 = load_desc(cur.hs, BLOCK, cur.nd);
        gather(ce, dc, 0u); if (PIPE == 2) gather(ce1, unpack(d1), 0u);
    }
    for (uint32_t rnd = 0; qidx(rnd) < nrows; ++rnd) {
        const bool has_n = qidx(rnd + 1) < nrows;
        if (has_n) request_hdr(id_n);                                           // header of the next row (taken after the accumulate loop); id of the one after it
        const uint32_t id_nn = qidx(rnd + 2) < nrows ? queue[qidx(rnd + 2)] : 0u;
        const uint32_t i = cur.i, rs = cur.rs, hs = cur.hs, nd = cur.nd;
        const uint32_t ub_i = cur.work;                      // products of the row's descriptors: bounds its distinct partners
        // leaves this row early: the next row's prefetch is issued back to back
#define ELBA_NEXT_ROW() do { if (has_n) { nxt = take_hdr(id_n); dc = unpack(load_desc(nxt.hs, 0u, nxt.nd)); if (PIPE >= 2) d1 = load_desc(nxt.hs, BLOCK, nxt.nd); gather(ce, dc, 0u); if (PIPE == 2) gather(ce1, unpack(d1), 0u); } cur = nxt; id_n = id_nn; } while (0)
        if (!GLOBAL && p.use_feedback) {
            // Self-correction inside a call: rows already done (here or on lower tiers) tell how many distinct partners a product
            // brings on THIS data; a row that is predicted not to fit is forwarded without an attempt.
            // (the sums live on one hot L2 line: lane 0 reads them once per 8 rows of this workgroup and broadcasts through LDS —
            //  the decision below must be workgroup-uniform, so every lane has to see the SAME snapshot)
            if ((fb_seen++ & 7u) == 0) {
                if (tid == 0) {
                    const unsigned long long u = __hip_atomic_load(&p.ctr->fb_ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long c = __hip_atomic_load(&p.ctr->fb_claims, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    misc[16] = (uint32_t)u; misc[17] = (uint32_t)(u >> 32); misc[18] = (uint32_t)c; misc[19] = (uint32_t)(c >> 32);
                }
                __syncthreads();
            }
            const unsigned long long gu = ((unsigned long long)misc[17] << 32) | misc[16], gc = ((unsigned long long)misc[19] << 32) | misc[18];
            if (gu >= (1ull << 20)) {
                const double pred = 1.25 * (double)ub_i * (double)gc / (double)gu;
                // forward only on strong evidence (short rows finish first and have a higher partner/product ratio: the running sums
                // are biased high early in a call): predicted partners beyond 1.2x the abandon limit
                if (pred > 1.2 * (double)p.tier_limit[tier] && guaranteed_tbits(ub_i, p.Mcols) > lds_tbits) {
                    int t2 = tier + 1;
                    while (t2 < NUM_LDS_TIERS && pred > (double)p.tier_limit[t2]) ++t2;
                    if (tid == 0) {
                        const uint32_t at = atomicAdd(&p.ctr->tier_count[t2], 1u);
                        p.lists[(size_t)t2 * p.M + at] = i;
                    }
                    ELBA_NEXT_ROW();
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
            tab.limit = p.tier_limit[tier];          // abandon point: at most limit + 2*BLOCK slots are ever claimed, < T
            tab.keys = smem; tab.cnt = smem + T; tab.smin = smem + 2 * T; tab.smax = smem + 3 * T;
            tab.vmin = reinterpret_cast<unsigned long long *>(smem + 2 * T); tab.vmax = tab.vmin + T;
            list16 = reinterpret_cast<uint16_t *>(smem + (PAY ? 6 : 4) * T);
        }
        const uint32_t T = tab.size();
        ELBA_STAMP(0);
        if (!(DIAG && (dbg & 64u)))                      // (64: ablation, no table initialisation)
        for (uint32_t s = tid; s < T; s += BLOCK) {
            tab.keys[s] = EMPTY; tab.cnt[s] = 0;
            if (PAY) { tab.vmin[s] = ~0ull; tab.vmax[s] = 0ull; } else { tab.smin[s] = 0xFFFFFFFFu; tab.smax[s] = 0; }
        }
        // the diagonal of B(i,i): every row entry pairs with the run of read i in its own column.  Count = nnz + the descriptors' w
        // (added below); first / last product of the fold = first / last row entry at its own place in its column
        if (tid < 16) misc[tid] = tid == 0 ? cur.nnz : (tid == 1 ? cur.own0 : (tid == 2 ? (((cur.nnz - 1u) << p.fbits) | cur.ownl) : 0u));
        // (the LDS tiers use barriers that order LDS only: prefetched loads stay in flight across them; the spill tier's table is
        //  global memory and keeps full barriers)
        if (GLOBAL) __syncthreads(); else if (!(DIAG && (dbg & 512u))) lds_barrier();      // (512: ablation, no workgroup barriers in the row loop)
        ELBA_STAMP(1);

        // ---- accumulate: one descriptor per lane and trip, two trips in flight ----
        // The first two trips' descriptors and partner entries are already in registers or in flight (prefetched during the previous
        // row).  Per trip: the descriptor two trips ahead is requested, the accumulator takes this trip's products, the pipeline
        // advances and the gathers of the trip after next are issued: they have a whole trip's accumulator updates to arrive.
        bool full = false;
        uint32_t dup = 0;
        if (PIPE == 3 && !GLOBAL && !(DIAG && (dbg & 3u))) {
            // Drain-aware order.  Across control flow the compiler consumes a load behind `s_waitcnt vmcnt(0)`, which also waits for every load
            // issued since — so loads issued BEFORE a consuming point are not overlapped with anything.  Each trip therefore (1) consumes
            // what the previous trip requested — this trip's partner entries and the next trip's descriptor: ONE wait — then (2) requests
            // the next trip's partner entries and the descriptor after next, and only then (3) runs its accumulator updates, which touch
            // LDS only: the requests have the whole update phase to land.
#pragma unroll 1
            for (uint32_t t0 = 0; t0 < nd; t0 += BLOCK) {
                const bool more = t0 + BLOCK < nd, more2 = t0 + 2 * BLOCK < nd;
                uint32_t jv[PK];
#pragma unroll
                for (int k = 0; k < PK; ++k) jv[k] = ce[k];
                const uint32_t c = dc.z, sy = dc.y, x0 = dc.x, qs = (dc.w >> 16) << 16;      // qs: the row entry's position, where the payload wants it
                dup += dc.w & 0xFFFFu;
                uint4 draw = d1;
                static_assert(PK == 4, "the consume point names PK registers");
                asm volatile("" : "+v"(jv[0]), "+v"(jv[1]), "+v"(jv[2]), "+v"(jv[3]), "+v"(draw.x) : : "memory");      // (1) everything requested so far has landed
                const uint4 dnx = unpack(draw);
                uint4 d2 = make_uint4(0u, 0u, 0u, 0u);
                if (more) gather(ce, dnx, 0u);                                                                       // (2)
                if (more2) d2 = load_desc(hs, t0 + 2 * BLOCK, nd);
                if (PAY) {                                                                                           // (3)
                    // gathered word = partner read << pb | its position: sequence number, then both positions, into the 64-bit extremes
#pragma unroll
                    for (int k = 0; k < PK; ++k) {
                        const unsigned long long v = ((unsigned long long)(sy + (uint32_t)k) << 32) | (qs | (jv[k] & pmask));
                        tab.insert_lds64(jv[k] >> pb, v, v, 1u, (uint32_t)k < c, full);
                    }
                } else {
                    if (pb) {
#pragma unroll
                        for (int k = 0; k < PK; ++k) jv[k] >>= pb;
                    }
                    const uint32_t jl = (uint32_t)__builtin_amdgcn_update_dpp((int)~jv[0], (int)jv[0], 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                    if (__popcll(__ballot(0u < c && jv[0] == jl)) >= DENSE_LANES) {
#pragma unroll
                        for (int k = 0; k < PK; ++k) tab.insert_runs(jv[k], sy + (uint32_t)k, (uint32_t)k < c, full);
                    } else {
#pragma unroll
                        for (int k = 0; k < PK; ++k) { const uint32_t sq = sy + (uint32_t)k; tab.insert_lds(jv[k], sq, sq, 1u, (uint32_t)k < c, full); }
                    }
                }
#pragma unroll 1
                for (uint32_t r0 = PK; __ballot(c > r0) != 0; r0 += PK) {      // ranges longer than PK entries: they come first in a row
                    uint32_t cx[PK];
                    { uint4 dx = make_uint4(x0, 0u, c, 0u); gather(cx, dx, r0); }
#pragma unroll
                    for (int k = 0; k < PK; ++k) {
                        const uint32_t sq = sy + r0 + (uint32_t)k;
                        if (PAY) { const unsigned long long v = ((unsigned long long)sq << 32) | (qs | (cx[k] & pmask)); tab.insert_lds64(cx[k] >> pb, v, v, 1u, r0 + (uint32_t)k < c, full); }
                        else tab.insert_lds(cx[k] >> pb, sq, sq, 1u, r0 + (uint32_t)k < c, full);
                    }
                }
                if (tab.abandoned()) {
                    if (tid == 0) { const uint32_t done = t0 + BLOCK; misc[11] = done < nd ? done : nd; }
                    break;
                }
                dc = dnx; d1 = d2;
            }
        } else
#pragma unroll 1
        for (uint32_t t0 = 0; t0 < nd; t0 += BLOCK) {
            uint4 d2 = make_uint4(0u, 0u, 0u, 0u);
            const bool more = t0 + BLOCK < nd, more2 = t0 + 2 * BLOCK < nd;
            if (PIPE == 2) { if (more2) d2 = load_desc(hs, t0 + 2 * BLOCK, nd); }
            else if (more) d2 = load_desc(hs, t0 + BLOCK, nd);
            ELBA_STAMP(6);
            dup += dc.w & 0xFFFFu;
            const uint32_t c = dc.z;
            if (pb) {                                       // position-carrying words: this path keeps the partner reads only
#pragma unroll
                for (int k = 0; k < PK; ++k) ce[k] >>= pb;
            }
#pragma unroll 1
            for (uint32_t r0 = 0;;) {
                if (DIAG && (dbg & 1u)) {                                   // ablation: gathers only, keep the loads alive
                    uint32_t sink = 0;
#pragma unroll
                    for (int k = 0; k < PK; ++k) sink ^= ce[k];
                    if (sink == 0xFFFFFFFFu) misc[15] = 1;
                } else if (GLOBAL) {
#pragma unroll
                    for (int k = 0; k < PK; k += 2) {
                        if (r0 + (uint32_t)k < c)
                            tab.insert2(ce[k], dc.y + r0 + (uint32_t)k, ce[k + 1], dc.y + r0 + (uint32_t)k + 1u, r0 + (uint32_t)k + 1u < c, full);
                    }
                } else {
                    // dense trip?  (most lanes hold the same partner as their left neighbour: decided per trip and wavefront, on the first entries)
                    const uint32_t j0 = ce[0];
                    const uint32_t jl = (uint32_t)__builtin_amdgcn_update_dpp((int)~j0, (int)j0, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                    if (__popcll(__ballot(r0 < c && j0 == jl)) >= DENSE_LANES) {
#pragma unroll
                        for (int k = 0; k < PK; ++k) tab.insert_runs(ce[k], dc.y + r0 + (uint32_t)k, r0 + (uint32_t)k < c, full);
                    } else {
#pragma unroll
                        for (int k = 0; k < PK; ++k) { const uint32_t sq = dc.y + r0 + (uint32_t)k; tab.insert_lds(ce[k], sq, sq, 1u, r0 + (uint32_t)k < c, full); }
                    }
                }
                r0 += PK;
                if (__ballot(c > r0) == 0) break;                            // wave-uniform: ranges longer than PK entries come first in a row
                gather(ce, dc, r0);
                if (pb) {
#pragma unroll
                    for (int k = 0; k < PK; ++k) ce[k] >>= pb;
                }
            }
            ELBA_STAMP(8);
            if (tab.abandoned()) {
                if (tid == 0) { const uint32_t done = t0 + BLOCK; misc[11] = done < nd ? done : nd; }
                break;
            }
            if (more) {
                if (PIPE == 2) {
                    dc = unpack(d1);
#pragma unroll
                    for (int k = 0; k < PK; ++k) ce[k] = ce1[k];
                    d1 = d2;
                    if (more2) gather(ce1, unpack(d1), 0u);
                } else {
                    dc = unpack(d2);
                    gather(ce, dc, 0u);
                }
            }
        }
        if (__ballot(dup != 0) != 0) {                      // the diagonal's count: nnz (in misc[0] already) + the runs beyond the entry itself
            dup = wave_sum_u32(dup);
            if (lane == 0) lds_add32(&misc[0], dup);
        }
        if (GLOBAL) __syncthreads(); else if (!(DIAG && (dbg & 512u))) lds_barrier();
        ELBA_STAMP(2);
        if (tab.abandoned()) {
            // the optimistic table was too small: hand the row to the next tier (its kernel starts after this one ends)
            if (tid == 0) {
                const uint32_t at = atomicAdd(&p.ctr->tier_count[tier + 1], 1u);
                p.lists[(size_t)(tier + 1) * p.M + at] = i;
                // the table filled after `done` of the row's descriptors: extrapolate its distinct-partner count for the feedback
                const unsigned long long all = nd ? nd : 1u, done = misc[11] ? misc[11] : all;
                lds_add64(w64(W_FB_C), (unsigned long long)misc[9] * all / done); lds_add64(w64(W_FB_U), (unsigned long long)ub_i); lds_add32(&misc[W_FB_N], 1u);
            }
            __syncthreads();
            ELBA_NEXT_ROW();
            continue;
        }
        uint4 dc_raw = make_uint4(0u, 0u, 0u, 0u);
        if (has_n) { nxt = take_hdr(id_n); dc_raw = load_desc(nxt.hs, 0u, nxt.nd); if (PIPE >= 2) d1 = load_desc(nxt.hs, BLOCK, nxt.nd); }   // next row, first two trips' descriptors: in flight during the sweep

        // ---- level 3: one table sweep: nnz before prune + ballot-compacted survivor list ----
        uint32_t yraw = 0;
        for (uint32_t b0 = (DIAG && (dbg & 128u)) ? T : 0u; b0 < T; b0 += BLOCK) {                       // wave-uniform trip count: ballots are safe  (128: ablation, no sweep)
            const uint32_t s0 = b0 + tid;
            bool keep = false;
            if (s0 < T) {
                const uint32_t j = tab.ld(tab.keys, s0);
                // a pair whose partner row is computed here too (half schedule) stands for both (i,j) and (j,i)
                if (j != EMPTY) { yraw += (p.half && j >= p.row_lo && j < p.row_hi) ? 2u : 1u; keep = tab.ld(tab.cnt, s0) >= 2; }
            }
            const uint64_t bal = __ballot(keep);
            if (bal == 0) continue;
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&misc[3], (uint32_t)__popcll(bal));
            at = __shfl(at, 0, 64) + (uint32_t)__popcll(bal & lt);
            if (keep) { if (GLOBAL) list[at] = s0; else list16[at] = (uint16_t)s0; }
        }
        yraw = wave_sum_u32(yraw);
        if (lane == 0 && yraw) atomicAdd(&misc[5], yraw);
        if (GLOBAL) __syncthreads(); else if (!(DIAG && (dbg & 512u))) lds_barrier();
        ELBA_STAMP(3);
        if (tid == 0) {
            const uint32_t dcount = misc[0];
            const uint32_t ytot = misc[3] + (dcount >= 2 ? 1u : 0u);
            // staging space: the workgroup draws CHUNK-sized pieces from the global cursor and sub-allocates its rows
            // from them (one hot 64-bit counter sustains ~10^8 atomics/s; one atomic per row would cap the kernel)
            unsigned long long off;
            if (ytot <= chunk_left) { off = chunk_off; chunk_off += ytot; chunk_left -= ytot; }
            else if (ytot >= STAGE_CHUNK / 2) off = atomicAdd(&p.ctr->cursor, (unsigned long long)ytot);
            else { off = atomicAdd(&p.ctr->cursor, (unsigned long long)STAGE_CHUNK); chunk_off = off + ytot; chunk_left = STAGE_CHUNK - ytot; }
            const bool fits = off + ytot <= p.tmp_cap;
            if (!fits) atomicOr(&p.ctr->overflow, 1u);
            p.row_cnt[i] = ytot;
            p.row_off[i] = off;
            misc[6] = (uint32_t)off; misc[7] = (uint32_t)(off >> 32); misc[8] = fits ? 1u : 0u;
            lds_add64(w64(W_ACC_YRAW), (unsigned long long)(misc[5] + (dcount >= 1 ? 1u : 0u)));
            lds_add32(&misc[W_ACC_DONE], 1u);
            lds_add32(&misc[W_ACC_NDIAG], dcount >= 2 ? 1u : 0u);
            lds_add64(w64(W_ACC_Y), (unsigned long long)ytot);
            if (!GLOBAL) { lds_add64(w64(W_FB_C), (unsigned long long)misc[9]); lds_add64(w64(W_FB_U), (unsigned long long)ub_i); lds_add32(&misc[W_FB_N], 1u); }
            if (p.use_feedback && misc[W_FB_N] >= 8) {          // first call for a matrix only: push this workgroup's share to the hot sums
                const unsigned long long fc = *w64(W_FB_C), fu = *w64(W_FB_U);
                atomicAdd(&p.ctr->fb_claims, fc); atomicAdd(&p.ctr->fb_ub, fu);
                *w64(W_TOT_C) += fc; *w64(W_TOT_U) += fu; *w64(W_FB_C) = 0; *w64(W_FB_U) = 0; misc[W_FB_N] = 0;
            }
        }
        if (!(DIAG && (dbg & 512u))) lds_barrier();          // row_cnt / row_off stores stay in flight
        ELBA_STAMP(4);
        if (has_n) { dc = unpack(dc_raw); gather(ce, dc, 0u); if (PIPE == 2) gather(ce1, unpack(d1), 0u); }    // next row, first two trips' partner entries: in flight during the decode
        if (misc[8]) {
            // ---- level 4: all survivors decode their seeds in parallel ----
            const unsigned long long off = ((unsigned long long)misc[7] << 32) | misc[6];
            const uint32_t ysurv = misc[3];
            uint32_t nup = 0, mx = 0, nmir = 0;
            // the diagonal entry (its count / min / max come with A) is decoded in the same pass, by the lane after the last survivor
            const uint32_t hasd = misc[0] >= 2 ? 1u : 0u;
            for (uint32_t t = (DIAG && (dbg & 256u)) ? ysurv + hasd : tid; t < ysurv + hasd; t += BLOCK) {      // (256: ablation, no decode / staging stores)
                uint32_t j = i, n = misc[0], a = misc[1], b = misc[2];
                unsigned long long va = 0, vb = 0;
                if (t < ysurv) {
                    const uint32_t s0 = GLOBAL ? __hip_atomic_load(&list[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)list16[t];
                    j = tab.ld(tab.keys, s0); n = tab.ld(tab.cnt, s0);
                    if (PAY) { va = tab.vmin[s0]; vb = tab.vmax[s0]; } else { a = tab.ld(tab.smin, s0); b = tab.ld(tab.smax, s0); }
                }
                // the partner's row gets the mirrored entry: draw its slot there now (the round trip overlaps the decode's loads);
                // k_mirror places it once the row pointers are known
                uint32_t tick = 0xFFFFFFFFu;
                if (p.half && j != i && j >= p.row_lo && j < p.row_hi) { tick = (DIAG && (dbg & 8u)) ? 0u : atomicAdd(&p.low_cnt[j], 1u); ++nmir; }     // (8: ablation, no ticket atomics)
                elba_seed_t v;
                if (DIAG && (dbg & 4u)) { v.q0 = a; v.t0 = b; v.q1 = a; v.t1 = b; v.numshared = (int32_t)n; }          // ablation: no seed decoding loads
                else if (PAY && t < ysurv) {                // the positions came with the extremes: payload = position in this read << 16 | position in the partner
                    v.q0 = (uint32_t)va >> 16; v.t0 = (uint32_t)va & 0xFFFFu; v.q1 = (uint32_t)vb >> 16; v.t1 = (uint32_t)vb & 0xFFFFu; v.numshared = (int32_t)n;
                    // (a partner position that did not fit the gathered word arrives as all ones: that seed is looked up — rare by construction, matrix.hip)
                    if (v.t0 == pmask || v.t1 == pmask) v = decode_seed(p, rs, (uint32_t)(va >> 32), (uint32_t)(vb >> 32), n, fmask);
                }
                else v = decode_seed(p, rs, a, b, n, fmask);       // (the diagonal entry, one per row, is still looked up)
                p.tmp[off + t].a = make_uint4(j, tick, v.q0, v.t0);
                p.tmp[off + t].b = make_uint4(v.q1, v.t1, (uint32_t)v.numshared, 0u);
                if (j > i) ++nup;
                mx = n > mx ? n : mx;
            }
            if (nup) lds_add64(w64(W_NUP), (unsigned long long)nup);
            if (nmir) lds_add64(w64(W_MIR), (unsigned long long)nmir);
            if (mx) lds_max32(&misc[W_MX], mx);
        }
        if (!(DIAG && (dbg & 512u))) lds_barrier();       // table and misc are re-initialised by the next row; staging stores stay in flight
        cur = nxt; id_n = id_nn;
        ELBA_STAMP(5);
    }
#undef ELBA_NEXT_ROW
    if (stamp && tid == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) atomicAdd(&p.ctr->phase[k], ph[k]);
        atomicAdd(&p.ctr->phase[10], 1ull);
    }
#undef ELBA_STAMP
    // flush the workgroup's statistics: a handful of atomics per workgroup instead of six per row
    __syncthreads();
    OvShard *sh = &p.ctr->shard[blockIdx.x & (NUM_SHARDS - 1)];
    if (tid == 0) {
        if (*w64(W_NUP)) atomicAdd(&sh->nupper, *w64(W_NUP));
        if (*w64(W_MIR)) atomicAdd(&sh->nnz, *w64(W_MIR));
        if (misc[W_MX]) atomicMax(&sh->maxshared, misc[W_MX]);
        unsigned long long fc = *w64(W_FB_C), fu = *w64(W_FB_U);
        if (p.use_feedback && misc[W_FB_N]) { atomicAdd(&p.ctr->fb_claims, fc); atomicAdd(&p.ctr->fb_ub, fu); }
        fc += *w64(W_TOT_C); fu += *w64(W_TOT_U);
        if (fu) { atomicAdd(&sh->fb_claims, fc); atomicAdd(&sh->fb_ub, fu); }
        if (misc[W_ACC_DONE]) {
            atomicAdd(&sh->yraw, *w64(W_ACC_YRAW));
            atomicAdd(&sh->nnz, *w64(W_ACC_Y));
            atomicAdd(&sh->tier_done[tier], misc[W_ACC_DONE]);
            if (misc[W_ACC_NDIAG]) atomicAdd(&sh->ndiag, (unsigned long long)misc[W_ACC_NDIAG]);
        }
    }
}
