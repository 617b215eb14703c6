// spgemm_table.hpp — the open-addressed accumulator of the overlap SpGEMM and the small LDS helpers around it (included by spgemm.hip
// inside its anonymous namespace, before spgemm_direct.hpp).
// misc words in LDS: 0 diag n, 1 diag smin, 2 diag smax, 3 survivors, 4 y, 5 yraw, 6/7 staging offset lo/hi, 8 fits,
//                    9 claimed slots, 10 abandon flag, 11 row entries consumed when the row was abandoned   (reset per row)
//                    16..19 feedback snapshot; from 32: the workgroup's statistics across rows (W_* below).  They are only ever ADDED to,
//                    by one lane per row: kept in LDS and updated with no-return ds_add they cost no latency, while as registers they
//                    cost every lane of the kernel ~20 VGPRs
enum : uint32_t {
    W_ACC_YRAW = 32 /*u64*/, W_ACC_Y = 34 /*u64*/, W_ACC_DONE = 36, W_ACC_NDIAG = 37, W_FB_N = 38,
    W_FB_C = 40 /*u64*/, W_FB_U = 42 /*u64*/, W_TOT_C = 44 /*u64*/, W_TOT_U = 46 /*u64*/, W_NUP = 48 /*u64*/, W_MIR = 50 /*u64*/, W_MX = 52, W_END = 54
};
// (A/B hook: a pre-mix of the key in front of the multiplicative hash — consecutive labels of the dense path land 316 slots apart on a 512-slot
//  table, i.e. on 8 of the 32 banks; profiles/r03_notes.md)
#if defined(ELBA_HMIX_SEL) && ELBA_HMIX_SEL == 1
#define ELBA_HMIX(j) ((j) ^ ((j) >> 5))
#elif defined(ELBA_HMIX_SEL) && ELBA_HMIX_SEL == 2
#define ELBA_HMIX(j) (((j) * 0x85EBCA6Bu) ^ ((j) >> 3))
#elif defined(ELBA_HMIX_SEL) && ELBA_HMIX_SEL == 3
#define ELBA_HMIX(j) ((j) ^ ((j) << 7))
#else
#define ELBA_HMIX(j) (j)
#endif
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
        uint32_t slot = (ELBA_HMIX(j) * 0x9E3779B1u) >> (32 - tbits);
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
        uint32_t slot = lds_slot(j);
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
        uint32_t slot = lds_slot(j);
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
    // Dense path, the common case of a product: its partner is in the table already and its sequence number is not below the pair's minimum.
    // Dense matrices meet the same hundred partners tens of thousands of times per row and sequence numbers arrive in ascending order per
    // wavefront, so after a slot's first visits this is what every product is.  FOUR look-ups (the partner ids of one 16-byte piece of a column;
    // j == EMPTY: no product in that place) are in flight together, each ONE plain LDS read of key and current minimum (equal addresses are a
    // broadcast where a compare-and-swap serialises), and a hit costs two atomics — count and maximum — where insert_lds spends four and exposes a
    // round trip per product.  No loop, no claim: a product that misses (another key or nothing in its first slot, or a new minimum) is
    // returned in the mask (bit r) and goes through insert_lds (the caller's retry ring).
    // (profiles/r03_notes.md: the dense path was bound by the LDS pipeline, 63 % of its busy cycles bank conflicts, and by instruction issue.)
    __device__ __forceinline__ uint32_t hit4_lds(const uint32_t (&j)[4], uint32_t s0) const
    {
        typedef __attribute__((address_space(3))) uint32_t lds_u32;
        const lds_u32 *lk = (const lds_u32 *)keys;
        const uint32_t T = size(), tb = T * 4;
        uint32_t slot[4], k[4], m[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[r] = lds_slot(j[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            k[r] = __hip_atomic_load(&lk[slot[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            m[r] = __hip_atomic_load(&lk[slot[r] + 2u * T], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t miss = 0;
        const uint32_t base = (uint32_t)(uintptr_t)keys, one = 1u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t s = s0 + (uint32_t)r;
            if (k[r] == j[r] && s >= m[r] && j[r] != EMPTY) {
                uint32_t a;
                asm volatile("v_lshl_add_u32 %[a], %[slot], 2, %[cb]\n\t"
                             "ds_add_u32 %[a], %[one]\n\t"
                             "v_add_u32_e32 %[a], %[tb2], %[a]\n\t"
                             "ds_max_u32 %[a], %[s]\n"
                             : [a] "=&v"(a) : [slot] "v"(slot[r]), [cb] "s"(base + tb), [tb2] "s"(2u * tb), [one] "v"(one), [s] "v"(s) : "memory");
            } else if (j[r] != EMPTY) miss |= 1u << r;
        }
        return miss;
    }
    // The same look-up with its outcome in flags instead of per-lane bits: slot r of the piece takes part when
    // r >= vr (the piece's slots in front of the first owned one do not; vr = 4: none), miss[r] = product r was not settled (a
    // divergent flag lives in a scalar register pair: the caller's ballot of it costs nothing).  TB != 0: the table size is known when the kernel is compiled (T = 1 << TB), the four arrays are then immediate offsets
    // of ONE address per product (keys | counts | minima | maxima, T words apart).
    template <int TB>
    __device__ __forceinline__ void hit4m_lds(const uint32_t (&j)[4], uint32_t s0, uint32_t vr, bool (&miss)[4]) const
    {
        typedef __attribute__((address_space(3))) uint32_t lds_u32;
        const lds_u32 *lk = (const lds_u32 *)keys;
        const uint32_t T = TB ? (1u << TB) : size(), tb = T * 4;
        uint32_t slot[4], k[4], m[4];
#ifndef ELBA_DENSE_NO_CMAX
        uint32_t mx[4];
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[r] = TB ? (ELBA_HMIX(j[r]) * 0x9E3779B1u) >> (32 - (TB ? TB : 1)) : lds_slot(j[r]);
        const uint32_t base = (uint32_t)(uintptr_t)keys, one = 1u;
#ifndef ELBA_DENSE_QUADS      // two look-ups in flight at a time (four: twelve more live registers in a kernel that is held to 64 — 36 bytes of scratch against 16; config 5 at 1/25: 8.89 vs 8.70 ms)
#pragma unroll
        for (int r0 = 0; r0 < 4; r0 += 2) {
#else
        {
        constexpr int r0 = 0;
#endif
#pragma unroll
        for (int r = r0; r < (
#ifndef ELBA_DENSE_QUADS
                              r0 + 2
#else
                              4
#endif
                              ); ++r) {
            k[r] = __hip_atomic_load(&lk[slot[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            m[r] = __hip_atomic_load(&lk[slot[r] + 2u * T], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef ELBA_DENSE_NO_CMAX
            mx[r] = __hip_atomic_load(&lk[slot[r] + 3u * T], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        }
#pragma unroll
        for (int r = r0; r < (
#ifndef ELBA_DENSE_QUADS
                              r0 + 2
#else
                              4
#endif
                              ); ++r) {
            const uint32_t s = s0 + (uint32_t)r;
            const bool v = (uint32_t)r >= vr;
            const bool h = v && k[r] == j[r] && s >= m[r];
            if (h) {
#ifndef ELBA_DENSE_NO_CMAX
                // The largest sequence number is only raised by a product that exceeds what the slot was READ to hold: eight wavefronts share a table and walk the
                // row's chunks in order — the one furthest ahead raises it, the others read a value above theirs (equal addresses of a read broadcast, those of
                // an atomic serialise).  Config 5 at 1/25: numeric 9.50 -> 8.95 ms (round 4 measured 7.3 ms with no ds_max at all: profiles/r04_notes.md).
                if (TB) {
                    const uint32_t a = base + (slot[r] << 2);
                    asm volatile("ds_add_u32 %[a], %[one] offset:%[o1]\n" : : [a] "v"(a), [one] "v"(one), [o1] "n"(4 << TB) : "memory");
                    if (s > mx[r]) asm volatile("ds_max_u32 %[a], %[s] offset:%[o3]\n" : : [a] "v"(a), [s] "v"(s), [o3] "n"(12 << TB) : "memory");
                } else
#endif
                if (TB) {
                    const uint32_t a = base + (slot[r] << 2);
                    asm volatile("ds_add_u32 %[a], %[one] offset:%[o1]\n\t"
                                 "ds_max_u32 %[a], %[s] offset:%[o3]\n"
                                 : : [a] "v"(a), [one] "v"(one), [s] "v"(s), [o1] "n"(4 << TB), [o3] "n"(12 << TB) : "memory");
                } else {
                    uint32_t a;
                    asm volatile("v_lshl_add_u32 %[a], %[slot], 2, %[cb]\n\t"
                                 "ds_add_u32 %[a], %[one]\n\t"
                                 "v_add_u32_e32 %[a], %[tb2], %[a]\n\t"
                                 "ds_max_u32 %[a], %[s]\n"
                                 : [a] "=&v"(a) : [slot] "v"(slot[r]), [cb] "s"(base + tb), [tb2] "s"(2u * tb), [one] "v"(one), [s] "v"(s) : "memory");
                }
            }
            miss[r] = v && !h;
        }
        }
    }
    __device__ __forceinline__ uint32_t lds_slot(uint32_t j) const { return (ELBA_HMIX(j) * 0x9E3779B1u) >> (32 - tbits); }      // (a 24-bit multiply, v_mul_u32_u24, was measured: no difference)
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

