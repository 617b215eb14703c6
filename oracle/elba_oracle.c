/*
 * oracle/elba_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of ELBA's overlap-detection hot path,
 * used as the checker by tests/, by __graft_entry__.smoke() and as the `cpu_baseline` leg of
 * bench.py.  The product (elba_amd/csrc, libelba_amd.so) never includes, links or calls this.
 *
 * All file:line citations are into /root/reference (PASSIONLab/ELBA @ v2).
 *
 * PARITY PINNING (what this oracle has been checked against; see tests/test_oracle_*.py):
 *   a1  2-bit encode ............ PINNED: reference DnaSeq::compress compiled from source (oracle/_ref) + committed vectors
 *   a2  pack / roll / twin / rep  PINNED: reference Kmer<1> compiled from source (oracle/_ref) + committed vectors
 *   a3  murmur3 x64_128 seed 313  PINNED: reference HashFuncs.cpp compiled from source (oracle/_ref) + committed vectors
 *   a4  owner ................... restated from src/KmerOps.cpp:352-359 (KmerOps.cpp needs CombBLAS: unbuildable here)
 *   a7/a8 reliable k-mers ....... PINNED two ways: (i) a replay of KmerOps.cpp's two-pass control flow on the reference's own
 *                                 Bloom + Kmer code (oracle/ref_shim.cpp: ref_replay_count); (ii) the counts the survey measured
 *                                 from the reference's KmerOps.cpp on its bundled reads.fa (I, N, Z, dups)
 *   a12 SharedSeeds semiring .... restated from include/SharedSeeds.hpp:36-58 (header needs CombBLAS types: unbuildable here);
 *                                 fold shapes checked against SURVEY.md App. B ("left fold p1..p4 -> {(1,1),(4,4),4}")
 *   a13 B = A*A^T + prune ....... pattern, numshared, P, Y_raw, Y, diagonal, strict-upper, max numshared PINNED by SURVEY.md
 *                                 App. B's figures for reads.fa at (17,2,8) and (31,15,35).
 *                                 SEED VALUES: **parity unpinned** — the fold order lives in CombBLAS (un-vendored, un-pinned,
 *                                 absent; github.com/PASSIONLab/CombBLAS, plain clone per usage.txt:5-6).  The canonical rule of
 *                                 SURVEY.md §8c-2 is implemented: k-mer ids = rank of the packed canonical k-mer value; seeds[0] =
 *                                 product with minimal (kid,posQ,posT), seeds[1] = maximal — i.e. what an ascending-k left fold
 *                                 of Semiring::add produces.
 *
 * Scope of this restatement: k <= 32 (NLONGS == 1, include/Kmer.hpp:95-97), LOWER >= 2.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * a1. 2-bit encoding.  include/DnaSeq.hpp:136-154 (codetab), src/DnaSeq.cpp:7-29 (compress).
 * ---------------------------------------------------------------------------------------------- */
static uint8_t orc_charcode(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;            /* "X": the reference ORs the overflowing code in (UB by its own comment) */
    }
}

size_t orc_bytes_needed(size_t len) { return (len + 3) / 4; }   /* include/DnaSeq.hpp:131 */

/* Writes (len+3)/4 bytes; first base in bits 7-6; unused low bits of the last byte are zero. */
size_t orc_encode_read(const char *s, size_t len, uint8_t *mem)
{
    size_t nbytes = orc_bytes_needed(len);
    int remain = (int)(4 * nbytes - len);
    for (size_t b = 0; b < nbytes; ++b) {
        uint8_t byte = 0;
        int left = (b != nbytes - 1) ? 4 : 4 - remain;
        for (int i = 0; i < left; ++i) {
            uint8_t code = orc_charcode((unsigned char)s[4 * b + i]);
            uint8_t shift = (uint8_t)(code << (6 - 2 * i));      /* truncated to 8 bits like the reference's uint8_t */
            byte |= shift;
        }
        mem[b] = byte;
    }
    return nbytes;
}

static inline int orc_base_at(const uint8_t *mem, size_t i)     /* src/DnaSeq.cpp:48-54 */
{
    return (mem[i / 4] >> (6 - 2 * (i % 4))) & 3;
}

/* ------------------------------------------------------------------------------------------------
 * a2. Packed k-mers, k <= 32.  src/Kmer.cpp:67-87 (set_kmer), :149-165 (GetExtension),
 *     :167-198 (GetTwin), :200-205 (GetRep), :118-131 (operator<).
 *     Base i sits at bits 2*(31-i); the low 64-2k bits are zero.
 * ---------------------------------------------------------------------------------------------- */
uint64_t orc_kmer_from_ascii(const char *s, int k)
{
    uint64_t w = 0;
    for (int i = 0; i < k; ++i) w |= (uint64_t)orc_charcode((unsigned char)s[i]) << (2 * (31 - i));
    return w;
}

uint64_t orc_kmer_extend(uint64_t w, int code, int k)           /* roll one base in */
{
    return (w << 2) | ((uint64_t)code << (2 * (32 - (k % 32))));
}

uint64_t orc_kmer_twin(uint64_t w, int k)
{
    /* reverse the 32 two-bit groups and complement them (what the 256-entry tetramer table + byte reversal do) */
    uint64_t x = ~w;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    /* x now holds the reverse complement of all 32 groups; the k real bases are the LOW 2k bits' mirror:
       the (32-k) padding groups (zero -> complemented to 3 -> reversed) occupy the TOP; shift them out. */
    int shift = (k % 32) ? 2 * (32 - (k % 32)) : 0;
    return x << shift;
}

uint64_t orc_kmer_rep(uint64_t w, int k)
{
    uint64_t t = orc_kmer_twin(w, k);
    return t < w ? t : w;
}

/* Two-word k-mers, 32 < k <= 64 (NLONGS == 2, include/Kmer.hpp:95-97): base i at bits 2*(31 - i%32) of longs[i/32]; twin and rep
 * as above over 128 bits; rep compares longs[0] first (src/Kmer.cpp:118-131).  Built base by base: the oracle favours the obvious. */
/* General form, NLONGS = ceil(k / 32) <= 3 words (k < 96, include/compiletime.h:10): out[0..nl-1], unused words zero. */
void orc_kmerN_at(const uint8_t *mem, size_t pos, int k, uint64_t out[3])
{
    uint64_t f[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    for (int i = 0; i < k; ++i) {
        const uint64_t b = (uint64_t)((mem[(pos + (size_t)i) / 4] >> (6 - 2 * ((pos + (size_t)i) % 4))) & 3);
        f[i / 32] |= b << (2 * (31 - i % 32));
        const int j = k - 1 - i;
        t[j / 32] |= (3 - b) << (2 * (31 - j % 32));
    }
    int twin_smaller = 0;
    for (int w = 0; w < 3; ++w) { if (t[w] != f[w]) { twin_smaller = t[w] < f[w]; break; } }
    for (int w = 0; w < 3; ++w) out[w] = twin_smaller ? t[w] : f[w];
}

void orc_kmer2_at(const uint8_t *mem, size_t pos, int k, uint64_t out[2])
{
    uint64_t f[2] = {0, 0}, t[2] = {0, 0};
    for (int i = 0; i < k; ++i) {
        const uint64_t b = (uint64_t)((mem[(pos + (size_t)i) / 4] >> (6 - 2 * ((pos + (size_t)i) % 4))) & 3);
        f[i / 32] |= b << (2 * (31 - i % 32));
        const int j = k - 1 - i;                                   /* base i of the forward strand is base k-1-i of the twin, complemented */
        t[j / 32] |= (3 - b) << (2 * (31 - j % 32));
    }
    const int twin_smaller = t[0] < f[0] || (t[0] == f[0] && t[1] < f[1]);
    out[0] = twin_smaller ? t[0] : f[0]; out[1] = twin_smaller ? t[1] : f[1];
}

/* ------------------------------------------------------------------------------------------------
 * a3. MurmurHash3 x64_128, seed 313, first word.  src/HashFuncs.cpp:40-117, :231-236; src/Kmer.cpp:207-213.
 *     Key = the 8 bytes of the packed k-mer as stored (little-endian u64).
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k;
}

void orc_murmur3_x64_128(const void *key, uint32_t len, uint32_t seed, uint64_t out[2])
{
    const uint8_t *data = (const uint8_t *)key;
    const uint32_t nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (uint32_t i = 0; i < nblocks; ++i) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8); memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + 16 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    uint32_t rem = len & 15;
    for (uint32_t b = rem; b > 8; --b) k2 ^= (uint64_t)tail[b - 1] << (8 * (b - 9));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (uint32_t b = (rem > 8 ? 8 : rem); b > 0; --b) k1 ^= (uint64_t)tail[b - 1] << (8 * (b - 1));
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

uint64_t orc_kmer_hash(uint64_t w)
{
    uint64_t o[2];
    orc_murmur3_x64_128(&w, 8, 313, o);
    return o[0];
}

/* a4. src/KmerOps.cpp:352-359. */
int orc_kmer_owner(uint64_t hash, int nprocs)
{
    double range = (double)hash * (double)nprocs;
    size_t owner = (size_t)(range / (double)UINT64_MAX);
    return (int)owner;
}

/* ------------------------------------------------------------------------------------------------
 * a12. SharedSeeds + Semiring.  include/SharedSeeds.hpp:8-58.  Explicit field order (never the
 *      std::tuple memory layout).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t q0, t0, q1, t1; int32_t numshared; } orc_seed_t;

orc_seed_t orc_sr_multiply(uint32_t a, uint32_t b)               /* :48-52 */
{
    orc_seed_t r = { a, b, 0, 0, 1 };
    return r;
}
orc_seed_t orc_sr_add(orc_seed_t l, orc_seed_t r)                /* :41-46 */
{
    orc_seed_t o = { l.q0, l.t0, r.q0, r.t0, l.numshared + r.numshared };
    return o;
}

/* ------------------------------------------------------------------------------------------------
 * The oracle context: reads -> reliable k-mers -> A (CSC + CSR) -> B.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int k, lower, upper;
    int64_t M;                 /* reads (rows of A, rows/cols of B) */
    int64_t first_read_id;
    /* k-mer stage */
    int64_t I;                 /* k-mer instances */
    int64_t N;                 /* reliable k-mers */
    int64_t Z;                 /* nnz(A) */
    int64_t ndistinct;         /* distinct canonical k-mers */
    uint64_t *kmers;           /* [N] packed canonical k-mer values, ascending == k-mer id order */
    uint64_t *kmers_lo;        /* [N] second word of the k-mers when k > 32 (NULL otherwise) */
    uint64_t *kmers_lo2;       /* [N] third word when k > 64 */
    int64_t *colptr;           /* [N+1] */
    uint32_t *csc_read;        /* [Z] local read index, within column sorted by (read,pos) */
    uint32_t *csc_pos;         /* [Z] */
    int64_t *rowptr;           /* [M+1] */
    uint32_t *csr_kid;         /* [Z] within row sorted by (kid,pos) */
    uint32_t *csr_pos;         /* [Z] */
    int64_t *hist;             /* [upper+2] histogram of column counts (src/main.cpp:449-485) */
    /* B */
    int64_t P, Yraw, Y, ndiag, nupper, maxshared;
    int64_t *b_rowptr;         /* [M+1] */
    uint32_t *b_col;           /* [Y] ascending within row */
    orc_seed_t *b_val;         /* [Y] */
} orc_ctx;

orc_ctx *orc_create(int k, int lower, int upper)
{
    if (k < 3 || k > 95 || !(k & 1) || lower < 1 || lower > upper || upper > 65535) return NULL;  /* include/compiletime.h:10,21 */
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->k = k; c->lower = lower; c->upper = upper;
    return c;
}

static void orc_free_A(orc_ctx *c)
{
    free(c->kmers); free(c->kmers_lo); free(c->kmers_lo2); free(c->colptr); free(c->csc_read); free(c->csc_pos);
    free(c->rowptr); free(c->csr_kid); free(c->csr_pos); free(c->hist);
    c->kmers = NULL; c->kmers_lo = NULL; c->kmers_lo2 = NULL; c->colptr = NULL; c->csc_read = c->csc_pos = NULL; c->rowptr = NULL; c->csr_kid = c->csr_pos = NULL; c->hist = NULL;
}
static void orc_free_B(orc_ctx *c)
{
    free(c->b_rowptr); free(c->b_col); free(c->b_val);
    c->b_rowptr = NULL; c->b_col = NULL; c->b_val = NULL;
}
void orc_destroy(orc_ctx *c) { if (!c) return; orc_free_A(c); orc_free_B(c); free(c); }

/* Enumerate the canonical k-mers of one packed read (a2; include/KmerOps.hpp:105-137 ForeachKmer:
 * reads shorter than k contribute nothing; position = forward start index). Returns count. */
int64_t orc_read_kmers(const uint8_t *mem, uint32_t len, int k, uint64_t *out)
{
    if ((int64_t)len < k) return 0;
    uint64_t w = 0;
    for (int i = 0; i < k; ++i) w |= (uint64_t)orc_base_at(mem, (size_t)i) << (2 * (31 - i));
    int64_t n = (int64_t)len - k + 1;
    out[0] = orc_kmer_rep(w, k);
    for (int64_t i = 1; i < n; ++i) {
        w = orc_kmer_extend(w, orc_base_at(mem, (size_t)(i + k - 1)), k);
        out[i] = orc_kmer_rep(w, k);
    }
    return n;
}

/* Test support for checks at sizes where the whole pipeline is too slow on the host: the canonical k-mer instances (k <= 31) of the reads
 * [r0, r1) whose VALUE CLASS — the right-aligned 2k-bit value modulo 4096 — is set in the 4096-bit bitmap `classes`; the enumeration itself is
 * orc_read_kmers' (include/KmerOps.hpp:105-137).  vals == NULL: count only.  Returns the number of instances (written: at most cap). */
int64_t orc_enumerate_classes(const uint8_t *buf, const uint64_t *byte_off, const uint32_t *lens, int64_t r0, int64_t r1, int k, const uint64_t *classes,
                              uint64_t *vals, uint32_t *reads, uint32_t *pos, int64_t cap)
{
    int64_t n = 0;
    for (int64_t r = r0; r < r1; ++r) {
        const uint8_t *mem = buf + byte_off[r];
        const uint32_t len = lens[r];
        if ((int64_t)len < k) continue;
        uint64_t w = 0;
        for (int i = 0; i < k; ++i) w |= (uint64_t)orc_base_at(mem, (size_t)i) << (2 * (31 - i));
        const int64_t ni = (int64_t)len - k + 1;
        for (int64_t i = 0; i < ni; ++i) {
            if (i) w = orc_kmer_extend(w, orc_base_at(mem, (size_t)(i + k - 1)), k);
            const uint64_t v = orc_kmer_rep(w, k) >> (64 - 2 * k);
            const uint32_t c = (uint32_t)(v & 4095u);
            if ((classes[c >> 6] >> (c & 63u)) & 1ull) {
                if (vals && n < cap) { vals[n] = v; reads[n] = (uint32_t)r; pos[n] = (uint32_t)i; }
                ++n;
            }
        }
    }
    return n;
}

typedef struct { uint64_t kmer, kmer2, kmer3; uint32_t read, pos; } orc_inst_t;      /* kmer2 / kmer3: second / third word when k > 32 / 64, else 0 */

/* stable LSD radix sort of instances by the k-mer value (11-bit digits): the bits of the second word first (k > 32), then the first word's */
static int orc_sort_word(orc_inst_t **pa, orc_inst_t **ptmp, int64_t n, int word, int shift0, int bits)
{
    orc_inst_t *a = *pa, *tmp = *ptmp;
    int passes = 0;
    for (int lo = 0; lo < bits; lo += 11, ++passes) {
        int64_t cnt[2049];
        memset(cnt, 0, sizeof(cnt));
#define ORC_WORD(x) (word == 0 ? (x).kmer : (word == 1 ? (x).kmer2 : (x).kmer3))
        for (int64_t i = 0; i < n; ++i) cnt[(((ORC_WORD(a[i]) >> shift0) >> lo) & 2047) + 1]++;
        for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; ++i) tmp[cnt[((ORC_WORD(a[i]) >> shift0) >> lo) & 2047]++] = a[i];
#undef ORC_WORD
        orc_inst_t *t = a; a = tmp; tmp = t;
    }
    *pa = a; *ptmp = tmp;
    return passes;
}
/* returns the buffer that holds the sorted instances */
static orc_inst_t *orc_sort_instances(orc_inst_t *a, orc_inst_t *tmp, int64_t n, int k)
{
    if (k > 64) { orc_sort_word(&a, &tmp, n, 2, 64 - 2 * (k - 64), 2 * (k - 64)); orc_sort_word(&a, &tmp, n, 1, 0, 64); orc_sort_word(&a, &tmp, n, 0, 0, 64); }
    else if (k > 32) { orc_sort_word(&a, &tmp, n, 1, 64 - 2 * (k - 32), 2 * (k - 32)); orc_sort_word(&a, &tmp, n, 0, 0, 64); }
    else orc_sort_word(&a, &tmp, n, 0, 64 - 2 * k, 2 * k);
    return a;
}

/*
 * a7 + a8 + a9 + a10 in one call: exact counting of canonical k-mers over all reads, keep those with
 * LOWER <= count <= UPPER (SURVEY.md App. A.4 — the net effect of src/KmerOps.cpp:158-187,283-318,335-340 for
 * LOWER >= 2), k-mer ids by ascending packed value, every instance of a reliable k-mer becomes an entry
 * (read, kid, pos) — duplicates within a read are KEPT (src/KmerOps.cpp:400 SumDuplicates=false).
 * Builds CSC (the reference's AT, src/main.cpp:272-273) and CSR (the reference's A).
 */
int orc_count_and_build(orc_ctx *c, const uint8_t *buf, const uint64_t *byte_off, const uint32_t *lens, int64_t nreads)
{
    orc_free_A(c); orc_free_B(c);
    const int k = c->k;
    c->M = nreads;
    int64_t I = 0;
    for (int64_t r = 0; r < nreads; ++r) if ((int64_t)lens[r] >= k) I += (int64_t)lens[r] - k + 1;
    c->I = I;
    orc_inst_t *a = (orc_inst_t *)malloc((size_t)(I > 0 ? I : 1) * sizeof(orc_inst_t));
    orc_inst_t *b = (orc_inst_t *)malloc((size_t)(I > 0 ? I : 1) * sizeof(orc_inst_t));
    uint32_t maxlen = 0;
    for (int64_t r = 0; r < nreads; ++r) if (lens[r] > maxlen) maxlen = lens[r];
    uint64_t *scratch = (uint64_t *)malloc((size_t)(maxlen + 1) * sizeof(uint64_t));
    if (!a || !b || !scratch) { free(a); free(b); free(scratch); return -1; }
    int64_t z = 0;
    for (int64_t r = 0; r < nreads; ++r) {
        if (k > 32) {
            const int64_t n = (int64_t)lens[r] >= k ? (int64_t)lens[r] - k + 1 : 0;
            for (int64_t p = 0; p < n; ++p) { uint64_t w[3]; orc_kmerN_at(buf + byte_off[r], (size_t)p, k, w); a[z].kmer = w[0]; a[z].kmer2 = w[1]; a[z].kmer3 = w[2]; a[z].read = (uint32_t)r; a[z].pos = (uint32_t)p; ++z; }
        } else {
            int64_t n = orc_read_kmers(buf + byte_off[r], lens[r], k, scratch);
            for (int64_t p = 0; p < n; ++p) { a[z].kmer = scratch[p]; a[z].kmer2 = 0; a[z].kmer3 = 0; a[z].read = (uint32_t)r; a[z].pos = (uint32_t)p; ++z; }
        }
    }
    free(scratch);
    orc_inst_t *s = orc_sort_instances(a, b, I, k);        /* sorted by (kmer, read, pos): the sort is stable */

    /* run-length pass 1: count N, Z */
    int64_t N = 0, Z = 0, nd = 0;
    c->hist = (int64_t *)calloc((size_t)c->upper + 2, sizeof(int64_t));
    for (int64_t i = 0; i < I; ) {
        int64_t j = i + 1;
        while (j < I && s[j].kmer == s[i].kmer && s[j].kmer2 == s[i].kmer2 && s[j].kmer3 == s[i].kmer3) ++j;
        int64_t cnt = j - i;
        ++nd;
        if (cnt >= c->lower && cnt <= c->upper) { ++N; Z += cnt; c->hist[cnt]++; }
        i = j;
    }
    c->N = N; c->Z = Z; c->ndistinct = nd;
    c->kmers = (uint64_t *)malloc((size_t)(N + 1) * sizeof(uint64_t));
    c->kmers_lo = k > 32 ? (uint64_t *)malloc((size_t)(N + 1) * sizeof(uint64_t)) : NULL;
    c->kmers_lo2 = k > 64 ? (uint64_t *)malloc((size_t)(N + 1) * sizeof(uint64_t)) : NULL;
    c->colptr = (int64_t *)malloc((size_t)(N + 1) * sizeof(int64_t));
    c->csc_read = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csc_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->rowptr = (int64_t *)calloc((size_t)(nreads + 2), sizeof(int64_t));
    c->csr_kid = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csr_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    int64_t kid = 0, e = 0;
    for (int64_t i = 0; i < I; ) {
        int64_t j = i + 1;
        while (j < I && s[j].kmer == s[i].kmer && s[j].kmer2 == s[i].kmer2 && s[j].kmer3 == s[i].kmer3) ++j;
        int64_t cnt = j - i;
        if (cnt >= c->lower && cnt <= c->upper) {
            c->kmers[kid] = s[i].kmer;
            if (c->kmers_lo) c->kmers_lo[kid] = s[i].kmer2;
            if (c->kmers_lo2) c->kmers_lo2[kid] = s[i].kmer3;
            c->colptr[kid] = e;
            for (int64_t t = i; t < j; ++t) { c->csc_read[e] = s[t].read; c->csc_pos[e] = s[t].pos; c->rowptr[s[t].read + 1]++; ++e; }
            ++kid;
        }
        i = j;
    }
    c->colptr[N] = e;
    free(a); free(b);
    /* CSR by a stable counting transpose of the CSC stream: rows come out sorted by (kid, pos). */
    for (int64_t r = 0; r < nreads; ++r) c->rowptr[r + 1] += c->rowptr[r];
    int64_t *cur = (int64_t *)malloc((size_t)(nreads + 1) * sizeof(int64_t));
    memcpy(cur, c->rowptr, (size_t)(nreads + 1) * sizeof(int64_t));
    for (int64_t kk = 0; kk < N; ++kk)
        for (int64_t t = c->colptr[kk]; t < c->colptr[kk + 1]; ++t) {
            int64_t d = cur[c->csc_read[t]]++;
            c->csr_kid[d] = (uint32_t)kk; c->csr_pos[d] = c->csc_pos[t];
        }
    free(cur);
    return 0;
}

/*
 * The same on `nthreads` host threads (bench.py's all-cores CPU figure for the k-mer stage; the one-thread function above stays the plain statement
 * and tests/test_oracle_golden.py holds the two against each other).  Same algorithm, cut by k-mer VALUE: reads are split into contiguous ranges
 * (one per thread, balanced on instances); every thread counts, then writes, its instances into 256 buckets by the leading 8 value bits — thread t's
 * share of a bucket lies behind thread t-1's, so a bucket holds its instances in (read, pos) order as the one-thread pass does; buckets are sorted
 * (the same stable LSD sort), run-length counted and emitted independently; k-mer ids and column pointers come from a prefix over the buckets.  CSR: the
 * same stable counting transpose, columns split over the threads with per-thread row counts.
 */
int orc_count_and_build_mt(orc_ctx *c, const uint8_t *buf, const uint64_t *byte_off, const uint32_t *lens, int64_t nreads, int nthreads)
{
    if (nthreads <= 1) return orc_count_and_build(c, buf, byte_off, lens, nreads);
    orc_free_A(c); orc_free_B(c);
    const int k = c->k, T = nthreads;
    enum { NBK = 256 };
    c->M = nreads;
    int64_t I = 0;
    uint32_t maxlen = 0;
    for (int64_t r = 0; r < nreads; ++r) { if ((int64_t)lens[r] >= k) I += (int64_t)lens[r] - k + 1; if (lens[r] > maxlen) maxlen = lens[r]; }
    c->I = I;
    orc_inst_t *a = (orc_inst_t *)malloc((size_t)(I > 0 ? I : 1) * sizeof(orc_inst_t));
    orc_inst_t *b = (orc_inst_t *)malloc((size_t)(I > 0 ? I : 1) * sizeof(orc_inst_t));
    int64_t *rbound = (int64_t *)malloc((size_t)(T + 1) * sizeof(int64_t));          /* read ranges of the threads */
    int64_t *cnt = (int64_t *)calloc((size_t)T * NBK, sizeof(int64_t));              /* [thread][bucket] */
    int64_t *bstart = (int64_t *)calloc(NBK + 1, sizeof(int64_t));
    if (!a || !b || !rbound || !cnt || !bstart) { free(a); free(b); free(rbound); free(cnt); free(bstart); return -1; }
    {
        int64_t acc = 0; int t = 0;
        rbound[0] = 0;
        for (int64_t r = 0; r < nreads; ++r) {
            if ((int64_t)lens[r] >= k) acc += (int64_t)lens[r] - k + 1;
            while (t + 1 < T && acc >= (I * (t + 1)) / T && r + 1 <= nreads) rbound[++t] = r + 1;
        }
        while (t < T) rbound[++t] = nreads;
    }
#define ORC_BK(w0) ((int)((w0) >> 56))
    int fail = 0;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        uint64_t *scratch = (uint64_t *)malloc((size_t)(maxlen + 1) * sizeof(uint64_t));
        if (!scratch) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
            int64_t *mycnt = cnt + (size_t)t * NBK;
            for (int64_t r = rbound[t]; r < rbound[t + 1]; ++r) {
                if (k > 32) {
                    const int64_t n = (int64_t)lens[r] >= k ? (int64_t)lens[r] - k + 1 : 0;
                    for (int64_t p = 0; p < n; ++p) { uint64_t w[3]; orc_kmerN_at(buf + byte_off[r], (size_t)p, k, w); mycnt[ORC_BK(w[0])]++; }
                } else {
                    const int64_t n = orc_read_kmers(buf + byte_off[r], lens[r], k, scratch);
                    for (int64_t p = 0; p < n; ++p) mycnt[ORC_BK(scratch[p])]++;
                }
            }
        }
#pragma omp barrier
#pragma omp single
        {
            int64_t run = 0;
            for (int bk = 0; bk < NBK; ++bk) {
                bstart[bk] = run;
                for (int tt = 0; tt < T; ++tt) { const int64_t x = cnt[(size_t)tt * NBK + bk]; cnt[(size_t)tt * NBK + bk] = run; run += x; }      /* cnt becomes the write cursor */
            }
            bstart[NBK] = run;
        }
        if (!fail) {
            int64_t *cur = cnt + (size_t)t * NBK;
            for (int64_t r = rbound[t]; r < rbound[t + 1]; ++r) {
                if (k > 32) {
                    const int64_t n = (int64_t)lens[r] >= k ? (int64_t)lens[r] - k + 1 : 0;
                    for (int64_t p = 0; p < n; ++p) {
                        uint64_t w[3]; orc_kmerN_at(buf + byte_off[r], (size_t)p, k, w);
                        orc_inst_t *d = &a[cur[ORC_BK(w[0])]++];
                        d->kmer = w[0]; d->kmer2 = w[1]; d->kmer3 = w[2]; d->read = (uint32_t)r; d->pos = (uint32_t)p;
                    }
                } else {
                    const int64_t n = orc_read_kmers(buf + byte_off[r], lens[r], k, scratch);
                    for (int64_t p = 0; p < n; ++p) {
                        orc_inst_t *d = &a[cur[ORC_BK(scratch[p])]++];
                        d->kmer = scratch[p]; d->kmer2 = 0; d->kmer3 = 0; d->read = (uint32_t)r; d->pos = (uint32_t)p;
                    }
                }
            }
        }
        free(scratch);
    }
    if (fail) { free(a); free(b); free(rbound); free(cnt); free(bstart); return -1; }
    /* every bucket on its own: sort (the number of passes does not depend on the bucket: all of them end in the same buffer), count */
    int64_t bN[NBK + 1], bZ[NBK + 1], bD[NBK];
    int64_t *hists = (int64_t *)calloc((size_t)NBK * ((size_t)c->upper + 2), sizeof(int64_t));
    orc_inst_t *sorted_in = NULL;
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
    for (int bk = 0; bk < NBK; ++bk) {
        const int64_t n = bstart[bk + 1] - bstart[bk];
        orc_inst_t *s = orc_sort_instances(a + bstart[bk], b + bstart[bk], n, k);
        if (bk == 0) sorted_in = (s == a + bstart[bk]) ? a : b;
        int64_t N = 0, Z = 0, nd = 0;
        for (int64_t i = 0; i < n; ) {
            int64_t j = i + 1;
            while (j < n && s[j].kmer == s[i].kmer && s[j].kmer2 == s[i].kmer2 && s[j].kmer3 == s[i].kmer3) ++j;
            const int64_t cn = j - i;
            ++nd;
            if (cn >= c->lower && cn <= c->upper) { ++N; Z += cn; hists[(size_t)bk * ((size_t)c->upper + 2) + (size_t)cn]++; }
            i = j;
        }
        bN[bk] = N; bZ[bk] = Z; bD[bk] = nd;
    }
    int64_t N = 0, Z = 0, nd = 0;
    for (int bk = 0; bk < NBK; ++bk) { const int64_t x = bN[bk], y = bZ[bk]; bN[bk] = N; bZ[bk] = Z; N += x; Z += y; nd += bD[bk]; }
    bN[NBK] = N; bZ[NBK] = Z;
    c->N = N; c->Z = Z; c->ndistinct = nd;
    c->hist = (int64_t *)calloc((size_t)c->upper + 2, sizeof(int64_t));
    for (int bk = 0; bk < NBK; ++bk) for (int64_t q = 0; q < (int64_t)c->upper + 2; ++q) c->hist[q] += hists[(size_t)bk * ((size_t)c->upper + 2) + (size_t)q];
    free(hists);
    c->kmers = (uint64_t *)malloc((size_t)(N + 1) * sizeof(uint64_t));
    c->kmers_lo = k > 32 ? (uint64_t *)malloc((size_t)(N + 1) * sizeof(uint64_t)) : NULL;
    c->kmers_lo2 = k > 64 ? (uint64_t *)malloc((size_t)(N + 1) * sizeof(uint64_t)) : NULL;
    c->colptr = (int64_t *)malloc((size_t)(N + 1) * sizeof(int64_t));
    c->csc_read = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csc_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->rowptr = (int64_t *)calloc((size_t)(nreads + 2), sizeof(int64_t));
    c->csr_kid = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csr_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    const orc_inst_t *sall = sorted_in ? sorted_in : a;
#pragma omp parallel for num_threads(T) schedule(dynamic, 1)
    for (int bk = 0; bk < NBK; ++bk) {
        const int64_t n = bstart[bk + 1] - bstart[bk];
        const orc_inst_t *s = sall + bstart[bk];
        int64_t kid = bN[bk], e = bZ[bk];
        for (int64_t i = 0; i < n; ) {
            int64_t j = i + 1;
            while (j < n && s[j].kmer == s[i].kmer && s[j].kmer2 == s[i].kmer2 && s[j].kmer3 == s[i].kmer3) ++j;
            const int64_t cn = j - i;
            if (cn >= c->lower && cn <= c->upper) {
                c->kmers[kid] = s[i].kmer;
                if (c->kmers_lo) c->kmers_lo[kid] = s[i].kmer2;
                if (c->kmers_lo2) c->kmers_lo2[kid] = s[i].kmer3;
                c->colptr[kid] = e;
                for (int64_t t2 = i; t2 < j; ++t2) { c->csc_read[e] = s[t2].read; c->csc_pos[e] = s[t2].pos; ++e; }
                ++kid;
            }
            i = j;
        }
    }
    c->colptr[N] = Z;
    free(a); free(b); free(rbound); free(cnt); free(bstart);
    /* CSR: stable counting transpose, columns split over the threads (thread t's entries of a row lie behind those of the threads before it) */
    int64_t *rc = (int64_t *)calloc((size_t)T * (size_t)(nreads + 1), sizeof(int64_t));
    if (!rc) return -1;
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const int64_t k0 = N * t / T, k1 = N * (t + 1) / T;
        int64_t *mine = rc + (size_t)t * (size_t)(nreads + 1);
        for (int64_t e = c->colptr[k0]; e < c->colptr[k1]; ++e) mine[c->csc_read[e]]++;
#pragma omp barrier
#pragma omp for schedule(static)
        for (int64_t r = 0; r < nreads; ++r) { int64_t tot = 0; for (int tt = 0; tt < T; ++tt) tot += rc[(size_t)tt * (size_t)(nreads + 1) + (size_t)r]; c->rowptr[r + 1] = tot; }
#pragma omp single
        { for (int64_t r = 0; r < nreads; ++r) c->rowptr[r + 1] += c->rowptr[r]; }
#pragma omp for schedule(static)
        for (int64_t r = 0; r < nreads; ++r) {
            int64_t run = c->rowptr[r];
            for (int tt = 0; tt < T; ++tt) { int64_t *x = &rc[(size_t)tt * (size_t)(nreads + 1) + (size_t)r]; const int64_t v = *x; *x = run; run += v; }
        }
        for (int64_t kk = k0; kk < k1; ++kk)
            for (int64_t e = c->colptr[kk]; e < c->colptr[kk + 1]; ++e) {
                const int64_t d = mine[c->csc_read[e]]++;
                c->csr_kid[d] = (uint32_t)kk; c->csr_pos[d] = c->csc_pos[e];
            }
    }
    free(rc);
    return 0;
}
#undef ORC_BK

/*
 * Alternative entry for a13 alone: take A as triples (row, col, val) — what create_seed_matrix's caller holds
 * (src/KmerOps.cpp:380-400 emits exactly such triples) — and build CSC/CSR with the canonical entry order:
 * within a column by (row,val), within a row by (col,val).  Duplicates kept.
 */
static int orc_cmp_csc(const void *x, const void *y)
{
    const int64_t *a = (const int64_t *)x, *b = (const int64_t *)y;   /* (col,row,val) */
    for (int i = 0; i < 3; ++i) { if (a[i] < b[i]) return -1; if (a[i] > b[i]) return 1; }
    return 0;
}
int orc_set_triples(orc_ctx *c, int64_t M, int64_t N, int64_t Z, const int64_t *rows, const int64_t *cols, const uint32_t *vals)
{
    orc_free_A(c); orc_free_B(c);
    c->M = M; c->N = N; c->Z = Z; c->I = 0; c->ndistinct = 0;
    int64_t *t = (int64_t *)malloc((size_t)(Z + 1) * 3 * sizeof(int64_t));
    for (int64_t i = 0; i < Z; ++i) {
        if (rows[i] < 0 || rows[i] >= M || cols[i] < 0 || cols[i] >= N) { free(t); return -2; }
        t[3 * i] = cols[i]; t[3 * i + 1] = rows[i]; t[3 * i + 2] = vals[i];
    }
    qsort(t, (size_t)Z, 3 * sizeof(int64_t), orc_cmp_csc);
    c->kmers = NULL; c->kmers_lo = NULL; c->kmers_lo2 = NULL;
    c->colptr = (int64_t *)calloc((size_t)(N + 2), sizeof(int64_t));
    c->csc_read = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csc_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->rowptr = (int64_t *)calloc((size_t)(M + 2), sizeof(int64_t));
    c->csr_kid = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csr_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    for (int64_t i = 0; i < Z; ++i) {
        c->colptr[t[3 * i] + 1]++; c->rowptr[t[3 * i + 1] + 1]++;
        c->csc_read[i] = (uint32_t)t[3 * i + 1]; c->csc_pos[i] = (uint32_t)t[3 * i + 2];
    }
    for (int64_t i = 0; i < N; ++i) c->colptr[i + 1] += c->colptr[i];
    for (int64_t i = 0; i < M; ++i) c->rowptr[i + 1] += c->rowptr[i];
    int64_t *cur = (int64_t *)malloc((size_t)(M + 1) * sizeof(int64_t));
    memcpy(cur, c->rowptr, (size_t)(M + 1) * sizeof(int64_t));
    for (int64_t i = 0; i < Z; ++i) { int64_t d = cur[t[3 * i + 1]]++; c->csr_kid[d] = (uint32_t)t[3 * i]; c->csr_pos[d] = (uint32_t)t[3 * i + 2]; }
    free(cur); free(t);
    return 0;
}

/*
 * The same hand-over for a matrix of BASELINE size (bench.py's cpu_baseline leg on the whole headline matrix: qsort of 5 * 10^8 triples would
 * take minutes).  Input: the columns of A already in the order of the reference's AT (src/main.cpp:272-273) — colptr[N+1] (u32) and
 * csc[Z] = read << 32 | pos, entries of a column in (read, pos) order.  CSR is derived here: a stable counting sort by read keeps, for
 * every row, the columns ascending and the positions ascending within a column, i.e. (kid, pos) order.  Threads own disjoint row ranges
 * (each scans all columns and takes its rows' entries).  Returns -2 on an index out of range or a column out of order.
 */
int orc_set_csc(orc_ctx *c, int64_t M, int64_t N, int64_t Z, const uint32_t *colptr32, const uint64_t *csc, int nthreads)
{
    orc_free_A(c); orc_free_B(c);
    if (nthreads < 1) nthreads = 1;
    c->M = M; c->N = N; c->Z = Z; c->I = 0; c->ndistinct = 0;
    c->kmers = NULL; c->kmers_lo = NULL; c->kmers_lo2 = NULL;
    c->colptr = (int64_t *)calloc((size_t)(N + 2), sizeof(int64_t));
    c->csc_read = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csc_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->rowptr = (int64_t *)calloc((size_t)(M + 2), sizeof(int64_t));
    c->csr_kid = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    c->csr_pos = (uint32_t *)malloc((size_t)(Z + 1) * sizeof(uint32_t));
    int bad = 0;
    if (colptr32[0] != 0 || (int64_t)colptr32[N] != Z) return -2;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(|:bad)
#endif
    for (int64_t kcol = 0; kcol < N; ++kcol) {
        c->colptr[kcol] = colptr32[kcol];
        if (colptr32[kcol + 1] < colptr32[kcol]) { bad = 1; continue; }
        for (int64_t f = colptr32[kcol]; f < colptr32[kcol + 1]; ++f) {
            const uint64_t e = csc[f];
            if ((int64_t)(e >> 32) >= M) bad = 1;
            if (f > colptr32[kcol] && csc[f - 1] > e) bad = 1;         /* (read, pos) ascending inside a column */
            c->csc_read[f] = (uint32_t)(e >> 32); c->csc_pos[f] = (uint32_t)e;
        }
    }
    c->colptr[N] = Z;
    if (bad) return -2;
    /* row counts: every thread counts the rows of its own range */
    int64_t *cnt = (int64_t *)calloc((size_t)(M + 1), sizeof(int64_t));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        const int64_t r0 = M * t / T, r1 = M * (t + 1) / T;
        for (int64_t f = 0; f < Z; ++f) { const int64_t r = (int64_t)(csc[f] >> 32); if (r >= r0 && r < r1) cnt[r]++; }
    }
    for (int64_t i = 0; i < M; ++i) c->rowptr[i + 1] = c->rowptr[i] + cnt[i];
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        const int64_t r0 = M * t / T, r1 = M * (t + 1) / T;
        for (int64_t i = r0; i < r1; ++i) cnt[i] = c->rowptr[i];
        for (int64_t kcol = 0; kcol < N; ++kcol)
            for (int64_t f = colptr32[kcol]; f < colptr32[kcol + 1]; ++f) {
                const int64_t r = (int64_t)(csc[f] >> 32);
                if (r >= r0 && r < r1) { const int64_t d = cnt[r]++; c->csr_kid[d] = (uint32_t)kcol; c->csr_pos[d] = (uint32_t)csc[f]; }
            }
    }
    free(cnt);
    return 0;
}

/*
 * Entry-by-entry comparison of a CSR B handed in (row pointers i64[M+1], columns u32[Y], values 20 bytes each in orc_seed_t's field order)
 * with this context's B: the number of rows whose extent differs + the number of entries whose column or value differs; -1 if M or Y differ.
 * (bench.py: the GPU's B of the whole headline matrix against the oracle's, without a Python-side copy of either.)
 */
int64_t orc_compare_B(const orc_ctx *c, int64_t M, int64_t Y, const int64_t *rowptr, const uint32_t *col, const orc_seed_t *val, int nthreads)
{
    if (!c->b_rowptr || M != c->M || Y != c->Y) return -1;
    if (nthreads < 1) nthreads = 1;
    int64_t diff = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+:diff)
#endif
    for (int64_t i = 0; i < M; ++i) {
        if (rowptr[i] != c->b_rowptr[i] || rowptr[i + 1] != c->b_rowptr[i + 1]) { ++diff; continue; }
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
            if (col[e] != c->b_col[e] || memcmp(&val[e], &c->b_val[e], sizeof(orc_seed_t)) != 0) ++diff;
    }
    return diff;
}

/*
 * a13: B = A*A^T over SharedSeeds::Semiring, then Prune(numshared <= 1)  (src/SharedSeeds.cpp:4-10).
 * Row-by-row Gustavson with a dense accumulator; for row i the products are visited in ascending
 * (kid, posQ, posT) order and combined by a LEFT fold of Semiring::add (include/SharedSeeds.hpp:41-46), so
 * seeds[0] = first product, seeds[1] = last product, numshared = #products: the canonical rule of SURVEY §8c-2.
 * Both triangles and the diagonal are computed (CombBLAS computes all of B).  Columns ascending within a row.
 * nthreads > 1 splits rows over OpenMP threads (each with its own accumulator); results are identical.
 */
static int orc_cmp_u32(const void *x, const void *y)
{
    uint32_t a = *(const uint32_t *)x, b = *(const uint32_t *)y;
    return a < b ? -1 : a > b;
}

int orc_spgemm(orc_ctx *c, int nthreads)
{
    orc_free_B(c);
    const int64_t M = c->M;
    if (nthreads < 1) nthreads = 1;
    int64_t *rowcnt = (int64_t *)calloc((size_t)(M + 2), sizeof(int64_t));
    uint32_t **rcol = (uint32_t **)calloc((size_t)(M + 1), sizeof(uint32_t *));
    orc_seed_t **rval = (orc_seed_t **)calloc((size_t)(M + 1), sizeof(orc_seed_t *));
    int64_t P = 0, Yraw = 0, ndiag = 0, nupper = 0, maxshared = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+:P,Yraw,ndiag,nupper) reduction(max:maxshared)
#endif
    {
        orc_seed_t *acc = (orc_seed_t *)calloc((size_t)(M + 1), sizeof(orc_seed_t));
        uint32_t *touched = (uint32_t *)malloc((size_t)(M + 1) * sizeof(uint32_t));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < M; ++i) {
            int64_t nt = 0;
            for (int64_t e = c->rowptr[i]; e < c->rowptr[i + 1]; ++e) {
                uint32_t kid = c->csr_kid[e], q = c->csr_pos[e];
                for (int64_t f = c->colptr[kid]; f < c->colptr[kid + 1]; ++f) {
                    uint32_t j = c->csc_read[f];
                    orc_seed_t prod = orc_sr_multiply(q, c->csc_pos[f]);
                    if (acc[j].numshared == 0) { acc[j] = prod; touched[nt++] = j; }
                    else acc[j] = orc_sr_add(acc[j], prod);
                    ++P;
                }
            }
            Yraw += nt;
            int64_t keep = 0;
            for (int64_t t = 0; t < nt; ++t) if (acc[touched[t]].numshared > 1) touched[keep++] = touched[t]; else acc[touched[t]].numshared = 0;
            qsort(touched, (size_t)keep, sizeof(uint32_t), orc_cmp_u32);
            rowcnt[i + 1] = keep;
            rcol[i] = (uint32_t *)malloc((size_t)(keep + 1) * sizeof(uint32_t));
            rval[i] = (orc_seed_t *)malloc((size_t)(keep + 1) * sizeof(orc_seed_t));
            for (int64_t t = 0; t < keep; ++t) {
                uint32_t j = touched[t];
                rcol[i][t] = j; rval[i][t] = acc[j];
                if ((int64_t)j == i) ++ndiag;
                if ((int64_t)j > i) ++nupper;
                if (acc[j].numshared > maxshared) maxshared = acc[j].numshared;
                acc[j].numshared = 0;
            }
        }
        free(acc); free(touched);
    }
    for (int64_t i = 0; i < M; ++i) rowcnt[i + 1] += rowcnt[i];
    int64_t Y = rowcnt[M];
    c->b_rowptr = rowcnt;
    c->b_col = (uint32_t *)malloc((size_t)(Y + 1) * sizeof(uint32_t));
    c->b_val = (orc_seed_t *)malloc((size_t)(Y + 1) * sizeof(orc_seed_t));
    for (int64_t i = 0; i < M; ++i) {
        int64_t n = rowcnt[i + 1] - rowcnt[i];
        memcpy(c->b_col + rowcnt[i], rcol[i], (size_t)n * sizeof(uint32_t));
        memcpy(c->b_val + rowcnt[i], rval[i], (size_t)n * sizeof(orc_seed_t));
        free(rcol[i]); free(rval[i]);
    }
    free(rcol); free(rval);
    c->P = P; c->Yraw = Yraw; c->Y = Y; c->ndiag = ndiag; c->nupper = nupper; c->maxshared = maxshared;
    return 0;
}

/*
 * a14: the consumer's view.  Fill a CombBLAS-shaped DCSC (jc, cp, ir, numx) of the block
 * [row_lo,row_hi) x [col_lo,col_hi) of B with LOCAL indices, column-major, rows ascending within a column —
 * the arrays src/PairwiseAlignment.cpp:28-32 walks.  Caller frees with orc_free_ptr.
 */
int orc_export_dcsc(const orc_ctx *c, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi,
                    int64_t *nnz, int64_t *nzc, int64_t **jc, int64_t **cp, int64_t **ir, orc_seed_t **numx)
{
    const int64_t ncols = col_hi - col_lo;
    int64_t *cnt = (int64_t *)calloc((size_t)(ncols + 2), sizeof(int64_t));
    int64_t total = 0;
    for (int64_t i = row_lo; i < row_hi; ++i)
        for (int64_t e = c->b_rowptr[i]; e < c->b_rowptr[i + 1]; ++e) {
            int64_t j = c->b_col[e];
            if (j >= col_lo && j < col_hi) { cnt[j - col_lo + 1]++; ++total; }
        }
    int64_t nz = 0;
    for (int64_t j = 0; j < ncols; ++j) if (cnt[j + 1]) ++nz;
    *jc = (int64_t *)malloc((size_t)(nz + 1) * sizeof(int64_t));
    *cp = (int64_t *)malloc((size_t)(nz + 2) * sizeof(int64_t));
    *ir = (int64_t *)malloc((size_t)(total + 1) * sizeof(int64_t));
    *numx = (orc_seed_t *)malloc((size_t)(total + 1) * sizeof(orc_seed_t));
    int64_t *start = (int64_t *)malloc((size_t)(ncols + 1) * sizeof(int64_t));
    int64_t run = 0, ci = 0;
    for (int64_t j = 0; j < ncols; ++j) {
        start[j] = run;
        if (cnt[j + 1]) { (*jc)[ci] = j; (*cp)[ci] = run; ++ci; }
        run += cnt[j + 1];
    }
    (*cp)[nz] = run;
    for (int64_t i = row_lo; i < row_hi; ++i)
        for (int64_t e = c->b_rowptr[i]; e < c->b_rowptr[i + 1]; ++e) {
            int64_t j = c->b_col[e];
            if (j >= col_lo && j < col_hi) { int64_t d = start[j - col_lo]++; (*ir)[d] = i - row_lo; (*numx)[d] = c->b_val[e]; }
        }
    free(start); free(cnt);
    *nnz = total; *nzc = nz;
    return 0;
}
void orc_free_ptr(void *p) { free(p); }

/*
 * test.py:57-65 property: a stored seed pair names the same k-mer in both reads, forward or reverse-complement.
 * Returns 1 if valid.
 */
int orc_seed_is_valid(const uint8_t *qmem, uint32_t qlen, const uint8_t *tmem, uint32_t tlen, uint32_t q, uint32_t t, int k)
{
    if ((int64_t)q + k > (int64_t)qlen || (int64_t)t + k > (int64_t)tlen) return 0;
    if (k > 32) { uint64_t x[3], y[3]; orc_kmerN_at(qmem, q, k, x); orc_kmerN_at(tmem, t, k, y); return x[0] == y[0] && x[1] == y[1] && x[2] == y[2]; }   /* same canonical k-mer */
    uint64_t a = 0, b = 0;
    for (int i = 0; i < k; ++i) { a |= (uint64_t)orc_base_at(qmem, q + (size_t)i) << (2 * (31 - i)); b |= (uint64_t)orc_base_at(tmem, t + (size_t)i) << (2 * (31 - i)); }
    return a == b || a == orc_kmer_twin(b, k);
}


/* ====================================================================================================================
 * f1: x-drop seed-and-extend, the consumer of seeds[0] right after the path.
 * Restates src/XDropAligner.cpp:46-222 (_extend_seed_one_direction), :224-282 (xdrop_aligner), :7-44 (classify_alignment)
 * and the field mapping of Overlap::extend_overlap (src/Overlap.cpp:24-73).  Pinned against the reference's own
 * XDropAligner.cpp through oracle/_ref (tests/golden/xdrop_*.txt + live random cross-checks).
 * ==================================================================================================================== */
typedef struct { int32_t begQ, endQ, begT, endT, score, rc; } orc_xseed_t;
typedef struct {
    int32_t begQ, begT, endQ, endT;   /* Overlap::beg / end (src/Overlap.cpp:38-41) */
    int32_t score, suffix, suffixT;
    int8_t direction, directionT;
    uint8_t rc, passed, containedQ, containedT;
    uint8_t kind, pad;                /* OverlapClass (include/XDropAligner.hpp:10-17) */
} orc_overlap_t;

static inline int orc_revcomp_at(const uint8_t *mem, size_t len, size_t i) { return 3 - orc_base_at(mem, len - 1 - i); }   /* include/DnaSeq.hpp:119 */

/* src/XDropAligner.cpp:46-208.  Three antidiagonals ad1 (n-2), ad2 (n-1), ad3 (n), each stored from column offsetK on;
 * only best_ext_{col,row,score} leave the function (the ext_* tail of the reference, :166-205, has no effect on them). */
/* diagnostics: widest band (max_col - min_col + 2 stored columns) and antidiagonals of the most recent extension on this thread */
static _Thread_local int orc_dbg_band = 0, orc_dbg_ads = 0;
void orc_xdrop_last_shape(int *band, int *ads) { *band = orc_dbg_band; *ads = orc_dbg_ads; }

static int orc_extend_one_direction(const uint8_t *q, int lenQ, const uint8_t *t, int lenT, int extleft, orc_xseed_t *xs,
                                    int mat, int mis, int gap, int dropoff, int64_t *cells)
{
    const int lenQ_ext = extleft ? xs->begQ : lenQ - xs->endQ;
    const int lenT_ext = extleft ? xs->begT : lenT - xs->endT;
    const int cols = lenQ_ext + 1, rows = lenT_ext + 1;
    if (rows == 1 || cols == 1) return 0;
    const int int_min = INT32_MIN;
    const int len = 2 * (cols > rows ? cols : rows);
    const int min_err_score = int_min / len;
    if (gap < min_err_score) gap = min_err_score;
    if (mis < min_err_score) mis = min_err_score;
    const int undef = int_min - gap - mis;
    const int cap = (cols > rows ? cols : rows) + 4;
    int *b1 = (int *)malloc(sizeof(int) * (size_t)cap), *b2 = (int *)malloc(sizeof(int) * (size_t)cap), *b3 = (int *)malloc(sizeof(int) * (size_t)cap);
    int *ad1 = b1, *ad2 = b2, *ad3 = b3;
    int n1 = 0, n2 = 0, n3 = 0;                 /* vector sizes */
    int min_col = 1, max_col = 2;
    int offset1 = 0, offset2 = 0, offset3 = 0;
    ad2[0] = 0; n2 = 1;
    int best_ext_col = 0, best_ext_row = 0, best_ext_score = 0;
    n3 = 2; ad3[0] = ad3[1] = (-gap > dropoff) ? undef : gap;
    int ad_no = 1, best = 0;
    const int offsetQ = xs->endQ, offsetT = xs->endT;
    orc_dbg_band = 0; orc_dbg_ads = 0;
    while (min_col < max_col) {
        ++ad_no;
        if (max_col + 1 - (min_col - 1) > orc_dbg_band) orc_dbg_band = max_col + 1 - (min_col - 1);
        ++orc_dbg_ads;
        { int *tb = ad1; ad1 = ad2; n1 = n2; ad2 = ad3; n2 = n3; ad3 = tb; }
        offset1 = offset2; offset2 = offset3; offset3 = min_col - 1;
        n3 = max_col + 1 - offset3;             /* resize: new cells hold whatever; every one is written below before it is read */
        for (int i = 0; i < n3; ++i) ad3[i] = 0;
        ad3[0] = ad3[max_col - offset3] = undef;
        if ((int64_t)ad_no * gap > (int64_t)best - dropoff) {
            if (offset3 == 0) ad3[0] = ad_no * gap;
            if (ad_no - max_col == 0) ad3[max_col - offset3] = ad_no * gap;
        }
        int ad_best = ad_no * gap;
        for (int col = min_col; col < max_col; ++col) {
            const int i3 = col - offset3, i2 = col - offset2, i1 = col - offset1;
            const int posQ = extleft ? cols - 1 - col : col - 1 + offsetQ;
            const int posT = extleft ? rows - 1 + col - ad_no : ad_no - col - 1 + offsetT;
            int temp = (ad2[i2 - 1] > ad2[i2] ? ad2[i2 - 1] : ad2[i2]) + gap;
            const int tb = xs->rc ? orc_revcomp_at(t, (size_t)lenT, (size_t)posT) : orc_base_at(t, (size_t)posT);
            const int temp2 = ad1[i1 - 1] + (orc_base_at(q, (size_t)posQ) == tb ? mat : mis);
            if (temp2 > temp) temp = temp2;
            if (temp < best - dropoff) ad3[i3] = undef;
            else { ad3[i3] = temp; if (temp > ad_best) ad_best = temp; }
            if (temp > best) { best_ext_col = col; best_ext_row = ad_no - best_ext_col; best_ext_score = ad3[best_ext_col - offset3]; }
            if (cells) ++*cells;
        }
        if (ad_best > best) best = ad_best;
        while (min_col - offset3 < n3 && ad3[min_col - offset3] == undef && min_col - offset2 - 1 < n2 && ad2[min_col - offset2 - 1] == undef) ++min_col;
        while (max_col - offset3 > 0 && ad3[max_col - offset3 - 1] == undef && ad2[max_col - offset2 - 1] == undef) --max_col;
        ++max_col;
        if (min_col < ad_no + 2 - rows) min_col = ad_no + 2 - rows;
        if (max_col > cols) max_col = cols;
    }
    (void)n1;
    if (best_ext_score != undef) {
        if (extleft) { xs->begT -= best_ext_row; xs->begQ -= best_ext_col; }
        else { xs->endT += best_ext_row; xs->endQ += best_ext_col; }
    }
    free(b1); free(b2); free(b3);
    return best_ext_score;
}

/* src/XDropAligner.cpp:224-282.  Returns the score, or -1 with *res left at XSeed's defaults (:22) when the seed is rejected. */
int orc_xdrop_aligner(const uint8_t *q, int lenQ, const uint8_t *t, int lenT, int begQ, int begT, int k,
                      int mat, int mis, int gap, int dropoff, orc_xseed_t *res, int64_t *cells)
{
    res->begQ = res->endQ = res->begT = res->endT = 0; res->score = -1; res->rc = 0;
    if (begQ < 0 || begQ + k > lenQ) return -1;
    if (begT < 0 || begT + k > lenT) return -1;
    if (begQ == 0 && begT == 0) return -1;
    const int rc = orc_base_at(q, (size_t)(begQ + (k >> 1))) != orc_base_at(t, (size_t)(begT + (k >> 1)));
    for (int i = 0; i < k; ++i) {
        const int tb = rc ? orc_revcomp_at(t, (size_t)lenT, (size_t)(lenT - begT - k + i)) : orc_base_at(t, (size_t)(begT + i));
        if (orc_base_at(q, (size_t)(begQ + i)) != tb) return -1;
    }
    orc_xseed_t xs;
    xs.begQ = begQ; xs.endQ = begQ + k;
    xs.begT = rc ? lenT - begT - k : begT; xs.endT = xs.begT + k;
    xs.rc = rc; xs.score = -1;
    orc_xseed_t l = xs, r = xs;
    const int lscore = orc_extend_one_direction(q, lenQ, t, lenT, 1, &l, mat, mis, gap, dropoff, cells);
    const int rscore = orc_extend_one_direction(q, lenQ, t, lenT, 0, &r, mat, mis, gap, dropoff, cells);
    const int score = lscore + rscore + mat * k;
    res->begQ = l.begQ; res->endQ = r.endQ;
    res->begT = rc ? lenT - r.endT : l.begT;
    res->endT = rc ? lenT - l.begT : r.endT;
    res->rc = rc; res->score = score;
    return score;
}

/* src/XDropAligner.cpp:7-44 */
int orc_classify_alignment(const orc_xseed_t *ai, int lenQ, int lenT)
{
    if (ai->score <= 0) return 0;                                   /* BAD_ALIGNMENT */
    const int begTr = ai->rc ? lenT - ai->endT : ai->begT;
    const int endTr = ai->rc ? lenT - ai->begT : ai->endT;
    const int maplen = ((ai->endT - ai->begT) + (ai->endQ - ai->begQ)) / 2;
    const int a = ai->begQ < begTr ? ai->begQ : begTr, b = (lenQ - ai->endQ) < (lenT - endTr) ? (lenQ - ai->endQ) : (lenT - endTr);
    const int overhang = a + b;
    const int overlap = maplen + overhang;
    const float my_thr = (float)((1.0 - 0.1) * (0.99 * overlap));   /* DELTACHERNOFF 0.1, include/XDropAligner.hpp:8 */
    if (ai->begQ <= begTr && lenQ - ai->endQ <= lenT - endTr) return 1;           /* FIRST_CONTAINED */
    if (ai->begQ >= begTr && lenQ - ai->endQ >= lenT - endTr) return 2;           /* SECOND_CONTAINED */
    if ((float)ai->score < my_thr || overlap < 500) return 0;
    if (ai->begQ > begTr) return 3;                                                /* FIRST_TO_SECOND_OVERLAP */
    return 4;                                                                      /* SECOND_TO_FIRST_OVERLAP */
}

/* Overlap::Overlap + Overlap::extend_overlap, src/Overlap.cpp:4-11, :24-73 */
void orc_overlap_extend(const uint8_t *q, int lenQ, const uint8_t *t, int lenT, int seedQ, int seedT, int k,
                        int mat, int mis, int gap, int dropoff, orc_overlap_t *o, int64_t *cells)
{
    orc_xseed_t r;
    orc_xdrop_aligner(q, lenQ, t, lenT, seedQ, seedT, k, mat, mis, gap, dropoff, &r, cells);
    const int kind = orc_classify_alignment(&r, lenQ, lenT);
    memset(o, 0, sizeof *o);
    o->direction = -1; o->directionT = -1;
    o->rc = (uint8_t)r.rc; o->score = r.score; o->kind = (uint8_t)kind;
    o->begQ = r.begQ; o->begT = r.begT; o->endQ = r.endQ; o->endT = r.endT;
    const int begQr = r.begQ, endQr = r.endQ;
    const int begTr = r.rc ? lenT - r.endT : r.begT, endTr = r.rc ? lenT - r.begT : r.endT;
    if (kind != 0) {
        o->passed = 1;
        if (kind == 1) o->containedQ = 1;
        else if (kind == 2) o->containedT = 1;
        else if (kind == 3) { o->direction = r.rc ? 0 : 1; o->directionT = r.rc ? 0 : 2; o->suffix = (lenT - endTr) - (lenQ - endQr); o->suffixT = begQr - begTr; }
        else { o->direction = r.rc ? 3 : 2; o->directionT = r.rc ? 3 : 1; o->suffix = begTr - begQr; o->suffixT = (lenQ - endQr) - (lenT - endTr); }
    }
}

/* PairwiseAlignment's loop on one rank (src/PairwiseAlignment.cpp:28-95): every stored B(i,j) with i < j is aligned from seeds[0];
 * out[e] for the e-th such entry in CSR order (rows ascending, columns ascending); returns their number. */
/* (the omp loop below needs a canonical increment: stride is applied through an index map) */
int64_t orc_align_upper(const orc_ctx *c, const uint8_t *buf, const uint64_t *byte_off, const uint32_t *lens,
                        int mat, int mis, int gap, int dropoff, int nthreads, int64_t stride, int64_t *rows, int64_t *cols, orc_overlap_t *out, int64_t cap, int64_t *cells_total)
{
    int64_t n = 0;
    for (int64_t i = 0; i < c->M; ++i)
        for (int64_t e = c->b_rowptr[i]; e < c->b_rowptr[i + 1]; ++e)
            if ((int64_t)c->b_col[e] > i) { if (n < cap) { rows[n] = i; cols[n] = (int64_t)c->b_col[e]; } ++n; }
    if (n > cap) return -n;
    int64_t cells = 0;
    if (nthreads < 1) nthreads = 1;
    if (stride < 1) stride = 1;                 /* stride > 1: only every stride-th pair is aligned (bounded CPU baseline samples) */
    const int64_t nsample = (n + stride - 1) / stride;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) reduction(+ : cells)
    for (int64_t sidx = 0; sidx < nsample; ++sidx) {
        const int64_t a = sidx * stride;
        const int64_t i = rows[a], j = cols[a];
        int64_t e = c->b_rowptr[i];
        while ((int64_t)c->b_col[e] != j) ++e;
        int64_t mine = 0;
        orc_overlap_extend(buf + byte_off[i], (int)lens[i], buf + byte_off[j], (int)lens[j], (int)c->b_val[e].q0, (int)c->b_val[e].t0, c->k,
                           mat, mis, gap, dropoff, &out[a], &mine);
        cells += mine;
    }
    if (cells_total) *cells_total = cells;
    return n;
}

/* getters (ctypes-friendly) */
int64_t orc_get_i64(const orc_ctx *c, int what)
{
    switch (what) {
    case 0: return c->M; case 1: return c->I; case 2: return c->N; case 3: return c->Z; case 4: return c->ndistinct;
    case 5: return c->P; case 6: return c->Yraw; case 7: return c->Y; case 8: return c->ndiag; case 9: return c->nupper; case 10: return c->maxshared;
    default: return -1;
    }
}
const void *orc_get_ptr(const orc_ctx *c, int what)
{
    switch (what) {
    case 0: return c->kmers; case 1: return c->colptr; case 2: return c->csc_read; case 3: return c->csc_pos;
    case 4: return c->rowptr; case 5: return c->csr_kid; case 6: return c->csr_pos; case 7: return c->hist;
    case 8: return c->b_rowptr; case 9: return c->b_col; case 10: return c->b_val; case 11: return c->kmers_lo; case 12: return c->kmers_lo2;
    default: return NULL;
    }
}

/* ====================================================================================================================
 * f2: from the aligned pairs to the string graph — the statements of src/main.cpp:305-312:
 *   find_bad_reads (:553-571), R->Prune(!passed), PruneFull(bad), find_contained_reads (:573-583), PruneFull(contained),
 *   TransitiveReduction (src/TransitiveReduction.cpp:3-90) over MinPlusSR (include/TransitiveReduction.hpp:78-110).
 * Followed literally, loop and all: P = R; do { N = P (x) R; prune no-path; P = N; I = (F >= N[dir]); I |= I^T; T |= I } while
 * nnz(T) changes; S = R \ T.  The product is computed in full (every (i,j) reachable through some k, all four path slots), with a
 * dense per-row accumulator — nothing is masked by R's pattern here; the HIP path computes only what the comparison reads.
 * **parity unpinned**: the matrix algebra (Mult_AnXBn_DoubleBuff, EWiseApply, Prune, PruneFull, Reduce, operator+=) lives in
 * CombBLAS (absent) and the reference holds no fixture for this stage; min and + over int are order-free, so the fold order that
 * leaves the seed values of a13 unpinned does not matter here.  Known-answer cases: tests/test_oracle_string_graph.py.
 * ==================================================================================================================== */
typedef struct { int32_t dir, suffix; int32_t paths[4]; } orc_mp_t;     /* the fields of Overlap MinPlusSR touches */
#define ORC_INF 0x7fffffff

/* Overlap::Transpose, include/Overlap.hpp:43-69 (len is not carried: it is the lengths of the two reads, swapped with them) */
static orc_overlap_t orc_overlap_transpose(orc_overlap_t o)
{
    orc_overlap_t t = o;
    t.begQ = o.begT; t.begT = o.begQ; t.endQ = o.endT; t.endT = o.endQ;
    t.suffix = o.suffixT; t.suffixT = o.suffix;
    t.direction = o.directionT; t.directionT = o.direction;
    t.containedQ = o.containedT; t.containedT = o.containedQ;
    return t;
}

/* MinPlusSR::multiply, include/TransitiveReduction.hpp:88-104; Overlap::arrows include/Overlap.hpp:33-41 */
static int orc_mp_multiply(const orc_mp_t *e1, const orc_mp_t *e2, int32_t out[4])
{
    out[0] = out[1] = out[2] = out[3] = ORC_INF;
    if (e1->dir == -1 || e2->dir == -1) return 0;
    const int t1 = (e1->dir >> 1) & 1, h1 = e1->dir & 1, t2 = (e2->dir >> 1) & 1, h2 = e2->dir & 1;
    if (t2 == h1) return 0;
    out[2 * t1 + h2] = e1->suffix + e2->suffix;
    return 1;
}

typedef struct { int64_t row, col; orc_overlap_t v; } orc_edge_t;
static int orc_cmp_i64(const void *x, const void *y) { const int64_t a = *(const int64_t *)x, b = *(const int64_t *)y; return a < b ? -1 : (a > b); }
static int orc_cmp_edge_rc(const void *x, const void *y)
{
    const orc_edge_t *a = (const orc_edge_t *)x, *b = (const orc_edge_t *)y;
    if (a->row != b->row) return a->row < b->row ? -1 : 1;
    return a->col < b->col ? -1 : (a->col > b->col);
}
static int orc_cmp_edge_cr(const void *x, const void *y)
{
    const orc_edge_t *a = (const orc_edge_t *)x, *b = (const orc_edge_t *)y;
    if (a->col != b->col) return a->col < b->col ? -1 : 1;
    return a->row < b->row ? -1 : (a->row > b->row);
}

/* In: the n aligned pairs (rows[a] < cols[a], no pair twice) of an M-read set.  Out: the entries of S in the order parallel_write_paf
 * walks them (DCSC: columns ascending, rows ascending within a column; src/main.cpp:527-541); read_flags[v] bit 0 = bad read, bit 1 =
 * contained read; stats[0..9] = bad reads, entries after the passed/bad prune, contained reads, entries handed to TransitiveReduction
 * (upper triangle), semiring products of the first P (x) R, nnz(N) of it after the no-path prune, nnz(I) before symmetrising,
 * nnz(T), nnz(S), loop iterations.  Returns nnz(S), or -(needed) when cap is too small, or INT64_MIN on bad input. */
int64_t orc_string_graph(int64_t M, int64_t n, const int64_t *rows, const int64_t *cols, const orc_overlap_t *vals, double cutoff, int fuzz,
                         int64_t *out_rows, int64_t *out_cols, orc_overlap_t *out_vals, int64_t cap, uint8_t *read_flags, int64_t *stats)
{
    for (int64_t a = 0; a < n; ++a) if (rows[a] < 0 || cols[a] >= M || rows[a] >= cols[a]) return INT64_MIN;
    int64_t st[10] = {0};
    /* find_bad_reads: (passed entries in row + column + 1) / (entries in row + column + 1) <= cutoff */
    int32_t *deg = calloc((size_t)M + 1, 4), *pas = calloc((size_t)M + 1, 4);
    uint8_t *bad = calloc((size_t)M + 1, 1), *cont = calloc((size_t)M + 1, 1);
    for (int64_t a = 0; a < n; ++a) { ++deg[rows[a]]; ++deg[cols[a]]; if (vals[a].passed) { ++pas[rows[a]]; ++pas[cols[a]]; } }
    for (int64_t v = 0; v < M; ++v) { const double r = ((double)pas[v] + 1) / ((double)deg[v] + 1); if (r <= cutoff) { bad[v] = 1; ++st[0]; } }
    /* R->Prune(!passed); R->PruneFull(bad, bad) */
    orc_edge_t *e = malloc(sizeof(orc_edge_t) * (size_t)(2 * n + 1));
    int64_t m = 0;
    for (int64_t a = 0; a < n; ++a) if (vals[a].passed && !bad[rows[a]] && !bad[cols[a]]) { e[m].row = rows[a]; e[m].col = cols[a]; e[m].v = vals[a]; ++m; }
    st[1] = m;
    /* find_contained_reads: containedQ marks the row's read, containedT the column's */
    for (int64_t a = 0; a < m; ++a) { if (e[a].v.containedQ) cont[e[a].row] = 1; if (e[a].v.containedT) cont[e[a].col] = 1; }
    for (int64_t v = 0; v < M; ++v) st[2] += cont[v];
    int64_t m2 = 0;
    for (int64_t a = 0; a < m; ++a) if (!cont[e[a].row] && !cont[e[a].col]) e[m2++] = e[a];
    st[3] = m2;
    if (read_flags) for (int64_t v = 0; v < M; ++v) read_flags[v] = (uint8_t)(bad[v] | (cont[v] << 1));
    /* RT = R^T with Overlap::Transpose applied; R += RT */
    for (int64_t a = 0; a < m2; ++a) { e[m2 + a].row = e[a].col; e[m2 + a].col = e[a].row; e[m2 + a].v = orc_overlap_transpose(e[a].v); }
    const int64_t E = 2 * m2;
    qsort(e, (size_t)E, sizeof *e, orc_cmp_edge_rc);
    int64_t *rp = calloc((size_t)M + 2, 8);
    for (int64_t a = 0; a < E; ++a) ++rp[e[a].row + 1];
    for (int64_t v = 0; v < M; ++v) rp[v + 1] += rp[v];
    /* P: pattern + values as a CSR that changes per iteration; R fixed */
    orc_mp_t *Rv = malloc(sizeof(orc_mp_t) * (size_t)(E + 1));
    for (int64_t a = 0; a < E; ++a) { Rv[a].dir = e[a].v.direction; Rv[a].suffix = e[a].v.suffix; for (int s = 0; s < 4; ++s) Rv[a].paths[s] = ORC_INF; }
    int64_t *Prp = malloc(8 * (size_t)(M + 2)); memcpy(Prp, rp, 8 * (size_t)(M + 2));
    int64_t Pn = E; int64_t *Pcol = malloc(8 * (size_t)(E + 1)); orc_mp_t *Pv = malloc(sizeof(orc_mp_t) * (size_t)(E + 1));
    for (int64_t a = 0; a < E; ++a) { Pcol[a] = e[a].col; Pv[a] = Rv[a]; }
    uint8_t *T = calloc((size_t)E + 1, 1), *I = malloc((size_t)E + 1);     /* T, I: flags on R's entries (I and T are subsets of R's pattern) */
    int32_t (*acc)[4] = malloc(sizeof(int32_t[4]) * (size_t)(M + 1));
    uint8_t *seen = calloc((size_t)M + 1, 1);
    int64_t *touched = malloc(8 * (size_t)(M + 1));
    int64_t nnzT = 0, prev, iters = 0;
    do {
        prev = nnzT;
        /* N = P (x) R, row by row; N.Prune(NoPathSRing) */
        int64_t Ncap = E + 16, Nn = 0, products = 0;
        int64_t *Nrp = calloc((size_t)M + 2, 8), *Ncol = malloc(8 * (size_t)Ncap);
        orc_mp_t *Nv = malloc(sizeof(orc_mp_t) * (size_t)Ncap);
        for (int64_t i = 0; i < M; ++i) {
            int64_t nt = 0;
            for (int64_t x = Prp[i]; x < Prp[i + 1]; ++x) {
                const int64_t k = Pcol[x];
                for (int64_t y = rp[k]; y < rp[k + 1]; ++y) {
                    const int64_t j = e[y].col;
                    int32_t prod[4];
                    orc_mp_multiply(&Pv[x], &Rv[y], prod);
                    ++products;
                    if (!seen[j]) { seen[j] = 1; touched[nt++] = j; for (int s = 0; s < 4; ++s) acc[j][s] = ORC_INF; }
                    for (int s = 0; s < 4; ++s) if (prod[s] < acc[j][s]) acc[j][s] = prod[s];            /* opmin, src/TransitiveReduction.cpp:92-100 */
                }
            }
            /* columns ascending within the row */
            qsort(touched, (size_t)nt, 8, orc_cmp_i64);
            for (int64_t a = 0; a < nt; ++a) {
                const int64_t j = touched[a];
                seen[j] = 0;
                if (acc[j][0] == ORC_INF && acc[j][1] == ORC_INF && acc[j][2] == ORC_INF && acc[j][3] == ORC_INF) continue;   /* NoPathSRing */
                if (Nn == Ncap) { Ncap *= 2; Ncol = realloc(Ncol, 8 * (size_t)Ncap); Nv = realloc(Nv, sizeof(orc_mp_t) * (size_t)Ncap); }
                Ncol[Nn] = j; Nv[Nn].dir = -1; Nv[Nn].suffix = 0;                                         /* Overlap() of multiply: direction -1, suffix 0 */
                for (int s = 0; s < 4; ++s) Nv[Nn].paths[s] = acc[j][s];
                ++Nn;
            }
            Nrp[i + 1] = Nn;
        }
        if (iters == 0) { st[4] = products; st[5] = Nn; }
        /* I = EWiseApply(F, N, GreaterThanSR) on the intersection, F = R with suffix + FUZZ; prune false */
        int64_t nI = 0;
        memset(I, 0, (size_t)E + 1);
        for (int64_t i = 0; i < M; ++i) {
            int64_t y = Nrp[i];
            for (int64_t x = rp[i]; x < rp[i + 1]; ++x) {
                while (y < Nrp[i + 1] && Ncol[y] < e[x].col) ++y;
                if (y < Nrp[i + 1] && Ncol[y] == e[x].col) {
                    const int dir = Rv[x].dir;
                    if (dir != -1 && Rv[x].suffix + fuzz >= Nv[y].paths[dir]) { I[x] = 1; ++nI; }
                }
            }
        }
        if (iters == 0) st[6] = nI;
        /* I += I^T; T += I */
        for (int64_t x = 0; x < E; ++x) if (I[x]) {
            orc_edge_t key; key.row = e[x].col; key.col = e[x].row;
            const orc_edge_t *r = bsearch(&key, e, (size_t)E, sizeof *e, orc_cmp_edge_rc);
            if (!T[x]) { T[x] = 1; ++nnzT; }
            if (!T[r - e]) { T[r - e] = 1; ++nnzT; }
        }
        free(Prp); free(Pcol); free(Pv);
        Prp = Nrp; Pcol = Ncol; Pv = Nv; Pn = Nn;
        ++iters;
    } while (nnzT != prev);
    (void)Pn;
    st[7] = nnzT; st[9] = iters;
    /* R = EWiseApply(R, T, TransitiveRemoval, not-T); R.Prune(direction == -1) */
    int64_t ns = 0;
    for (int64_t x = 0; x < E; ++x) if (!T[x] && e[x].v.direction != -1) e[ns++] = e[x];
    st[8] = ns;
    int64_t ret = ns;
    if (ns > cap) ret = -ns;
    else {
        qsort(e, (size_t)ns, sizeof *e, orc_cmp_edge_cr);
        for (int64_t a = 0; a < ns; ++a) { out_rows[a] = e[a].row; out_cols[a] = e[a].col; out_vals[a] = e[a].v; }
    }
    if (stats) memcpy(stats, st, sizeof st);
    free(deg); free(pas); free(bad); free(cont); free(e); free(rp); free(Rv); free(Prp); free(Pcol); free(Pv); free(T); free(I); free(acc); free(seen); free(touched);
    return ret;
}
