// synth.cpp — synthetic long-read generator (see include/elba_synth.h).  Host-only, deterministic, counter-based RNG so
// that any shard [first_read, first_read+num_reads) of a read set can be generated independently on each rank.
#include "elba_synth.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t hash2(uint64_t a, uint64_t b)
{
    uint64_t s = a * 0xD6E8FEB86659FD93ULL + b;
    splitmix(s);
    return splitmix(s);
}
inline double u01(uint64_t &s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
inline double gauss(uint64_t &s)
{
    double u1 = u01(s), u2 = u01(s);
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}

// genome base at position p: a pure function of (seed, p) with repeat families folded in, so no genome array is needed
struct Genome {
    uint64_t seed; int64_t L; int32_t fams; int64_t rlen;
    std::vector<int64_t> copy_start, copy_fam;   // sorted copy intervals
    inline int raw(int64_t p) const { return (int)(hash2(seed ^ 0xA5A5A5A5ULL, (uint64_t)(p >> 5)) >> (2 * (p & 31)) & 3); }
    inline int fam_base(int f, int64_t o) const { return (int)(hash2(seed ^ (0xF00DULL + (uint64_t)f), (uint64_t)(o >> 5)) >> (2 * (o & 31)) & 3); }
    int at(int64_t p) const
    {
        if (!copy_start.empty()) {
            // last copy with start <= p
            size_t lo = 0, hi = copy_start.size();
            while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (copy_start[mid] <= p) lo = mid; else hi = mid; }
            if (copy_start[lo] <= p && p < copy_start[lo] + rlen) return fam_base((int)copy_fam[lo], p - copy_start[lo]);
        }
        return raw(p);
    }
};

Genome make_genome(const elba_synth_cfg &c)
{
    Genome g{c.seed, c.genome_length, c.repeat_families, c.repeat_len > 0 ? c.repeat_len : 0, {}, {}};
    if (c.repeat_families > 0 && c.repeat_fraction > 0 && g.rlen > 0) {
        int64_t ncopies = (int64_t)(c.repeat_fraction * (double)c.genome_length / (double)g.rlen);
        int64_t slot = ncopies > 0 ? c.genome_length / ncopies : 0;
        uint64_t s = c.seed ^ 0x5EEDULL;
        for (int64_t i = 0; i < ncopies && slot > g.rlen; ++i) {
            int64_t st = i * slot + (int64_t)(u01(s) * (double)(slot - g.rlen));
            g.copy_start.push_back(st);
            g.copy_fam.push_back((int64_t)(splitmix(s) % (uint64_t)c.repeat_families));
        }
    }
    return g;
}

}  // namespace

extern "C" {

int64_t elba_synth_num_reads(const elba_synth_cfg *c)
{
    if (!c || c->avg_len <= 0) return 0;
    return (int64_t)((double)c->genome_length * c->depth / c->avg_len);     // runs/simfor.py:10
}

int elba_synth_generate(const elba_synth_cfg *c, elba_synth_reads *out)
{
    if (!c || !out || c->genome_length <= 0 || c->avg_len <= 0) return 1;
    memset(out, 0, sizeof(*out));
    const int64_t total = elba_synth_num_reads(c);
    const int64_t first = c->first_read < 0 ? 0 : c->first_read;
    int64_t n = c->num_reads < 0 ? total - first : c->num_reads;
    if (first + n > total) n = total - first;
    if (n < 0) n = 0;
    Genome g = make_genome(*c);
    const int64_t min_len = c->min_len > 0 ? c->min_len : 1;
    std::vector<std::vector<uint8_t>> codes((size_t)n);
    out->len = (uint32_t *)malloc((size_t)(n + 1) * 4);
    out->byte_off = (uint64_t *)malloc((size_t)(n + 1) * 8);
    out->genome_pos = (int64_t *)malloc((size_t)(n + 1) * 8);
    out->strand = (uint8_t *)malloc((size_t)(n + 1));
    int64_t bytes = 0, bases = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bases)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t s = hash2(c->seed, (uint64_t)(first + i));
        int64_t ln = (int64_t)(c->avg_len + c->sd_len * gauss(s));
        if (ln < min_len) ln = min_len;
        int64_t maxpos = c->genome_length - min_len;
        int64_t pos = maxpos > 0 ? (int64_t)(u01(s) * (double)maxpos) : 0;
        if (pos + ln > c->genome_length) ln = c->genome_length - pos;                 // runs/simfor.py:24-25
        int strand = (int)(splitmix(s) & 1);
        std::vector<uint8_t> &v = codes[(size_t)i];
        v.reserve((size_t)(ln + ln / 8 + 8));
        const double e = c->error_rate;
        for (int64_t p = 0; p < ln; ++p) {
            int b = g.at(pos + p);
            if (e > 0) {
                double r = u01(s);
                if (r < e) {
                    int kind = (int)(splitmix(s) % 3);
                    if (kind == 0) { v.push_back((uint8_t)((b + 1 + (int)(splitmix(s) % 3)) & 3)); }       // substitution
                    else if (kind == 1) { v.push_back((uint8_t)b); v.push_back((uint8_t)(splitmix(s) & 3)); }  // insertion
                    /* kind == 2: deletion */
                    continue;
                }
            }
            v.push_back((uint8_t)b);
        }
        if (strand) {                                                                 // reverse complement (runs/simfor.py:103)
            size_t m = v.size();
            for (size_t a = 0; a < m / 2; ++a) { uint8_t t = v[a]; v[a] = (uint8_t)(3 - v[m - 1 - a]); v[m - 1 - a] = (uint8_t)(3 - t); }
            if (m & 1) v[m / 2] = (uint8_t)(3 - v[m / 2]);
        }
        out->len[i] = (uint32_t)v.size();
        out->genome_pos[i] = pos;
        out->strand[i] = (uint8_t)strand;
        bases += (int64_t)v.size();
    }
    for (int64_t i = 0; i < n; ++i) { out->byte_off[i] = (uint64_t)bytes; bytes += ((int64_t)out->len[i] + 3) / 4; }
    out->packed = (uint8_t *)calloc((size_t)bytes + 16, 1);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        uint8_t *m = out->packed + out->byte_off[i];
        const std::vector<uint8_t> &v = codes[(size_t)i];
        for (size_t p = 0; p < v.size(); ++p) m[p >> 2] |= (uint8_t)(v[p] << (6 - 2 * (p & 3)));   // src/DnaSeq.cpp:18-24 layout
    }
    out->nreads = n; out->total_reads = total; out->packed_bytes = bytes; out->total_bases = bases;
    return 0;
}

void elba_synth_free(elba_synth_reads *r)
{
    if (!r) return;
    free(r->packed); free(r->byte_off); free(r->len); free(r->genome_pos); free(r->strand);
    memset(r, 0, sizeof(*r));
}

}  // extern "C"
