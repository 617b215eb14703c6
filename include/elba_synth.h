/*
 * elba_synth.h — synthetic long-read generator (host code, part of the benchmark/test harness, not of ELBA's API).
 *
 * A native restatement of the *statistical* spec of the reference's runs/simfor.py:8-32 (uniform random genome, reads at
 * uniform positions with N(avg, sd) lengths, random strand; `np.random.seed(313)` there — the numpy RNG stream is NOT
 * reproduced) extended with the substitution/insertion/deletion error model and repeat families that SURVEY.md §8d's
 * configs 2-5 ask for.  Output is the DnaBuffer 2-bit layout that elba_set_reads takes.
 */
#ifndef ELBA_SYNTH_H_
#define ELBA_SYNTH_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t seed;
    int64_t genome_length;
    double depth;
    double avg_len, sd_len;
    int64_t min_len;
    double error_rate;        /* per-base; split evenly into substitution / insertion / deletion */
    int32_t repeat_families;  /* number of repeat families copied around the genome */
    double repeat_fraction;   /* fraction of the genome covered by repeat copies */
    int64_t repeat_len;
    int64_t first_read, num_reads;  /* generate reads [first_read, first_read+num_reads) of the set; num_reads<0 = all */
} elba_synth_cfg;

typedef struct {
    int64_t nreads, total_reads, packed_bytes, total_bases;
    uint8_t *packed;      /* DnaBuffer bytes */
    uint64_t *byte_off;   /* [nreads] */
    uint32_t *len;        /* [nreads] */
    int64_t *genome_pos;  /* [nreads] ground truth */
    uint8_t *strand;      /* [nreads] */
} elba_synth_reads;

int64_t elba_synth_num_reads(const elba_synth_cfg *cfg);
int  elba_synth_generate(const elba_synth_cfg *cfg, elba_synth_reads *out);
void elba_synth_free(elba_synth_reads *r);

#ifdef __cplusplus
}
#endif
#endif
