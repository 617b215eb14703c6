"""Small numpy generator of synthetic long-read sets for tests (statistical spec of the reference's
runs/simfor.py:8-32 — random genome, reads at uniform positions with N(avg, sd) lengths, random strand — plus an
optional substitution/insertion/deletion error model).  Not the reference's RNG stream; fixtures are committed."""
import numpy as np

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(s: bytes) -> bytes:
    return s.translate(_COMP)[::-1]


def make_reads(seed, genome_length, depth, avg_len, sd_len, error=0.0, min_len=50, n_frac=0.0, repeats=0):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_length, dtype=np.uint8)
    for _ in range(repeats):  # copy a segment elsewhere: creates high-multiplicity k-mers
        ln = int(min(genome_length // 10, 2000))
        a = int(rng.integers(0, genome_length - ln)); b = int(rng.integers(0, genome_length - ln))
        genome[b:b + ln] = genome[a:a + ln]
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    gbytes = letters[genome].tobytes()
    nreads = int(genome_length * depth / avg_len)
    reads, truth = [], []
    for i in range(nreads):
        ln = max(min_len, int(rng.normal(avg_len, sd_len)))
        pos = int(rng.integers(0, max(1, genome_length - min_len)))
        ln = min(ln, genome_length - pos)
        s = bytearray(gbytes[pos:pos + ln])
        if error > 0:
            out = bytearray()
            r = rng.random(len(s))
            kinds = rng.integers(0, 3, len(s))
            subs = rng.integers(0, 4, len(s))
            for j, ch in enumerate(s):
                if r[j] < error:
                    if kinds[j] == 0:
                        out.append(b"ACGT"[subs[j]])
                    elif kinds[j] == 1:
                        out.append(ch); out.append(b"ACGT"[subs[j]])
                    # kinds == 2: deletion
                else:
                    out.append(ch)
            s = out
        if n_frac > 0:
            mask = rng.random(len(s)) < n_frac
            for j in np.nonzero(mask)[0]:
                s[j] = ord("N") if (j & 1) else ord("n")
        strand = int(rng.integers(0, 2))
        b = bytes(s)
        if strand:
            b = revcomp(b.upper().replace(b"N", b"A")) if n_frac == 0 else revcomp(b.upper().replace(b"N", b"A"))
        reads.append(b)
        truth.append((pos, ln, strand))
    return reads, truth
