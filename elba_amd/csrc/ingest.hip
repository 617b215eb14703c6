// ingest.hip — FASTA chunk -> 2-bit DnaBuffer on the device (the step right before the path).
//
// Replaces the per-record loop of FastaIndex::getmydna (src/FastaIndex.cpp:256-283: copy `bases` characters per line into a
// temporary buffer, then DnaBuffer::push_back -> DnaSeq::compress, src/DnaSeq.cpp:7-29) by one kernel over the raw file chunk:
// record (len, pos, bases) of the .fai (include/FastaIndex.hpp:10) places base b of the read at file offset pos + b + b / bases
// (one newline byte per full line, src/FastaIndex.cpp:264-271); four bases make one output byte, first base in bits 7-6, codes
// A/a/N/n 0, C/c 1, G/g 2, T/t 3 (include/DnaSeq.hpp:136-154).  Any other character gets code 4, whose shifted value is ORed in
// truncated to 8 bits — what the reference's compress does with it (its own comment calls it undefined) — so the bytes are identical
// for every input, not only for clean ones.
//
// HBM streaming work: 4 (+ newlines) bytes read per byte written.  One workgroup per read, 16 bases (4 output bytes, one 32-bit
// store) per lane and trip.
#include "common.hpp"

namespace elba {

namespace {

__device__ __forceinline__ uint32_t char_code(uint8_t ch)
{
    switch (ch) {
    case 'A': case 'a': case 'N': case 'n': return 0u;
    case 'C': case 'c': return 1u;
    case 'G': case 'g': return 2u;
    case 'T': case 't': return 3u;
    default: return 4u;
    }
}

// One lane encodes 16 consecutive bases into one 32-bit word of the output (the packed bytes of a read are byte-ordered, first base in
// bits 7-6 of byte 0: the word is assembled byte by byte and stored little-endian; a read's buffer starts on a 4-byte boundary only
// by accident, so the tail and unaligned reads fall back to byte stores).  Per trip a workgroup handles 256 words = 4096 bases: the
// bytes of the file that hold them (4096 + one newline per line: at most 8192 for one-base lines) are first brought into LDS with
// aligned 16-byte loads, consecutive lanes consecutive addresses — the file is read exactly once, coalesced — and the lanes then
// pick their characters from LDS.  The line of a lane's first base costs one 32-bit division per trip; the character codes come from
// a 256-entry table in LDS.
constexpr uint32_t ENC_THREADS = 256, ENC_BASES = 16 * ENC_THREADS, ENC_SPAN = 2 * ENC_BASES + 32;
__global__ __launch_bounds__(ENC_THREADS) void k_fasta_encode(const uint8_t *chunk, uint64_t chunk_off, uint64_t chunk_bytes, const elba_fasta_record_t *recs,
                                                              const uint64_t *byte_off, uint32_t nreads, uint8_t *packed)
{
    __shared__ uint8_t lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t span[ENC_SPAN];
    lut[threadIdx.x] = (uint8_t)char_code((uint8_t)threadIdx.x);
    __syncthreads();
    for (uint32_t r = blockIdx.x; r < nreads; r += gridDim.x) {
        const uint32_t len = (uint32_t)recs[r].len;
        const uint64_t bases64 = recs[r].bases;
        const uint32_t bases = bases64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bases64;      // a line longer than the read: never wraps
        const uint64_t rec0 = recs[r].pos - chunk_off;                                          // offset of the record's first base in the chunk
        uint8_t *out = packed + byte_off[r];
        const uint32_t nbytes = (len + 3) / 4, nwords = (nbytes + 3) / 4;
        const bool aligned = (reinterpret_cast<uintptr_t>(out) & 3u) == 0;
        for (uint32_t w0 = 0; w0 < nwords; w0 += ENC_THREADS) {
            // file bytes of bases [16 w0, 16 w0 + 4096): from the first base's offset to the last one's, aligned down to 16
            const uint32_t pb = 16 * w0, pe = pb + ENC_BASES < len ? pb + ENC_BASES : len;      // [pb, pe)
            const uint64_t f0 = rec0 + pb + pb / bases, f1 = rec0 + (pe - 1) + (pe - 1) / bases + 1;       // [f0, f1) in the chunk
            const uint64_t a0 = f0 & ~15ull;
            const uint32_t nvec = (uint32_t)((f1 - a0 + 15) / 16);                               // <= ENC_SPAN / 16
            for (uint32_t v = threadIdx.x; v < nvec; v += ENC_THREADS) {
                const uint64_t at = a0 + 16ull * v;
                uint4 x = make_uint4(0x58585858u, 0x58585858u, 0x58585858u, 0x58585858u);       // 'X' beyond the chunk: code 4, like a missing byte
                if (at + 16 <= chunk_bytes) x = *reinterpret_cast<const uint4 *>(chunk + at);
                else for (uint32_t q = 0; q < 16 && at + q < chunk_bytes; ++q) reinterpret_cast<uint8_t *>(&x)[q] = chunk[at + q];
                *reinterpret_cast<uint4 *>(span + 16u * v) = x;
            }
            __syncthreads();
            const uint32_t w = w0 + threadIdx.x;
            if (w < nwords) {
                const uint32_t p0 = 16 * w;
                uint32_t line = p0 / bases, rem = p0 - line * bases;
                uint32_t word = 0;
                const uint32_t sbase = (uint32_t)(rec0 - a0);                                    // span offset of the record's first base (mod the trip)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (p0 + (uint32_t)i < len) {
                        const uint32_t code = lut[span[sbase + p0 + (uint32_t)i + line]];
                        const uint32_t byte = (code << (6 - 2 * (i & 3))) & 0xFFu;
                        word |= byte << (8 * (i >> 2));
                        if (++rem == bases) { rem = 0; ++line; }
                    }
                }
                const uint32_t b0 = 4 * w;
                if (aligned && b0 + 4 <= nbytes) *reinterpret_cast<uint32_t *>(out + b0) = word;
                else for (uint32_t x = 0; x < 4 && b0 + x < nbytes; ++x) out[b0 + x] = (uint8_t)(word >> (8 * x));
            }
            __syncthreads();
        }
    }
}

}  // namespace

void stage_set_reads_fasta(Ctx &c, const char *chunk, int64_t chunk_bytes, uint64_t chunk_file_offset, const elba_fasta_record_t *recs, int64_t nreads,
                           int64_t first_global_id, elba_ingest_stats *stats)
{
    ELBA_REQUIRE(nreads >= 0 && chunk_bytes >= 0 && (nreads == 0 || (chunk && recs)), ELBA_ERR_INVALID_ARG, "set_reads_fasta: null array");
    ELBA_REQUIRE(nreads < 0xFFFFFFFFll, ELBA_ERR_UNSUPPORTED, "set_reads_fasta: more than 2^32-1 reads on one GPU");
    hipStream_t s = c.stream;
    std::vector<uint64_t> off((size_t)nreads + 1);
    std::vector<uint32_t> len((size_t)nreads);
    uint64_t pb = 0, totbases = 0;
    for (int64_t r = 0; r < nreads; ++r) {
        ELBA_REQUIRE(recs[r].len < 0x7FFFFFF0ull && recs[r].bases > 0, ELBA_ERR_INVALID_ARG, "set_reads_fasta: bad .fai record (length >= 2^31 or zero line width)");
        ELBA_REQUIRE(recs[r].pos >= chunk_file_offset, ELBA_ERR_INVALID_ARG, "set_reads_fasta: record starts before the chunk");
        // last base of the record must lie inside the chunk (src/FastaIndex.cpp:222-224 sizes the chunk the same way)
        const uint64_t last = recs[r].len ? recs[r].pos - chunk_file_offset + (recs[r].len - 1) + (recs[r].len - 1) / recs[r].bases : 0;
        ELBA_REQUIRE(recs[r].len == 0 || last < (uint64_t)chunk_bytes, ELBA_ERR_INVALID_ARG, "set_reads_fasta: record runs past the chunk");
        off[(size_t)r] = pb; len[(size_t)r] = (uint32_t)recs[r].len;
        pb += (recs[r].len + 3) / 4;                 // every read starts on a byte boundary (src/DnaBuffer.cpp:22-29)
        totbases += recs[r].len;
    }
    off[(size_t)nreads] = pb;
    c.t_total.start(s);
    DevBuf d_chunk, d_recs;
    d_chunk.reserve((size_t)chunk_bytes + 16); d_recs.reserve((size_t)(nreads + 1) * sizeof(elba_fasta_record_t));
    c.own_packed.reserve((size_t)pb + 16);          // +16: the enumerate kernel reads whole 8-byte windows
    c.own_byte_off.reserve((size_t)(nreads + 1) * 8);
    c.own_len.reserve((size_t)(nreads + 1) * 4);
    ELBA_HIP(hipMemsetAsync(c.own_packed.p, 0, (size_t)pb + 16, s));
    if (chunk_bytes) ELBA_HIP(hipMemcpyAsync(d_chunk.p, chunk, (size_t)chunk_bytes, hipMemcpyHostToDevice, s));
    if (nreads) {
        ELBA_HIP(hipMemcpyAsync(d_recs.p, recs, (size_t)nreads * sizeof(elba_fasta_record_t), hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.own_byte_off.p, off.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, s));
        ELBA_HIP(hipMemcpyAsync(c.own_len.p, len.data(), (size_t)nreads * 4, hipMemcpyHostToDevice, s));
        c.t_a.start(s);
        int64_t nb = nreads < (int64_t)c.num_cus * 16 ? nreads : (int64_t)c.num_cus * 16;
        hipLaunchKernelGGL(k_fasta_encode, dim3((unsigned)nb), dim3(256), 0, s, d_chunk.as<uint8_t>(), chunk_file_offset, (uint64_t)chunk_bytes, d_recs.as<elba_fasta_record_t>(),
                           c.own_byte_off.as<uint64_t>(), (uint32_t)nreads, c.own_packed.as<uint8_t>());
        c.t_a.stop(s);
        ELBA_HIP(hipGetLastError());
    }
    c.t_total.stop(s);
    ELBA_HIP(hipStreamSynchronize(s));
    c.d_packed = c.own_packed.as<uint8_t>(); c.d_byte_off = c.own_byte_off.as<uint64_t>(); c.d_len = c.own_len.as<uint32_t>();
    c.h_len = len; off.pop_back(); c.h_byte_off = off;
    c.nreads = nreads; c.first_global_id = first_global_id; c.packed_bytes = (int64_t)pb;
    c.have_reads = true; c.have_counts = false; c.have_aln = false; c.have_edges = false; c.have_S = false;
    if (c.A_has_kmers) { c.have_A = false; c.have_B = false; }
    if (stats) {
        stats->nreads = nreads; stats->bases = (int64_t)totbases; stats->packed_bytes = (int64_t)pb; stats->chunk_bytes = chunk_bytes;
        stats->ms_total = c.t_total.ms(); stats->ms_encode = nreads ? c.t_a.ms() : 0.f;
    }
}

}  // namespace elba
