"""Host side of the FASTA ingest: the .fai records and the file chunk a rank needs — the parts of the reference's FastaIndex that stay on
the host (include/FastaIndex.hpp, src/FastaIndex.cpp).  The 2-bit encoding itself runs on the GPU (elba_set_reads_fasta, ingest.hip)."""
import numpy as np

# elba_fasta_record_t == FastaIndex::Record (include/FastaIndex.hpp:10)
FAI_DTYPE = np.dtype([("len", "<u8"), ("pos", "<u8"), ("bases", "<u8")])


def read_fai(path):
    """`name len pos bases [width]` per line -> (names, records); get_faidx_record, src/FastaIndex.cpp:15-23."""
    names, recs = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 4:
                continue
            names.append(t[0]); recs.append((int(t[1]), int(t[2]), int(t[3])))
    return names, np.array(recs, dtype=FAI_DTYPE).reshape(-1)


def write_fai(fasta_path, fai_path=None):
    """samtools faidx for plain FASTA (samtools is not in the image): name, length, offset of the first base, bases per line, bytes per line."""
    fai_path = fai_path or fasta_path + ".fai"
    out = []
    with open(fasta_path, "rb") as f:
        data = f.read()
    i, n = 0, len(data)
    while i < n:
        assert data[i:i + 1] == b">", "not a FASTA record"
        e = data.index(b"\n", i)
        name = data[i + 1:e].split()[0].decode()
        pos = e + 1
        j = pos
        length, bases, width = 0, 0, 0
        while j < n and data[j:j + 1] != b">":
            le = data.find(b"\n", j)
            if le < 0:
                le = n
            if bases == 0:
                bases, width = le - j, le - j + 1
            length += le - j
            j = le + 1
        out.append("%s\t%d\t%d\t%d\t%d" % (name, length, pos, bases, width))
        i = j
    with open(fai_path, "w") as f:
        f.write("\n".join(out) + "\n")
    return fai_path


def chunk_bounds(recs, file_size):
    """[startpos, endpos) of the file bytes holding `recs` (consecutive records): src/FastaIndex.cpp:222-224."""
    first, last = recs[0], recs[-1]
    start = int(first["pos"])
    end = int(last["pos"]) + int(last["len"]) + int(last["len"]) // int(last["bases"])
    return start, min(end, file_size)


def load_chunk(fasta_path, recs):
    import os
    start, end = chunk_bounds(recs, os.path.getsize(fasta_path))
    with open(fasta_path, "rb") as f:
        f.seek(start)
        return f.read(end - start), start
