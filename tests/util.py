"""Shared helpers: small random FASTA sets with all the reference's corner cases."""
import os
import subprocess

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_reads(rng, n, len_lo, len_hi, n_rate=0.01, lower_rate=0.1, other_rate=0.002):
    """list of bytes objects"""
    out = []
    for _ in range(n):
        L = int(rng.integers(len_lo, len_hi + 1))
        s = ACGT[rng.integers(0, 4, size=L)].copy()
        if L:
            m = rng.random(L)
            s[m < n_rate] = ord("N")
            s[(m >= n_rate) & (m < n_rate + other_rate)] = rng.choice(np.frombuffer(b"RYKM-.*", dtype=np.uint8))
            if rng.random() < lower_rate:
                s = np.frombuffer(s.tobytes().lower(), dtype=np.uint8)
        out.append(s.tobytes())
    return out


def mutate(rng, read, sub_rate=0.02):
    s = np.frombuffer(read, dtype=np.uint8).copy()
    m = rng.random(len(s)) < sub_rate
    s[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
    return s.tobytes()


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(read):
    return read.translate(_COMP)[::-1]


def related_reads(rng, pool, n, len_lo, len_hi, share=0.4, **kw):
    """n reads, a `share` of them derived from `pool` (copies, mutated copies, reverse complements, substrings)."""
    fresh = random_reads(rng, n, len_lo, len_hi, **kw)
    out = []
    for i in range(n):
        if pool and rng.random() < share:
            r = pool[int(rng.integers(0, len(pool)))]
            mode = int(rng.integers(0, 4))
            if mode == 1:
                r = mutate(rng, r)
            elif mode == 2:
                r = revcomp(r)
            elif mode == 3 and len(r) > 8:
                a = int(rng.integers(0, len(r) // 2))
                r = r[a:a + max(4, len(r) // 2)] + fresh[i][:len(r) // 3]
            out.append(r)
        else:
            out.append(fresh[i])
    return out


def write_fasta(path, reads, rng=None, multiline=False, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    with open(path, "wb") as fh:
        for i, r in enumerate(reads):
            fh.write(b">r%d some text" % i + nl)
            if multiline and rng is not None and len(r) > 10 and rng.random() < 0.5:
                w = int(rng.integers(5, max(6, len(r))))
                for j in range(0, len(r), w):
                    fh.write(r[j:j + w] + nl)
                if rng.random() < 0.2:
                    fh.write(nl)            # stray blank line inside a record
            else:
                fh.write(r + nl)


def to_batch(reads):
    """(bases uint8, offsets uint64) of a list of bytes reads"""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, dtype=np.uint8)
    return bases, offs


def bits_from_bools(b):
    b = np.asarray(b, dtype=bool)
    out = np.zeros(len(b) // 8 + 1, dtype=np.uint8)
    packed = np.packbits(b, bitorder="little")
    out[:len(packed)] = packed
    return out


def bools_from_bits(bits, n):
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def write_bv(path, comment, bools):
    bits = bits_from_bools(bools)
    with open(path, "wb") as fh:
        fh.write(comment.encode() + b"\n#%d\n" % len(bools))
        fh.write(bits.tobytes())


def read_bv(path):
    data = open(path, "rb").read()
    h = data.index(b"#")
    nl = data.index(b"\n", h)
    n = int(data[h + 1:nl])
    raw = data[nl + 1:nl + 1 + n // 8 + 1]
    return data[:h - 1].decode(errors="replace"), n, np.frombuffer(raw, dtype=np.uint8)


def run(cmd, cwd=None, check=True):
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if check and p.returncode != 0:
        raise RuntimeError(f"{cmd} failed rc={p.returncode}\n{p.stdout.decode()[-2000:]}\n{p.stderr.decode()[-2000:]}")
    return p


def last_log_line(path):
    lines = open(path).read().strip().split("\n")
    return lines[-1]


def parse_fasta(path):
    """Records the way the reference reads them (fasta_file.h:155-175): header
    lines start with '>', every other non-empty line is appended verbatim
    (only the '\\n' is stripped, a '\\r' stays in the sequence)."""
    reads = []
    cur = None
    with open(path, "rb") as fh:
        for line in fh.read().split(b"\n"):
            if line.startswith(b">"):
                if cur is not None:
                    reads.append(b"".join(cur))
                cur = []
            elif cur is not None and line:
                cur.append(line)
    if cur is not None:
        reads.append(b"".join(cur))
    return reads


def write_fastq(path, reads, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    with open(path, "wb") as fh:
        for i, r in enumerate(reads):
            fh.write(b"@r%d text" % i + nl + r + nl + b"+" + nl + b"I" * len(r) + nl)


def write_reads(path, reads, fmt, rng=None, multiline=False, crlf=False):
    """fmt: fa | fq | fa.gz | fq.gz"""
    import gzip
    import shutil
    plain = path[:-3] + ".tmp" if fmt.endswith(".gz") else path
    if fmt.startswith("fq"):
        write_fastq(plain, reads, crlf=crlf)
    else:
        write_fasta(plain, reads, rng=rng, multiline=multiline, crlf=crlf)
    if fmt.endswith(".gz"):
        with open(plain, "rb") as fi, gzip.GzipFile(path, "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
        os.remove(plain)


def parse_reads(path):
    """sequences of a read file of any supported format, as the tools see them"""
    import gzip
    data = open(path, "rb").read()
    if data[:2] == b"\x1f\x8b":
        data = gzip.decompress(data)
    if data[:1] == b"@":
        lines = data.split(b"\n")
        out, i = [], 0
        n = sum(1 for ln in lines if ln) // 4
        for _ in range(n):
            while i < len(lines) and not lines[i]:
                i += 1
            i += 1
            out.append(lines[i] if i < len(lines) else b"")
            i += 1
            while i < len(lines) and not lines[i]:
                i += 1
            i += 1
            while i < len(lines) and not lines[i]:
                i += 1
            i += 1
        return out
    tmp = path + ".parse_tmp"
    open(tmp, "wb").write(data)
    try:
        return parse_fasta(tmp)
    finally:
        os.remove(tmp)


def gen_set_fasta(args):
    """(set id, reads, read length, path): writes one synthetic set (commet_amd.synth) as FASTA.
    Top-level so that a spawn pool can run it."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from commet_amd import synth
    return synth.write_set_fasta(args)
