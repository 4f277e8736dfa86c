"""Synthetic read sets of SURVEY §8d, regenerated identically on every box.

Set s uses numpy.random.default_rng(seed_base + s).  Reads are uniform i.i.d.
over ACGT, fixed length L.  For s > 0 the first `copy_frac` of the reads are
copies of the same-index reads of set 0 with `sub_rate` per-base substitutions,
every `rc_every`-th copied read reverse-complemented.  `n_rate` of all bases
become 'N'.  Substitution and N positions are drawn as (count ~ Binomial,
positions uniform with replacement) so the generator stays O(output) in memory.
"""
import numpy as np

# byte -> ASCII base of its low two bits (A,C,G,T = 0,1,2,3)
_FOLD = bytes(b"ACGT"[i & 3] for i in range(256))


def _raw(seed, n_bytes):
    """n_bytes of the PCG64 byte stream of `seed` (a prefix of any longer request)."""
    rng = np.random.default_rng(seed)
    return rng, bytearray(rng.bytes(n_bytes))


def synth_set(set_id, n_reads, read_len, seed_base=1000, copy_frac=0.25, sub_rate=0.01, rc_every=20, n_rate=0.001,
              base_set=0):
    """Returns (bases uint8[n*L] ASCII, offsets uint64[n+1]).  Base code = random byte & 3.
    Copies are taken from set `base_set` (0 in SURVEY 8d; bench.py gives every GPU its own pair)."""
    total = n_reads * read_len
    rng, raw = _raw(seed_base + set_id, total)
    codes = np.frombuffer(raw, dtype=np.uint8)          # writable view
    if set_id != base_set and copy_frac > 0:
        ncopy = int(n_reads * copy_frac)
        if ncopy:
            _, raw0 = _raw(seed_base + base_set, ncopy * read_len)  # the base set's first ncopy reads
            cp = np.frombuffer(raw0, dtype=np.uint8).reshape(ncopy, read_len)
            nsub = int(rng.binomial(ncopy * read_len, sub_rate))
            pos = rng.integers(0, ncopy * read_len, size=nsub)
            delta = rng.integers(1, 4, size=nsub).astype(np.uint8)
            flat = cp.reshape(-1)
            flat[pos] = (flat[pos] + delta) & 3
            if rc_every:
                idx = np.arange(0, ncopy, rc_every)
                cp[idx] = 3 - (cp[idx, ::-1] & 3)        # A<->T, C<->G
            codes[:ncopy * read_len] = flat
            del cp, flat, raw0
    # codes -> ASCII in place, a block at a time (a 50 M-read set is 5 GB: no second copy of it)
    lut = np.frombuffer(_FOLD, dtype=np.uint8)
    for i in range(0, total, 1 << 26):
        codes[i:i + (1 << 26)] = lut[codes[i:i + (1 << 26)]]
    bases = codes
    nn = int(rng.binomial(total, n_rate)) if n_rate > 0 else 0
    if nn:
        bases[rng.integers(0, total, size=nn)] = ord("N")
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return bases, offsets


def skew_set(bases, n_reads, read_len, set_id, frac=0.10, seed_base=7000, library_reads=1000):
    """Overwrites a `frac` of the reads of a synthetic set, in place, with what real data is full of and i.i.d. reads
    never show (SURVEY 7's keyd / hot-bucket warning): a third poly-A (one base in a hundred substituted), a third short
    tandem repeats (a random unit of 2..6 bases repeated over the read), a third reads drawn from a library of
    `library_reads` fixed reads that every set shares (seed_base alone), i.e. ~frac/3 * n / library_reads copies of each.
    Which reads are replaced depends on the set (seed_base + set_id).  Returns the indices replaced."""
    rng = np.random.default_rng(seed_base + 1 + set_id)
    m = int(n_reads * frac)
    if m == 0:
        return np.zeros(0, dtype=np.int64)
    idx = rng.choice(n_reads, m, replace=False)
    view = np.asarray(bases).reshape(n_reads, read_len)
    a, b = m // 3, 2 * (m // 3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    # poly-A with 1 % substitutions
    pa = np.full((a, read_len), ord("A"), dtype=np.uint8)
    nsub = int(rng.binomial(a * read_len, 0.01))
    pa.reshape(-1)[rng.integers(0, a * read_len, size=nsub)] = acgt[rng.integers(1, 4, size=nsub)]
    view[idx[:a]] = pa
    # short tandem repeats: unit length 2..6
    ulen = rng.integers(2, 7, size=b - a)
    units = acgt[rng.integers(0, 4, size=(b - a, 6))]
    pos = np.arange(read_len)[None, :] % ulen[:, None]
    view[idx[a:b]] = np.take_along_axis(units, pos, axis=1)
    # the shared repeat library
    lib = acgt[np.random.default_rng(seed_base).integers(0, 4, size=(library_reads, read_len))]
    view[idx[b:]] = lib[rng.integers(0, library_reads, size=m - b)]
    return idx


def synth_set_skewed(set_id, n_reads, read_len, frac=0.10, **kw):
    """synth_set with a `frac` of the reads replaced by low-complexity / repeated ones (skew_set)"""
    bases, offsets = synth_set(set_id, n_reads, read_len, **kw)
    if not bases.flags.writeable:
        bases = np.array(bases)
    skew_set(bases, n_reads, read_len, set_id, frac)
    return bases, offsets


def write_fasta(path, bases, offsets, width=0, lowercase_every=0):
    """Header '>i', one sequence line (or `width`-column lines)."""
    b = np.asarray(bases, dtype=np.uint8).tobytes()
    with open(path, "wb") as fh:
        for i in range(len(offsets) - 1):
            s = b[int(offsets[i]):int(offsets[i + 1])]
            if lowercase_every and i % lowercase_every == 0:
                s = s.lower()
            fh.write(b">%d\n" % i)
            if width and len(s) > width:
                for j in range(0, len(s), width):
                    fh.write(s[j:j + width] + b"\n")
            else:
                fh.write(s + b"\n")


def write_fasta_fast(path, bases, n_reads, read_len, digits=9):
    """Vectorised writer for fixed-length sets: header '>%0<digits>d', one sequence line."""
    rec = 1 + digits + 1 + read_len + 1
    view = np.asarray(bases, dtype=np.uint8).reshape(n_reads, read_len)
    block = 1 << 20                                       # records per write: the file image is never held whole
    with open(path, "wb") as fh:
        for r0 in range(0, n_reads, block):
            m = min(block, n_reads - r0)
            buf = np.empty((m, rec), dtype=np.uint8)
            buf[:, 0] = ord(">")
            idx = np.arange(r0, r0 + m, dtype=np.int64)
            for d in range(digits):
                buf[:, digits - d] = (idx % 10 + ord("0")).astype(np.uint8)
                idx //= 10
            buf[:, 1 + digits] = ord("\n")
            buf[:, 2 + digits:2 + digits + read_len] = view[r0:r0 + m]
            buf[:, -1] = ord("\n")
            fh.write(buf.data)


def write_set_fasta(args):
    """(set id, reads, read length, path): generates one synthetic set and writes it as FASTA.  Top-level so that a
    spawn pool can run it (bench.py's matrix leg, the full-size tests)."""
    s, n, L, path = args
    b, _ = synth_set(s, n, L)
    write_fasta_fast(path, b, n, L)
    return s


# ---- ragged sets: trimmed reads of many lengths (what a .fq.gz run really holds, fastq_file.h:139-190) ---------------------

def ragged_lengths(set_id, n_reads, lo, hi, seed_base=1000, copy_frac=0.25, base_set=0):
    """Read lengths of ragged set `set_id`: uniform in [lo, hi], seeded per set; the copied reads (the first `copy_frac`
    of a set other than `base_set`) have the lengths of the reads they copy."""
    lens = np.random.default_rng(seed_base + 5_000_000 + set_id).integers(lo, hi + 1, size=n_reads, dtype=np.int64)
    if set_id != base_set and copy_frac > 0:
        ncopy = int(n_reads * copy_frac)
        if ncopy:
            lens[:ncopy] = np.random.default_rng(seed_base + 5_000_000 + base_set).integers(lo, hi + 1, size=n_reads, dtype=np.int64)[:ncopy]
    return lens


def synth_set_ragged(set_id, n_reads, lo, hi, seed_base=1000, copy_frac=0.25, sub_rate=0.01, rc_every=20, n_rate=0.001,
                     base_set=0):
    """synth_set with read lengths uniform in [lo, hi] — the same copy / substitution / reverse-complement / N rules.
    Returns (bases uint8[total] ASCII, offsets uint64[n+1])."""
    lens = ragged_lengths(set_id, n_reads, lo, hi, seed_base, copy_frac, base_set)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:].view(np.int64))
    total = int(offsets[-1])
    rng, raw = _raw(seed_base + set_id, total)
    codes = np.frombuffer(raw, dtype=np.uint8)
    if set_id != base_set and copy_frac > 0:
        ncopy = int(n_reads * copy_frac)
        if ncopy:
            nb = int(offsets[ncopy])                      # the base set's first ncopy reads have the same lengths: the same bytes
            _, raw0 = _raw(seed_base + base_set, nb)
            flat = np.frombuffer(raw0, dtype=np.uint8)
            nsub = int(rng.binomial(nb, sub_rate))
            pos = rng.integers(0, nb, size=nsub)
            delta = rng.integers(1, 4, size=nsub).astype(np.uint8)
            flat[pos] = (flat[pos] + delta) & 3
            if rc_every:
                for i in range(0, ncopy, rc_every):
                    a, b = int(offsets[i]), int(offsets[i + 1])
                    flat[a:b] = 3 - (flat[a:b][::-1] & 3)
            codes[:nb] = flat
            del flat, raw0
    lut = np.frombuffer(_FOLD, dtype=np.uint8)
    for i in range(0, total, 1 << 26):
        codes[i:i + (1 << 26)] = lut[codes[i:i + (1 << 26)]]
    nn = int(rng.binomial(total, n_rate)) if n_rate > 0 else 0
    if nn:
        codes[rng.integers(0, total, size=nn)] = ord("N")
    return codes, offsets


def write_fasta_ragged(path, bases, offsets, digits=9):
    """Vectorised writer for ragged sets: header '>%0<digits>d', one sequence line per read."""
    n = len(offsets) - 1
    offs = np.asarray(offsets, dtype=np.int64)
    src = np.asarray(bases, dtype=np.uint8)
    hdr = 1 + digits + 1                                   # '>' + digits + '\n'
    block = 1 << 18
    with open(path, "wb") as fh:
        for r0 in range(0, n, block):
            m = min(block, n - r0)
            o = offs[r0:r0 + m + 1] - offs[r0]
            lens = np.diff(o)
            rec0 = o[:-1] + np.arange(m, dtype=np.int64) * (hdr + 1)      # start of record i in the block's image
            out = np.empty(int(o[-1]) + m * (hdr + 1), dtype=np.uint8)
            out[rec0] = ord(">")
            idx = np.arange(r0, r0 + m, dtype=np.int64)
            for d in range(digits):
                out[rec0 + (digits - d)] = (idx % 10 + ord("0")).astype(np.uint8)
                idx //= 10
            out[rec0 + digits + 1] = ord("\n")
            out[rec0 + hdr + lens] = ord("\n")
            # the sequence bytes fill, in order, every place that is not part of a header or a line end
            seq = np.ones(out.size, dtype=bool)
            seq[(rec0[:, None] + np.arange(hdr, dtype=np.int64)[None, :]).reshape(-1)] = False
            seq[rec0 + hdr + lens] = False
            out[seq] = src[int(offs[r0]):int(offs[r0 + m])]
            fh.write(out.data)


def write_set_fasta_ragged(args):
    """(set id, reads, lo, hi, path): one ragged synthetic set as FASTA (top-level: a spawn pool runs it)"""
    s, n, lo, hi, path = args
    b, o = synth_set_ragged(s, n, lo, hi)
    write_fasta_ragged(path, b, o)
    return s
