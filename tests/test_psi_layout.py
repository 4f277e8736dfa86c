"""The strand-paired layout of filter plane A (psi_a, commet_amd/csrc/kernels.hpp), mirrored in numpy:
a bijection of [0, 2^k) with psi(T(key)) == psi(key) ^ 1, T(x) = ~bitreverse_k(x) being the map from a
window's forward keya to its reverse-complement keya (hash_key.h:63-123).  The device implementation itself
is exercised by the GPU parity tests (filter export + search results)."""
import numpy as np
import pytest

import oracle_binding as ob


def brev(x, bits):
    out = np.zeros_like(x)
    for i in range(bits):
        out |= ((x >> i) & 1) << (bits - 1 - i)
    return out


def psi(key, k):
    key = key.astype(np.uint64)
    h, odd = k >> 1, k & 1
    hmask = np.uint64((1 << h) - 1)
    L = key & hmask
    m = (key >> np.uint64(h)) & np.uint64(1) if odd else np.zeros_like(key)
    u = (key >> np.uint64(h + odd)) & hmask
    v = (~brev(L, h)) & hmask
    s = u ^ v
    b = np.zeros_like(s)
    nz = s != 0
    b[nz] = np.log2((s[nz] & (~s[nz] + np.uint64(1))).astype(np.float64)).astype(np.uint64)   # lowest set bit
    ub = (u >> b) & np.uint64(1)
    y = u ^ np.where(ub == 1, s & ~(np.uint64(1) << b), np.uint64(0))
    d = ((y >> b) ^ y) & np.uint64(1)
    y = y ^ (d | (d << b))
    if odd:
        addr = (s << np.uint64(h + 1)) | ((m ^ (y & np.uint64(1))) << np.uint64(h)) | y
        addr0 = (u << np.uint64(1)) | m
    else:
        addr = (s << np.uint64(h)) | y
        addr0 = u
    return np.where(nz, addr, addr0), (~nz) & (odd == 0)


@pytest.mark.parametrize("k", list(range(2, 17)))
def test_psi_is_a_bijection_pairing_the_two_strands(k):
    keys = np.arange(1 << k, dtype=np.uint64)
    a, selfp = psi(keys, k)
    assert np.array_equal(np.sort(a), keys)                      # bijection of [0, 2^k)
    T = (~brev(keys, k)) & np.uint64((1 << k) - 1)
    aT, _ = psi(T, k)
    assert np.array_equal(aT[~selfp], a[~selfp] ^ np.uint64(1))  # partner = neighbouring bit
    assert np.array_equal(T[selfp], keys[selfp])                 # self-paired keys are the fixed points of T


def test_T_is_the_forward_to_reverse_key_map():
    rng = np.random.default_rng(0)
    for k in (5, 8, 13, 20, 31, 32, 33):
        seq = "".join(rng.choice(list("ACGT"), size=k + 40))
        fw, _ = ob.keys_of_read(seq, k, reverse=False)
        rv, _ = ob.keys_of_read(seq, k, reverse=True)
        mask = (1 << k) - 1
        for (a, *_), (ar, *_) in zip(fw, rv):
            assert int(ar) == (~int(brev(np.array([a], dtype=np.uint64), k)[0])) & mask
