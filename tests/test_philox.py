"""Philox4x32-10 + uniform contract: the numpy oracle and the C oracle against
the Random123 known-answer vectors (kat_vectors of Random123 1.x)."""
import ctypes as C

import numpy as np

from oracle.philox import philox4x32_10, uniform53, TrackUniforms
from oracle import c_oracle

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_numpy_philox_kat():
    for ctr, key, want in KAT:
        got = tuple(int(v) for v in philox4x32_10(*ctr, *key))
        assert got == want


def test_c_philox_kat():
    lib = c_oracle.lib()
    for ctr, key, want in KAT:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        out = (C.c_uint32 * 4)()
        lib.orc_philox4x32_10(c, k, out)
        assert tuple(out) == want


def test_uniform_contract_layout():
    """u = ((a>>5)*2^26 + (b>>6)) / 2^53 with (a,b) = words (0,1) / (2,3) of the
    block ctr = (step>>1 lo, hi, track lo, hi), key = seed (rocRAND layout)."""
    seed, track = 0xa4093822299f31d0 >> 0, 0x0370734413198a2e
    # block index 0x85a308d3243f6a88 -> steps 2*blk, 2*blk+1
    blk = 0x85a308d3243f6a88 & ((1 << 63) - 1)
    w = philox4x32_10(blk & 0xffffffff, blk >> 32, track & 0xffffffff, track >> 32,
                      seed & 0xffffffff, seed >> 32)
    w = [int(v) for v in w]
    u0 = ((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) / 9007199254740992.0
    u1 = ((w[2] >> 5) * 67108864.0 + (w[3] >> 6)) / 9007199254740992.0
    assert uniform53(seed, track, 2 * blk) == u0
    assert uniform53(seed, track, 2 * blk + 1) == u1


def test_c_and_numpy_uniforms_agree():
    rng = np.random.default_rng(0)
    tracks = rng.integers(0, 2**62, 500, dtype=np.uint64)
    steps = rng.integers(0, 2**40, 500, dtype=np.uint64)
    want = uniform53(12345678901234567, tracks, steps)
    got = np.array([c_oracle.uniform(12345678901234567, int(t), int(s))
                    for t, s in zip(tracks, steps)])
    assert np.array_equal(got, want)
    assert want.min() >= 0 and want.max() < 1
    tu = TrackUniforms(7, 3, chunk=16)
    assert [tu(k) for k in (0, 1, 15, 16, 40)] == \
        [float(uniform53(7, 3, k)) for k in (0, 1, 15, 16, 40)]


def test_numpy_legacy_mapping():
    """The word->double map is numpy's legacy random_sample (rk_double): the
    same two MT19937 words read through RandomState.bytes and random_sample."""
    raw = np.random.RandomState(5).bytes(8)
    a = int.from_bytes(raw[0:4], 'little')
    b = int.from_bytes(raw[4:8], 'little')
    u = np.random.RandomState(5).random_sample()
    assert u == ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0
