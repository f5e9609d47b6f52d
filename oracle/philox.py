"""ORACLE (test infrastructure only -- never imported by the product path).

Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw, "Parallel
random numbers: as easy as 1, 2, 3", SC'11; Random123 1.x) restated in numpy,
plus the SSRS-MI355X uniform contract built on it.

Why it exists: the reference draws one legacy-MT19937 double per step from the
*global* numpy state (`np.random.choice`, /root/reference/ssrs/movmodel.py:312),
which a parallel implementation cannot reproduce.  The
build therefore defines

    u(seed, track_id, step) in [0, 1)

from Philox4x32-10 with
    key     = (seed & 0xffffffff, seed >> 32)
    counter = (blk & 0xffffffff, blk >> 32, track & 0xffffffff, track >> 32),
              blk = step >> 1
    (a, b)  = words (0, 1) of the block for even steps, words (2, 3) for odd
    u       = ((a >> 5) * 2**26 + (b >> 6)) / 2**53

The word->double mapping is numpy's legacy `random_sample` mapping
(`rk_double`, used by `np.random.choice`); the counter layout is exactly
rocRAND's `rocrand_init(seed, subsequence=track, offset=2*step)` followed by two
`rocrand()` calls (rocrand_philox4x32_10.h), so the device kernel can use the
rocRAND engine itself.

Pinned by the Random123 known-answer vectors (tests/test_philox.py).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. All inputs broadcastable uint32-valued arrays.

    Returns the four output words as uint64 arrays holding 32-bit values.
    """
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0          # 64-bit products of 32-bit values: no overflow
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform53(seed, track_id, step):
    """u(seed, track_id, step) of the contract above; vectorised over
    track_id/step. Returns float64 in [0, 1)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    track = np.asarray(track_id, dtype=np.uint64)
    step = np.asarray(step, dtype=np.uint64)
    blk = step >> np.uint64(1)
    w = philox4x32_10(blk & MASK32, blk >> np.uint64(32),
                      track & MASK32, track >> np.uint64(32),
                      seed & 0xFFFFFFFF, seed >> 32)
    odd = (step & np.uint64(1)).astype(bool)
    a = np.where(odd, w[2], w[0])
    b = np.where(odd, w[3], w[1])
    hi = (a >> np.uint64(5)).astype(np.float64)
    lo = (b >> np.uint64(6)).astype(np.float64)
    return (hi * 67108864.0 + lo) / 9007199254740992.0


class TrackUniforms:
    """Callable step -> u for one track, generating blocks of steps lazily so
    the pure-Python stepper oracle does not pay numpy overhead per step."""

    def __init__(self, seed, track_id, chunk=4096):
        self.seed = seed
        self.track_id = track_id
        self.chunk = chunk
        self._base = -1
        self._buf = None

    def __call__(self, step):
        base = (step // self.chunk) * self.chunk
        if base != self._base:
            steps = np.arange(base, base + self.chunk, dtype=np.uint64)
            self._buf = uniform53(self.seed, self.track_id, steps)
            self._base = base
        return float(self._buf[step - base])
