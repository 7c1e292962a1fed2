"""CPU restatement of the dropout mask rule of include/bmnas_hip.h (bmnas_dropout_t).  TEST INFRASTRUCTURE ONLY.

The reference draws its masks from torch's generator (nn.Dropout at node_operations.py:38, :55, :105,
node_search.py:64, aux_models.py:74, :114); the HIP path draws them from a counter-based Philox4x32-10 stream so
that the backward can regenerate them.  This module restates that stream in numpy, from the header's
definition, so the masks `bmnas_dropout_mask` exports can themselves be checked against an independent
implementation: element e is kept iff philox(seed, base + offset + e // 4)[e % 4] >= thr.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57          # Philox4x32 multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85          # key schedule (Weyl) increments
C2, C3 = 0x2545F491, 0x9E3779B1          # fixed upper half of the 128-bit counter (csrc/common.hpp)


def philox4x32_10(ctr, seed, c2=C2, c3=C3):
    """ctr: uint64 array (counter words 0 and 1); seed: python int (key words 0 and 1); c2, c3: counter words
    2 and 3 -> (n, 4) uint32 array of the four output words.  Standard Philox4x32-10 (Salmon et al., SC'11):
    with c2 = c3 = 0 it reproduces the Random123 known-answer vectors (tests/test_oracle_golden.py)."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1 = ctr & mask, ctr >> np.uint64(32)
    c2 = np.full_like(c0, c2)
    c3 = np.full_like(c0, c3)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = np.uint64(M0) * c0                       # 32 x 32 -> 64 bit products
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def dropout_multipliers(p, seed, offset, numel, step=0):
    """float32 array of `numel` multipliers: 0 where dropped, 1/(1-p) where kept (thr = int(p * 2^32))."""
    thr = min(int(p * 4294967296.0), 0xFFFFFFFF)
    if thr <= 0:
        return np.ones(numel, np.float32)
    n4 = (numel + 3) // 4
    ctr = (np.arange(n4, dtype=np.uint64) + np.uint64((offset + step) & 0xFFFFFFFFFFFFFFFF))
    words = philox4x32_10(ctr, seed & 0xFFFFFFFFFFFFFFFF).reshape(-1)[:numel]
    return np.where(words >= np.uint32(thr), np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)
