"""Oracle: Philox4x32-R (Salmon, Moraes, Dror, Shaw - "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
counter-based generator behind tf.random_normal / cuRAND / torch on GPUs, which use R = 10) and the Box-Muller map, in numpy.
The product's in-kernel stream uses R = 7 (Random123's philox4x32_7, the smallest round count its authors report as
Crush-resistant) and takes FOUR Box-Muller pairs from every 128-bit block - one per 32-bit word: the top 20 bits are the radius
uniform, the low 12 bits the angle (round 5: three pairs of 21 + 21 bits).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws its noise inside the step with TF's own Philox
stream (models/svae.py:113-114); TF 1.3's exact counter layout is a TensorFlow internal that is absent from the reference
tree, so the product's in-kernel generator defines its own layout (csrc/vmp_svae.hip: counter = (cell_lo, cell_hi,
block, 0), key = seed) and parity is distributional.  This file pins the generator itself: the known-answer vectors of
the Random123 distribution (kat_vectors: philox4x32, 7 and 10 rounds) and the element layout of a cell's noise block."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


ROUNDS = 7          # the product's stream (csrc/vmp_svae.hip VMP_PHILOX_ROUNDS)


def philox4x32(ctr, key, rounds=ROUNDS):
    """ctr (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        for r in range(rounds):
            if r:
                k0 = (k0 + W0).astype(np.uint32)
                k1 = (k1 + W1).astype(np.uint32)
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
    return np.stack(c, axis=-1)


def philox4x32_10(ctr, key):
    return philox4x32(ctr, key, 10)


def box_muller8(u):
    """(..., 4) uint32 -> (..., 4, 2) float64: one Box-Muller pair r (cos, sin) per 32-bit word;
    radius uniform (a + 1/2) 2^-20 from the word's top 20 bits, angle b 2^-12 revolutions from its low 12 bits."""
    u = np.asarray(u, dtype=np.uint32)
    a = (u >> np.uint32(12)).astype(np.float64)
    b = (u & np.uint32(0xFFF)).astype(np.float64)
    r = np.sqrt(-2.0 * np.log((a + 0.5) * 2.0 ** -20))
    th = 2 * np.pi * b * 2.0 ** -12
    return np.stack([r * np.cos(th), r * np.sin(th)], axis=-1)


def box_muller4(u):
    """(..., 4) uint32 -> (..., 4) float64 standard normals: pairs (u0,u1) and (u2,u3);
    radius from the top 24 bits of the first word, angle (in revolutions) from the top 24 bits of the second."""
    u = np.asarray(u, dtype=np.uint32)
    out = []
    for a, b in ((0, 1), (2, 3)):
        u1 = ((u[..., a] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
        ang = (u[..., b] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
        r = np.sqrt(-2.0 * np.log(u1))
        out += [r * np.cos(2 * np.pi * ang), r * np.sin(2 * np.pi * ang)]
    return np.stack(out, axis=-1)


def cell_noise(seed, cells, L, S):
    """The (len(cells), L, S) noise blocks of the given cell ids (n*K + k) under `seed`.
    Block b = (s >> 1) * ceil(L/4) + j of a cell holds the pairs (eps[i, s], eps[i, s+1]) of coordinates i = 4j .. 4j+3."""
    cells = np.asarray(cells, dtype=np.uint64)
    SP, L4 = (S + 1) // 2, (L + 3) // 4
    blk = np.arange(SP * L4, dtype=np.uint32)
    ctr = np.zeros((cells.size, blk.size, 4), dtype=np.uint32)
    ctr[..., 0] = (cells & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    ctr[..., 1] = (cells >> np.uint64(32)).astype(np.uint32)[:, None]
    ctr[..., 2] = blk[None, :]
    key = np.zeros((cells.size, blk.size, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    z = box_muller8(philox4x32(ctr, key))                    # (cells, blocks, 4, 2)
    out = np.zeros((cells.size, L, S))
    for sp in range(SP):
        for j in range(L4):
            for t in range(4):
                i = 4 * j + t
                if i < L:
                    out[:, i, 2 * sp] = z[:, sp * L4 + j, t, 0]
                    if 2 * sp + 1 < S:
                        out[:, i, 2 * sp + 1] = z[:, sp * L4 + j, t, 1]
    return out


SUBSAMPLE_TAG = 0x5bb5a3c1


def subsample_uniforms(seed, N, S_out):
    """(N, S_out) float32 uniforms in [0,1) of the categorical draw (csrc/vmp_svae.hip subsample_kernel, rng mode): top 24
    bits of word 0 of Philox4x32-7(key = seed, counter = (n low, n high, s, SUBSAMPLE_TAG)) * 2^-24."""
    n = np.arange(N, dtype=np.uint64)
    ctr = np.zeros((N, S_out, 4), dtype=np.uint32)
    ctr[..., 0] = (n & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    ctr[..., 1] = (n >> np.uint64(32)).astype(np.uint32)[:, None]
    ctr[..., 2] = np.arange(S_out, dtype=np.uint32)[None, :]
    ctr[..., 3] = np.uint32(SUBSAMPLE_TAG)
    key = np.zeros((N, S_out, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    w = philox4x32(ctr, key)[..., 0]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
