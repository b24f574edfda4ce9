"""Oracle: Philox4x32-R (Salmon, Moraes, Dror, Shaw - "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
counter-based generator behind tf.random_normal / cuRAND / torch on GPUs, which use R = 10) and the Box-Muller map, in numpy.
The product's in-kernel stream uses R = 7 (Random123's philox4x32_7, the smallest round count its authors report as
Crush-resistant) and takes three Box-Muller pairs of 21-bit uniforms from every 128-bit block.

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


def box_muller6(u):
    """(..., 4) uint32 -> (..., 3, 2) float64: three Box-Muller pairs r (cos, sin) from 21-bit uniforms.
    pair 0 = (c0[0..20], c1[11..31]); pair 1 = (c2[0..20], c3[11..31]); pair 2 = (c0[21..31] | c1[0..9] << 11, c2[21..31] | c3[0..9] << 11)."""
    u = np.asarray(u, dtype=np.uint32)
    m21, s11, s21 = np.uint32(0x1FFFFF), np.uint32(11), np.uint32(21)
    c0, c1, c2, c3 = (u[..., i] for i in range(4))
    rad = [c0 & m21, c2 & m21, ((c0 >> s21) | (c1 << s11)) & m21]
    ang = [c1 >> s11, c3 >> s11, ((c2 >> s21) | (c3 << s11)) & m21]
    out = []
    for a, b in zip(rad, ang):
        u1 = (a.astype(np.float64) + 0.5) * 2.0 ** -21
        th = b.astype(np.float64) * 2.0 ** -21
        r = np.sqrt(-2.0 * np.log(u1))
        out.append(np.stack([r * np.cos(2 * np.pi * th), r * np.sin(2 * np.pi * th)], axis=-1))
    return np.stack(out, axis=-2)


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
    Block b = (s >> 1) * ceil(L/3) + j of a cell holds the pairs (eps[i, s], eps[i, s+1]) of coordinates i = 3j, 3j+1, 3j+2."""
    cells = np.asarray(cells, dtype=np.uint64)
    SP, L3 = (S + 1) // 2, (L + 2) // 3
    blk = np.arange(SP * L3, dtype=np.uint32)
    ctr = np.zeros((cells.size, blk.size, 4), dtype=np.uint32)
    ctr[..., 0] = (cells & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    ctr[..., 1] = (cells >> np.uint64(32)).astype(np.uint32)[:, None]
    ctr[..., 2] = blk[None, :]
    key = np.zeros((cells.size, blk.size, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    z = box_muller6(philox4x32(ctr, key))                    # (cells, blocks, 3, 2)
    out = np.zeros((cells.size, L, S))
    for sp in range(SP):
        for j in range(L3):
            for t in range(3):
                i = 3 * j + t
                if i < L:
                    out[:, i, 2 * sp] = z[:, sp * L3 + j, t, 0]
                    if 2 * sp + 1 < S:
                        out[:, i, 2 * sp + 1] = z[:, sp * L3 + j, t, 1]
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
