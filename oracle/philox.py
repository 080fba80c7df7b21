"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the device RNG contract of include/gcnvae.h (gv_rng_fill).

Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11;
multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85), pinned against the paper's
known-answer vectors in tests/test_oracle_golden.py.  The reference draws its randomness from torch's CPU generator
(nn.Dropout, randn_like: kgvae/utils.py:342-361); that stream cannot be reproduced on a device, so the product defines
its own counter-based stream and this file is its checker: masks must match bit for bit, normals to float rounding.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(key, ctr):
    """key: (k0, k1) python ints; ctr: uint32 array (..., 4).  Returns uint32 array (..., 4)."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def _raw(seed, tick, stream, n):
    groups = (n + 3) // 4
    ctr = np.zeros((groups, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(groups, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = stream & 0xFFFFFFFF
    ctr[:, 2] = tick & 0xFFFFFFFF
    ctr[:, 3] = (tick >> 32) & 0xFFFFFFFF
    return philox4x32_10((seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF), ctr)


def keep_mask(seed, tick, stream, n, drop_p):
    """uint8 (n,): byte = (u32 >= floor(float32(drop_p) * 2^32))."""
    th = min(int(float(np.float32(drop_p)) * 4294967296.0), 4294967295)
    return (_raw(seed, tick, stream, n).reshape(-1)[:n] >= np.uint32(th)).astype(np.uint8)


def normals(seed, tick, stream, n):
    """float32 (n,): Box-Muller on (x0, x1) -> outputs 0, 1 and (x2, x3) -> outputs 2, 3 of each group."""
    x = _raw(seed, tick, stream, n).astype(np.float32)          # the kernel converts u32 -> f32 (round to nearest)
    s = np.float32(2.3283064365386963e-10)
    out = np.empty((x.shape[0], 4), dtype=np.float32)
    for a, b, o in ((0, 1, 0), (2, 3, 2)):
        u1 = (x[:, a] + np.float32(1.0)) * s
        r = np.sqrt(np.float32(-2.0) * np.log(u1))
        th = np.float32(6.283185307179586) * (x[:, b] * s)
        out[:, o], out[:, o + 1] = r * np.cos(th), r * np.sin(th)
    return out.reshape(-1)[:n]


# ---- the batch sampler's draws (include/gcnvae.h: gv_perm_sample, gv_negative_sampling) ---------------------------------
def _philox_x(seed, tick, stream, c0):
    ctr = np.zeros((len(c0), 4), dtype=np.uint32)
    ctr[:, 0] = np.asarray(c0, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = np.uint32(stream & 0xFFFFFFFF)
    ctr[:, 2] = tick & 0xFFFFFFFF
    ctr[:, 3] = (tick >> 32) & 0xFFFFFFFF
    return philox4x32_10((seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF), ctr)


def feistel_permute(x, bits, seed, tick, stream):
    """One application of the keyed permutation of [0, 2^bits): 4-round unbalanced Feistel network."""
    x = np.asarray(x, dtype=np.uint64)
    la, rb = bits // 2, bits - bits // 2
    l, r = x >> np.uint64(rb), x & np.uint64((1 << rb) - 1)
    for rnd in range(4):
        f = _philox_x(seed, tick, (stream + 0x10000 * (rnd + 1)) & 0xFFFFFFFF, r)[:, 0].astype(np.uint64) & np.uint64((1 << la) - 1)
        l, r = r, l ^ f
        la, rb = rb, la
    return (l << np.uint64(rb)) | r


def perm_sample(n, k, seed, tick, stream):
    """First k outputs of the keyed permutation of [0, n) (cycle walking), int64 (k,)."""
    bits = 2
    while (1 << bits) < n:
        bits += 1
    x = np.arange(k, dtype=np.uint64)
    todo = np.ones(k, dtype=bool)
    while todo.any():
        x[todo] = feistel_permute(x[todo], bits, seed, tick, stream)
        todo = x >= n
    return x.astype(np.int64)


def negative_draws(total, n_entities, seed, tick, stream):
    """(values int64 (total,), hit_subject bool (total,)): value = mulhi(u32, n_entities), coin = top bit of the second u32."""
    raw = _philox_x(seed, tick, stream, np.arange(total, dtype=np.uint64))
    values = (raw[:, 0].astype(np.uint64) * np.uint64(n_entities)) >> np.uint64(32)
    return values.astype(np.int64), (raw[:, 1] >> np.uint32(31)).astype(bool)


def neighborhood_draw(seed, tick, stream):
    """draw(i, attempt) -> uint32 of gv_neighborhood_sample (include/gcnvae.h): Philox counter
    (i, stream + 0x10000 * attempt, tick lo, tick hi), output word x."""
    def draw(i, attempt):
        return int(_philox_x(seed, tick, stream + 0x10000 * attempt, [i])[0, 0])
    return draw
