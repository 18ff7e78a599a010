"""numpy restatement of the library's dropout mask (csrc/common.hpp): Threefry4x32-12 keyed by (seed, step, site), eight 16-bit
lots per call.  Test infrastructure only: tests/test_dropout_cpu.py pins the generator to the published Threefry4x32-20
known-answer vectors, tests/test_kernels_gpu.py compares slnlp_dropout_mask with keep_mask() bit for bit."""
import numpy as np

ROT = [(10, 26), (11, 21), (13, 27), (23, 5), (6, 20), (17, 11), (25, 10), (18, 20)]
PHI = 0x9E3779B97F4A7C15


def _rotl(x, n):
    return ((x << np.uint32(n)) | (x >> np.uint32(32 - n))).astype(np.uint32)


def threefry4x32(ctr, key, rounds):
    """Threefry-4x32 with `rounds` rounds (Salmon et al., SC'11; Random123's threefry4x32_R): ctr, key = 4 uint32 arrays each."""
    X = [np.asarray(c, dtype=np.uint32).copy() for c in ctr]
    k = [np.asarray(v, dtype=np.uint32) for v in key]
    ks = k + [np.uint32(0x1BD11BDA) ^ k[0] ^ k[1] ^ k[2] ^ k[3]]
    with np.errstate(over="ignore"):
        for i in range(4):
            X[i] = (X[i] + ks[i]).astype(np.uint32)
        for r in range(rounds):
            a, b = ROT[r % 8]
            if r % 2 == 0:
                X[0] = (X[0] + X[1]).astype(np.uint32); X[1] = _rotl(X[1], a) ^ X[0]
                X[2] = (X[2] + X[3]).astype(np.uint32); X[3] = _rotl(X[3], b) ^ X[2]
            else:
                X[0] = (X[0] + X[3]).astype(np.uint32); X[3] = _rotl(X[3], a) ^ X[0]
                X[2] = (X[2] + X[1]).astype(np.uint32); X[1] = _rotl(X[1], b) ^ X[2]
            if r % 4 == 3:
                s = (r + 1) // 4
                for i in range(4):
                    X[i] = (X[i] + ks[(s + i) % 5]).astype(np.uint32)
                X[3] = (X[3] + np.uint32(s)).astype(np.uint32)
    return X


def threshold(p):
    return min(int(float(np.float32(p)) * 65536.0 + 0.5), 65535)


def keep_mask(R, C, p, site, seed, step):
    """[R, C] float32 mask (1 = kept) of a dropout site, as slnlp_dropout_mask returns it."""
    k = (seed + PHI * step) % (1 << 64)
    r, c = np.meshgrid(np.arange(R, dtype=np.uint32), np.arange(C, dtype=np.uint32), indexing="ij")
    cc = ((c >> np.uint32(5)) << np.uint32(4)) | (c & np.uint32(15))
    zero = np.zeros_like(r)
    key = [zero + np.uint32(k & 0xFFFFFFFF), zero + np.uint32(k >> 32), zero + np.uint32(site), zero]
    X = threefry4x32([cc, r >> np.uint32(2), zero, zero], key, 12)
    f = ((c >> np.uint32(4)) & np.uint32(1)) * np.uint32(4) + (r & np.uint32(3))          # lot index 0 .. 7
    word = np.choose(f >> np.uint32(1), X)
    lot = (word >> (np.uint32(16) * (f & np.uint32(1)))) & np.uint32(0xFFFF)
    return (lot >= np.uint32(threshold(p))).astype(np.float32)
