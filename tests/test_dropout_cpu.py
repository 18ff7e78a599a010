"""The dropout generator's numpy restatement (tests/threefry_ref.py) against the published Threefry4x32-20 known-answer vectors
(Random123 kat_vectors: zero / all-ones / pi-digits counter and key), and the statistics of the 12-round, 16-bit-lot masks the
library draws (the GPU test compares slnlp_dropout_mask with this restatement bit for bit)."""
import numpy as np

import threefry_ref as tf


def _hex(X):
    return [int(x) for x in X]


def test_threefry4x32_20_known_answers():
    assert _hex(tf.threefry4x32([0, 0, 0, 0], [0, 0, 0, 0], 20)) == [0x9c6ca96a, 0xe17eae66, 0xfc10ecd4, 0x5256a7d8]
    f = 0xffffffff
    assert _hex(tf.threefry4x32([f] * 4, [f] * 4, 20)) == [0x2a881696, 0x57012287, 0xf6c7446e, 0xa16a6732]
    assert _hex(tf.threefry4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                [0xa4093822, 0x299f31d0, 0x082efa98, 0xec4e6c89], 20)) == [0x59cd1dbb, 0xb8879579, 0x86b5d00c, 0xac8b6d84]


def test_mask_statistics_and_independence():
    p = 0.1
    m = tf.keep_mask(512, 256, p, 3, seed=7, step=2)
    n = m.size
    assert abs(1.0 - m.mean() - p) < 4 * np.sqrt(p * (1 - p) / n)                      # keep rate
    assert abs(tf.threshold(p) / 65536.0 - p) <= 2.0 ** -17                             # quantisation of p
    # rows, columns, the two columns of a call and the four rows of a call are uncorrelated
    z = m - m.mean()
    for a, b in ((z[:, :-1], z[:, 1:]), (z[:-1], z[1:]), (z[:, :-16], z[:, 16:]), (z[:-4], z[4:])):
        assert abs((a * b).mean()) < 4 * p * (1 - p) / np.sqrt(a.size)
    # another site, step or seed is another mask
    for other in (tf.keep_mask(512, 256, p, 4, 7, 2), tf.keep_mask(512, 256, p, 3, 7, 3), tf.keep_mask(512, 256, p, 3, 8, 2)):
        assert 0.1 < (other != m).mean() < 0.26                                         # ~ 2 p (1 - p) = 0.18
    assert tf.keep_mask(8, 8, 0.0, 1, 1, 1).all() and tf.threshold(0.999999) == 65535
