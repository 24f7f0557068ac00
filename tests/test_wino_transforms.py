"""The Winograd transforms the fp32 conv kernels hard-code (dodt_amd/csrc/wino_kernels.h: F(2x2,3x3), points
0, +-1, inf; wino43_kernel.h: F(4x4,3x3), points 0, +-2/3, +-3/2, inf; filter transforms in conv.hip),
restated in numpy float64 exactly as the kernels compute them and checked against the direct correlation.
CPU only: pins the algebra, the kernels are held to the oracle on the GPU (tests/test_gpu_conv.py)."""
import numpy as np

A, B = 2.0 / 3.0, 1.5
A2, B2, S, A3, B3 = A * A, B * B, A * A + B * B, A ** 3, B ** 3


def bt43(x):
    """wino43_bt: six values -> six (B^T x)."""
    x0, x1, x2, x3, x4, x5 = x
    e1, e2 = x4 - B2 * x2, x4 - A2 * x2
    o1, o2 = A * x3 - B * x1, B * x3 - A * x1
    return np.array([x0 + (x4 - S * x2), e1 + o1, e1 - o1, e2 + o2, e2 - o2, x1 + (x5 - S * x3)])


def at43(m):
    """wino43_at: six values -> four (A^T m)."""
    m0, m1, m2, m3, m4, m5 = m
    s1, d1, s2, d2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    return np.array([(m0 + s1) + s2, A * d1 + B * d2, A2 * s1 + B2 * s2, A3 * d1 + B3 * d2 + m5])


def g43():
    """conv.hip: G[j][k] = p_j^k / prod_{l != j} (p_j - p_l), last row (0, 0, 1)."""
    p = [0.0, A, -A, B, -B]
    g = np.zeros((6, 3))
    for j in range(5):
        n = np.prod([p[j] - p[l] for l in range(5) if l != j])
        g[j] = [1.0 / n, p[j] / n, p[j] ** 2 / n]
    g[5] = [0, 0, 1]
    return g


def bt23(x):
    x0, x1, x2, x3 = x
    return np.array([x0 - x2, x1 + x2, x2 - x1, x1 - x3])


def at23(m):
    return np.array([(m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]])


G23 = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1.0]])


def _check(bt, at, G, m):
    rng = np.random.default_rng(m)
    n = m + 2
    # the matrices the functions stand for
    BT = np.stack([bt(e) for e in np.eye(n)], axis=1)
    AT = np.stack([at(e) for e in np.eye(n)], axis=1)
    for _ in range(20):
        d, g = rng.normal(size=(n, n)), rng.normal(size=(3, 3))
        want = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
        got = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    return AT, BT


def test_f43_transforms_compute_the_correlation():
    AT, BT = _check(bt43, at43, g43(), 4)
    # the form written in wino43_kernel.h's header comment
    assert np.allclose(BT[0], [1, 0, -S, 0, 1, 0]) and np.allclose(BT[1], [0, -B, -B2, A, 1, 0])
    assert np.allclose(BT[3], [0, -A, -A2, B, 1, 0]) and np.allclose(BT[5], [0, 1, 0, -S, 0, 1])
    assert np.allclose(AT[3], [0, A3, -A3, B3, -B3, 1])
    # the amplification the points were chosen for: sum |A^T| |G| |B^T| per output, against (1, 2)
    amp = (np.abs(AT) @ (np.abs(g43()).sum(1)[:, None] * np.abs(BT))).max()
    assert amp < 40        # Lavin & Gray's points: 130


def test_f23_transforms_compute_the_correlation():
    _check(bt23, at23, G23, 2)
