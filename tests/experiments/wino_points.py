"""Empirical fp32 error of 2-D Winograd F(m x m, 3 x 3) for symmetric point sets {0, +-a, +-b, inf}."""
import numpy as np, itertools, sys
from fractions import Fraction

def matrices(points, m=4, r=3):
    """A^T (m,n), G (n,r), B^T (n,n) in float64 for finite points + infinity (Cook-Toom)."""
    n = m + r - 1
    pts = [float(p) for p in points]
    assert len(pts) == n - 1
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for j, a in enumerate(pts):
        N = np.prod([a - b for l, b in enumerate(pts) if l != j])
        for i in range(m): AT[i, j] = a ** i
        for k in range(r): G[j, k] = a ** k / N
    AT[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    # solve for B^T from the bilinear identity
    rows, rhs = [], []
    for i in range(m):
        for k in range(r):
            for l in range(n):
                row = np.zeros((n, n))
                row[:, l] = AT[i, :] * G[:, k]
                rows.append(row.ravel()); rhs.append(1.0 if l == i + k else 0.0)
    BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(n, n)
    res = np.abs(np.array(rows) @ BT.ravel() - np.array(rhs)).max()
    assert res < 1e-9, res
    return AT, G, BT

def wino_layer(x, w, AT, G, BT, m, scale_pow2=True):
    """x (H,W,Cin) f32, w (3,3,Cin,Cout) f32 -> (H,W,Cout) f32; H, W multiples of m.  U in f64 rounded
    once, V = B^T d B in f32, per-point f32 GEMM, Y = A^T M A in f32."""
    H, W, Cin = x.shape; Cout = w.shape[3]; n = m + 2
    U = np.einsum('ik,klcd,jl->ijcd', G, w.astype(np.float64), G).astype(np.float32)
    xp = np.zeros((H + 2, W + 2, Cin), np.float32); xp[1:-1, 1:-1] = x
    th, tw = H // m, W // m
    # tiles (th, tw, n, n, Cin)
    idx_y = (np.arange(th) * m)[:, None] + np.arange(n)[None]
    idx_x = (np.arange(tw) * m)[:, None] + np.arange(n)[None]
    d = xp[idx_y[:, None, :, None], idx_x[None, :, None, :]]     # th,tw,n,n,Cin
    BT32 = BT.astype(np.float32)
    V = np.einsum('ik,abklc->abilc', BT32, d).astype(np.float32)
    V = np.einsum('abilc,jl->abijc', V, BT32).astype(np.float32)
    M = np.empty((th, tw, n, n, Cout), np.float32)
    for i in range(n):
        for j in range(n):
            M[:, :, i, j] = (V[:, :, i, j].reshape(-1, Cin) @ U[i, j]).reshape(th, tw, Cout)
    AT32 = AT.astype(np.float32)
    Y = np.einsum('ik,abklc->abilc', AT32, M).astype(np.float32)
    Y = np.einsum('abilc,jl->abijc', Y, AT32).astype(np.float32)
    return Y.transpose(0, 2, 1, 3, 4).reshape(H, W, Cout)

def direct64(x, w):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin)); xp[1:-1, 1:-1] = x
    out = np.zeros((H * W, w.shape[3]))
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + H, kx:kx + W].reshape(-1, Cin) @ w[ky, kx].astype(np.float64)
    return out.reshape(H, W, -1)

def direct32(x, w):
    H, W, Cin = x.shape
    xp = np.zeros((H + 2, W + 2, Cin), np.float32); xp[1:-1, 1:-1] = x
    out = np.zeros((H * W, w.shape[3]), np.float32)
    for ky in range(3):
        for kx in range(3):
            out += np.ascontiguousarray(xp[ky:ky + H, kx:kx + W]).reshape(-1, Cin) @ w[ky, kx]
    return out.reshape(H, W, -1)

if __name__ == '__main__':
    rng = np.random.default_rng(0)
    Cin, Cout, H, W = 128, 128, 48, 48
    x = np.maximum(rng.normal(size=(H, W, Cin)), 0).astype(np.float32)
    w = rng.normal(0, np.sqrt(2 / (9 * Cin)), size=(3, 3, Cin, Cout)).astype(np.float32)
    ref = direct64(x, w); sc = np.abs(ref).max()
    def rep(name, y):
        e = np.abs(y - ref)
        print('%-36s max %.3e  rms %.3e' % (name, e.max() / sc, np.sqrt((e ** 2).mean()) / sc)); sys.stdout.flush()
    rep('direct f32', direct32(x, w))
    AT, G, BT = matrices([0, 1, -1], m=2)
    rep('F(2,3) {0,1,-1}', wino_layer(x, w, AT, G, BT, 2))
    cands = [(1, 2), (1, .5), (.5, 2), (.5, 1.5), (0.75, 1.5), (2/3., 1.5), (.6, 1.2), (.7, 1.4), (0.5, 1), (.5,1.25), (.6,1.5), (.8, 1.6), (1, 1.5), (.7, 1.2), (.6, 1.1), (.5, .9), (.4,.9), (.45, 1.0), (.5,1.1), (.55,1.2)]
    for a, b in cands:
        AT, G, BT = matrices([0, a, -a, b, -b], m=4)
        rep('F(4,3) {0,+-%.3g,+-%.3g}' % (a, b), wino_layer(x, w, AT, G, BT, 4))
