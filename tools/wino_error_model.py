#!/usr/bin/env python3
"""Numerical error model of Winograd F(m, 3) on the host: which interpolation points for F(4x4,3x3)?

fp32 transforms (weights U = G g G^T, inputs V = B^T d B), fp32 elementwise products accumulated over the input
channels in order (what an MFMA k-chain does), fp32 output transform — against the fp64 direct convolution, next to
the direct fp32 sum on the same data (inputs >= 0 like SPADE's actv = ReLU(mlp_shared(seg)), weights ~ N(0, 1/fan_in)).

    python tools/wino_error_model.py [Cin]        (Cin = 128: seconds; 1024: a few minutes)

Result (max |y - fp64| / max |fp64|, rms in brackets) at Cin = 128 / 1024:
    F(2,3)  {0, 1, -1}            4.0e-7 / 1.3e-6     direct fp32   0.9e-6 / 2.3e-6
    F(4,3)  {0, +-1, +-2}         6.3e-6 / 1.4e-5
    F(4,3)  {0, +-1, +-1/2}       8.6e-6 / 1.6e-5
    F(4,3)  {0, 1, -1, 1/2, -2}   2.2e-6 / 5.5e-6     <- csrc/wino4.hip
    F(4,3)  {0, -1, 1, 1/2, -3}   3.6e-6 / 1.2e-5
The matrices are built by the Toom-Cook construction (Barabasz et al., "Error analysis and improving the accuracy of
Winograd convolution for deep neural networks") with exact rational arithmetic and checked against a direct 1-D
correlation before use."""
import numpy as np, itertools, sys
from fractions import Fraction as Fr

def toom_cook(points, m, r):
    """Winograd/Toom-Cook matrices AT (m x n), G (n x r), BT (n x n) for F(m, r) with n = m + r - 1 points, last = infinity."""
    n = m + r - 1
    pts = points[:n-1]
    # Lagrange-based construction (Barabasz et al.)
    # f_i = prod_{j != i} (p_i - p_j)
    def poly_mul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    M = [Fr(1)]
    for p in pts:
        M = poly_mul(M, [-p, Fr(1)])        # M(x) = prod (x - p_i), degree n-1
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * r for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    for i, p in enumerate(pts):
        f = Fr(1)
        for j, q in enumerate(pts):
            if j != i:
                f *= (p - q)
        for k in range(m):
            AT[k][i] = p ** k
        for k in range(r):
            G[i][k] = p ** k / f
        # BT row i: coefficients of M(x)/(x - p_i)
        Mi = [Fr(1)]
        for j, q in enumerate(pts):
            if j != i:
                Mi = poly_mul(Mi, [-q, Fr(1)])
        for k in range(n - 1):
            BT[i][k] = Mi[k]
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][r - 1] = Fr(1)
    for k in range(n):
        BT[n - 1][k] = M[k]
    f = lambda A: np.array([[float(x) for x in row] for row in A], dtype=np.float64)
    return f(AT), f(G), f(BT)

def check(AT, G, BT, m, r):
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

def conv_err(AT, G, BT, m, r, Cin=128, ntile=256, Cout=16, seed=0, relu_in=True):
    """fp32 Winograd 2D (transform in fp32, elementwise products summed over Cin in fp32 sequentially like an MFMA chain) vs fp64 direct."""
    rng = np.random.default_rng(seed)
    n = m + r - 1
    d = rng.standard_normal((ntile, Cin, n, n))
    if relu_in: d = np.maximum(d, 0)          # SPADE: actv = ReLU(mlp_shared(seg))
    g = rng.standard_normal((Cout, Cin, r, r)) / np.sqrt(Cin * r * r)
    # fp64 direct
    ref = np.zeros((ntile, Cout, m, m))
    for a in range(m):
        for b in range(m):
            patch = d[:, :, a:a + r, b:b + r]
            ref[:, :, a, b] = np.einsum('tcij,ocij->to', patch, g)
    # fp32 direct (sequential fp32 accumulate over taps+channels)
    d32, g32 = d.astype(np.float32), g.astype(np.float32)
    dir32 = np.zeros((ntile, Cout, m, m), np.float32)
    for a in range(m):
        for b in range(m):
            acc = np.zeros((ntile, Cout), np.float32)
            for i in range(r):
                for j in range(r):
                    for c0 in range(0, Cin, 1):
                        acc += d32[:, None, c0, a + i, b + j] * g32[None, :, c0, i, j]
            dir32[:, :, a, b] = acc
    AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
    U = np.einsum('ai,ocij,bj->ocab', G32, g32, G32).astype(np.float32)        # weights transformed in fp32
    V = np.einsum('ai,tcij,bj->tcab', BT32, d32, BT32).astype(np.float32)
    Mm = np.zeros((ntile, Cout, n, n), np.float32)
    for c0 in range(Cin):
        Mm += V[:, None, c0] * U[None, :, c0]
    Y = np.einsum('ai,toij,bj->toab', AT32, Mm, AT32).astype(np.float32)
    scale = np.abs(ref).max()
    return np.abs(Y - ref).max() / scale, np.sqrt(((Y - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()), \
           np.abs(dir32 - ref).max() / scale, np.sqrt(((dir32 - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())

F = Fr
sets = {
  "F(2,3) 0,1,-1": (2, [F(0), F(1), F(-1)]),
  "F(4,3) 0,1,-1,2,-2": (4, [F(0), F(1), F(-1), F(2), F(-2)]),
  "F(4,3) 0,1,-1,1/2,-1/2": (4, [F(0), F(1), F(-1), F(1,2), F(-1,2)]),
  "F(4,3) 0,1,-1,1/2,-2": (4, [F(0), F(1), F(-1), F(1,2), F(-2)]),
  "F(4,3) 0,-1,1,1/2,-3": (4, [F(0), F(-1), F(1), F(1,2), F(-3)]),
  "F(3,3) 0,1,-1,2": (3, [F(0), F(1), F(-1), F(2)]),
  "F(3,3) 0,1,-1,1/2": (3, [F(0), F(1), F(-1), F(1,2)]),
  "F(3,3) 0,1,-1,-1/2": (3, [F(0), F(1), F(-1), F(-1,2)]),
}
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, (m, pts) in sets.items():
    AT, G, BT = toom_cook(pts, m, 3)
    e = check(AT, G, BT, m, 3)
    w = conv_err(AT, G, BT, m, 3, Cin=Cin, ntile=64, Cout=8)
    print("%-28s exact-check %.1e | wino max/scale %.2e rms %.2e | direct32 max/scale %.2e rms %.2e" % ((name, e) + w))
