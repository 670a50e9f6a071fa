"""CPU study behind the choice of Winograd F(4x4, 3x3) interpolation points: builds the
Cook-Toom matrices for a point set (sympy), checks them in 1-D, and measures the f32 error of F(2x2) / F(4x4)
on a 768 -> 768 channel 3x3 layer against float64 direct convolution."""
import numpy as np, torch, itertools
torch.manual_seed(0)
def cook_toom(points, m, r, dtype=np.float64):
    """Winograd F(m, r) matrices via Cook-Toom with the given finite points + infinity: returns AT (m x a), G (a x r), BT (a x a), a = m+r-1."""
    from fractions import Fraction as Fr
    a = m + r - 1
    pts = [Fr(p) for p in points]  # a-1 finite points
    assert len(pts) == a - 1
    # Using the standard construction (Lavin): y = AT [(G g) * (BT d)]
    # Build via polynomial interpolation: M(x) = prod (x - p_i)
    import sympy as sp
    x = sp.symbols('x')
    ps = [sp.Rational(p.numerator, p.denominator) for p in pts]
    # Vandermonde-like matrices
    AT = sp.zeros(m, a); G = sp.zeros(a, r); BT = sp.zeros(a, a)
    # G: rows i<a-1: [p_i^j / N_i], N_i = prod_{k!=i}(p_i - p_k); last row: [0..0,1]
    for i, pi in enumerate(ps):
        Ni = sp.Integer(1)
        for k, pk in enumerate(ps):
            if k != i: Ni *= (pi - pk)
        for j in range(r): G[i, j] = pi**j / Ni
    G[a-1, r-1] = 1
    # AT: columns i<a-1: [p_i^j], last column: [0..0,1]
    for i, pi in enumerate(ps):
        for j in range(m): AT[j, i] = pi**j
    AT[m-1, a-1] = 1
    # BT: rows i<a-1: coefficients of M(x)/(x-p_i) ; last row: coefficients of M(x)
    M = sp.Integer(1)
    for pk in ps: M *= (x - pk)
    for i, pi in enumerate(ps):
        q = sp.Poly(sp.cancel(M / (x - pi)), x).all_coeffs()[::-1]
        for j in range(len(q)): BT[i, j] = q[j]
    q = sp.Poly(sp.expand(M), x).all_coeffs()[::-1]
    for j in range(len(q)): BT[a-1, j] = q[j]
    f = lambda Mx: np.array(Mx.tolist(), dtype=np.float64)
    return f(AT), f(G), f(BT)
try:
    import sympy
except Exception as e:
    print("no sympy", e); raise SystemExit
def check(points, m=4, r=3):
    AT, G, BT = cook_toom(points, m, r)
    # verify 1-D correctness in f64
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return AT, G, BT, np.abs(y - ref).max()
for pts in ([0, 1, -1, 2, -2], [0, 1, -1, '1/2', -2], [0, 1, -1, '1/2', '-1/2'], [0,'1/2','-1/2',2,-2]):
    AT, G, BT, e = check(pts)
    print(pts, "1-D check err", e)

import torch.nn.functional as F
def wino_conv(x, w, AT, G, BT, m, dtype=torch.float32, filt64=True):
    """x: (C,H,W) ; w: (N,C,3,3); valid conv via F(m x m, 3x3); transforms+GEMM in `dtype`."""
    a = m + 2
    C, H, W = x.shape; N = w.shape[0]
    OH, OW = H - 2, W - 2
    th, tw = -(-OH // m), -(-OW // m)
    xp = torch.zeros(C, th * m + 2, tw * m + 2, dtype=x.dtype); xp[:, :H, :W] = x
    tiles = xp.unfold(1, a, m).unfold(2, a, m)            # C, th, tw, a, a
    ATt, Gt, BTt = (torch.tensor(M_, dtype=torch.float64) for M_ in (AT, G, BT))
    U = torch.einsum('ik,nckl,jl->ijnc', Gt, w.double(), Gt)      # a,a,N,C  (f64 filter transform)
    U = U.to(dtype) if filt64 else torch.einsum('ik,nckl,jl->ijnc', Gt.to(dtype), w.to(dtype), Gt.to(dtype))
    t = tiles.to(dtype)
    B_ = BTt.to(dtype)
    V = torch.einsum('ik,cxykl,jl->ijcxy', B_, t, B_)             # a,a,C,th,tw
    Mm = torch.einsum('ijnc,ijcxy->ijnxy', U, V)                  # GEMM over C in dtype
    A_ = ATt.to(dtype)
    Y = torch.einsum('ki,ijnxy,lj->nxkyl', A_, Mm, A_)            # N,th,m,tw,m
    Y = Y.reshape(N, th * m, tw * m)[:, :OH, :OW]
    return Y
torch.manual_seed(1)
C = N = 768; H = W = 34
x = torch.relu(torch.randn(C, H, W)) * 1.0
w = torch.randn(N, C, 3, 3) * (2.0 / (C * 9)) ** 0.5
ref = F.conv2d(x.double()[None], w.double())[0]
d32 = F.conv2d(x[None], w)[0]
scale = ref.abs().max().item()
def rep(name, y):
    e = (y.double() - ref)
    print(f"{name:42s} max abs err / max|ref| = {e.abs().max().item() / scale:.2e}   rel L2 = {e.norm().item() / ref.norm().item():.2e}")
rep("direct f32", d32)
AT2, G2, BT2 = cook_toom([0, 1, -1], 2, 3)
rep("F(2x2) f32, pts 0,1,-1", wino_conv(x, w, AT2, G2, BT2, 2))
for pts in ([0, 1, -1, 2, -2], [0, 1, -1, '1/2', -2], [0, 1, -1, '1/2', '-1/2'], [0, '1/2', '-1/2', 2, -2], [0, 1, -1, '1/2', '-1/2'][:5]):
    AT, G, BT = cook_toom(pts, 4, 3)
    rep(f"F(4x4) f32, pts {pts}", wino_conv(x, w, AT, G, BT, 4))
AT, G, BT = cook_toom([0, 1, -1, 2, -2], 4, 3)
rep("F(4x4) f32 pts std, f32 filter transform", wino_conv(x, w, AT, G, BT, 4, filt64=False))
