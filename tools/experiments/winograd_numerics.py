#!/usr/bin/env python3
"""Float32 error of the Winograd formulations considered for the FC layer, measured on the host BEFORE any HIP
was written: Cook-Toom matrices from sympy for a set of interpolation points, forward convolution and weight gradient
evaluated in float32 / float64 with torch and compared with the float64 direct form.

    python tools/experiments/winograd_numerics.py 2 5      # F(2x2,5x5), several point sets
    python tools/experiments/winograd_numerics.py 2 3
    python tools/experiments/winograd_numerics.py wgrad    # forward + weight gradient, F(2,5) / F(4,3) / F(2,3)
"""
import numpy as np, torch, itertools, sys
torch.manual_seed(0)
from fractions import Fraction as Fr

def winograd_mats(m, r, pts):
    """Cook-Toom F(m, r) with alpha = m+r-1 points (last = infinity). Returns AT (m x a), G (a x r), BT (a x a) as Fractions."""
    a = m + r - 1
    assert len(pts) == a - 1
    import sympy as sp
    x = sp.symbols('x')
    P = [sp.Rational(p.numerator, p.denominator) for p in pts]
    # Following Lavin/ wincnn
    def At(a_, m_):
        return sp.Matrix(m_, a_, lambda i, j: (P[j] ** i if j < a_ - 1 else (1 if i == m_ - 1 else 0)))
    # use wincnn formulation
    n = a - 1
    f = [sp.Integer(1)] * n
    for i in range(n):
        for j in range(n):
            if i != j:
                f[i] *= (P[i] - P[j])
    AT = sp.zeros(m, a)
    for i in range(m):
        for j in range(n):
            AT[i, j] = P[j] ** i
    AT[m - 1, n] = 1
    G = sp.zeros(a, r)
    for i in range(n):
        for j in range(r):
            G[i, j] = P[i] ** j / f[i]
    G[n, r - 1] = 1
    # B^T from polynomial: rows i<n: coefficients of prod_{j != i}(x - P[j]) ... ; row n: coefficients of prod_j (x - P[j])
    M = sp.Poly(sp.prod([x - p for p in P]), x)
    BT = sp.zeros(a, a)
    for i in range(n):
        q = sp.Poly(sp.prod([x - P[j] for j in range(n) if j != i]), x)
        c = q.all_coeffs()[::-1]
        for j, v in enumerate(c):
            BT[i, j] = v
    c = M.all_coeffs()[::-1]
    for j, v in enumerate(c):
        BT[n, j] = v
    return AT, G, BT

def to_np(Mx, dt):
    return np.array(Mx.tolist(), dtype=np.float64).astype(dt)

def check_exact(AT, G, BT, m, r):
    import sympy as sp
    a = m + r - 1
    d = sp.Matrix(a, 1, lambda i, j: sp.symbols('d%d' % i))
    g = sp.Matrix(r, 1, lambda i, j: sp.symbols('g%d' % i))
    y = AT * sp.matrix_multiply_elementwise(G * g, BT * d)
    for i in range(m):
        want = sum(d[i + j] * g[j] for j in range(r))
        assert sp.simplify(y[i] - want) == 0, (i, sp.simplify(y[i] - want))

def conv_direct(x, w, dt):
    return torch.nn.functional.conv2d(x.to(dt), w.to(dt))

def conv_wino(x, w, AT, G, BT, m, r, dt=torch.float32):
    # x (B,C,H,W), w (N,C,r,r); valid conv; H-r+1 divisible by m
    a = m + r - 1
    B, C, H, W = x.shape
    N = w.shape[0]
    Ho, Wo = H - r + 1, W - r + 1
    assert Ho % m == 0 and Wo % m == 0
    th, tw = Ho // m, Wo // m
    ATt, Gt, BTt = (torch.tensor(to_np(M_, np.float64)).to(dt) for M_ in (AT, G, BT))
    U = torch.einsum('ai,ncij,bj->abnc', Gt, w.to(dt), Gt)            # (a,a,N,C)
    # tiles
    xt = x.to(dt).unfold(2, a, m).unfold(3, a, m)                     # (B,C,th,tw,a,a)
    V = torch.einsum('ai,bcyxij,ej->aebyxc', BTt, xt, BTt)            # (a,a,B,th,tw,C)
    Mm = torch.einsum('aebyxc,aenc->aebyxn', V, U)                    # f32 accumulate over C (torch may use higher internally; fine)
    Y = torch.einsum('ia,aebyxn,je->bnyixj', ATt, Mm, ATt)            # (B,N,th,m,tw,m)
    return Y.reshape(B, N, Ho, Wo)


def wgrad_report():
    torch.manual_seed(0)

    def wgrad_wino(x, dy, AT, G, BT, m, r, dt):
        a = m + r - 1
        B, C, H, W = x.shape
        N = dy.shape[1]
        ATt, Gt, BTt = (torch.tensor(to_np(M_, np.float64)).to(dt) for M_ in (AT, G, BT))
        xt = x.to(dt).unfold(2, a, m).unfold(3, a, m)                     # (B,C,th,tw,a,a)
        V = torch.einsum('ai,bcyxij,ej->aebyxc', BTt, xt, BTt)
        dyt = dy.to(dt).unfold(2, m, m).unfold(3, m, m)                   # (B,N,th,tw,m,m)
        Zh = torch.einsum('ia,bnyxij,je->aebyxn', ATt, dyt, ATt)          # A dY A^T : (a,a,...)
        dU = torch.einsum('aebyxc,aebyxn->aenc', V, Zh)
        return torch.einsum('ai,aenc,ej->ncij', Gt, dU, Gt)               # G^T dU G

    for (m, r, pts, C, N, Ho, Wo) in [(2, 5, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-1, 2)], 128, 128, 68, 48),
                                      (2, 5, [Fr(0), Fr(1), Fr(-1), Fr(1, 2), Fr(-2)], 128, 128, 68, 48),
                                      (4, 3, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-1, 2)], 256, 128, 36, 24),
                                      (4, 3, [Fr(0), Fr(1), Fr(-1), Fr(1, 2), Fr(-2)], 256, 128, 36, 24),
                                      (4, 3, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)], 256, 128, 36, 24),
                                      (4, 3, [Fr(0), Fr(1), Fr(-1), Fr(1,2), Fr(-1,2)], 256, 128, 36, 24),
                                      (2, 3, [Fr(0), Fr(1), Fr(-1)], 256, 128, 36, 24)]:
        AT, G, BT = winograd_mats(m, r, pts)
        B = 4
        x = torch.randn(B, C, Ho + r - 1, Wo + r - 1, dtype=torch.float64)
        w = torch.randn(N, C, r, r, dtype=torch.float64) / (C * r * r) ** 0.5
        dy = torch.randn(B, N, Ho, Wo, dtype=torch.float64)
        ref = conv_direct(x, w, torch.float64)
        y32 = conv_wino(x, w, AT, G, BT, m, r, torch.float32).double()
        d32 = conv_direct(x, w, torch.float32).double()
        xr = x.clone().requires_grad_(); wr = w.clone().requires_grad_()
        torch.nn.functional.conv2d(xr, wr).backward(dy)
        gw32 = wgrad_wino(x, dy, AT, G, BT, m, r, torch.float32).double()
        gw64 = wgrad_wino(x, dy, AT, G, BT, m, r, torch.float64)
        xf = x.float().requires_grad_(); wf = w.float().requires_grad_()
        torch.nn.functional.conv2d(xf, wf).backward(dy.float())
        rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
        print('F(%d,%d) pts %s: fwd wino %.2e direct %.2e | wgrad wino f32 %.2e (f64 %.1e) direct f32 %.2e' %
              (m, r, [str(p) for p in pts], rel(y32, ref), rel(d32, ref), rel(gw32, wr.grad), rel(gw64, wr.grad), rel(wf.grad.double(), wr.grad)))

if __name__ == '__main__':
    if sys.argv[1] == 'wgrad':
        wgrad_report()
        sys.exit(0)

    m, r = int(sys.argv[1]), int(sys.argv[2])
    cands = {
      'std': [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)],
      'half': [Fr(0), Fr(1), Fr(-1), Fr(1,2), Fr(-1,2)],
      'mix': [Fr(0), Fr(1), Fr(-1), Fr(1,2), Fr(-2)],
      'mix2': [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-1,2)],
      'q': [Fr(0), Fr(1,2), Fr(-1,2), Fr(3,2), Fr(-3,2)],
    }
    if m + r - 1 == 4:
        cands = {'std': [Fr(0), Fr(1), Fr(-1)], 'half': [Fr(0), Fr(1,2), Fr(-1,2)]}
    C, N = 128, 128
    B, H, W = 2, 64 + r - 1, 44 + r - 1
    if (64 % m) or (44 % m):
        H, W = 66 + r - 1 - (66 % m), 44 + r - 1
    x = torch.randn(B, C, H, W, dtype=torch.float64)
    w = torch.randn(N, C, r, r, dtype=torch.float64) / (C * r * r) ** 0.5
    ref = conv_direct(x, w, torch.float64)
    d32 = conv_direct(x, w, torch.float32).double()
    print('direct f32 err', ((d32 - ref).abs().max() / ref.abs().max()).item())
    for name, pts in cands.items():
        AT, G, BT = winograd_mats(m, r, pts)
        if name == 'std': check_exact(AT, G, BT, m, r)
        y64 = conv_wino(x, w, AT, G, BT, m, r, torch.float64)
        y32 = conv_wino(x, w, AT, G, BT, m, r, torch.float32).double()
        print(name, 'f64 err %.2e' % ((y64 - ref).abs().max() / ref.abs().max()).item(), 'f32 err %.2e' % ((y32 - ref).abs().max() / ref.abs().max()).item(),
              'rms %.2e' % (((y32 - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item())
