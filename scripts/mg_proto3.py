"""Round-4 design experiment for the S2 preconditioner (numpy/scipy; not part of the product or the tests).
Vertex-centred coarsening (coarse point I = fine point 2I), operator-dependent (black-box / Dendy) interpolation, Galerkin 9-point coarse operators,
Chebyshev-weighted Jacobi smoothing. Compared with the shipped 2x2-aggregation V-cycle.   usage: python scripts/mg_proto3.py [size]"""
import sys
sys.path.insert(0, '/root/repo/tests')
import numpy as np, scipy.sparse as sp, time
import oracle_bind, synth
orc = oracle_bind.load(); orc._decl_color()


def system(H, W, lamda, seed=3, rough_frac=0.1):
    img = synth.image(seed, H, W); lab = np.ascontiguousarray(orc.bgr2lab(img).astype(np.float64) / 255.0)
    rng = np.random.default_rng(1); rough = np.where(rng.random(H * W) < rough_frac, 1e-6, 1.0)
    d = np.empty(H * W); wx = np.empty(H * W); wy = np.empty(H * W)
    orc.l.orc_wls_system(lab.reshape(-1), H, W, lamda, 1.2, rough, d, wx, wy)
    wx = wx.reshape(H, W).copy(); wy = wy.reshape(H, W).copy(); wx[:, -1] = 0; wy[-1, :] = 0
    return rough.reshape(H, W), wx, wy


def assemble(r, wx, wy):
    H, W = r.shape; n = H * W; idx = np.arange(n).reshape(H, W)
    d = r.copy(); d += wx; d += wy; d[:, 1:] += wx[:, :-1]; d[1:, :] += wy[:-1, :]
    rows = [idx.ravel()]; cols = [idx.ravel()]; vals = [d.ravel()]
    a, b, v = idx[:, :-1].ravel(), idx[:, 1:].ravel(), -wx[:, :-1].ravel(); rows += [a, b]; cols += [b, a]; vals += [v, v]
    a, b, v = idx[:-1, :].ravel(), idx[1:, :].ravel(), -wy[:-1, :].ravel(); rows += [a, b]; cols += [b, a]; vals += [v, v]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


def stencil(A, H, W):
    """9 planes S[dy+1][dx+1][y, x] = A[(y,x), (y+dy, x+dx)]"""
    A = A.tocoo(); S = np.zeros((3, 3, H, W))
    y, x = A.row // W, A.row % W; yy, xx = A.col // W, A.col % W
    S[yy - y + 1, xx - x + 1, y, x] = A.data
    return S


def interp_opdep(A, H, W, mode="opdep"):
    """P (n x nc): coarse points = even (y, x). Dendy's collapse for 9-point stencils (5-point is the special case)."""
    S = stencil(A, H, W); Hc, Wc = (H + 1) // 2, (W + 1) // 2
    idx = np.arange(H * W).reshape(H, W); cidx = np.arange(Hc * Wc).reshape(Hc, Wc)
    rows, cols, vals = [], [], []
    # coarse points: injection
    rows.append(idx[0::2, 0::2].ravel()); cols.append(cidx.ravel()); vals.append(np.ones(Hc * Wc))
    PW = np.zeros((H, W)); PE = np.zeros((H, W)); PN = np.zeros((H, W)); PS = np.zeros((H, W))
    # horizontal edge points (even y, odd x): collapse the stencil in y
    if mode == "opdep":
        w_ = -(S[0, 0] + S[1, 0] + S[2, 0]); e_ = -(S[0, 2] + S[1, 2] + S[2, 2]); c_ = S[1, 1] + S[0, 1] + S[2, 1]
        PW = w_ / c_; PE = e_ / c_
        n_ = -(S[0, 0] + S[0, 1] + S[0, 2]); s_ = -(S[2, 0] + S[2, 1] + S[2, 2]); c2 = S[1, 1] + S[1, 0] + S[1, 2]
        PN = n_ / c2; PS = s_ / c2
    else:   # bilinear
        PW[:] = 0.5; PE[:] = 0.5; PN[:] = 0.5; PS[:] = 0.5
        PE[:, -1] = 0; PS[-1, :] = 0
        if mode == "bilin1":   # renormalise at the boundary
            PW[:, -1] = 1.0; PN[-1, :] = 1.0
    ys, xs = np.mgrid[0:H, 0:W]
    m = (ys % 2 == 0) & (xs % 2 == 1)
    rows.append(idx[m]); cols.append(cidx[ys[m] // 2, (xs[m] - 1) // 2]); vals.append(PW[m])
    m2 = m & (xs + 1 < W)
    rows.append(idx[m2]); cols.append(cidx[ys[m2] // 2, (xs[m2] + 1) // 2]); vals.append(PE[m2])
    m = (ys % 2 == 1) & (xs % 2 == 0)
    rows.append(idx[m]); cols.append(cidx[(ys[m] - 1) // 2, xs[m] // 2]); vals.append(PN[m])
    m2 = m & (ys + 1 < H)
    rows.append(idx[m2]); cols.append(cidx[(ys[m2] + 1) // 2, xs[m2] // 2]); vals.append(PS[m2])
    P1 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(H * W, Hc * Wc))
    # centre points (odd, odd): x_c = -(sum offdiag * x_nbr) / diag, neighbours already interpolated
    m = ((ys % 2 == 1) & (xs % 2 == 1)).ravel()
    if mode == "opdep":
        D = A.diagonal(); Aoff = A - sp.diags(D)
        Cm = sp.diags(np.where(m, -1.0 / D, 0.0)) @ Aoff
    else:
        # bilinear: average of the 4 edge neighbours' interpolants = 1/4 each corner (interior)
        Hh = assemble(np.zeros((H, W)), np.pad(np.ones((H, W - 1)), ((0, 0), (0, 1))), np.pad(np.ones((H - 1, W)), ((0, 1), (0, 0))))
        D = Hh.diagonal(); Aoff = Hh - sp.diags(D)
        Cm = sp.diags(np.where(m, -1.0 / D, 0.0)) @ Aoff
    P = P1 + Cm @ P1
    return P.tocsr(), Hc, Wc


class VMG:
    """vertex-centred Galerkin hierarchy"""
    def __init__(s, A, H, W, mode="opdep", ws=(0.5346, 0.9677, 5.0974), coarse=8, f32=False, wsc=None):
        s.lv = []; s.ws = ws; s.wsc = wsc or ws
        while True:
            Dinv = 1.0 / A.diagonal()
            if min(H, W) <= coarse:
                s.lv.append((A, Dinv, None, H, W)); s.Ac = np.linalg.inv(A.toarray()); break
            P, Hc, Wc = interp_opdep(A, H, W, mode)
            s.lv.append((A, Dinv, P, H, W))
            A = (P.T @ A @ P).tocsr(); H, W = Hc, Wc

    def vcycle(s, l, b):
        A, Dinv, P, H, W = s.lv[l]
        if P is None:
            return s.Ac @ b
        ws = s.ws if l == 0 else s.wsc
        x = ws[0] * Dinv * b
        for w in ws[1:]: x = x + w * Dinv * (b - A @ x)
        rc = P.T @ (b - A @ x)
        x = x + P @ s.vcycle(l + 1, rc)
        for w in ws: x = x + w * Dinv * (b - A @ x)
        return x


class AMG2:
    """shipped: 2x2 aggregation, piecewise-constant transfer (Galerkin = sum of crossing edges)"""
    def __init__(s, A, H, W, ws=(0.5346, 0.9677, 5.0974), coarse=8):
        s.lv = []; s.ws = ws
        while True:
            Dinv = 1.0 / A.diagonal()
            if min(H, W) <= coarse:
                s.lv.append((A, Dinv, None, H, W)); s.Ac = np.linalg.inv(A.toarray()); break
            Hc, Wc = (H + 1) // 2, (W + 1) // 2
            ys, xs = np.mgrid[0:H, 0:W]
            P = sp.csr_matrix((np.ones(H * W), (np.arange(H * W), ((ys // 2) * Wc + xs // 2).ravel())), shape=(H * W, Hc * Wc))
            s.lv.append((A, Dinv, P, H, W)); A = (P.T @ A @ P).tocsr(); H, W = Hc, Wc
    vcycle = VMG.vcycle
    wsc = property(lambda s: s.ws)


def pcg(A, b, x0, prec, rtol=1e-7, maxit=500):
    x = x0.copy(); r = b - A @ x; z = prec(r); p = z.copy(); rz = r @ z; bb = b @ b; it = 0
    while r @ r > rtol ** 2 * bb and it < maxit:
        Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap; z = prec(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2; it += 1
    return x, it


if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    W2 = (0.5808, 2.6437); W3 = (0.5346, 0.9677, 5.0974)
    for lam_f in (253.0, 63.3, 16.0, 4.0):
        r, wx, wy = system(S, S, 0.024 * lam_f); A = assemble(r, wx, wy)
        rng = np.random.default_rng(5); x0 = rng.random(S * S); b = r.ravel() * x0
        res = []
        for name, mk in [("agg W3", lambda: AMG2(A, S, S, W3)), ("vc bilin W3", lambda: VMG(A, S, S, "bilin", W3)), ("vc opdep W2", lambda: VMG(A, S, S, "opdep", W2)),
                         ("vc opdep W3", lambda: VMG(A, S, S, "opdep", W3)), ("vc opdep fine W3 coarse W2", lambda: VMG(A, S, S, "opdep", W3, wsc=W2))]:
            t = time.time(); mg = mk(); _, it = pcg(A, b, x0, lambda v: mg.vcycle(0, v)); res.append("%s: %d (%.1fs)" % (name, it, time.time() - t))
        print("S %d lambda factor %6.1f | " % (S, lam_f) + " | ".join(res), flush=True)
