"""Round-3 design experiment for the S2 preconditioner (numpy; not part of the product or the tests).
The shipped V-cycle uses 2x2 aggregation with piecewise-constant transfer: its coarse operator (sum of the two crossing edges) is the Galerkin operator of a
piecewise-constant interpolation and is too stiff by ~2x for smooth error, which floors PCG at 30-70 iterations. Variants tried here keep every level a 5-point
operator (same kernels): rediscretised coarse edges (scaled sums / series-parallel conductances) and bilinear (cell-centred, 9/3/3/1) prolongation with constant or
bilinear-transposed restriction.   usage: python scripts/mg_proto2.py"""
import sys
sys.path.insert(0, '/root/repo/tests')
import numpy as np, oracle_bind, synth
orc = oracle_bind.load(); orc._decl_color()


def system(H, W, lamda, seed=3, rough_frac=0.1):
    img = synth.image(seed, H, W); lab = np.ascontiguousarray(orc.bgr2lab(img).astype(np.float64) / 255.0)
    rng = np.random.default_rng(1); rough = np.where(rng.random(H * W) < rough_frac, 1e-6, 1.0)
    d = np.empty(H * W); wx = np.empty(H * W); wy = np.empty(H * W)
    orc.l.orc_wls_system(lab.reshape(-1), H, W, lamda, 1.2, rough, d, wx, wy)
    wx = wx.reshape(H, W).copy(); wy = wy.reshape(H, W).copy(); wx[:, -1] = 0; wy[-1, :] = 0
    return rough.reshape(H, W), wx, wy


def diag_of(r, wx, wy):
    d = r.copy(); d += wx; d += wy; d[:, 1:] += wx[:, :-1]; d[1:, :] += wy[:-1, :]; return d


def apply(d, wx, wy, x):
    y = d * x; y[:, :-1] -= wx[:, :-1] * x[:, 1:]; y[:, 1:] -= wx[:, :-1] * x[:, :-1]; y[:-1, :] -= wy[:-1, :] * x[1:, :]; y[1:, :] -= wy[:-1, :] * x[:-1, :]; return y


def pad2(a, Hc, Wc):
    p = np.zeros((Hc * 2, Wc * 2)); p[:a.shape[0], :a.shape[1]] = a; return p


def coarsen(r, wx, wy, mode):
    H, W = r.shape; Hc, Wc = (H + 1) // 2, (W + 1) // 2
    rp = pad2(r, Hc, Wc); rc = rp[0::2, 0::2] + rp[0::2, 1::2] + rp[1::2, 0::2] + rp[1::2, 1::2]
    wxp = pad2(wx, Hc, Wc); wyp = pad2(wy, Hc, Wc)
    cx0, cx1 = wxp[0::2, 1::2], wxp[1::2, 1::2]          # the two fine edges crossing the coarse face (x direction)
    cy0, cy1 = wyp[1::2, 0::2], wyp[1::2, 1::2]
    if mode == "sum":
        wxc, wyc = cx0 + cx1, cy0 + cy1
    elif mode == "half":
        wxc, wyc = 0.5 * (cx0 + cx1), 0.5 * (cy0 + cy1)
    elif mode == "series":
        # conductance between coarse cell centres: per fine row, the crossing edge in series with half of each in-cell edge on either side (in-cell edge = the
        # edge between the two cells of the aggregate in that row), rows in parallel
        ix = wxp[:, 0::2]                                  # in-cell x edges (from even col to odd col), all rows
        def ser(*c):
            inv = sum(1.0 / np.maximum(ci, 1e-300) for ci in c); return 1.0 / inv
        inl0, inl1 = ix[0::2, :], ix[1::2, :]              # in-cell edge of the left aggregate, row 0 / 1
        inr0 = np.zeros_like(inl0); inr1 = np.zeros_like(inl1); inr0[:, :-1] = inl0[:, 1:]; inr1[:, :-1] = inl1[:, 1:]
        wxc = ser(2 * inl0, cx0, 2 * np.where(inr0 > 0, inr0, 1e300)) * (cx0 > 0) + ser(2 * inl1, cx1, 2 * np.where(inr1 > 0, inr1, 1e300)) * (cx1 > 0)
        iy = wyp[0::2, :]
        int0, int1 = iy[:, 0::2], iy[:, 1::2]
        inb0 = np.zeros_like(int0); inb1 = np.zeros_like(int1); inb0[:-1, :] = int0[1:, :]; inb1[:-1, :] = int1[1:, :]
        wyc = ser(2 * int0, cy0, 2 * np.where(inb0 > 0, inb0, 1e300)) * (cy0 > 0) + ser(2 * int1, cy1, 2 * np.where(inb1 > 0, inb1, 1e300)) * (cy1 > 0)
    wxc = wxc.copy(); wyc = wyc.copy(); wxc[:, -1] = 0; wyc[-1, :] = 0
    return rc, wxc, wyc


def restrict_const(v):
    H, W = v.shape; Hc, Wc = (H + 1) // 2, (W + 1) // 2; p = pad2(v, Hc, Wc)
    return p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]


def prolong_const(vc, H, W):
    return np.repeat(np.repeat(vc, 2, 0), 2, 1)[:H, :W]


def prolong_bilin(vc, H, W):
    """cell-centred bilinear: fine cell (2I+a, 2J+b) = 9/16 own + 3/16 + 3/16 + 1/16 of the coarse cells towards its corner; clamped at the boundary"""
    Hc, Wc = vc.shape
    e = np.pad(vc, 1, mode="edge")
    out = np.empty((Hc * 2, Wc * 2))
    for a in (0, 1):
        for b in (0, 1):
            dy, dx = (-1 if a == 0 else 1), (-1 if b == 0 else 1)
            own = e[1:-1, 1:-1]; ny = e[1 + dy:Hc + 1 + dy, 1:-1]; nx = e[1:-1, 1 + dx:Wc + 1 + dx]; nd = e[1 + dy:Hc + 1 + dy, 1 + dx:Wc + 1 + dx]
            out[a::2, b::2] = (9 * own + 3 * ny + 3 * nx + nd) / 16.0
    return out[:H, :W]


def restrict_bilin(v):
    """transpose of prolong_bilin (without the boundary clamp's renormalisation: weights that would fall outside are added to the clamped cell)"""
    H, W = v.shape; Hc, Wc = (H + 1) // 2, (W + 1) // 2; p = pad2(v, Hc, Wc)
    acc = np.zeros((Hc + 2, Wc + 2))
    for a in (0, 1):
        for b in (0, 1):
            dy, dx = (-1 if a == 0 else 1), (-1 if b == 0 else 1)
            f = p[a::2, b::2] / 16.0
            acc[1:-1, 1:-1] += 9 * f; acc[1 + dy:Hc + 1 + dy, 1:-1] += 3 * f; acc[1:-1, 1 + dx:Wc + 1 + dx] += 3 * f; acc[1 + dy:Hc + 1 + dy, 1 + dx:Wc + 1 + dx] += f
    acc[1, :] += acc[0, :]; acc[-2, :] += acc[-1, :]; acc[:, 1] += acc[:, 0]; acc[:, -2] += acc[:, -1]
    return acc[1:-1, 1:-1].copy()


class MG:
    def __init__(s, r, wx, wy, coarse_mode="sum", P="const", R="const", w1=0.5808, w2=2.6437, coarse=8, kappa=1.0, nu=2):
        s.lv = []; s.P = P; s.R = R; s.w = (w1, w2) if nu == 2 else tuple([0.8] * nu); s.kap = kappa
        while True:
            s.lv.append((diag_of(r, wx, wy), wx, wy))
            if min(r.shape) <= coarse: break
            r, wx, wy = coarsen(r, wx, wy, coarse_mode)

    def smooth(s, l, x, b, order):
        d, wx, wy = s.lv[l]
        for w in order:
            x = x + w * (b - apply(d, wx, wy, x)) / d
        return x

    def vcycle(s, l, b):
        d, wx, wy = s.lv[l]
        if l == len(s.lv) - 1:
            x = np.zeros_like(b)
            for _ in range(60): x = x + 0.8 * (b - apply(d, wx, wy, x)) / d
            return x
        x = s.smooth(l, np.zeros_like(b), b, s.w)
        res = b - apply(d, wx, wy, x)
        rc = restrict_const(res) if s.R == "const" else restrict_bilin(res)
        ec = s.vcycle(l + 1, rc)
        x = x + s.kap * (prolong_const(ec, *b.shape) if s.P == "const" else prolong_bilin(ec, *b.shape))
        return s.smooth(l, x, b, s.w)


def pcg(d, wx, wy, b, x0, prec, rtol=1e-7, maxit=500):
    x = x0.copy(); r = b - apply(d, wx, wy, x); z = prec(r); p = z.copy(); rz = (r * z).sum(); bb = (b * b).sum(); it = 0
    while (r * r).sum() > rtol ** 2 * bb and it < maxit:
        Ap = apply(d, wx, wy, p); al = rz / (p * Ap).sum(); x += al * p; r -= al * Ap; z = prec(r); rz2 = (r * z).sum(); p = z + (rz2 / rz) * p; rz = rz2; it += 1
    return x, it


if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    variants = [("shipped: sum/const/const", dict()),
                ("half/const/const", dict(coarse_mode="half")),
                ("sum/bilin/const", dict(P="bilin")),
                ("half/bilin/const", dict(coarse_mode="half", P="bilin")),
                ("half/bilin/bilin", dict(coarse_mode="half", P="bilin", R="bilin")),
                ("series/bilin/const", dict(coarse_mode="series", P="bilin")),
                ("series/const/const", dict(coarse_mode="series")),
                ("sum/const/const kappa 2", dict(kappa=2.0)),
                ("half/bilin/const V(1,1)x0.8", dict(coarse_mode="half", P="bilin", nu=1))]
    for lam_f in (253.0, 63.3, 16.0, 4.0):
        r, wx, wy = system(S, S, 0.024 * lam_f); d = diag_of(r, wx, wy)
        rng = np.random.default_rng(5); x0 = rng.random((S, S)); b = r * x0
        row = []
        for name, kw in variants:
            mg = MG(r, wx, wy, **kw)
            try:
                _, it = pcg(d, wx, wy, b, x0, lambda v: mg.vcycle(0, v))
            except Exception as e:      # noqa
                it = -1
            row.append(it)
        print("lambda factor %6.1f" % lam_f, " | ".join("%s: %d" % (n, i) for (n, _), i in zip(variants, row)), flush=True)
