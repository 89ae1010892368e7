"""S2 preconditioner on NATURAL weights (numpy prototype, scripts/mg_proto3.py's hierarchy): PCG iterations of the shipped cycle on crops of the reference's demo photographs
against the synthetic image, and of candidate smoothers.   usage: python scripts/mg_proto_natural.py [crop=256]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl, time
from PIL import Image
import mg_proto3 as M
orc = M.orc
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W3 = (0.5346, 0.9677, 5.0974)


def system_img(img_bgr, lamda):
    H, W = img_bgr.shape[:2]
    lab = np.ascontiguousarray(orc.bgr2lab(img_bgr).astype(np.float64) / 255.0)
    rough = np.ones(H * W)
    d = np.empty(H * W); wx = np.empty(H * W); wy = np.empty(H * W)
    orc.l.orc_wls_system(lab.reshape(-1), H, W, lamda, 1.2, rough, d, wx, wy)
    wx = wx.reshape(H, W).copy(); wy = wy.reshape(H, W).copy(); wx[:, -1] = 0; wy[-1, :] = 0
    return rough.reshape(H, W), wx, wy


class VMGx(M.VMG):
    """variants of the cycle: smoother = 'jac' (shipped), 'jac4' (one more sweep, same weights recycled), 'line' (x-line then y-line Jacobi, one pair per leg), 'gs' (symmetric Gauss-Seidel)"""
    def __init__(s, A, H, W, smoother="jac", **kw):
        s.tile = kw.pop("tile", 0); s.line_levels = kw.pop("line_levels", 99); s.njac = kw.pop("njac", 3); s.om = kw.pop("om", 0.9)
        super().__init__(A, H, W, **kw); s.sm = smoother; s.fac = {}

    def smooth(s, l, x, b, pre):
        A, Dinv, P, H, W = s.lv[l]
        if s.sm == "jac" or ((s.sm.startswith("line") or s.sm.startswith("jl")) and l >= s.line_levels) or (s.sm.startswith("adi") and l > 0):
            if l not in s.fac:                                      # the product's safe diagonal: dt = max(d, (|d| + sum |w|) / 2)
                d = A.diagonal(); off = np.asarray(abs(A).sum(1)).ravel() - abs(d)
                s.fac[l] = 1.0 / np.maximum(d, (abs(d) + off) / 2)
            Di = s.fac[l]; ws = s.ws
            if x is None: x = ws[0] * Di * b; ws = ws[1:]
            for w in ws: x = x + w * Di * (b - A @ x)
            return x
        if x is None: x = np.zeros_like(b)
        if s.sm == "gs":
            if l not in s.fac: s.fac[l] = (sp.tril(A, format="csr"), sp.triu(A, format="csr"))
            L, U = s.fac[l]
            x = x + spl.spsolve_triangular(L, b - A @ x, lower=True)
            x = x + spl.spsolve_triangular(U, b - A @ x, lower=False)
            return x
        if s.sm.startswith("adi") and l == 0:                          # tile-local ADI block step FIRST in the pre-smoother (from zero: no residual needed, no halo), LAST in the post-smoother (mirrored)
            if "adi" not in s.fac:
                d = A.diagonal(); off = np.asarray(abs(A).sum(1)).ravel() - abs(d)
                coo = A.tocoo(); tx, ty = s.tile
                r0, c0, r1, c1 = coo.row // W, coo.row % W, coo.col // W, coo.col % W
                same_tile = (r0 // ty == r1 // ty) & (c0 // tx == c1 // tx)
                mk = lambda m: sp.csr_matrix((coo.data[m], (coo.row[m], coo.col[m])), shape=A.shape)
                s.fac["adi"] = (1.0 / np.maximum(d, (abs(d) + off) / 2), spl.splu(mk(same_tile & (r0 == r1)).tocsc()), spl.splu(mk(same_tile & (c0 == c1)).tocsc()), mk(same_tile))
            Di, Lx, Ly, Att = s.fac["adi"]; om = s.om
            def block(r, first, second):
                e1 = first.solve(r); e2 = second.solve(r - Att @ e1); return om * (e1 + e2)
            ws = list(s.ws[:s.njac])
            if pre:
                x = block(b, Lx, Ly)
                for w in ws: x = x + w * Di * (b - A @ x)
                return x
            for w in ws: x = x + w * Di * (b - A @ x)
            return x + block(b - A @ x, Ly, Lx)
        if s.sm.startswith("jl") and l < s.line_levels:                # Jacobi sweeps + one tile-cut x/y line pair (symmetric: lines last before the coarse grid, first after)
            key = ("jl", l)
            if key not in s.fac:
                d = A.diagonal(); off = np.asarray(abs(A).sum(1)).ravel() - abs(d)
                coo = A.tocoo(); same_row = (coo.row // W) == (coo.col // W); same_col = (coo.row % W) == (coo.col % W)
                if s.tile:
                    same_row &= (coo.row % W) // s.tile == (coo.col % W) // s.tile
                    same_col &= (coo.row // W) // s.tile == (coo.col // W) // s.tile
                Ar = sp.csr_matrix((coo.data[same_row], (coo.row[same_row], coo.col[same_row])), shape=A.shape)
                Ac = sp.csr_matrix((coo.data[same_col], (coo.row[same_col], coo.col[same_col])), shape=A.shape)
                s.fac[key] = (1.0 / np.maximum(d, (abs(d) + off) / 2), spl.splu(Ar.tocsc()), spl.splu(Ac.tocsc()))
            Di, Lr, Lc = s.fac[key]; ws = s.ws[:s.njac] if s.njac < 3 else s.ws
            def jac(x):
                w_ = list(ws)
                if x is None: x = w_[0] * Di * b; w_ = w_[1:]
                for w in w_: x = x + w * Di * (b - A @ x)
                return x
            def lines(x, order):
                for F in order: x = x + 0.9 * F.solve(b - A @ x)
                return x
            if pre: return lines(jac(x), (Lr, Lc))
            return jac(lines(x, (Lc, Lr)))
        if s.sm.startswith("line"):
            if l not in s.fac:
                idx = np.arange(H * W).reshape(H, W)
                Ax = A.multiply(sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape).multiply(0) + 0)  # placeholder
                coo = A.tocoo(); same_row = (coo.row // W) == (coo.col // W); same_col = (coo.row % W) == (coo.col % W)
                if s.tile:                                      # lines cut at tile boundaries (what a tile-fused GPU leg can do): couplings across a cut stay outside the line solve
                    same_row &= (coo.row % W) // s.tile == (coo.col % W) // s.tile
                    same_col &= (coo.row // W) // s.tile == (coo.col // W) // s.tile
                Ar = sp.csr_matrix((coo.data[same_row], (coo.row[same_row], coo.col[same_row])), shape=A.shape)
                Ac = sp.csr_matrix((coo.data[same_col], (coo.row[same_col], coo.col[same_col])), shape=A.shape)
                s.fac[l] = (spl.splu(Ar.tocsc()), spl.splu(Ac.tocsc()))
            Lr, Lc = s.fac[l]
            order = (Lr, Lc) if pre else (Lc, Lr)
            for F in order: x = x + 0.9 * F.solve(b - A @ x)
            return x

    def vcycle(s, l, b):
        A, Dinv, P, H, W = s.lv[l]
        if P is None: return s.Ac @ b
        x = s.smooth(l, None, b, True)
        x = x + P @ s.vcycle(l + 1, P.T @ (b - A @ x))
        return s.smooth(l, x, b, False)


def crop(name, y0, x0):
    im = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "natural", name + ".png")).convert("RGB"))[..., ::-1]
    return np.ascontiguousarray(im[y0:y0 + S, x0:x0 + S])


import synth
imgs = {"synthetic": synth.image(3, S, S), "in0 crop": crop("in0", 100, 200), "in1 crop": crop("in1", 150, 200), "in4 crop": crop("in4", 120, 220)}
for lam_f in (253.0, 4.0):
    for name, img in imgs.items():
        r, wx, wy = system_img(img, 0.024 * lam_f); A = M.assemble(r, wx, wy)
        rng = np.random.default_rng(5); x0 = rng.random(S * S); b = r.ravel() * x0
        res = []
        W2 = (0.5808, 2.6437)
        for sm, kw in (("jac", {}), ("jl3 L0 t32", dict(line_levels=1, tile=32)), ("adi3 32x16 .9", dict(tile=(32, 16))), ("adi3 32x16 1.", dict(tile=(32, 16), om=1.0)), ("adi3 32x14", dict(tile=(32, 14))), ("adi2 32x16", dict(tile=(32, 16), njac=2)), ("adi3 32x32", dict(tile=(32, 32)))):
            t = time.time(); mg = VMGx(A, S, S, sm, mode="opdep", ws=W3, **kw)
            _, it = M.pcg(A, b, x0, lambda v: mg.vcycle(0, v)); res.append("%s %d (%.0fs)" % (sm, it, time.time() - t))
        print("lambda x%-5.0f %-10s | " % (lam_f, name) + " | ".join(res), flush=True)
