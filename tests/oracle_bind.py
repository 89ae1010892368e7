"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os
import subprocess
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(REPO, "oracle")
ORC_LIB = os.path.join(ORC_DIR, "liboracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
I = C.c_int


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, l):
        self.l = l
        l.orc_feat_normalize.argtypes = [_f32p, _f32p, C.c_void_p, I, I, I]
        l.orc_nnf_init.argtypes = [_u32p, I, I, I, I]
        l.orc_nnf_upsample.argtypes = [_u32p, _u32p, I, I, I, I, I, I]
        l.orc_patchmatch.argtypes = [_f32p, _f32p, I, I, I, I, I, I, I, I, C.c_uint32, _u32p, _f32p]
        l.orc_patchmatch_last_evals.restype = C.c_longlong
        l.orc_feature_distance.argtypes = [_f32p, _f32p, _f32p, I, I, I]
        l.orc_bds_vote_features.argtypes = [_u32p, _u32p, _f32p, _f32p, C.c_void_p, I, I, I, I, I, I, C.c_float, C.c_float]
        l.orc_bds_vote_image.argtypes = [_u8p, I, I, _u8p, I, I, _u32p, _u32p, I, C.c_double, C.c_double, _u8p]

    # ---- VGG19 (orc_vgg.c)
    def _decl_vgg(self):
        l = self.l
        if getattr(self, "_vgg_declared", False):
            return
        l.orc_vgg_preprocess.argtypes = [_u8p, I, I, _f32p]
        l.orc_conv3x3.argtypes = [_f32p, I, I, I, _f32p, _f32p, I, _f32p, I]
        l.orc_maxpool2x2.argtypes = [_f32p, I, I, I, _f32p]
        l.orc_maxpool_generic.argtypes = [_f32p, I, I, I, I, I, _f32p, C.POINTER(I), C.POINTER(I)]
        l.orc_vgg19_features.argtypes = [_u8p, I, I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), I, C.POINTER(C.c_void_p), C.c_void_p]
        self._vgg_declared = True

    def vgg_preprocess(self, bgr):
        self._decl_vgg()
        img = np.ascontiguousarray(bgr, np.uint8)
        out = np.empty((3,) + img.shape[:2], np.float32)
        self.l.orc_vgg_preprocess(img, img.shape[0], img.shape[1], out)
        return out

    def conv3x3(self, x, w, b, relu=True):
        self._decl_vgg()
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
        cin, H, W = x.shape
        out = np.empty((w.shape[0], H, W), np.float32)
        self.l.orc_conv3x3(x, cin, H, W, w, b, w.shape[0], out, 1 if relu else 0)
        return out

    def maxpool2x2(self, x):
        self._decl_vgg()
        x = np.ascontiguousarray(x, np.float32)
        c, H, W = x.shape
        out = np.empty((c, (H - 1) // 2 + 1, (W - 1) // 2 + 1), np.float32)
        self.l.orc_maxpool2x2(x, c, H, W, out)
        return out

    def maxpool_generic(self, x, k, s):
        self._decl_vgg()
        x = np.ascontiguousarray(x, np.float32)
        c, H, W = x.shape
        out = np.empty((c, H, W), np.float32)
        ho, wo = I(), I()
        self.l.orc_maxpool_generic(x, c, H, W, k, s, out, C.byref(ho), C.byref(wo))
        return out.reshape(-1)[: c * ho.value * wo.value].reshape(c, ho.value, wo.value).copy()

    def vgg19_features(self, bgr, weights, biases, deepest_tap=5):
        self._decl_vgg()
        img = np.ascontiguousarray(bgr, np.uint8)
        h, w = img.shape[:2]
        ws = [np.ascontiguousarray(x, np.float32) for x in weights]
        bs = [np.ascontiguousarray(x, np.float32) for x in biases]
        tapc = [64, 128, 256, 512, 512]
        outs, hh, ww = [], h, w
        for t in range(deepest_tap):
            outs.append(np.empty((tapc[t], hh, ww), np.float32))
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        wp = (C.c_void_p * len(ws))(*[x.ctypes.data for x in ws])
        bp = (C.c_void_p * len(bs))(*[x.ctypes.data for x in bs])
        tp = (C.c_void_p * 5)(*([o.ctypes.data for o in outs] + [None] * (5 - deepest_tap)))
        dims = np.zeros(15, np.int32)
        self.l.orc_vgg19_features(img, h, w, wp, bp, deepest_tap, tp, _ptr(dims))
        return outs

    # ---- colour stage (orc_cvt.c, orc_color.c)
    def _decl_color(self):
        l = self.l
        if getattr(self, "_color_declared", False):
            return
        l.orc_bgr2lab_u8.argtypes = [_u8p, C.c_size_t, _u8p]
        l.orc_lab2bgr_u8.argtypes = [_u8p, C.c_size_t, _u8p]
        l.orc_resize_u8c3.argtypes = [_u8p, I, I, _u8p, I, I]
        l.orc_resize_f64c3.argtypes = [_f64p, I, I, _f64p, I, I]
        l.orc_kmeans_labels.argtypes = [_f32p, I, I, I, I, C.c_uint64, _i32p]
        l.orc_kmeans_labels.restype = I
        l.orc_knn_graph.argtypes = [_f64p, I, I, _i32p, I, I, I, I, I, _i32p, _f64p]
        l.orc_local_color_transfer.argtypes = [_f32p, _u8p, _u8p, _u8p, _i32p, _f64p, I, I, I, I, I, C.c_void_p, _u8p, C.c_void_p, I]
        l.orc_local_color_transfer.restype = I
        l.orc_wls_solve.argtypes = [_f64p, _f64p, _f64p, I, I, C.c_double, C.c_double, _f64p, I]
        l.orc_wls_solve.restype = I
        l.orc_wls_system.argtypes = [_f64p, I, I, C.c_double, C.c_double, _f64p, _f64p, _f64p, _f64p]
        l.orc_wls_vcycle_apply.argtypes = [_f64p, I, I, C.c_double, C.c_double, _f64p, _f64p, _f64p, I]
        l.orc_wls_vcycle_apply.restype = I
        l.orc_wls_hierarchy_stats.argtypes = [_f64p, I, I, C.c_double, C.c_double, _f64p, _f64p]
        l.orc_wls_hierarchy_stats.restype = I
        self._color_declared = True

    def bgr2lab(self, bgr):
        self._decl_color()
        a = np.ascontiguousarray(bgr, np.uint8); out = np.empty_like(a)
        self.l.orc_bgr2lab_u8(a.reshape(-1, 3), a.size // 3, out.reshape(-1, 3))
        return out

    def lab2bgr(self, lab, form=None):
        """form None = the oracle's current default (0 = piecewise form unless orc_set_lab2bgr_form changed it; 1 = plain-cube form); 0 / 1 explicit"""
        self._decl_color()
        a = np.ascontiguousarray(lab, np.uint8); out = np.empty_like(a)
        if form is None:
            self.l.orc_lab2bgr_u8(a.reshape(-1, 3), a.size // 3, out.reshape(-1, 3))
        else:
            self.l.orc_lab2bgr_u8_form.argtypes = [_u8p, C.c_size_t, _u8p, I]
            self.l.orc_lab2bgr_u8_form(a.reshape(-1, 3), a.size // 3, out.reshape(-1, 3), form)
        return out

    def wls_vcycle(self, lab, lamda, alpha, rough, r):
        """z = V-cycle(r) of the S2 preconditioner for the WLS system of (lab [H][W][3] in [0,1], roughness [H*W]); r: [nv][H*W][6]. Returns (z, number of levels)."""
        self._decl_color()
        lab = np.ascontiguousarray(lab, np.float64); H, W = lab.shape[:2]
        r = np.ascontiguousarray(r, np.float64); z = np.empty_like(r)
        nl = self.l.orc_wls_vcycle_apply(lab.reshape(-1), H, W, lamda, alpha, np.ascontiguousarray(rough, np.float64).reshape(-1), r.reshape(-1), z.reshape(-1), r.shape[0])
        return z, nl

    def wls_hierarchy_stats(self, lab, lamda, alpha, rough):
        """per level: [n, min diagonal, min row sum (d - sum of couplings), number of couplings of the wrong sign, max safe-diagonal / diagonal]"""
        self._decl_color()
        lab = np.ascontiguousarray(lab, np.float64); H, W = lab.shape[:2]
        out = np.zeros((16, 5))
        nl = self.l.orc_wls_hierarchy_stats(lab.reshape(-1), H, W, lamda, alpha, np.ascontiguousarray(rough, np.float64).reshape(-1), out.reshape(-1))
        return out[:nl]

    def resize_u8c3(self, img, dh, dw):
        self._decl_color()
        a = np.ascontiguousarray(img, np.uint8); out = np.empty((dh, dw, 3), np.uint8)
        self.l.orc_resize_u8c3(a, a.shape[0], a.shape[1], out, dh, dw)
        return out

    def resize_f64c3(self, img, dh, dw):
        self._decl_color()
        a = np.ascontiguousarray(img, np.float64); out = np.empty((dh, dw, 3), np.float64)
        self.l.orc_resize_f64c3(a, a.shape[0], a.shape[1], out, dh, dw)
        return out

    def cluster_features(self, feat_chw, K=10, iters=11, seed=1):
        """Same contract as nct_cluster_features: un-normalised CHW in, labels out (normalisation = orc_feat_normalize)."""
        self._decl_color()
        f = self.feat_normalize(feat_chw)
        Cc, h, w = f.shape
        hwc = np.ascontiguousarray(f.reshape(Cc, h * w).T)
        labels = np.empty(h * w, np.int32)
        nl = self.l.orc_kmeans_labels(hwc, h * w, Cc, K, iters, seed, labels)
        return labels.reshape(h, w), nl

    def knn_graph(self, lab_u8, labels, nlabels, samples, k=8):
        self._decl_color()
        lab = np.ascontiguousarray(lab_u8, np.uint8)
        h, w = lab.shape[:2]
        labd = lab.astype(np.float64) * (1.0 / 255.0)
        lb = np.ascontiguousarray(labels, np.int32)
        ids = np.empty((h * w, k), np.int32); ws = np.empty((h * w, k), np.float64)
        self.l.orc_knn_graph(labd.reshape(-1), h, w, lb.reshape(-1), lb.shape[0], lb.shape[1], nlabels, samples, k, ids.reshape(-1), ws.reshape(-1))
        return ids, ws

    def local_color_transfer(self, err, s_level, g_level, s_full, knn_id, knn_w, layer, params=None, want_stages=False, s2_exact=False):
        self._decl_color()
        err = np.ascontiguousarray(err, np.float32)
        h, w = err.shape
        s_full = np.ascontiguousarray(s_full, np.uint8)
        H, W = s_full.shape[:2]
        p = params or dict(eps=0.60, nonlocal_weight=2.0, local_weight=0.125, wls_lambda_init=0.024, wls_alpha=1.2, k_num=8.0)
        prm = (C.c_double * 6)(p["eps"], p["nonlocal_weight"], p["local_weight"], p["wls_lambda_init"], p["wls_alpha"], p["k_num"])
        out = np.empty((H, W, 3), np.uint8)
        keep = {"ab_local": np.empty((2, h * w, 3)), "ab_nonlocal": np.empty((2, h * w, 3)), "ab_up": np.empty((2, H * W, 3)),
                "roughness": np.empty(H * W), "ab_wls": np.empty((2, H * W, 3)), "cg_iters": np.zeros(3, np.int32), "wls_iters": np.zeros(6, np.int32)}
        st = (C.c_void_p * 7)(*[keep[k].ctypes.data for k in ("ab_local", "ab_nonlocal", "ab_up", "roughness", "ab_wls", "cg_iters", "wls_iters")])
        rc = self.l.orc_local_color_transfer(err.reshape(-1), np.ascontiguousarray(s_level, np.uint8).reshape(-1, 3),
                                             np.ascontiguousarray(g_level, np.uint8).reshape(-1, 3), s_full.reshape(-1, 3),
                                             np.ascontiguousarray(knn_id, np.int32).reshape(-1), np.ascontiguousarray(knn_w, np.float64).reshape(-1),
                                             layer, h, w, H, W, C.addressof(prm), out.reshape(-1, 3), C.addressof(st), 1 if s2_exact else 0)
        assert rc == 0
        return (out, keep) if want_stages else out

    # ---- whole pair (orc_pipeline.c)
    def process_pair(self, src, ref, weights, biases, params=None, want_levels=False, s2_exact=False, want_nnf=False):
        """want_levels: also return the 5 intermediate results [5][H][W][3]; want_nnf: also return a dict of per-level
        ann/bnn/annd/bnnd/guide/err/result lists (level 0 = coarsest), the same intermediates nct_pair_run_levels exposes."""
        import ctypes as Cc

        class P(Cc.Structure):
            _fields_ = [("bds_weight", Cc.c_double), ("eps", Cc.c_double), ("nonlocal_weight", Cc.c_double), ("local_weight", Cc.c_double),
                        ("wls_lambda_init", Cc.c_double), ("cluster_num", Cc.c_int), ("k_num", Cc.c_int), ("patch_size", Cc.c_int),
                        ("wls_alpha", Cc.c_double), ("pm_iters", Cc.c_int), ("seed", Cc.c_uint32), ("levels", Cc.c_int), ("flags", Cc.c_uint32)]
        d = dict(bds_weight=2.0, eps=0.60, nonlocal_weight=2.0, local_weight=0.125, wls_lambda_init=0.024, cluster_num=10, k_num=8,
                 patch_size=3, wls_alpha=1.2, pm_iters=10, seed=1, levels=5, flags=0)
        if params:
            d.update(params)
        prm = P(**d)
        s = np.ascontiguousarray(src, np.uint8); r = np.ascontiguousarray(ref, np.uint8)
        H, W = s.shape[:2]; RH, RW = r.shape[:2]
        ws = [np.ascontiguousarray(x, np.float32) for x in weights]; bs = [np.ascontiguousarray(x, np.float32) for x in biases]
        wp = (C.c_void_p * len(ws))(*[x.ctypes.data for x in ws]); bp = (C.c_void_p * len(bs))(*[x.ctypes.data for x in bs])
        out = np.empty_like(s)
        lv = np.empty((5, H, W, 3), np.uint8) if want_levels else None
        keep, plv = None, None
        if want_nnf:
            dims, h, w, h2, w2 = [], H, W, RH, RW
            for _ in range(5):
                dims.insert(0, (h, w, h2, w2))
                h, w, h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1, (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
            keep = {k: [] for k in ("ann", "bnn", "annd", "bnnd", "guide", "err", "result")}
            for (ah, aw, bh, bw) in dims:
                keep["ann"].append(np.zeros((ah, aw), np.uint32)); keep["bnn"].append(np.zeros((bh, bw), np.uint32))
                keep["annd"].append(np.zeros((ah, aw), np.float32)); keep["bnnd"].append(np.zeros((bh, bw), np.float32))
                keep["guide"].append(np.zeros((ah, aw, 3), np.uint8)); keep["err"].append(np.zeros((ah, aw), np.float32))
                keep["result"].append(np.zeros((H, W, 3), np.uint8))

            class LV(Cc.Structure):
                _fields_ = [(k, Cc.c_void_p * 5) for k in ("ann", "bnn", "annd", "bnnd", "guide", "err", "result")]
            plv = LV()
            for k in keep:
                setattr(plv, k, (Cc.c_void_p * 5)(*[a.ctypes.data for a in keep[k]]))
            keep["dims"] = dims
        self.l.orc_process_pair_levels.argtypes = [_u8p, I, I, _u8p, I, I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, _u8p, C.c_void_p, I, C.c_void_p]
        self.l.orc_process_pair_levels.restype = I
        rc = self.l.orc_process_pair_levels(s.reshape(-1, 3), H, W, r.reshape(-1, 3), RH, RW, wp, bp, C.addressof(prm), out.reshape(-1, 3), _ptr(lv), 1 if s2_exact else 0,
                                            C.addressof(plv) if plv is not None else None)
        assert rc == 0
        res = (out,) + ((lv,) if want_levels else ()) + ((keep,) if want_nnf else ())
        return res if len(res) > 1 else out

    def feat_normalize(self, src, want_resp=False):
        src = np.ascontiguousarray(src, np.float32)
        Cc, H, W = src.shape
        dst = np.empty_like(src)
        resp = np.empty((H, W), np.float32) if want_resp else None
        self.l.orc_feat_normalize(src, dst, _ptr(resp), Cc, H, W)
        return (dst, resp) if want_resp else dst

    def nnf_init(self, ah, aw, bh, bw):
        nnf = np.empty((ah, aw), np.uint32)
        self.l.orc_nnf_init(nnf, ah, aw, bh, bw)
        return nnf

    def nnf_upsample(self, half, ah, aw, bh, bw):
        half = np.ascontiguousarray(half, np.uint32)
        nnf = np.empty((ah, aw), np.uint32)
        self.l.orc_nnf_upsample(half, nnf, ah, aw, bh, bw, half.shape[0], half.shape[1])
        return nnf

    def patchmatch(self, a, b, nnf, iters=10, rs_max=32, seed=0, patch=3):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        Cc, ah, aw = a.shape
        _, bh, bw = b.shape
        nnf = np.array(nnf, np.uint32, order="C", copy=True).reshape(ah, aw)
        dist = np.empty((ah, aw), np.float32)
        self.l.orc_patchmatch(a, b, Cc, ah, aw, bh, bw, patch, iters, rs_max, seed, nnf, dist)
        return nnf, dist

    def last_evals(self):
        return int(self.l.orc_patchmatch_last_evals())

    def patchmatch_inplace(self, a, b, nnf, iters=10, rs_max=32, seed=0, patch=3, schedule=1):
        """the reference's own in-place schedule (orc_nnf_inplace.c): schedule 1 = thread-sequential, 2 = lock step."""
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        Cc, ah, aw = a.shape
        _, bh, bw = b.shape
        nnf = np.array(nnf, np.uint32, order="C", copy=True).reshape(ah, aw)
        dist = np.empty((ah, aw), np.float32)
        self.l.orc_patchmatch_inplace.argtypes = [_f32p, _f32p, I, I, I, I, I, I, I, I, C.c_uint32, I, _u32p, _f32p]
        rc = self.l.orc_patchmatch_inplace(a, b, Cc, ah, aw, bh, bw, patch, iters, rs_max, seed, schedule, nnf, dist)
        assert rc == 0
        return nnf, dist

    def field_stats(self, d):
        """mean, p5, p25, p50, p75, p95 of a distance field"""
        d = np.ascontiguousarray(d, np.float32)
        out = np.empty(6, np.float64)
        self.l.orc_field_stats.argtypes = [_f32p, I, _f64p]
        self.l.orc_field_stats(d.reshape(-1), d.size, out)
        return out

    def set_pm_schedule(self, s):
        """which PatchMatch schedule process_pair runs: 0 = the product's (default), 1 / 2 = the reference's in-place schedule"""
        self.l.orc_set_pm_schedule(int(s))

    def set_s1_form(self, f):
        """which S1 recurrence process_pair runs: 0 = canonical (default, what the GPU reproduces), 1 = the literal textbook CG on the assembled A (SparseSolver_GPU.cu:132-159)"""
        self.l.orc_set_s1_form(int(f))

    def feature_distance(self, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        Cc, H, W = a.shape
        err = np.empty((H, W), np.float32)
        self.l.orc_feature_distance(a, b, err, Cc, H, W)
        return err

    def bds_vote_features(self, ann, bnn, pin, w_coh=1.0, w_comp=2.0, patch=3, want_pw=False):
        pin = np.ascontiguousarray(pin, np.float32)
        Cc, bh, bw = pin.shape
        ann = np.ascontiguousarray(ann, np.uint32)
        bnn = np.ascontiguousarray(bnn, np.uint32)
        ah, aw = ann.shape
        pout = np.empty((Cc, ah, aw), np.float32)
        pw = np.empty((ah, aw), np.float32) if want_pw else None
        self.l.orc_bds_vote_features(ann, bnn, pin, pout, _ptr(pw), Cc, ah, aw, bh, bw, patch, w_coh, w_comp)
        return (pout, pw) if want_pw else pout

    def bds_vote_image(self, a, b, ann, bnn, w_coh=1.0, w_comp=2.0, patch=3):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        ah, aw = a.shape[:2]
        bh, bw = b.shape[:2]
        out = np.empty((ah, aw, 3), np.uint8)
        self.l.orc_bds_vote_image(a, ah, aw, b, bh, bw, np.ascontiguousarray(ann, np.uint32), np.ascontiguousarray(bnn, np.uint32),
                                  patch, w_coh, w_comp, out)
        return out


_cached = None


def load(build=True):
    global _cached
    if _cached is None:
        if build:
            subprocess.run(["make", "-C", ORC_DIR, "-s"], check=True)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        l = C.CDLL(ORC_LIB)
        l.orc_set_threads(int(os.environ.get("ORC_THREADS", min(16, os.cpu_count() or 1))))
        _cached = Oracle(l)
    return _cached
