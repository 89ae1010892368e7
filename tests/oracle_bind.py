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
        _cached = Oracle(C.CDLL(ORC_LIB))
    return _cached
