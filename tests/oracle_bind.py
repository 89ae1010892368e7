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
