"""nct — thin ctypes binding over libnct.so (the C ABI declared in include/nct.h).

This is plumbing for tests / bench.py only; the product is the shared library + the C++ CLI. There is no CPU
fallback anywhere in this package: if the library is missing or no HIP device is usable, calls raise NctError.
Function names and argument meaning mirror include/nct.h, which cites the reference seam each one replaces
(code/windows/neural_color_transfer/source/main.cu).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))
REPO_ROOT = os.path.normpath(os.path.join(PKG_ROOT, ".."))
LIB_PATH = os.environ.get("NCT_LIB") or os.path.join(PKG_ROOT, "lib", "libnct.so")      # NCT_LIB: kernel-tuning experiments only


class NctError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nct error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libnct.so (in-tree build only — never a site-packages copy)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NctError(-4, f"{LIB_PATH} not found — run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.nct_version() != NCT_VERSION:        # the structs below mirror include/nct.h at this version
            v = _lib.nct_version(); _lib = None
            raise NctError(-2, f"{LIB_PATH} is nct version {v}, this binding was written against {NCT_VERSION}: rebuild")
    return _lib


NCT_VERSION = 110        # include/nct.h
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")

# name -> (restype, argtypes); the single source of truth for the symbol-export test
SIGNATURES = {
    "nct_version": (C.c_int, []),
    "nct_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nct_destroy": (None, [C.c_void_p]),
    "nct_last_error": (C.c_char_p, [C.c_void_p]),
    "nct_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "nct_ctx_counter": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "nct_synchronize": (C.c_int, [C.c_void_p]),
    "nct_feat_normalize": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nct_nnf_init": (C.c_int, [C.c_void_p, _u32p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nct_nnf_upsample": (C.c_int, [C.c_void_p, _u32p, _u32p] + [C.c_int] * 6),
    "nct_patchmatch": (C.c_int, [C.c_void_p, _f32p, _f32p] + [C.c_int] * 8 + [C.c_uint32, _u32p, _f32p]),
    "nct_bds_vote_features": (C.c_int, [C.c_void_p, _u32p, _u32p, _f32p, _f32p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_float]),
    "nct_feature_distance": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]),
    "nct_bds_vote_image": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, _u32p, _u32p, C.c_int, C.c_double, C.c_double, _u8p]),
    "nct_vgg19_load_caffemodel": (C.c_int, [C.c_void_p, C.c_char_p]),
    "nct_model_parse_caffemodel": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "nct_model_free": (None, [C.c_void_p]),
    "nct_model_layer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nct_model_last_error": (C.c_char_p, []),
    "nct_vgg19_load_model": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nct_vgg19_share_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nct_vgg19_weights_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "nct_vgg19_check_prototxt": (C.c_int, [C.c_void_p, C.c_char_p]),
    "nct_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nct_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "nct_vgg19_load_raw": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]),
    "nct_vgg19_features": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "nct_conv3x3_relu": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, _f32p, C.c_int]),
    "nct_maxpool2x2": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "nct_params_default": (None, [C.c_void_p]),
    "nct_bgr2lab_u8": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p]),
    "nct_lab2bgr_u8": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p]),
    "nct_lab2bgr_u8_form": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_int]),
    "nct_resize_u8c3": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]),
    "nct_resize_f64c3": (C.c_int, [C.c_void_p, _f64p, C.c_int, C.c_int, _f64p, C.c_int, C.c_int]),
    "nct_cluster_features": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _i32p, C.POINTER(C.c_int)]),
    "nct_knn_graph": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _f64p]),
    "nct_local_color_transfer": (C.c_int, [C.c_void_p, _f32p, _u8p, _u8p, _u8p, _i32p, _f64p] + [C.c_int] * 5 + [C.c_void_p, _u8p, C.c_void_p]),
    "nct_process_pair": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_void_p, _u8p, C.c_void_p]),
    "nct_pair_upload": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]),
    "nct_pair_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nct_pair_run_levels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nct_pair_download": (C.c_int, [C.c_void_p, _u8p]),
    "nct_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "nct_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nct_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "nct_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "nct_chw_to_hwc_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nct_hwc_to_chw_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nct_vgg19_features_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "nct_vgg19_features_hwc_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "nct_feat_normalize_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nct_nnf_init_dev": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4),
    "nct_nnf_upsample_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6),
    "nct_patchmatch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_uint32, C.c_void_p, C.c_void_p]),
    "nct_patchmatch_bidir_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "nct_bds_vote_features_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_float]),
    "nct_bds_vote_image_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_double, C.c_double, C.c_void_p]),
    "nct_feature_distance_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nct_pm_bench_setup": (C.c_int, [C.c_void_p, _f32p, _f32p] + [C.c_int] * 5),
    "nct_pm_bench_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]),
    "nct_pm_bench_run_bidir": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def _declare(l):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Params(C.Structure):
    """struct nct_params (include/nct.h)."""
    _fields_ = [("bds_weight", C.c_double), ("eps", C.c_double), ("nonlocal_weight", C.c_double), ("local_weight", C.c_double),
                ("wls_lambda_init", C.c_double), ("cluster_num", C.c_int), ("k_num", C.c_int), ("patch_size", C.c_int),
                ("wls_alpha", C.c_double), ("pm_iters", C.c_int), ("seed", C.c_uint32), ("levels", C.c_int), ("flags", C.c_uint32)]

    @staticmethod
    def default():
        p = Params()
        lib().nct_params_default(C.byref(p))
        return p


class PairTiming(C.Structure):
    """struct nct_pair_timing (include/nct.h)."""
    _fields_ = [("total_ms", C.c_double), ("vgg_ms", C.c_double), ("cluster_ms", C.c_double), ("patchmatch_ms", C.c_double),
                ("vote_ms", C.c_double), ("knn_ms", C.c_double), ("color_ms", C.c_double), ("other_ms", C.c_double),
                ("nonlocal_ms", C.c_double), ("wls_ms", C.c_double), ("wls_iters", C.c_int * 5),
                ("pm_level_ms", C.c_double * 5), ("pm_level_launches", C.c_int * 5),
                ("vote_level_ms", C.c_double * 5), ("nonlocal_level_ms", C.c_double * 5), ("wls_level_ms", C.c_double * 5),
                ("pm_level_evals", C.c_ulonglong * 5), ("pm_level_accepted", C.c_ulonglong * 5),
                ("kernel_us", C.c_double * 10), ("kernel_samples", C.c_int * 10)]

    def as_dict(self):
        d = {}
        for k, t in self._fields_:
            v = getattr(self, k)
            d[k] = v if isinstance(v, (int, float)) else list(v)
        return d


CTR_ARENA_BYTES, CTR_S1_HUB_BLOCKS_L0 = 0, 1
FLAG_FEAT16 = 1
FLAG_COUNT_EVALS = 2
FLAG_LATENCY = 4
FLAG_LAB2BGR_CUBE = 8
FLAG_TIME_KERNELS = 16
KT_NAMES = ("s1_apply", "s1_scalars", "s1_update", "wls_down", "wls_up", "wls_apply", "wls_update", "wls_coarse", "wls_block_pre", "wls_block_post")      # nct.h NCT_KT_*
LAB2BGR_PIECEWISE, LAB2BGR_CUBE = 0, 1


class Model:
    """host-side parsed caffemodel (nct_model): parse the file once per process, upload it once per GPU (Context.vgg19_load_model)."""

    def __init__(self, path):
        self._m = None
        m = C.c_void_p()
        rc = lib().nct_model_parse_caffemodel(os.fsencode(path), C.byref(m))
        if rc != 0:
            raise NctError(rc, (lib().nct_model_last_error() or b"").decode())
        self._m = m

    def layer(self, i):
        """(weights [cout, cin, 3, 3], bias [cout]) of conv layer i (0 = conv1_1 … 12 = conv5_1) as numpy copies (nct_model_layer)"""
        w, b, co, ci = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_int(), C.c_int()
        rc = lib().nct_model_layer(self._m, i, C.byref(w), C.byref(b), C.byref(co), C.byref(ci))
        if rc != 0:
            raise NctError(rc, (lib().nct_model_last_error() or b"").decode())
        return (np.ctypeslib.as_array(w, (co.value, ci.value, 3, 3)).copy(), np.ctypeslib.as_array(b, (co.value,)).copy())

    def close(self):
        if self._m:
            lib().nct_model_free(self._m); self._m = None

    def __del__(self):
        self.close()


def check_prototxt(path):
    """nct_vgg19_check_prototxt without a context (no GPU needed): raises NctError with the reason when the file is not this library's VGG19"""
    rc = lib().nct_vgg19_check_prototxt(None, os.fsencode(path))
    if rc != 0:
        raise NctError(rc, (lib().nct_model_last_error() or b"").decode())


class PairLevels(C.Structure):
    """struct nct_pair_levels (include/nct.h)."""
    _fields_ = [(k, C.c_void_p * 5) for k in ("ann", "bnn", "annd", "bnnd", "guide", "err", "result", "color")] + [("labels", C.c_void_p)]


class ColorStages(C.Structure):
    """struct nct_color_stages (include/nct.h)."""
    _fields_ = [("ab_local", C.c_void_p), ("ab_nonlocal", C.c_void_p), ("ab_up", C.c_void_p), ("roughness", C.c_void_p),
                ("ab_wls", C.c_void_p), ("cg_iters", C.c_void_p), ("wls_iters", C.c_void_p)]


class Context:
    """One context per GPU / process (include/nct.h: not thread-safe)."""

    def __init__(self, device=0):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.nct_create(device, C.byref(h))
        if rc != 0:
            raise NctError(rc, self._l.nct_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.nct_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise NctError(rc, self._l.nct_last_error(self._h).decode())

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._chk(self._l.nct_device_name(self._h, buf, 256))
        return buf.value.decode()

    def counter(self, which):
        """nct_ctx_counter: CTR_ARENA_BYTES, CTR_S1_HUB_BLOCKS_L0 + level"""
        v = C.c_int64(0)
        self._chk(self._l.nct_ctx_counter(self._h, which, C.byref(v)))
        return v.value

    def synchronize(self):
        self._chk(self._l.nct_synchronize(self._h))

    # ---- N1
    def feat_normalize(self, src_chw, want_resp=False):
        src = np.ascontiguousarray(src_chw, np.float32)
        Cc, H, W = src.shape
        dst = np.empty_like(src)
        resp = np.empty((H, W), np.float32) if want_resp else None
        self._chk(self._l.nct_feat_normalize(self._h, src, dst, _ptr(resp), Cc, H, W))
        return (dst, resp) if want_resp else dst

    # ---- N2
    def nnf_init(self, ah, aw, bh, bw):
        nnf = np.empty((ah, aw), np.uint32)
        self._chk(self._l.nct_nnf_init(self._h, nnf, ah, aw, bh, bw))
        return nnf

    def nnf_upsample(self, nnf_half, ah, aw, bh, bw):
        half = np.ascontiguousarray(nnf_half, np.uint32)
        nnf = np.empty((ah, aw), np.uint32)
        self._chk(self._l.nct_nnf_upsample(self._h, half, nnf, ah, aw, bh, bw, half.shape[0], half.shape[1]))
        return nnf

    # ---- P1
    def patchmatch(self, a_chw, b_chw, nnf, iters=10, rs_max=32, seed=0, patch=3):
        a = np.ascontiguousarray(a_chw, np.float32)
        b = np.ascontiguousarray(b_chw, np.float32)
        Cc, ah, aw = a.shape
        _, bh, bw = b.shape
        nnf = np.array(nnf, np.uint32, order="C", copy=True).reshape(ah, aw)
        dist = np.empty((ah, aw), np.float32)
        self._chk(self._l.nct_patchmatch(self._h, a, b, Cc, ah, aw, bh, bw, patch, iters, rs_max, seed, nnf, dist))
        return nnf, dist

    # ---- B2
    def bds_vote_features(self, ann, bnn, pin_chw, w_coh=1.0, w_comp=2.0, patch=3, want_pw=False):
        pin = np.ascontiguousarray(pin_chw, np.float32)
        Cc, bh, bw = pin.shape
        ann = np.ascontiguousarray(ann, np.uint32)
        bnn = np.ascontiguousarray(bnn, np.uint32)
        ah, aw = ann.shape
        pout = np.empty((Cc, ah, aw), np.float32)
        pw = np.empty((ah, aw), np.float32) if want_pw else None
        self._chk(self._l.nct_bds_vote_features(self._h, ann, bnn, pin, pout, _ptr(pw), Cc, ah, aw, bh, bw, patch, w_coh, w_comp))
        return (pout, pw) if want_pw else pout

    def feature_distance(self, a_chw, b_chw):
        a = np.ascontiguousarray(a_chw, np.float32)
        b = np.ascontiguousarray(b_chw, np.float32)
        Cc, H, W = a.shape
        err = np.empty((H, W), np.float32)
        self._chk(self._l.nct_feature_distance(self._h, a, b, err, Cc, H, W))
        return err

    # ---- B1
    def bds_vote_image(self, a_bgr, b_bgr, ann, bnn, w_coh=1.0, w_comp=2.0, patch=3):
        a = np.ascontiguousarray(a_bgr, np.uint8)
        b = np.ascontiguousarray(b_bgr, np.uint8)
        ah, aw = a.shape[:2]
        bh, bw = b.shape[:2]
        out = np.empty((ah, aw, 3), np.uint8)
        self._chk(self._l.nct_bds_vote_image(self._h, a, ah, aw, b, bh, bw, np.ascontiguousarray(ann, np.uint32),
                                             np.ascontiguousarray(bnn, np.uint32), patch, w_coh, w_comp, out))
        return out

    # ---- V1 / V2
    VGG_CIN = [3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512]
    VGG_COUT = [64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512]
    VGG_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4",
                 "conv4_1", "conv4_2", "conv4_3", "conv4_4", "conv5_1", "conv5_2", "conv5_3", "conv5_4"]
    TAP_C = [64, 128, 256, 512, 512]

    def vgg19_load_caffemodel(self, path):
        self._chk(self._l.nct_vgg19_load_caffemodel(self._h, os.fsencode(path)))

    # ---- device-pointer seams (nct_dev.cpp): buffers are plain integers (device addresses); nothing synchronises until dev_download / synchronize
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self._l.nct_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, p):
        self._chk(self._l.nct_dev_free(self._h, p))

    def dev_upload(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.dev_alloc(a.nbytes)
        self._chk(self._l.nct_dev_upload(self._h, p, a.ctypes.data, a.nbytes))
        return p

    def dev_download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self._chk(self._l.nct_dev_download(self._h, out.ctypes.data, p, out.nbytes))
        return out

    def dev_call(self, name, *args):
        """nct_<name>_dev(ctx, *args)"""
        self._chk(getattr(self._l, "nct_" + name + "_dev")(self._h, *args))

    def vgg19_load_model(self, model):
        """model: a Model (nct_model_parse_caffemodel) — upload the host copy to this context's GPU"""
        self._chk(self._l.nct_vgg19_load_model(self._h, model._m))

    def vgg19_share_weights(self, other):
        """use `other`'s device copy of the weights (same GPU): no parse, no upload, no second copy"""
        self._chk(self._l.nct_vgg19_share_weights(self._h, other._h))

    def vgg19_weights_info(self):
        i, b, n = C.c_uint64(), C.c_size_t(), C.c_int()
        self._chk(self._l.nct_vgg19_weights_info(self._h, C.byref(i), C.byref(b), C.byref(n)))
        return {"id": i.value, "bytes": b.value, "sharers": n.value}

    def vgg19_check_prototxt(self, path):
        self._chk(self._l.nct_vgg19_check_prototxt(self._h, os.fsencode(path)))

    def vgg19_load_raw(self, weights, biases):
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, np.float32) for b in biases]
        n = len(ws)
        wp = (C.c_void_p * n)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        self._chk(self._l.nct_vgg19_load_raw(self._h, wp, bp, n))

    def vgg19_features(self, bgr, deepest_tap=5):
        """-> list of `deepest_tap` CHW fp32 arrays (tap 1 = conv1_1 … tap 5 = conv5_1)."""
        img = np.ascontiguousarray(bgr, np.uint8)
        h, w = img.shape[:2]
        outs, hh, ww = [], h, w
        for t in range(deepest_tap):
            outs.append(np.empty((self.TAP_C[t], hh, ww), np.float32))
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        ptrs = (C.c_void_p * 5)(*([o.ctypes.data for o in outs] + [None] * (5 - deepest_tap)))
        dims = np.zeros(15, np.int32)
        self._chk(self._l.nct_vgg19_features(self._h, img, h, w, w * 3, deepest_tap, ptrs, _ptr(dims)))
        for t, o in enumerate(outs):
            assert tuple(dims[3 * t:3 * t + 3]) == o.shape
        return outs

    def conv3x3_relu(self, x_chw, weights, bias, relu=True):
        x = np.ascontiguousarray(x_chw, np.float32)
        w = np.ascontiguousarray(weights, np.float32)
        b = np.ascontiguousarray(bias, np.float32)
        cin, H, W = x.shape
        cout = w.shape[0]
        out = np.empty((cout, H, W), np.float32)
        self._chk(self._l.nct_conv3x3_relu(self._h, x, cin, H, W, w, b, cout, out, 1 if relu else 0))
        return out

    def maxpool2x2(self, x_chw):
        x = np.ascontiguousarray(x_chw, np.float32)
        c, H, W = x.shape
        out = np.empty((c, (H - 1) // 2 + 1, (W - 1) // 2 + 1), np.float32)
        self._chk(self._l.nct_maxpool2x2(self._h, x, c, H, W, out))
        return out

    # ---- colour stage
    def bgr2lab(self, bgr):
        a = np.ascontiguousarray(bgr, np.uint8)
        out = np.empty_like(a)
        self._chk(self._l.nct_bgr2lab_u8(self._h, a.reshape(-1, 3), a.size // 3, out.reshape(-1, 3)))
        return out

    def lab2bgr(self, lab, form=None):
        a = np.ascontiguousarray(lab, np.uint8)
        out = np.empty_like(a)
        if form is None:
            self._chk(self._l.nct_lab2bgr_u8(self._h, a.reshape(-1, 3), a.size // 3, out.reshape(-1, 3)))
        else:
            self._chk(self._l.nct_lab2bgr_u8_form(self._h, a.reshape(-1, 3), a.size // 3, out.reshape(-1, 3), form))
        return out

    def resize_u8c3(self, img, dh, dw):
        a = np.ascontiguousarray(img, np.uint8)
        out = np.empty((dh, dw, 3), np.uint8)
        self._chk(self._l.nct_resize_u8c3(self._h, a, a.shape[0], a.shape[1], out, dh, dw))
        return out

    def resize_f64c3(self, img, dh, dw):
        a = np.ascontiguousarray(img, np.float64)
        out = np.empty((dh, dw, 3), np.float64)
        self._chk(self._l.nct_resize_f64c3(self._h, a, a.shape[0], a.shape[1], out, dh, dw))
        return out

    def cluster_features(self, feat_chw, K=10, iters=11, seed=1):
        f = np.ascontiguousarray(feat_chw, np.float32)
        Cc, h, w = f.shape
        labels = np.empty((h, w), np.int32)
        nl = C.c_int()
        self._chk(self._l.nct_cluster_features(self._h, f, Cc, h, w, K, iters, seed, labels.reshape(-1), C.byref(nl)))
        return labels, nl.value

    def knn_graph(self, lab_u8, labels, nlabels, samples, k=8):
        lab = np.ascontiguousarray(lab_u8, np.uint8)
        h, w = lab.shape[:2]
        lb = np.ascontiguousarray(labels, np.int32)
        ids = np.empty((h * w, k), np.int32)
        ws = np.empty((h * w, k), np.float64)
        self._chk(self._l.nct_knn_graph(self._h, lab, h, w, lb.reshape(-1), lb.shape[0], lb.shape[1], nlabels, samples, k, ids.reshape(-1), ws.reshape(-1)))
        return ids, ws

    def local_color_transfer(self, err, s_level, g_level, s_full, knn_id, knn_w, layer, params=None, want_stages=False):
        err = np.ascontiguousarray(err, np.float32)
        h, w = err.shape
        s_full = np.ascontiguousarray(s_full, np.uint8)
        H, W = s_full.shape[:2]
        prm = params or Params.default()
        out = np.empty((H, W, 3), np.uint8)
        st, keep = None, {}
        if want_stages:
            keep = {"ab_local": np.empty((2, h * w, 3)), "ab_nonlocal": np.empty((2, h * w, 3)), "ab_up": np.empty((2, H * W, 3)),
                    "roughness": np.empty(H * W), "ab_wls": np.empty((2, H * W, 3)), "cg_iters": np.zeros(3, np.int32), "wls_iters": np.zeros(6, np.int32)}
            st = ColorStages(*[keep[k].ctypes.data for k in ("ab_local", "ab_nonlocal", "ab_up", "roughness", "ab_wls", "cg_iters", "wls_iters")])
        self._chk(self._l.nct_local_color_transfer(self._h, err.reshape(-1), np.ascontiguousarray(s_level, np.uint8).reshape(-1, 3),
                                                   np.ascontiguousarray(g_level, np.uint8).reshape(-1, 3), s_full.reshape(-1, 3),
                                                   np.ascontiguousarray(knn_id, np.int32).reshape(-1), np.ascontiguousarray(knn_w, np.float64).reshape(-1),
                                                   layer, h, w, H, W, C.addressof(prm), out.reshape(-1, 3), C.addressof(st) if st else None))
        return (out, keep) if want_stages else out

    # ---- per-pair hot loop
    def process_pair(self, src_bgr, ref_bgr, params=None, want_timing=False):
        s = np.ascontiguousarray(src_bgr, np.uint8)
        r = np.ascontiguousarray(ref_bgr, np.uint8)
        prm = params or Params.default()
        out = np.empty_like(s)
        tm = PairTiming() if want_timing else None
        self._chk(self._l.nct_process_pair(self._h, s.reshape(-1, 3), s.shape[0], s.shape[1], r.reshape(-1, 3), r.shape[0], r.shape[1],
                                           C.addressof(prm), out.reshape(-1, 3), C.addressof(tm) if tm is not None else None))
        return (out, tm.as_dict()) if want_timing else out

    def pair_upload(self, src_bgr, ref_bgr):
        s = np.ascontiguousarray(src_bgr, np.uint8)
        r = np.ascontiguousarray(ref_bgr, np.uint8)
        self._pair_shape = s.shape
        self._chk(self._l.nct_pair_upload(self._h, s.reshape(-1, 3), s.shape[0], s.shape[1], r.reshape(-1, 3), r.shape[0], r.shape[1]))

    def pair_run(self, params=None, want_timing=False):
        prm = params or Params.default()
        tm = PairTiming() if want_timing else None
        self._chk(self._l.nct_pair_run(self._h, C.addressof(prm), C.addressof(tm) if tm is not None else None))
        return tm.as_dict() if want_timing else None

    def pair_run_levels(self, src_shape, ref_shape, params=None, want_color=False):
        """nct_pair_run_levels on the uploaded pair -> dict of per-level intermediates (lists indexed by level, 0 = coarsest).
        want_color adds "color" (per level the dict of coefficient maps local_color_transfer(want_stages=True) returns) and "labels"."""
        prm = params or Params.default()
        H, W = src_shape[:2]; RH, RW = ref_shape[:2]
        dims = []
        h, w, h2, w2 = H, W, RH, RW
        for _ in range(5):
            dims.insert(0, (h, w, h2, w2))
            h, w, h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1, (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
        keep = {"ann": [], "bnn": [], "annd": [], "bnnd": [], "guide": [], "err": [], "result": []}
        for (ah, aw, bh, bw) in dims:
            keep["ann"].append(np.zeros((ah, aw), np.uint32)); keep["bnn"].append(np.zeros((bh, bw), np.uint32))
            keep["annd"].append(np.zeros((ah, aw), np.float32)); keep["bnnd"].append(np.zeros((bh, bw), np.float32))
            keep["guide"].append(np.zeros((ah, aw, 3), np.uint8)); keep["err"].append(np.zeros((ah, aw), np.float32))
            keep["result"].append(np.zeros((H, W, 3), np.uint8))
        lv = PairLevels()
        for k in keep:
            setattr(lv, k, (C.c_void_p * 5)(*[a.ctypes.data for a in keep[k]]))
        if want_color:
            color, structs = [], []
            for (ah, aw, _, _) in dims:
                d = {"ab_local": np.empty((2, ah * aw, 3)), "ab_nonlocal": np.empty((2, ah * aw, 3)), "ab_up": np.empty((2, H * W, 3)),
                     "roughness": np.empty(H * W), "ab_wls": np.empty((2, H * W, 3)), "cg_iters": np.zeros(3, np.int32), "wls_iters": np.zeros(6, np.int32)}
                color.append(d)
                structs.append(ColorStages(*[d[k].ctypes.data for k in ("ab_local", "ab_nonlocal", "ab_up", "roughness", "ab_wls", "cg_iters", "wls_iters")]))
            lv.color = (C.c_void_p * 5)(*[C.addressof(st) if i < prm.levels else None for i, st in enumerate(structs)])
            labels = np.zeros(dims[0][:2], np.int32)
            lv.labels = labels.ctypes.data
        tm = PairTiming()
        self._chk(self._l.nct_pair_run_levels(self._h, C.addressof(prm), C.addressof(tm), C.addressof(lv)))
        if want_color:
            keep["color"] = color; keep["labels"] = labels
        keep["timing"] = tm.as_dict()
        keep["dims"] = dims
        return keep

    def pair_download(self):
        out = np.empty(self._pair_shape, np.uint8)
        self._chk(self._l.nct_pair_download(self._h, out.reshape(-1, 3)))
        return out

    # ---- measurement hooks
    def pm_bench_setup(self, a_chw, b_chw):
        a = np.ascontiguousarray(a_chw, np.float32)
        b = np.ascontiguousarray(b_chw, np.float32)
        Cc, ah, aw = a.shape
        _, bh, bw = b.shape
        self._pm_shape = (ah, aw)
        self._pm_shape_b = (bh, bw)
        self._chk(self._l.nct_pm_bench_setup(self._h, a, b, Cc, ah, aw, bh, bw))

    def pm_bench_run(self, iters=10, rs_max=32, seed=0, count_evals=False, fetch=False):
        ms = C.c_float()
        ev = C.c_uint64()
        ah, aw = self._pm_shape
        nnf = np.empty((ah, aw), np.uint32) if fetch else None
        dist = np.empty((ah, aw), np.float32) if fetch else None
        self._chk(self._l.nct_pm_bench_run(self._h, iters, rs_max, seed, C.byref(ms), C.byref(ev) if count_evals else None,
                                           _ptr(nnf), _ptr(dist)))
        return ms.value, (ev.value if count_evals else None), nnf, dist

    def pm_bench_run_bidir(self, iters=10, rs_max=32, seed=0, pm_mode=1, count=False, fetch=False, both=False):
        """The pipeline's form of the pass (both directions per launch; pm_mode 0 fp32 / 1 fp32 + row rejection / 2 fp16).
        -> (kernel ms, [evals, accepted] or None, ann, annd[, bnn, bnnd])."""
        ms = C.c_float()
        cnt = (C.c_uint64 * 2)()
        ah, aw = self._pm_shape
        bh, bw = self._pm_shape_b
        ann = np.empty((ah, aw), np.uint32) if fetch else None
        annd = np.empty((ah, aw), np.float32) if fetch else None
        bnn = np.empty((bh, bw), np.uint32) if fetch and both else None
        bnnd = np.empty((bh, bw), np.float32) if fetch and both else None
        self._chk(self._l.nct_pm_bench_run_bidir(self._h, iters, rs_max, seed, pm_mode, C.byref(ms), C.addressof(cnt) if count else None,
                                                 _ptr(ann), _ptr(annd), _ptr(bnn), _ptr(bnnd)))
        r = (ms.value, (list(cnt) if count else None), ann, annd)
        return r + (bnn, bnnd) if both else r
