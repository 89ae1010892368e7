#!/usr/bin/env python3
"""Generates tests/golden/natural/pair_<case>.npz: the CPU oracle on the REFERENCE'S OWN DEMO INPUTS (demo/example/in/*.png, staged by tests/natural_inputs.py, never committed, into
tests/golden/natural/ — natural photographs with 10^4-pixel groups of one colour, unlike tests/synth.py's cosines + noise), with the synthetic VGG19 (the
Oxford weights do not exist here), for the lines of demo/example/pairs.txt:1-9 named in CASES. Same record as gen_pair700_exact.py: the canonical-order oracle
result (the one the GPU must reproduce byte for byte) as CRC-32 per pyramid level and of the final image, and the run with the EXACT S2 solve (the reference's
direct-solve semantics, SparseSolver_CPU.cpp:104-286) as a sparse delta on it, plus per-level CRCs of the NNFs / guidance / matching error of the canonical run
so that a divergence names its stage. Also stores the kNN in-degree statistics of every level (max / p99 / >64 / >512), the quantity the S1 kernels are sized on.

PNG decode = PIL, alpha dropped (cv::imread's default flag does the same: main.cu:483,491). Runs where tests/golden/natural/*.png exist (python tests/natural_inputs.py stages them);
~10-40 min per case on 8 cores."""
import os, sys, time, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, oracle_bind
from PIL import Image
from caffemodel_io import synthetic_vgg19
orc = oracle_bind.load()
orc.l.orc_set_threads(min(32, os.cpu_count() or 1))
ws, bs = synthetic_vgg19(19)
CASES = {"in1_tar1_2": ("in1", "tar1", 2.0), "in4_tar4_2": ("in4", "tar4", 2.0), "in4_tar4_0": ("in4", "tar4", 0.0), "in4_tar4_8": ("in4", "tar4", 8.0),
         "in0_tar0_2": ("in0", "tar0", 2.0),
         # round 6: the remaining lines of demo/example/pairs.txt (3, 4, 6, 8) — all nine demo pairs are fixtures now
         "in2_tar2_2": ("in2", "tar2", 2.0), "in3_tar3_2": ("in3", "tar3", 2.0), "in4_tar4_1": ("in4", "tar4", 1.0), "in4_tar4_4": ("in4", "tar4", 4.0)}


def load_bgr(name):
    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "natural", name + ".png")).convert("RGB"))[..., ::-1])


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


if __name__ == "__main__":
    exact_too = os.environ.get("NCT_GEN_EXACT", "1") != "0"
    for name in (sys.argv[1:] or list(CASES)):
        s, r, bds = CASES[name]
        src, ref = load_bgr(s), load_bgr(r)
        prm = {"bds_weight": bds}
        t = time.time(); canon, canon_lv, keep = orc.process_pair(src, ref, ws, bs, params=prm, want_levels=True, s2_exact=False, want_nnf=True); t_c = time.time() - t
        rec = dict(shape=np.array(src.shape[:2] + ref.shape[:2]), bds=np.float64(bds), crc_src=np.uint32(crc(src)), crc_ref=np.uint32(crc(ref)),
                   crc_canonical=np.uint32(crc(canon)), sum_canonical=np.uint64(canon.astype(np.uint64).sum()),
                   level_crc_canonical=np.array([crc(canon_lv[l]) for l in range(5)], np.uint32))
        for k in ("ann", "bnn", "annd", "bnnd", "guide", "err"):
            rec["level_crc_" + k] = np.array([crc(keep[k][l]) for l in range(5)], np.uint32)
        secs = [t_c]
        if exact_too:
            # NCT_GEN_REUSE_EXACT=1 (a change of the canonical S2 order only; the exact-solve run does not depend on it): keep the stored exact run when the new canonical run has its CRCs
            fo = os.path.join(HERE, "natural", f"pair_{name}.npz")
            old = np.load(fo) if os.environ.get("NCT_GEN_REUSE_EXACT") == "1" and os.path.exists(fo) else None
            if old is not None and "crc_exact" in old and int(old["crc_exact"]) == crc(canon) and [int(v) for v in old["level_crc_exact"]] == [crc(canon_lv[l]) for l in range(5)]:
                exact, exact_lv = canon, canon_lv; secs.append(float(old["seconds"][1]))
            else:
                t = time.time(); exact, exact_lv = orc.process_pair(src, ref, ws, bs, params=prm, want_levels=True, s2_exact=True); secs.append(time.time() - t)
                if old is not None and "crc_exact" in old: assert int(old["crc_exact"]) == crc(exact), "the exact-solve run changed"
            d = exact.astype(np.int16).reshape(-1) - canon.astype(np.int16).reshape(-1)
            idx = np.flatnonzero(d).astype(np.uint32)
            rec.update(idx=idx, delta=d[idx].astype(np.int16), crc_exact=np.uint32(crc(exact)),
                       level_crc_exact=np.array([crc(exact_lv[l]) for l in range(5)], np.uint32),
                       level_linf_exact_vs_canonical=np.array([int(np.abs(exact_lv[l].astype(int) - canon_lv[l].astype(int)).max()) for l in range(5)]),
                       level_ndiff_exact_vs_canonical=np.array([int((exact_lv[l] != canon_lv[l]).sum()) for l in range(5)]))
        rec["seconds"] = np.array(secs)
        np.savez_compressed(os.path.join(HERE, "natural", f"pair_{name}.npz"), **rec)
        print(name, src.shape, ref.shape, "bds", bds, "crc", int(rec["crc_canonical"]), "seconds", [round(x, 1) for x in secs],
              ("exact-vs-canonical differing bytes %d, per-level ndiff %s" % (rec["idx"].size, rec["level_ndiff_exact_vs_canonical"].tolist())) if exact_too else "", flush=True)
