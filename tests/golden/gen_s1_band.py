#!/usr/bin/env python3
"""Generates tests/golden/s1_band.json — VERDICT r5 item 3c: by how much the two legal forms of the truncated nonlocal solve S1 differ END TO END.

S1 (ColorTransfer.cpp:548-911 -> SparseSolver_GPU.cu:132-159) runs a fixed 50 / 100 CG iterations per level on a system it does not converge on, and the iterate is
chaotic in its rounding (tests/test_oracle_color.py::test_truncated_cg_is_chaotic): two exact-arithmetic-equivalent recurrences give visibly different colours.
  form 0 = the canonical one (oracle/orc_color_canon.c; matrix-free operator, fixed summation trees, Chronopoulos-Gear single-reduction recurrence) — what the GPU
           reproduces bit for bit;
  form 1 = the literal one (orc_nonlocal_solve_explicit: A assembled, A^T(A p) as two sparse products, the textbook recurrence of SparseSolver_GPU.cu:132-159,
           sequential dot products) — the closest this container gets to the reference's cuSPARSE / cuBLAS arithmetic, whose own summation order is unspecified.
Everything else (VGG, PatchMatch, votes, kNN, S2, Lab conversions) is the same code in both runs, so the numbers below are S1's band alone: the distance a bit-faithful
port of the reference on another GPU / cuSPARSE version would also show. The fixture holds statistics only (PSNR per level result and of the final image, L-inf, share of
bytes that differ). CPU only, needs nothing from /root/reference.

    python tests/golden/gen_s1_band.py [--big]        (256x256 pair: ~2 min on 8 cores; --big adds the 700x700 bench pair: ~25 min)
"""
import json, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_bind, synth  # noqa: E402
from caffemodel_io import synthetic_vgg19  # noqa: E402


def cmp(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    per = [10 * np.log10(255.0 ** 2 / max(np.mean(d[..., c] ** 2), 1e-12)) for c in range(3)]
    return {"psnr_min_channel_db": round(float(min(per)), 2), "linf": int(np.abs(d).max()), "bytes_differing": round(float(np.mean(d != 0)), 4),
            "mean_abs": round(float(np.abs(d).mean()), 4)}


def main():
    orc = oracle_bind.load()
    ws, bs = synthetic_vgg19(19)
    path = os.path.join(HERE, "s1_band.json")
    out = json.load(open(path)) if os.path.exists(path) else {"generator": "tests/golden/gen_s1_band.py", "cases": {}}
    cases = [("pair256", 256, 256, 256, 256, 1000, 1001)]
    if "--big" in sys.argv:
        cases.append(("pair700", 700, 700, 700, 700, 1, 2))          # bench.py's pair
    for name, h, w, rh, rw, s1, s2 in cases:
        if name in out["cases"] and "--all" not in sys.argv:
            continue
        src, ref = synth.image(s1, h, w), synth.image(s2, rh, rw)
        res = {}
        for form in (0, 1):
            orc.set_s1_form(form)
            t = time.time()
            res[form] = orc.process_pair(src, ref, ws, bs, want_levels=True)
            print(name, "form", form, "%.1f s" % (time.time() - t), flush=True)
        orc.set_s1_form(0)
        (o0, l0), (o1, l1) = res[0], res[1]
        case = {"size": [h, w, rh, rw], "image_seeds": [s1, s2], "final": cmp(o0, o1), "levels_coarse_to_fine": [cmp(l0[i], l1[i]) for i in range(5)],
                "colour_change_of_the_transfer": cmp(o0, src)}          # for scale: how far the transfer moves the source at all
        out["cases"][name] = case
        print(json.dumps(case, indent=1))
        json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
