#!/usr/bin/env python3
"""Generate tests/golden/wls_pardiso.npz: the reference's actual WLS solver — Intel MKL PARDISO, mtype 2 (real
symmetric), upper-triangular 1-based CSR, the iparm settings of SparseSolver_CPU.cpp:135-160, phases 11/22/33/-1 —
applied to the system assembled by oracle/orc_color.c::orc_wls_system (ColorTransfer.cpp:996-1070).
Runs ONLY in the build container (needs /opt/conda/lib/libmkl_rt.so); the .npz it writes is the committed fixture."""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_bind  # noqa: E402
import synth  # noqa: E402


def pardiso_solve(n, ia, ja, a, rhs_list):
    mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so")
    pt = (C.c_void_p * 64)()
    iparm = (C.c_int * 64)()
    iparm[0] = 1; iparm[1] = 3; iparm[3] = 22; iparm[7] = 1; iparm[9] = 8; iparm[17] = -1; iparm[18] = -1; iparm[20] = 1
    maxfct, mnum, mtype, nrhs, msglvl, err = C.c_int(1), C.c_int(1), C.c_int(2), C.c_int(1), C.c_int(0), C.c_int(0)
    nn = C.c_int(n)
    idum = C.c_int(0)
    ddum = C.c_double(0)
    ap = a.ctypes.data_as(C.c_void_p); iap = ia.ctypes.data_as(C.c_void_p); jap = ja.ctypes.data_as(C.c_void_p)

    def call(phase, b=None, x=None):
        ph = C.c_int(phase)
        mkl.pardiso(pt, C.byref(maxfct), C.byref(mnum), C.byref(mtype), C.byref(ph), C.byref(nn), ap, iap, jap, C.byref(idum), C.byref(nrhs),
                    iparm, C.byref(msglvl), b.ctypes.data_as(C.c_void_p) if b is not None else C.byref(ddum),
                    x.ctypes.data_as(C.c_void_p) if x is not None else C.byref(ddum), C.byref(err))
        assert err.value == 0, f"pardiso phase {phase} error {err.value}"

    call(11); call(22)
    iparm[7] = 2
    outs = []
    for b in rhs_list:
        x = np.zeros(n)
        call(33, np.ascontiguousarray(b), x)
        outs.append(x)
    call(-1)
    return outs


def main():
    orc = oracle_bind.load()
    orc._decl_color()
    H, W, lamda, alpha = 20, 23, 1.536, 1.2
    rng = np.random.default_rng(7)
    lab = np.ascontiguousarray(orc.bgr2lab(synth.image(21, H, W)).astype(np.float64) / 255.0)
    rough = np.where(rng.random(H * W) < 0.15, 1e-6, 1.0)
    a0 = rng.random((H * W, 3)) * 1.5
    b0 = rng.random((H * W, 3)) - 0.5
    n = H * W
    diag = np.empty(n); wx = np.empty(n); wy = np.empty(n)
    orc.l.orc_wls_system(lab.reshape(-1), H, W, lamda, alpha, rough, diag, wx, wy)
    # upper-triangular one-based CSR exactly as ColorTransfer.cpp:1048-1075 lays it out
    vals, cols, rowp = [], [], [1]
    for i in range(n):
        vals.append(diag[i]); cols.append(i + 1)
        if i % W + 1 < W:
            vals.append(-wx[i]); cols.append(i + 2)
        if i + W < n:
            vals.append(-wy[i]); cols.append(i + W + 1)
        rowp.append(len(vals) + 1)
    a = np.array(vals); ja = np.array(cols, np.int32); ia = np.array(rowp, np.int32)
    sol = pardiso_solve(n, ia, ja, a, [rough * a0[:, c] for c in range(3)] + [rough * b0[:, c] for c in range(3)])
    np.savez_compressed(os.path.join(HERE, "wls_pardiso.npz"), H=H, W=W, lamda=lamda, alpha=alpha, lab=lab, rough=rough, a0=a0, b0=b0,
                        a_pardiso=np.stack(sol[:3], 1), b_pardiso=np.stack(sol[3:], 1))
    print("wrote wls_pardiso.npz")


if __name__ == "__main__":
    main()
