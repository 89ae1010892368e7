"""CPU study behind the PCA-sketch pre-rejection of PatchMatch's random samples (round 6; DESIGN / HISTORY 9): for a partly converged field of the oracle on real VGG taps,
which share of the random samples of radius 32 ... 1 is rejected (a) by the shipped first-row Cauchy-Schwarz test, (b) by the exact sketch bound
sum_taps (P a).(P b) + |res_a| |res_b| < need with the K dominant principal directions of the pooled features (or of every 4th pixel of the reference map only).
usage: python scripts/pm_sketch_probe.py <image side> <tap 1|2> [synth | 0..4 = demo photograph pair]"""
import sys, numpy as np
sys.path.insert(0, "tests")
import synth, oracle_bind
from caffemodel_io import synthetic_vgg19
orc = oracle_bind.load()
N = int(sys.argv[1]); tap = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "synth"
ws, bs = synthetic_vgg19(19)
if kind == "synth":
    S, R = synth.image(1000, N, N), synth.image(1001, N, N)
else:
    import natural_inputs
    from PIL import Image
    natural_inputs.stage(); d = natural_inputs.DIR
    S = np.array(Image.open(f"{d}/in{kind}.png").convert("RGB"))[:, :, ::-1]; R = np.array(Image.open(f"{d}/tar{kind}.png").convert("RGB"))[:, :, ::-1]
    S = np.ascontiguousarray(S[:N, :N]); R = np.ascontiguousarray(R[:N, :N])
fa = orc.vgg19_features(S, ws, bs, deepest_tap=tap)[tap - 1]
fb = orc.vgg19_features(R, ws, bs, deepest_tap=tap)[tap - 1]
a = orc.feat_normalize(fa); b = orc.feat_normalize(fb)
C, H, W = a.shape
print("features", a.shape)
nnf0 = orc.nnf_init(H, W, H, W)
nnf, dist = orc.patchmatch(a, b, nnf0, iters=4, rs_max=32, seed=0)
A = a.transpose(1, 2, 0).astype(np.float64); B = b.transpose(1, 2, 0).astype(np.float64)
def basis(X):
    cov = X.T @ X / len(X)
    ev, U = np.linalg.eigh(cov); o = np.argsort(-ev); return ev[o], U[:, o]
evp, Up = basis(np.concatenate([A.reshape(-1, C), B.reshape(-1, C)]))
evb, Ub = basis(B.reshape(-1, C)[::4])
print("energy pooled:", {k: round(float(evp[:k].sum() / evp.sum()), 4) for k in (4, 7, 8, 11, 15, 16, 31)})
rng = np.random.default_rng(5)
nq = 20000
qy = rng.integers(1, H - 1, nq); qx = rng.integers(1, W - 1, nq)
my = (nnf[qy, qx] >> 12).astype(int); mx = (nnf[qy, qx] & 4095).astype(int)
need = -9.0 * dist[qy, qx].astype(np.float64)
offs = [(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
for name, U in (("pooled", Up), ("B/4 only", Ub)):
  for K in (7, 8, 11, 15):
    P = U[:, :K]
    YA = A @ P; YB = B @ P
    RA = np.linalg.norm(A - YA @ P.T, axis=-1); RB = np.linalg.norm(B - YB @ P.T, axis=-1)
    out = []
    for rad in (32, 16, 8, 4, 2, 1):
        r2 = np.random.default_rng(rad)
        cy = np.clip(my + r2.integers(-rad, rad + 1, nq), 1, H - 2); cx = np.clip(mx + r2.integers(-rad, rad + 1, nq), 1, W - 2)
        Strue = np.zeros(nq); Srow = np.zeros(nq); Sk = np.zeros(nq)
        for (dy, dx) in offs:
            dots = (A[qy + dy, qx + dx] * B[cy + dy, cx + dx]).sum(-1)
            Strue += dots
            if dy == -1: Srow += dots
            Sk += (YA[qy + dy, qx + dx] * YB[cy + dy, cx + dx]).sum(-1) + RA[qy + dy, qx + dx] * RB[cy + dy, cx + dx]
        assert (Sk + 1e-9 >= Strue).all()
        out.append(f"r{rad}: row {(Srow + 6.0007 < need).mean():.2f} sk {(Sk + 2e-3 < need).mean():.3f}")
    print(f"  {name:9s} K={K:2d}  " + " | ".join(out))
