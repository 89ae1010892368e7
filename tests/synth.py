"""Seeded synthetic inputs shared by tests, smoke() and bench.py (SURVEY.md §8d 'Synthetic inputs')."""
import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64_stream(seed, n):
    """SplitMix64 -> n uint64 (identical in C++ and Python)."""
    out = np.empty(n, np.uint64)
    x = seed & MASK64
    for i in range(n):
        x = (x + 0x9E3779B97F4A7C15) & MASK64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        out[i] = z ^ (z >> 31)
    return out


def features(seed, C, H, W, smooth=True):
    """Non-negative (post-ReLU-like) CHW fp32 features with spatial structure, no all-zero pixels."""
    rng = np.random.default_rng(seed)
    f = rng.random((C, H, W), dtype=np.float32)
    if smooth:
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        for c in range(0, C, max(1, C // 16)):
            ph = rng.random(3)
            f[c] += 0.5 + 0.5 * np.cos(2 * np.pi * (xx / (8 + 40 * ph[0]) + yy / (8 + 40 * ph[1]) + ph[2])).astype(np.float32)
    return np.ascontiguousarray(f + np.float32(0.01))


def image(seed, H, W):
    """u8 BGR HWC: sum of low-frequency cosines + noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((H, W, 3))
    for c in range(3):
        acc = np.zeros((H, W))
        for _ in range(6):
            wl = 32 + rng.random() * 224
            th = rng.random() * 2 * np.pi
            ph = rng.random() * 2 * np.pi
            acc += np.cos(2 * np.pi * (xx * np.cos(th) + yy * np.sin(th)) / wl + ph)
        acc = (acc - acc.min()) / max(acc.max() - acc.min(), 1e-9)
        img[..., c] = 32 + acc * 192 + rng.integers(-8, 9, size=(H, W))
    return np.clip(img, 0, 255).astype(np.uint8)


def image_flat(seed, H, W):
    """image() plus what natural photographs have and cosines + noise do not: regions of ONE colour — a flat rectangle over a quarter of the image and a posterised
    band along the bottom (the reference's demo inputs hold groups of 10^4 identical pixels). Such groups make kNN hubs (the tie rule (distance, id) gives the group's
    lowest ids an in-degree of the group size), many-source targets of the BDS completeness vote, and stiff WLS systems."""
    img = image(seed, H, W).copy()
    rng = np.random.default_rng(seed + 77)
    y0, x0 = H // 8, W // 6
    img[y0:y0 + H // 2, x0:x0 + W // 2] = rng.integers(30, 220, 3)
    band = img[H - H // 4:, :]
    band[:] = (band // 64) * 64 + 20
    return img


def random_nnf(seed, ah, aw, bh, bw):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, bw, size=(ah, aw)).astype(np.uint32)
    y = rng.integers(0, bh, size=(ah, aw)).astype(np.uint32)
    return (y << 12) | x
