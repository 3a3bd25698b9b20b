"""Seeded synthetic frame stacks with known ground-truth homographies.

The reference's only data set (image_stacking_py, README.md:18) is fetched from the network by
the user and is not available here, so tests and bench.py use this generator instead: a
corner-rich base scene rendered once, each frame i > 0 re-sampled through a small random
homography G_i (frame-i pixel -> frame-0 pixel), plus gain and sensor noise, quantised to
u8 (or u16). Scene and motion parameters come from numpy PCG64(seed); the heavy resampling
runs in torch so the same code renders 4K stacks on the GPU in bench.py.
"""
from __future__ import annotations

import math

import numpy as np

SEED = 20251114
MARGIN = 64


def _gauss_blur_np(img: np.ndarray, sigma: float) -> np.ndarray:
    from scipy.ndimage import gaussian_filter
    if img.ndim == 3:
        return np.stack([gaussian_filter(img[..., c], sigma, mode="reflect") for c in range(img.shape[2])], -1)
    return gaussian_filter(img, sigma, mode="reflect")


def render_scene(width: int, height: int, seed: int = SEED) -> np.ndarray:
    """Base scene S, float32 BGR in [0,255], size (height+2M, width+2M)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W, H = width + 2 * MARGIN, height + 2 * MARGIN
    sc = max(width / 1920.0, 0.2)
    img = np.full((H, W, 3), 96.0, np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    # (c) smooth vignette / illumination gradient
    r2 = ((xx - W / 2) / W) ** 2 + ((yy - H / 2) / H) ** 2
    img *= (1.15 - 0.6 * r2)[..., None]
    # (a) 400 rotated rectangles + 200 discs
    for k in range(600):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        col = rng.uniform(20, 235, 3).astype(np.float32)
        if k < 400:
            hw, hh = rng.uniform(8, 70, 2) * sc
            th = rng.uniform(0, math.pi)
            rad = int(math.hypot(hw, hh)) + 2
        else:
            rad_f = rng.uniform(6, 45) * sc
            rad = int(rad_f) + 2
        x0, x1 = max(int(cx) - rad, 0), min(int(cx) + rad + 1, W)
        y0, y1 = max(int(cy) - rad, 0), min(int(cy) + rad + 1, H)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        if k < 400:
            c, s = math.cos(th), math.sin(th)
            m = (np.abs(dx * c + dy * s) <= hw) & (np.abs(-dx * s + dy * c) <= hh)
        else:
            m = dx * dx + dy * dy <= rad_f * rad_f
        img[y0:y1, x0:x1][m] = col
    # (b) band-limited texture
    tex = rng.standard_normal((H, W, 3)).astype(np.float32)
    tex = _gauss_blur_np(tex, 3.0)
    tex *= 6.0 / max(float(tex.std()), 1e-6)
    img += tex
    img = _gauss_blur_np(img, 1.0)
    return np.clip(img, 0, 255).astype(np.float32)


def random_homography(rng: np.random.Generator, width: int, height: int, strength: float = 1.0) -> np.ndarray:
    """G = C . T(tx,ty) . R(theta) . Sc(s) . P(p1,p2) . C^-1 about the image centre (3x3 f64)."""
    k = width / 1920.0
    tx, ty = rng.uniform(-8, 8, 2) * k * strength
    th = math.radians(rng.uniform(-0.5, 0.5)) * strength
    s = 1.0 + rng.uniform(-0.005, 0.005) * strength
    p1, p2 = rng.uniform(-2e-6, 2e-6, 2) / k * strength
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    C = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1]], np.float64)
    Ci = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], np.float64)
    T = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float64)
    R = np.array([[math.cos(th), -math.sin(th), 0], [math.sin(th), math.cos(th), 0], [0, 0, 1]], np.float64)
    S = np.diag([s, s, 1.0])
    P = np.array([[1, 0, 0], [0, 1, 0], [p1, p2, 1]], np.float64)
    G = C @ T @ R @ S @ P @ Ci
    return G / G[2, 2]


def make_stack(n: int, width: int, height: int, *, seed: int = SEED, depth: int = 8, device="cpu",
               noise_sigma: float = 2.0, strength: float = 1.0, scene: np.ndarray | None = None,
               indices=None):
    """Returns (frames, G): frames = torch uint8/int16-as-uint16 tensor [n,H,W,3] on `device`
    (BGR interleaved, the layout imread(UNCHANGED) yields, utils.rs:132), G = [n,3,3] float64
    with frame_i(x) ~= frame_0(G_i x). `indices` selects global frame numbers (default 0..n-1), so a
    rank can render just its slice of a larger stack; frame number 0 is always the unwarped scene."""
    import torch
    import torch.nn.functional as F

    dev = torch.device(device)
    if scene is None:
        scene = render_scene(width, height, seed)
    Hs, Ws, _ = scene.shape
    fdt = torch.float64 if dev.type == "cpu" else torch.float32
    S = torch.from_numpy(scene).to(dev, fdt).permute(2, 0, 1)[None]          # [1,3,Hs,Ws]
    ys, xs = torch.meshgrid(torch.arange(height, device=dev, dtype=fdt),
                            torch.arange(width, device=dev, dtype=fdt), indexing="ij")
    Gs = np.zeros((n, 3, 3), np.float64)
    frames = []
    gen = torch.Generator(device=dev)
    indices = list(range(n)) if indices is None else list(indices)
    n = len(indices)
    Gs = np.zeros((n, 3, 3), np.float64)
    for k, i in enumerate(indices):
        rng = np.random.Generator(np.random.PCG64(seed + 1 + i))
        G = np.eye(3) if i == 0 else random_homography(rng, width, height, strength)
        gain = 1.0 if i == 0 else rng.uniform(0.97, 1.03)
        Gs[k] = G
        g = torch.from_numpy(G).to(dev, fdt)
        den = g[2, 0] * xs + g[2, 1] * ys + g[2, 2]
        u = (g[0, 0] * xs + g[0, 1] * ys + g[0, 2]) / den + MARGIN
        v = (g[1, 0] * xs + g[1, 1] * ys + g[1, 2]) / den + MARGIN
        grid = torch.stack([2 * u / (Ws - 1) - 1, 2 * v / (Hs - 1) - 1], -1)[None]
        f = F.grid_sample(S, grid, mode="bicubic", padding_mode="border", align_corners=True)[0]
        f = f.permute(1, 2, 0) * gain
        if noise_sigma > 0:
            gen.manual_seed(seed + 1000 + i)
            f = f + torch.randn(f.shape, generator=gen, device=dev, dtype=fdt) * noise_sigma
        if depth == 8:
            q = torch.clamp(torch.round(f), 0, 255).to(torch.uint8)
        elif depth == 16:
            q = torch.clamp(torch.round(f * 257.0), 0, 65535).to(torch.int32).to(torch.uint16)
        else:
            raise ValueError("depth must be 8 or 16")
        frames.append(q.contiguous())
    return torch.stack(frames), Gs


def corner_error(A: np.ndarray, B: np.ndarray, width: int, height: int) -> float:
    """Max displacement (pixels) between the images of the four frame corners under A and B."""
    pts = np.array([[0, 0, 1], [width - 1, 0, 1], [0, height - 1, 1], [width - 1, height - 1, 1]], np.float64).T
    a = np.asarray(A, np.float64).reshape(3, 3) @ pts
    b = np.asarray(B, np.float64).reshape(3, 3) @ pts
    a = a[:2] / a[2]
    b = b[:2] / b[2]
    return float(np.sqrt(((a - b) ** 2).sum(0)).max())
