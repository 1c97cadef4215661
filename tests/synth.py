"""Synthetic inputs for the CoLoC hot path; all seeds fixed.

Shared by tests/, bench.py and __graft_entry__.smoke().  Pure numpy, no GPU, no oracle.
"""
import math

import numpy as np

KP_DTYPE = np.dtype(
    {"names": ["x", "y", "score", "angle", "scale"],
     "formats": ["<i4", "<i4", "u1", "<f4", "u1"],
     "offsets": [0, 4, 8, 12, 16], "itemsize": 20})  # Keypoint.h:155-163 (sizeof == 20)


def rect_image(w, h, n_rect=None, seed=1000, noise_sigma=0.0):
    """u8 image: random axis-aligned rectangles of random grey level over black (+ optional noise)."""
    rng = np.random.default_rng(seed)
    if n_rect is None:
        n_rect = 400 if w <= 640 else 1200
    img = np.zeros((h, w), dtype=np.float32)
    for _ in range(n_rect):
        x0 = int(rng.integers(0, w - 8)); y0 = int(rng.integers(0, h - 8))
        rw = int(rng.integers(6, max(7, w // 6))); rh = int(rng.integers(6, max(7, h // 6)))
        img[y0:y0 + rh, x0:x0 + rw] = float(rng.integers(16, 256))
    if noise_sigma > 0:
        img += rng.normal(0.0, noise_sigma, size=img.shape).astype(np.float32)
    return np.clip(img + 0.5, 0, 255).astype(np.uint8)


def pyramid_dims(W, H, scale_factor=1.2, levels=8):
    """GPUDetector.hpp:109-114 in fp32: f_i = f_{i-1}*sf; w_i = (uint32)((float)W / f_i + 0.5f)."""
    f = np.float32(1.0)
    sf = np.float32(scale_factor)
    ws, hs, fs = [W], [H], [np.float32(1.0)]
    for _ in range(1, levels):
        f = np.float32(f * sf)
        ws.append(int(np.float32(np.float32(W) / f + np.float32(0.5))))
        hs.append(int(np.float32(np.float32(H) / f + np.float32(0.5))))
        fs.append(f)
    return ws, hs, fs


def random_keypoints(n, W, H, seed=2000, levels=8, scale_factor=1.2, border=3):
    """n keypoints: scale ~ weighted by level area, level-local x,y ints, angle ~ U(-pi, pi] fp32."""
    rng = np.random.default_rng(seed)
    ws, hs, _ = pyramid_dims(W, H, scale_factor, levels)
    area = np.array([w * h for w, h in zip(ws, hs)], dtype=np.float64)
    scale = rng.choice(levels, size=n, p=area / area.sum()).astype(np.uint8)
    kps = np.zeros(n, dtype=KP_DTYPE)
    wv = np.array(ws)[scale]; hv = np.array(hs)[scale]
    kps["x"] = (border + rng.random(n) * (wv - 2 * border - 1)).astype(np.int32)
    kps["y"] = (border + rng.random(n) * (hv - 2 * border - 1)).astype(np.int32)
    ang = (rng.random(n) * 2.0 - 1.0) * math.pi
    kps["angle"] = ang.astype(np.float32)
    kps["scale"] = scale
    kps["score"] = 0
    return kps


def random_descriptors(n, seed=3000):
    """iid uniform 512-bit descriptors, n x 64 u8."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, 64), dtype=np.uint8)


def planted_descriptors(nq, nt, seed=3000, frac=0.3, max_flip=60):
    """(Q, T): T iid uniform; `frac` of the queries are near-duplicates of random train rows with
    k ~ U{0..max_flip} flipped bits, the rest iid uniform -- exercises accept and reject branches."""
    rng = np.random.default_rng(seed)
    T = rng.integers(0, 256, size=(nt, 64), dtype=np.uint8)
    Q = rng.integers(0, 256, size=(nq, 64), dtype=np.uint8)
    if nt == 0:
        return Q, T
    n_plant = int(frac * nq)
    rows = rng.choice(nq, size=n_plant, replace=False)
    src = rng.integers(0, nt, size=n_plant)
    for r, s in zip(rows, src):
        bits = np.unpackbits(T[s])
        k = int(rng.integers(0, max_flip + 1))
        flip = rng.choice(512, size=k, replace=False)
        bits[flip] ^= 1
        Q[r] = np.packbits(bits)
    return Q, T


def pnp_scene(n_pts, seed=4000, cam=0, outlier_frac=0.3, noise_sigma=0.5, W=1280, H=720):
    """3-D points ~ U([-5,5]^2 x [4,20]) seen by camera `cam` on a 2 m arc looking at the scene.
    Returns dict(X (N,3), x (N,2), K (3,3), R (3,3), t (3,), inlier mask)."""
    rng = np.random.default_rng(seed + cam)
    X = np.stack([rng.uniform(-5, 5, n_pts), rng.uniform(-5, 5, n_pts), rng.uniform(4, 20, n_pts)], 1)
    K = np.array([[1000.0, 0, W / 2.0], [0, 1000.0, H / 2.0], [0, 0, 1.0]])
    a = (cam - 1.5) * 0.35
    C = np.array([2.0 * math.sin(a), 0.1 * cam, 2.0 - 2.0 * math.cos(a)])
    yaw = -a * 0.5
    R = np.array([[math.cos(yaw), 0, -math.sin(yaw)], [0, 1, 0], [math.sin(yaw), 0, math.cos(yaw)]])
    t = -R @ C
    Xc = X @ R.T + t
    uvw = Xc @ K.T
    x = uvw[:, :2] / uvw[:, 2:3]
    x += rng.normal(0, noise_sigma, x.shape)
    inl = np.ones(n_pts, dtype=bool)
    n_out = int(outlier_frac * n_pts)
    out_idx = rng.choice(n_pts, size=n_out, replace=False)
    x[out_idx] = np.stack([rng.uniform(0, W, n_out), rng.uniform(0, H, n_out)], 1)
    inl[out_idx] = False
    return dict(X=np.ascontiguousarray(X), x=np.ascontiguousarray(x), K=K, R=R, t=t, inliers=inl)


def random_poses(n, seed=4100, base_R=None, base_t=None, jitter=0.2):
    """n x 12 row-major [R|t] hypotheses: random rotations near base (or arbitrary)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 12))
    for i in range(n):
        w = rng.normal(0, jitter, 3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        Rj = np.eye(3) + (math.sin(th) / th) * Kx + ((1 - math.cos(th)) / th ** 2) * Kx @ Kx if th > 1e-12 else np.eye(3)
        R = Rj @ (base_R if base_R is not None else np.eye(3))
        t = (base_t if base_t is not None else np.zeros(3)) + rng.normal(0, jitter, 3)
        out[i] = np.concatenate([R, t[:, None]], 1).reshape(-1)
    return out


# ---- a textured plane seen by pinhole cameras: geometrically consistent frames for end-to-end runs --------------------
def plane_texture(size=1400, seed=77, n_rect=900):
    """High-contrast random rectangles (corners for FAST) with mild noise, uint8 size x size."""
    rng = np.random.default_rng(seed)
    tex = np.full((size, size), 110.0)
    for _ in range(n_rect):
        w, h = rng.integers(12, 90, 2)
        x, y = rng.integers(0, size - w), rng.integers(0, size - h)
        tex[y:y + h, x:x + w] = rng.integers(20, 236)
    tex += rng.normal(0.0, 1.5, tex.shape)
    return np.clip(tex, 0, 255)


def look_at_plane_pose(center_xy, height, yaw=0.0, tilt=(0.0, 0.0)):
    """World -> camera [R|t] of a camera at (cx, cy, -height) looking along +z at the plane z = 0, rotated by yaw about
    its optical axis and tilted by (rx, ry) radians."""
    cz, sz = math.cos(yaw), math.sin(yaw)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    cx_, sx_ = math.cos(tilt[0]), math.sin(tilt[0])
    Rx = np.array([[1.0, 0, 0], [0, cx_, -sx_], [0, sx_, cx_]])
    cy_, sy_ = math.cos(tilt[1]), math.sin(tilt[1])
    Ry = np.array([[cy_, 0, sy_], [0, 1.0, 0], [-sy_, 0, cy_]])
    R = Rz @ Rx @ Ry
    C = np.array([center_xy[0], center_xy[1], -float(height)])
    return R, -R @ C


def smooth_relief(centres=((6.2, 6.4, 0.80, 0.9), (8.0, 7.6, 0.60, 0.7), (7.1, 8.3, 0.50, 0.6), (7.9, 6.0, 0.70, 0.8))):
    """A gentle height field h(X, Y) >= 0 (Gaussian bumps: x, y, height, radius) that lifts the surface z = -h(X, Y) towards the
    cameras: breaks the planar two-fold ambiguity of the essential matrix without creating occlusions."""
    def h(X, Y):
        out = np.zeros_like(X, dtype=np.float64)
        for cx, cy, a, r in centres:
            out = out + a * np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / (2 * r * r))
        return out
    return h


def _surface_hit(o, dw, relief):
    """Ray o + lam dw against the surface z = -relief(X, Y) (the plane z = 0 without relief): fixed-point iteration on lam,
    a contraction for slopes well below the rays' inclination."""
    lam = -o[2] / dw[2]
    if relief is not None:
        for _ in range(8):
            Xw = o[:, None] + lam * dw
            lam = (-relief(Xw[0], Xw[1]) - o[2]) / dw[2]
    return lam


def render_plane(tex, tex_px_per_unit, K, R, t, W=640, H=480, relief=None):
    """Image of the surface z = -relief(X, Y) (the plane z = 0 by default; texture pixel (u, v) <-> world (X, Y) =
    (u, v) / tex_px_per_unit) from camera [R|t], bilinear."""
    v, u = np.mgrid[0:H, 0:W]
    d = np.linalg.inv(K) @ np.stack([u.ravel() + 0.0, v.ravel() + 0.0, np.ones(W * H)])
    o = -R.T @ t
    dw = R.T @ d
    lam = _surface_hit(o, dw, relief)
    Xw = o[:, None] + lam * dw
    tu, tv = Xw[0] * tex_px_per_unit, Xw[1] * tex_px_per_unit
    x0 = np.clip(np.floor(tu).astype(np.int64), 0, tex.shape[1] - 2)
    y0 = np.clip(np.floor(tv).astype(np.int64), 0, tex.shape[0] - 2)
    fx, fy = np.clip(tu - x0, 0, 1), np.clip(tv - y0, 0, 1)
    img = (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy) + tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy)
    inside = (tu >= 0) & (tu < tex.shape[1] - 1) & (tv >= 0) & (tv < tex.shape[0] - 1) & (lam > 0)
    img = np.where(inside, img, 0.0)
    return np.clip(img + 0.5, 0, 255).astype(np.uint8).reshape(H, W)


def backproject_to_plane(px, K, R, t, relief=None):
    """World points (n x 3, on z = -relief(X, Y), the plane z = 0 by default) seen at pixels px (n x 2) by camera [R|t]."""
    d = np.linalg.inv(K) @ np.c_[px, np.ones(len(px))].T
    o = -R.T @ t
    dw = R.T @ d
    lam = _surface_hit(o, dw, relief)
    return (o[:, None] + lam * dw).T
