"""Seeded synthetic point clouds shaped like the reference's inputs (no dataset is available).

Shapes follow SURVEY.md section 8(d):
  * ``vkitti_frame``: Virtual-KITTI-shaped lidar-like cloud + light-curtain returns, camera
    1242x375 with 81.16 deg horizontal FOV (second/data/vkitti_dataset.py:51-54), x < 80
    (vkitti_dataset.py:71), range [0,-40,-3,70.4,40,1] (configs/vkitti/second/base.yaml:6);
    intensity 1.0 for lidar-like points (second/pytorch/builder/input_reader_builder.py:131)
    and U(50/255, 1) for curtain returns (second/light_curtain/utils.py:115,161); curtain
    height capped at 2.0 m (base.yaml:144); car boxes 1.6 x 3.9 x 1.56 (base.yaml:83).
  * ``synthia_frame``: SYNTHIA-shaped 640x480, 39.32 deg FOV (second/data/synthia_dataset.py:53-55)
    with 10 curtains, N = 120,000.
Points are shuffled, as second/data/preprocess.py:616-618 does.
"""
import numpy as np

PC_RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)
VOXEL_SIZE = (0.05, 0.05, 0.1)
MAX_POINTS_PER_VOXEL = 5


def _rays(rng, n, width, height, hfov_deg, lower_only=False):
    f = 0.5 * width / np.tan(np.radians(hfov_deg) * 0.5)
    u = rng.uniform(0.0, width, n)
    v = rng.uniform(height * 0.5 + 1.0 if lower_only else 0.0, height, n)
    dy = -(u - 0.5 * width) / f
    dz = -(v - 0.5 * height) / f
    return dy, dz, f


def _frame(seed, n, width, height, hfov, frac_ground, frac_curtain, n_curtains):
    rng = np.random.default_rng(seed)
    n_ground = int(n * frac_ground)
    n_curt = int(n * frac_curtain)
    n_box = n - n_ground - n_curt
    cam_h = 1.7
    parts = []
    # ground plane seen by the camera (dense near, sparse far)
    dy, dz, _ = _rays(rng, n_ground * 2, width, height, hfov, lower_only=True)
    x = cam_h / np.maximum(-dz, 1e-6)
    keep = (x > 2.0) & (x < 70.0)
    x, dy = x[keep][:n_ground], dy[keep][:n_ground]
    while x.shape[0] < n_ground:  # top up (rare)
        dy2, dz2, _ = _rays(rng, n_ground, width, height, hfov, lower_only=True)
        x2 = cam_h / np.maximum(-dz2, 1e-6)
        k2 = (x2 > 2.0) & (x2 < 70.0)
        x = np.concatenate([x, x2[k2]])[:n_ground]
        dy = np.concatenate([dy, dy2[k2]])[:n_ground]
    g = np.stack([x, x * dy, -cam_h + rng.normal(0.0, 0.03, n_ground),
                  np.ones(n_ground)], axis=1)
    parts.append(g)
    # light-curtain returns: smooth vertical ruled surfaces x = r(azimuth)
    per = [n_curt // n_curtains] * n_curtains
    per[-1] += n_curt - sum(per)
    half = np.tan(np.radians(hfov) * 0.5)
    for ci, m in enumerate(per):
        t = rng.uniform(-half, half, m)  # tan(azimuth) = y / x
        base = rng.uniform(8.0, 45.0)
        amp = rng.uniform(1.0, 6.0)
        freq = rng.uniform(1.0, 4.0)
        ph = rng.uniform(0.0, 2 * np.pi)
        r = base + amp * np.sin(freq * np.arctan(t) * 4.0 + ph)
        r = np.clip(r, 3.0, 69.0)
        z = rng.uniform(-cam_h, 2.0, m)
        inten = rng.uniform(50.0 / 255.0, 1.0, m)
        parts.append(np.stack([r, -r * t, z, inten], axis=1))
    # car-sized boxes
    n_cars = max(4, n_box // 300)
    per_car = [n_box // n_cars] * n_cars
    per_car[-1] += n_box - sum(per_car)
    for m in per_car:
        cx = rng.uniform(6.0, 60.0)
        cy = rng.uniform(-0.8 * half * cx, 0.8 * half * cx)
        yaw = rng.uniform(-np.pi, np.pi)
        w, l, h = 1.6, 3.9, 1.56
        face = rng.integers(0, 3, m)
        a = rng.uniform(-0.5, 0.5, m)
        b = rng.uniform(-0.5, 0.5, m)
        s = rng.choice([-0.5, 0.5], m)
        lx = np.where(face == 0, s, a) * l
        ly = np.where(face == 1, s, np.where(face == 0, a, b)) * w
        lz = np.where(face == 2, 0.5, np.where(face == 0, b, a) + 0.5) * h
        c, sn = np.cos(yaw), np.sin(yaw)
        parts.append(np.stack([cx + c * lx - sn * ly, cy + sn * lx + c * ly, -cam_h + lz,
                               np.ones(m)], axis=1))
    pts = np.concatenate(parts, axis=0).astype(np.float32)
    assert pts.shape[0] == n, (pts.shape, n)
    rng.shuffle(pts, axis=0)
    return np.ascontiguousarray(pts)


def vkitti_frame(seed=0, n=20000):
    """[N,4] float32 (x, y, z, intensity)."""
    return _frame(seed, n, 1242, 375, 81.16, 0.55, 0.30, 3)


def synthia_frame(seed=0, n=120000):
    return _frame(seed, n, 640, 480, 39.32, 0.20, 0.70, 10)


def lc_scene(seed, K=2, Y=200, X=176, bumps=12):
    """A detector's class-score map for the light-curtain placement step: K anchor
    types on the [Y, X] bird's-eye grid of the shipped configs (0.4 m cells over
    [0, 70.4] x [-40, 40] m, base.yaml), scores in [0, 1] = a low floor plus a few Gaussian bumps;
    an anchors mask that drops 10 % of the locations; anchor centres.  Returns (cls [K,Y,X] f32,
    mask [K*Y*X] u8, anchors [K*Y*X, 7] f32)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:Y, 0:X].astype(np.float32)
    cls = np.full((K, Y, X), 0.02, dtype=np.float32)
    for k in range(K):
        for _ in range(bumps):
            cy, cx, sg = rng.uniform(0, Y), rng.uniform(0, X), rng.uniform(2.0, 8.0)
            cls[k] += np.float32(rng.uniform(0.2, 0.9)) * np.exp(
                -((yy - cy) ** 2 + (xx - cx) ** 2) / np.float32(2 * sg * sg))
    cls = np.clip(cls, 0, 1).astype(np.float32)
    mask = (rng.uniform(size=(K, Y, X)) < 0.9).reshape(-1).astype(np.uint8)
    xs = (np.arange(X, dtype=np.float32) + 0.5) * np.float32(70.4 / X)
    ys = (np.arange(Y, dtype=np.float32) + 0.5) * np.float32(80.0 / Y) - 40
    anchors = np.zeros((K, Y, X, 7), dtype=np.float32)
    anchors[..., 0] = xs[None, None, :]
    anchors[..., 1] = ys[None, :, None]
    anchors[..., 2] = -1.0
    anchors[..., 3:6] = np.array([1.6, 3.9, 1.56], dtype=np.float32)
    for k in range(K):
        anchors[k, ..., 6] = np.float32(k * np.pi / K)
    return cls, mask, anchors.reshape(-1, 7)
