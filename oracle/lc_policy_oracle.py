"""CPU restatement of the light-curtain placement step next to the detector (SURVEY.md section
8f-4).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this; the product path (objectdetection_amd/light_curtain.py + csrc/lc_policy.hip)
never does.

Parity status: PINNED by tests/golden/lc_policy_golden.npz, produced by running the reference's
own second/light_curtain/policy.py (DPOptimizedPolicy) and second/light_curtain/utils.py
(_get_confidence_map) in the build container (tests/golden/make_lc_goldens.py).  The device
geometry (pylc/pylc.py:12-27, 117-146) needs `transformations` + the compiled `pylc_lib`, both
absent: `device_geometry` restates it and is "parity unpinned" on its own; the fixtures feed the
SAME geometry to the reference's policy and to this oracle, so the policy itself is pinned.

Follows:
  confidence_map        second/light_curtain/utils.py:177-237  (_get_confidence_map)
  entropy_lookup        second/light_curtain/policy.py:60-66, 154-158
  cam_thetas            policy.py:90-112
  interpolate_k         policy.py:114-152
  build_graph           policy.py:160-281
  dp_design             policy.py:283-434
  greedy_dp, max_entropy, parallel_max_entropy, fixed_parallel      policy.py:437-684
"""
import math

import numpy as np


# ------------------------------------------------------------------ device geometry (unpinned)
def _rot_sxyz(roll, pitch, yaw):
    """Static-frame x, y, z Euler rotation (what transformations.euler_matrix defaults to)."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=np.float64)
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=np.float64)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=np.float64)
    return rz @ ry @ rx


def _xyzrpy(x, y, z, roll, pitch, yaw):
    t = np.eye(4, dtype=np.float64)
    t[:3, :3] = _rot_sxyz(*(math.radians(a) for a in (roll, pitch, yaw)))
    t[:3, 3] = [x, y, z]
    return t.astype(np.float32)  # pylc.py:122


def device_geometry(width=512, height=512, fov=80.0, laser_y=-0.2):
    """CAMERA_PARAMS / TRANSFORMS of pylc.LCDevice's defaults (pylc.py:12-27, 42-84, 117-146)."""
    fx = (width / 2) / math.tan(math.radians(fov) / 2)
    matrix = np.array([[fx, 0., width / 2], [0., fx, height / 2], [0., 0., 1.]], dtype=np.float32)
    tr = {}
    tr["cTw"] = _xyzrpy(0.0, 0.0, 0.0, -90.0, 0.0, -90.0)
    tr["wTc"] = np.linalg.inv(tr["cTw"])
    tr["lTw"] = _xyzrpy(0.0, laser_y, 0.0, -90.0, 0.0, -90.0)
    tr["wTl"] = np.linalg.inv(tr["lTw"])
    tr["lTc"] = tr["wTc"].dot(tr["lTw"])
    tr["cTl"] = np.linalg.inv(tr["lTc"])
    cam = {"width": width, "height": height, "fov": fov, "matrix": matrix}
    return cam, tr


# ------------------------------------------------------------------ confidence map and look-up
def confidence_map(anchors, anchors_mask, cls_preds, vel2cam):
    """utils.py:203-237.  cls_preds [K, Y, X] in [0, 1]; returns [Y, X, 2 + K] with axis 0 flipped."""
    K, Y, X = cls_preds.shape
    c = np.array(cls_preds.transpose(1, 2, 0))  # a copy: the reference writes into its argument
    anc = anchors.reshape(K, Y, X, 7)
    if anchors_mask is not None:
        inside = np.all(anchors_mask.reshape(K, Y, X), axis=0)
        c[~inside] = 0
    xyz = anc[0, :, :, :3]
    xyz1 = np.concatenate([xyz, np.ones([Y, X, 1], dtype=np.float32)], axis=2)
    xz = (xyz1 @ vel2cam.T)[:, :, [0, 2]]
    return np.flip(np.concatenate((xz, c), axis=2), axis=0)


def entropy_lookup(cmap):
    """policy.py:60-66 + 154-158: [1 + X*Z] float32, element 0 = 0."""
    p = cmap[:, :, 2:].clip(1e-5, 1 - 1e-5)
    e = -p * np.log2(p) - (1 - p) * np.log2(1 - p)
    e = e.mean(axis=2)
    return np.hstack([np.zeros((1,), dtype=np.float32), e.reshape(-1)])


# ------------------------------------------------------------------ the graph
def cam_thetas(cam):
    """policy.py:90-106: one angle (radians) per pixel column, ascending."""
    w, h = cam["width"], cam["height"]
    xp = np.arange(w).reshape(w, 1) + 0.5
    yp = np.ones([w, 1], dtype=np.float32) * 0.5 * h
    xy1 = np.hstack([xp, yp, np.ones([w, 1], dtype=np.float32)]) @ np.linalg.inv(cam["matrix"]).T
    th = np.arctan2(xy1[:, 2], xy1[:, 0])
    th.sort()
    return th


def interpolate_k(x, z, qx, qz):
    """policy.py:135-152: 1-based nearest cell of the [X, Z] grid x[:, :], z[:, :]; 0 outside."""
    X, Z = x.shape
    dx = (x.max() - x.min()) / (X - 1)
    dz = (z.max() - z.min()) / (Z - 1)
    ix = np.round((qx - x.min()) / dx).astype(np.int64)
    iz = np.round((qz - z.min()) / dz).astype(np.int64)
    k = np.zeros(len(ix), dtype=np.int64)
    ok = (ix >= 0) & (ix < X) & (iz >= 0) & (iz < Z)
    k[ok] = ix[ok] * Z + iz[ok] + 1
    return k


def build_graph(xz, cam, cTl, pts_per_cam_ray=80):
    """policy.py:197-208, 253-266: nodes [T, N, 5] (x, z, r, theta, k) float64 and the laser
    angle of every node [T, N] float64 (degrees).  xz = cmap[:, :, :2]."""
    x, z = xz[:, :, 0], xz[:, :, 1]
    th = cam_thetas(cam)
    r = np.linspace(3.0, z.max(), pts_per_cam_ray)
    nodes = np.zeros((len(th), pts_per_cam_ray, 5), dtype=np.float64)
    for t in range(len(th)):
        tr = th[t] * np.ones([pts_per_cam_ray], dtype=np.float32)  # float32, as the reference
        xr, zr = r * np.cos(tr), r * np.sin(tr)
        nodes[t] = np.stack([xr, zr, r, tr, interpolate_k(x, z, xr, zr)], axis=1)
    T, N = nodes.shape[:2]
    p = nodes.reshape(T * N, 5)
    xyz1 = np.hstack([p[:, [0]], np.zeros([T * N, 1], dtype=np.float32), p[:, [1]],
                      np.ones([T * N, 1], dtype=np.float32)])
    las = xyz1 @ cTl.T
    theta_las = np.rad2deg(np.arctan2(las[:, 2], las[:, 0])).reshape(T, N)
    return nodes, theta_las


# ------------------------------------------------------------------ the dynamic programme
def dp_design(ulookup, node_k, theta_las, max_dtheta):
    """policy.py:330-415 with plain loops over the points of a ray (the arithmetic per element is
    the reference's: float32 sums of uncertainties, float64 laser costs, first arg-min).
    Returns choice [T], v_unc [N] float32, v_las [N] float64 of ray 0."""
    T, N = node_k.shape
    v_unc = ulookup[node_k[T - 1]].astype(np.float32)
    v_las = np.zeros(N, dtype=np.float64)
    nxt = np.zeros((T, N), dtype=np.int64)
    for t in range(T - 2, -1, -1):
        r_unc = ulookup[node_k[t]].astype(np.float32)
        nu = np.empty(N, dtype=np.float32)
        nl = np.empty(N, dtype=np.float64)
        for i in range(N):
            d = theta_las[t + 1] - theta_las[t, i]
            ok = (d < max_dtheta) & (d > -max_dtheta)
            q = np.where(ok, r_unc[i] + v_unc, np.float32(-np.inf)).astype(np.float32)
            best = q.max()
            cost = np.where(q == best, np.square(np.abs(d)) + v_las, np.inf)
            j = int(np.argmin(cost))
            nu[i], nl[i], nxt[t, i] = best, cost[j], j
        v_unc, v_las = nu, nl
    idx = np.arange(N)[v_unc == v_unc.max()]
    c = int(idx[np.argmin(v_las[idx])])
    choice = np.zeros(T, dtype=np.int64)
    for t in range(T):
        choice[t] = c
        if t < T - 1:
            c = int(nxt[t, c])
    return choice, v_unc, v_las


def dp_design_vec(ulookup, node_k, theta_las, max_dtheta):
    """dp_design with one [N, N] array operation per ray, the shape the reference computes in
    (policy.py:340-388): the CPU-timing leg; same results as dp_design."""
    T, N = node_k.shape
    v_unc = ulookup[node_k[T - 1]].astype(np.float32)
    v_las = np.zeros(N, dtype=np.float64)
    nxt = np.zeros((T, N), dtype=np.int64)
    rows = np.arange(N)
    for t in range(T - 2, -1, -1):
        d = theta_las[t + 1][None, :] - theta_las[t][:, None]
        ok = (d < max_dtheta) & (d > -max_dtheta)
        q = np.where(ok, ulookup[node_k[t]].astype(np.float32)[:, None] + v_unc[None, :],
                     np.float32(-np.inf))
        v_unc = q.max(axis=1)
        cost = np.where(q == v_unc[:, None], np.square(np.abs(d)) + v_las[None, :], np.inf)
        nxt[t] = cost.argmin(axis=1)
        v_las = cost[rows, nxt[t]]
    idx = rows[v_unc == v_unc.max()]
    c = int(idx[np.argmin(v_las[idx])])
    choice = np.zeros(T, dtype=np.int64)
    for t in range(T):
        choice[t] = c
        if t < T - 1:
            c = int(nxt[t, c])
    return choice, v_unc, v_las


def design_points(cmap, cam, cTl, pts_per_cam_ray=80, max_dtheta=14000 * 1.5e-5):
    """DPOptimizedPolicy.get_design_points end to end (policy.py:283-434): [T, 2] float64."""
    nodes, theta_las = build_graph(cmap[:, :, :2], cam, cTl, pts_per_cam_ray)
    ul = entropy_lookup(cmap)
    choice, _, _ = dp_design(ul, nodes[:, :, 4].astype(np.int64), theta_las, max_dtheta)
    pts = nodes[np.arange(len(choice)), choice, :2]
    return np.flip(pts, axis=0)


# ------------------------------------------------------------------ the other policies
def parallel_extent(fov, z):
    """policy.py:548-557."""
    x_max = z / np.tan(np.deg2rad(0.5 * fov))
    return -x_max, x_max


def fixed_parallel(fov, z, n=1600):
    """policy.py:560-591 (FixedParallel15/30/45) and :675-684 (RandomParallel, given its z)."""
    x_min, x_max = parallel_extent(fov, z)
    xs = np.linspace(x_min, x_max, n).reshape(-1, 1)
    return np.hstack((xs, np.ones_like(xs) * z))


def max_entropy(cmap):
    """policy.py:593-603: per camera-x row, the cell of maximum entropy."""
    e = entropy_lookup(cmap)[1:].reshape(cmap.shape[:2])
    xz = cmap[:, :, :2]
    return xz[np.arange(xz.shape[0]), np.argmax(e, axis=1)]


def parallel_max_entropy(cmap, fov):
    """policy.py:605-673: a fronto-parallel curtain at the depth of maximum summed entropy inside
    the camera's field of view, at least 5 m away."""
    x, z = cmap[:, :, 0], cmap[:, :, 1]
    th = np.rad2deg(np.arctan2(z, x))
    mask = (th > 90 - 0.5 * fov) & (th < 90 + 0.5 * fov)
    e = entropy_lookup(cmap)[1:].reshape(cmap.shape[:2]) * mask
    zc = max(cmap[0, np.argmax(e.sum(axis=0)), 1], 5.0)
    xs = cmap[:, 0, [0]]
    return np.hstack((xs, np.ones_like(xs) * zc))


def greedy_dp(ulookup, nodes, theta_las, max_dtheta, ties, rng=np.random):
    """policy.py:437-535 (GreedyDP): forward greedy walk over the DP's graph; ties "random"
    (rng.choice, same call sequence as the reference) or "minvel"."""
    k = nodes[:, :, 4].astype(np.int64)
    r = ulookup[k[0]]
    cur = rng.choice(np.where(r == r.max())[0])
    traj = [nodes[0][[cur]]]
    for t in range(1, len(nodes)):
        d = theta_las[t] - theta_las[t - 1, cur]
        nb = np.where((d < max_dtheta) & (d > -max_dtheta))[0]
        if len(nb) == 0:
            break
        r = ulookup[k[t][nb]]
        keep = r == r.max()
        nb, dd = nb[keep], d[nb][keep]
        cur = rng.choice(nb) if ties == "random" else nb[np.argmin(np.abs(dd))]
        traj.append(nodes[t][[cur]])
    return np.flip(np.vstack(traj)[:, :2], axis=0)
