"""Light-curtain placement next to the detector, on the device.

Host-side mirror of the reference's policy interface for the path LC_PROCESS walks between two
detector passes (second/light_curtain/utils.py:88-104): class scores -> confidence map ->
entropy -> DPOptimizedPolicy.get_design_points.  Same names, constructor arguments and return
values as second/light_curtain/policy.py; the arithmetic runs in liblcf.so
(csrc/lc_policy.hip) and this module fails loudly without it -- there is no CPU path.

    policy = get_lc_policy_class("DPOptimizedPolicy")(lc_device)
    design_pts = policy.get_design_points(confidence_map)            # the reference's call
    design_pts = policy.design_points_from_scores(cls_preds, mask)   # scores stay on the GPU

What stays on the host, once per map shape, exactly as the reference caches it
(policy.py:303-304): the graph -- camera-ray points, their map cells and laser angles
(policy.py:160-281), float64 numpy.  NOTE the node coordinates follow numpy's promotion rules at
run time, like the reference's expressions do: under NumPy >= 2 `theta * ones(float32)` is
float64, under the NumPy the reference was written for it was float32; cells on a rounding
boundary can differ between the two.  The fixtures were generated under NumPy 2.2.

All nine registered policies of policy.py exist under their names: DPOptimizedPolicy (entropy
table + DP on the device), MaxEntropy / ParallelMaxEntropy (entropy table on the device, arg-max
with torch), GreedyDPTiesRandom / GreedyDPTiesMinVel (entropy table on the device, the walk on the
host), FixedParallel15/30/45 and RandomParallel (geometry only).

`LCGeometry` carries the CAMERA_PARAMS / LASER_PARAMS / TRANSFORMS of pylc.LCDevice
(pylc/pylc.py:29-146) for callers that have no pylc build; the light-curtain SIMULATOR
(`get_return`) is outside SURVEY.md section 8 and is not provided.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check

REGISTERED_LC_POLICY_CLASSES = {}


def register_lc_policy(cls):
    """policy.py:7-12."""
    name = cls.__name__
    assert name not in REGISTERED_LC_POLICY_CLASSES, f"exist class: {REGISTERED_LC_POLICY_CLASSES}"
    REGISTERED_LC_POLICY_CLASSES[name] = cls
    return cls


def get_lc_policy_class(name):
    """policy.py:14-17."""
    assert name in REGISTERED_LC_POLICY_CLASSES, f"available class: {REGISTERED_LC_POLICY_CLASSES}"
    return REGISTERED_LC_POLICY_CLASSES[name]


# ------------------------------------------------------------------------------- geometry
def _euler_sxyz(roll, pitch, yaw):
    cr, sr, cp, sp = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    return m


class LCGeometry:
    """CAMERA_PARAMS, LASER_PARAMS and TRANSFORMS of a light-curtain device with the defaults of
    pylc.LCDevice (pylc/pylc.py:42-84, 117-146): velo frame = world frame, camera axes permuted
    (roll, pitch, yaw = -90, 0, -90), laser 0.2 m to the camera's right."""

    def __init__(self, LASER_PARAMS=None, CAMERA_PARAMS=None):
        self.LASER_PARAMS = dict(x=0.0, y=-0.2, z=0.0, roll=-90.0, pitch=0.0, yaw=-90.0, fov=80,
                                 laser_limit=14000, laser_timestep=1.5e-5)
        self.CAMERA_PARAMS = dict(x=0.0, y=0.0, z=0.0, roll=-90.0, pitch=0.0, yaw=-90.0,
                                  width=512, height=512, fov=80, matrix=None)
        for mine, given in ((self.LASER_PARAMS, LASER_PARAMS), (self.CAMERA_PARAMS, CAMERA_PARAMS)):
            for key in mine:
                if given is not None and key in given:
                    mine[key] = given[key]
        c = self.CAMERA_PARAMS
        if c["matrix"] is None:  # pylc.py:12-27
            f = (c["width"] / 2) / math.tan(c["fov"] * math.pi / 180.0 / 2)
            c["matrix"] = np.array([[f, 0., c["width"] / 2], [0., f, c["height"] / 2],
                                    [0., 0., 1.]], dtype=np.float32)
        t = {}
        t["cTw"] = self._pose(c)
        t["wTc"] = np.linalg.inv(t["cTw"])
        t["lTw"] = self._pose(self.LASER_PARAMS)
        t["wTl"] = np.linalg.inv(t["lTw"])
        t["lTc"] = t["wTc"].dot(t["lTw"])
        t["cTl"] = np.linalg.inv(t["lTc"])
        self.TRANSFORMS = t

    @staticmethod
    def _pose(p):
        m = _euler_sxyz(*(p[k] * np.pi / 180. for k in ("roll", "pitch", "yaw")))
        m[:3, 3] = [p["x"], p["y"], p["z"]]
        return m.astype(np.float32)

    def get_return(self, np_depth_image, np_design_pts):
        raise NotImplementedError("the light-curtain simulator (pylc_lib) is outside this "
                                  "library's scope; use pylc.LCDevice for get_return")


def confidence_grid(anchors, K, Y, X, vel2cam):
    """(x, z) in the camera frame of every map location, [Y, X, 2] float32 with axis 0 in
    increasing camera x: the first two channels of `_get_confidence_map`'s result
    (utils.py:207, 219-235).  Geometry only -- the scores stay on the device."""
    anchors = np.asarray(anchors).reshape(K, Y, X, -1)
    for k in range(1, K):
        assert np.all(anchors[0, :, :, :3] == anchors[k, :, :, :3])
    xyz1 = np.concatenate([anchors[0, :, :, :3], np.ones([Y, X, 1], dtype=np.float32)], axis=2)
    return np.flip((xyz1 @ vel2cam.T)[:, :, [0, 2]], axis=0)


# ------------------------------------------------------------------------------- policies
def _stream():
    return torch.cuda.current_stream().cuda_stream


class Policy:
    """policy.py:19-66.  `confidence2entropy` here returns the device look-up table
    (policy.py:154-158's `ulookup`) instead of a host map."""

    def __init__(self, lc_device, device="cuda"):
        self._lc_device = lc_device
        self._dev = torch.device(device)
        if self._dev.type != "cuda":
            raise RuntimeError("light-curtain policies run in liblcf.so on a GPU; no CPU path")

    def get_design_points(self, confidence_map):
        raise NotImplementedError

    def entropy_lookup(self, cls, anchors_mask=None, apply_sigmoid=False, out=None):
        """cls [K, Y, X] float32 device tensor laid out as the detector emits it (increasing velo
        y, x); anchors_mask [K*Y*X] uint8 or None.  Returns ulookup [1 + Y*X] float32."""
        K, Y, X = cls.shape
        cls = cls.contiguous()
        assert cls.is_cuda and cls.dtype == torch.float32
        if anchors_mask is not None:
            anchors_mask = anchors_mask.reshape(-1).contiguous()
            assert anchors_mask.is_cuda and anchors_mask.dtype == torch.uint8
            assert anchors_mask.numel() == K * Y * X
        if out is None:
            out = torch.empty(1 + Y * X, dtype=torch.float32, device=cls.device)
        check(_lib.lib().lcf_lc_entropy_lookup(
            cls.data_ptr(), K, Y, X, 0 if anchors_mask is None else anchors_mask.data_ptr(),
            int(bool(apply_sigmoid)), out.data_ptr(), _stream()))
        return out

    def confidence2entropy(self, confidence_map):
        """The reference's signature (policy.py:41-66): host map [X, Z, 2 + K] in, the entropy
        look-up table on the device out."""
        p = np.flip(np.asarray(confidence_map)[:, :, 2:], axis=0).transpose(2, 0, 1)
        cls = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).to(self._dev)
        return self.entropy_lookup(cls)


@register_lc_policy
class DPOptimizedPolicy(Policy):
    """policy.py:68-434: one design point per camera ray, maximising the summed entropy under
    the laser's velocity limit, ties broken by the smallest summed squared laser motion."""

    def __init__(self, lc_device, pts_per_cam_ray=80, debug=False, device="cuda"):
        super().__init__(lc_device, device)
        self._pts_per_cam_ray = pts_per_cam_ray
        self._debug = debug
        self._laser_vel_limit = 14000  # degrees per sec
        self._laser_timestep = 1.5e-5  # sec
        self._max_dtheta_las = self._laser_vel_limit * self._laser_timestep  # degrees
        self._cam_thetas = self._compute_cam_thetas()
        self._graph = None

    def _compute_cam_thetas(self):
        """policy.py:90-112: the angle of every pixel column, radians, ascending."""
        cp = self._lc_device.CAMERA_PARAMS
        w, h = cp["width"], cp["height"]
        cols = np.arange(w).reshape(w, 1) + 0.5
        rows = np.ones([w, 1], dtype=np.float32) * 0.5 * h
        rays = np.hstack([cols, rows, np.ones([w, 1], dtype=np.float32)]) \
            @ np.linalg.inv(cp["matrix"]).T
        assert np.all(np.abs(rays[:, 1]) < 1e-6)
        thetas = np.arctan2(rays[:, 2], rays[:, 0])
        thetas.sort()
        return thetas

    @staticmethod
    def _interpolate(x, z, query_x, query_z):
        """policy.py:135-152: 1-based flat index of the nearest map cell, 0 outside the map."""
        X, Z = x.shape
        step_x = (x.max() - x.min()) / (X - 1)
        step_z = (z.max() - z.min()) / (Z - 1)
        qx = np.round((query_x - x.min()) / step_x).astype(np.int64)
        qz = np.round((query_z - z.min()) / step_z).astype(np.int64)
        qk = np.zeros(len(qx), dtype=np.int64)
        inside = (qx >= 0) & (qx < X) & (qz >= 0) & (qz < Z)
        qk[inside] = qx[inside] * Z + qz[inside] + 1
        return qk

    def _preprocess_graph(self, xz):
        """policy.py:160-281 for the map geometry xz [X, Z, 2]: nodes [T, N, 5] = (x, z, r, theta,
        k) and the laser angle (degrees) of every node; uploaded once."""
        n = self._pts_per_cam_ray
        if n > 256:
            raise ValueError("pts_per_cam_ray > 256 is not supported by lcf_lc_dp_design")
        x, z = xz[:, :, 0], xz[:, :, 1]
        r_ray = np.linspace(3.0, z.max(), n)
        nodes = []
        for theta in self._cam_thetas:
            th_ray = theta * np.ones([n], dtype=np.float32)
            x_ray, z_ray = r_ray * np.cos(th_ray), r_ray * np.sin(th_ray)
            k_ray = self._interpolate(x, z, x_ray, z_ray)
            nodes.append(np.hstack([e.reshape(-1, 1) for e in (x_ray, z_ray, r_ray, th_ray, k_ray)]))
        nodes = np.stack(nodes)  # [T, N, 5] float64
        flat = nodes.reshape(-1, 5)
        zeros = np.zeros([len(flat)], dtype=np.float32)
        xyz1 = np.hstack([e.reshape(-1, 1) for e in (flat[:, 0], zeros, flat[:, 1], zeros + 1)])
        in_las = xyz1 @ self._lc_device.TRANSFORMS["cTl"].T
        theta_las = np.rad2deg(np.arctan2(in_las[:, 2], in_las[:, 0])).reshape(nodes.shape[:2])
        T = len(nodes)
        L = _lib.lib()
        g = {"umap_wh": tuple(xz.shape[:2]), "nodes": nodes, "theta_las": theta_las,
             "node_k": torch.from_numpy(nodes[:, :, 4].astype(np.int32)).to(self._dev),
             "theta_las_dev": torch.from_numpy(np.ascontiguousarray(theta_las)).to(self._dev),
             "xz_dev": torch.from_numpy(np.ascontiguousarray(nodes[:, :, :2])).to(self._dev),
             "choice": torch.empty(T, dtype=torch.int32, device=self._dev),
             "ws": torch.empty(L.lcf_lc_dp_workspace_bytes(T, n), dtype=torch.uint8,
                               device=self._dev)}
        return g

    def set_map_geometry(self, xz):
        """Build (or reuse) the graph for a map whose first two channels are xz [X, Z, 2]."""
        xz = np.asarray(xz)
        if self._graph is None or self._graph["umap_wh"] != tuple(xz.shape[:2]):
            self._graph = self._preprocess_graph(xz)
        return self._graph

    def _solve(self, ulookup):
        g = self._graph
        T, n = g["node_k"].shape
        assert ulookup.numel() == 1 + g["umap_wh"][0] * g["umap_wh"][1]
        check(_lib.lib().lcf_lc_dp_design(
            ulookup.data_ptr(), ulookup.numel(), g["node_k"].data_ptr(),
            g["theta_las_dev"].data_ptr(), T, n,
            float(self._max_dtheta_las), g["choice"].data_ptr(), g["ws"].data_ptr(),
            g["ws"].numel(), _stream()))
        return g["choice"]

    def design_points_from_scores(self, cls_preds, anchors_mask=None, apply_sigmoid=True):
        """The device path: cls_preds [K, Y, X] float32 logits (apply_sigmoid) or probabilities as
        the detector head emits them, anchors_mask [K*Y*X] uint8 or None, both on the GPU; the
        map geometry must have been given with set_map_geometry(confidence_grid(...)).
        Returns the design points [T, 2] float64 on the device, x ascending."""
        if self._graph is None:
            raise RuntimeError("call set_map_geometry(xz) first")
        choice = self._solve(self.entropy_lookup(cls_preds, anchors_mask, apply_sigmoid))
        g = self._graph
        pts = g["xz_dev"][torch.arange(len(choice), device=choice.device), choice.long()]
        return torch.flip(pts, dims=[0])

    def get_design_points(self, confidence_map):
        """policy.py:283-434: host confidence map [X, Z, 2 + K] in, design points [T, 2] float64
        (camera-frame x, z; x ascending) out."""
        confidence_map = np.asarray(confidence_map)
        g = self.set_map_geometry(confidence_map[:, :, :2])
        choice = self._solve(self.confidence2entropy(confidence_map)).cpu().numpy()
        return np.flip(g["nodes"][np.arange(len(choice)), choice, :2], axis=0)


class GreedyDP(DPOptimizedPolicy):
    """policy.py:437-535: a forward greedy walk over the DP's graph -- the most uncertain allowed
    successor of every ray, ties broken at random or by the smallest laser motion.  The entropy
    table comes from the device kernel; the walk is T dependent steps over a handful of points
    and runs on the host from the downloaded table (141 KB), as in the reference."""

    def __init__(self, lc_device, pts_per_cam_ray=80, ties=None, debug=False, device="cuda"):
        super().__init__(lc_device, pts_per_cam_ray, debug, device)
        self._ties = ties

    def _walk(self, ulookup):
        g = self._graph
        nodes, theta = g["nodes"], g["theta_las"]
        k = nodes[:, :, 4].astype(np.int64)
        r_unc = ulookup[k[0]]
        cur = np.random.choice(np.where(r_unc == r_unc.max())[0])  # first point: random tie-break
        trajectory = [nodes[0][[cur]]]
        for t in range(1, len(nodes)):
            d = theta[t] - theta[t - 1, cur]
            neighbours = np.where((d < self._max_dtheta_las) & (d > -self._max_dtheta_las))[0]
            if len(neighbours) == 0:
                break  # the curtain cannot proceed any further
            r_unc = ulookup[k[t][neighbours]]
            keep = r_unc == r_unc.max()
            neighbours, d_kept = neighbours[keep], d[neighbours][keep]
            if self._ties == "random":
                cur = np.random.choice(neighbours)
            elif self._ties == "minvel":
                cur = neighbours[np.argmin(np.abs(d_kept))]
            else:
                raise Exception("ties must be one of random/minvel")
            trajectory.append(nodes[t][[cur]])
        return np.flip(np.vstack(trajectory)[:, :2], axis=0)

    def get_design_points(self, confidence_map):
        confidence_map = np.asarray(confidence_map)
        self.set_map_geometry(confidence_map[:, :, :2])
        return self._walk(self.confidence2entropy(confidence_map).cpu().numpy())


@register_lc_policy
class GreedyDPTiesRandom(GreedyDP):
    def __init__(self, lc_device, pts_per_cam_ray=80, debug=False, device="cuda"):
        super().__init__(lc_device, pts_per_cam_ray, "random", debug, device)


@register_lc_policy
class GreedyDPTiesMinVel(GreedyDP):
    def __init__(self, lc_device, pts_per_cam_ray=80, debug=False, device="cuda"):
        super().__init__(lc_device, pts_per_cam_ray, "minvel", debug, device)


def _parallel_lc_extent(lc_device, z):
    """policy.py:548-557: the leftmost and rightmost x of the camera's field of view at depth z."""
    x_max = z / np.tan(np.deg2rad(0.5 * lc_device.CAMERA_PARAMS["fov"]))
    return -x_max, x_max


def _parallel_curtain(lc_device, z):
    x_min, x_max = _parallel_lc_extent(lc_device, z)
    xs = np.linspace(x_min, x_max, 1600).reshape(-1, 1)
    return np.hstack((xs, np.ones_like(xs) * z))


class _HostPolicy(Policy):
    """Policies that place a curtain without looking at the scores need no GPU."""

    def __init__(self, lc_device, device=None):
        self._lc_device = lc_device
        self._dev = None


@register_lc_policy
class FixedParallel15(_HostPolicy):
    def get_design_points(self, confidence_map):
        return _parallel_curtain(self._lc_device, 15)


@register_lc_policy
class FixedParallel30(_HostPolicy):
    def get_design_points(self, confidence_map):
        return _parallel_curtain(self._lc_device, 30)


@register_lc_policy
class FixedParallel45(_HostPolicy):
    def get_design_points(self, confidence_map):
        return _parallel_curtain(self._lc_device, 45)


@register_lc_policy
class RandomParallel(_HostPolicy):
    def get_design_points(self, confidence_map):
        return _parallel_curtain(self._lc_device, np.random.uniform(5, 60))


def _first_argmax(e, dim):
    """Index of the first maximum along dim (numpy's argmax rule), on the device."""
    n = e.shape[dim]
    idx = torch.arange(n, device=e.device).view([-1 if d == dim else 1 for d in range(e.dim())])
    top = e.max(dim=dim, keepdim=True).values
    return torch.where(e == top, idx, n).min(dim=dim).values


@register_lc_policy
class MaxEntropy(Policy):
    """policy.py:593-603: per camera-x row of the map, the cell of maximum entropy."""

    def design_points_from_lookup(self, ulookup, xz):
        X, Z = xz.shape[:2]
        z_inds = _first_argmax(ulookup[1:].view(X, Z), 1).cpu().numpy()
        return np.asarray(xz)[np.arange(X), z_inds]

    def get_design_points(self, confidence_map):
        confidence_map = np.asarray(confidence_map)
        return self.design_points_from_lookup(self.confidence2entropy(confidence_map),
                                              confidence_map[:, :, :2])


@register_lc_policy
class ParallelMaxEntropy(Policy):
    """policy.py:605-673: a fronto-parallel curtain at the depth whose cells inside the camera's
    field of view hold the most entropy, at least 5 m away."""

    def __init__(self, lc_device, debug=False, device="cuda"):
        super().__init__(lc_device, device)
        self._fov_mask = {"cmap_wh": None, "mask": None}
        self._debug = debug

    def _create_fov_mask(self, xz):
        theta = np.rad2deg(np.arctan2(xz[:, :, 1], xz[:, :, 0]))
        fov = self._lc_device.CAMERA_PARAMS["fov"]
        mask = (theta > 90 - 0.5 * fov) & (theta < 90 + 0.5 * fov)
        return {"cmap_wh": tuple(xz.shape[:2]),
                "mask": torch.from_numpy(np.ascontiguousarray(mask)).to(self._dev)}

    def design_points_from_lookup(self, ulookup, xz):
        xz = np.asarray(xz)
        if self._fov_mask["cmap_wh"] != tuple(xz.shape[:2]):
            self._fov_mask = self._create_fov_mask(xz)
        X, Z = xz.shape[:2]
        e = ulookup[1:].view(X, Z) * self._fov_mask["mask"]
        z_max_entropy = max(xz[0, int(_first_argmax(e.sum(dim=0), 0)), 1], 5.0)
        xs = xz[:, 0, [0]]
        return np.hstack((xs, np.ones_like(xs) * z_max_entropy))

    def get_design_points(self, confidence_map):
        confidence_map = np.asarray(confidence_map)
        return self.design_points_from_lookup(self.confidence2entropy(confidence_map),
                                              confidence_map[:, :, :2])
