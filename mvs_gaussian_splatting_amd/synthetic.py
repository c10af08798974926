"""Seeded synthetic scenes for tests and benchmarks: cameras that reproduce the
reference's matrix conventions (``scene/cameras.py:48-57``, ``utils/graphics_utils.py:38-71``) and a
duck-typed Gaussian model exposing the getters ``render()`` reads
(``scene/gaussian_model.py:151-194``).  Everything is generated on the CPU in float32 from a
``torch.Generator`` and then moved, so the CPU oracle and the GPU see identical inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch


# ---- camera matrices --------------------------------------------------------------------
def get_world2view2(R: np.ndarray, t: np.ndarray, translate=np.array([0.0, 0.0, 0.0]), scale: float = 1.0):
    """``utils/graphics_utils.py:38-49``: R is camera-to-world rotation, t the world-to-camera translation."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def get_projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """``utils/graphics_utils.py:51-71``."""
    top = math.tan(fovY / 2) * znear
    right = math.tan(fovX / 2) * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right + right)
    P[1, 1] = 2.0 * znear / (top + top)
    P[0, 2] = (right - right) / (right + right)
    P[1, 2] = (top - top) / (top + top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def focal2fov(focal: float, pixels: float) -> float:
    return 2 * math.atan(pixels / (2 * focal))


class SyntheticCamera:
    """The attributes ``render()`` reads from a ``Camera`` / ``MiniCam`` (``scene/cameras.py:17-70``)."""

    def __init__(self, width: int, height: int, fx: float, fy: float, R: Optional[np.ndarray] = None,
                 T: Optional[np.ndarray] = None, znear: float = 0.01, zfar: float = 100.0, device="cpu"):
        self.image_width, self.image_height = int(width), int(height)
        self.FoVx, self.FoVy = focal2fov(fx, width), focal2fov(fy, height)
        self.znear, self.zfar = znear, zfar
        R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
        T = np.zeros(3) if T is None else np.asarray(T, dtype=np.float64)
        self.R, self.T = R, T
        self.world_view_transform = torch.tensor(get_world2view2(R, T)).transpose(0, 1).contiguous()
        self.projection_matrix = get_projection_matrix(znear, zfar, self.FoVx, self.FoVy).transpose(0, 1).contiguous()
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0)
                                    .bmm(self.projection_matrix.unsqueeze(0))).squeeze(0).contiguous()
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()
        self.to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def orbit_camera(view: int, n_views: int, width: int, height: int, fx: float, fy: float,
                 centre=(0.0, 0.0, 6.0), device="cpu") -> SyntheticCamera:
    """View ``v``: the identity camera rotated by ``360/n_views * v`` degrees about the vertical axis through
    ``centre``.  View 0 is the camera at the origin looking down +z."""
    ang = 2.0 * math.pi * view / max(n_views, 1)
    c, s = math.cos(ang), math.sin(ang)
    Rc2w = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])           # camera-to-world
    centre = np.asarray(centre, dtype=np.float64)
    cam_pos = centre + Rc2w @ (np.zeros(3) - centre)                          # rotate the origin about centre
    T = -Rc2w.T @ cam_pos                                                     # world-to-camera translation
    return SyntheticCamera(width, height, fx, fy, R=Rc2w, T=T, device=device)


# ---- Gaussian clouds ----------------------------------------------------------------------
@dataclass
class SceneConfig:
    name: str
    P: int
    sh_degree: int
    width: int
    height: int
    fx: float
    fy: float
    log_scale_mean: float


CONFIGS = {
    # SURVEY §8d / BASELINE.json configs[1..3]
    "C2": SceneConfig("C2", 100_000, 0, 800, 800, 1111.1, 1111.1, math.log(0.030)),
    "C3": SceneConfig("C3", 1_000_000, 3, 1920, 1080, 1200.0, 1200.0, math.log(0.012)),
    "C4": SceneConfig("C4", 6_000_000, 3, 1920, 1080, 1200.0, 1200.0, math.log(0.006)),
}


class SyntheticGaussianModel:
    """Raw parameters + the activations of ``scene/gaussian_model.py:27-42,151-194``."""

    def __init__(self, P: int, sh_degree: int, seed: int = 0, log_scale_mean: float = math.log(0.01),
                 extent=(6.0, 3.4, 3.0), centre=(0.0, 0.0, 6.0), device="cpu", requires_grad: bool = False):
        g = torch.Generator().manual_seed(seed)
        M = (sh_degree + 1) ** 2
        ext = torch.tensor(extent, dtype=torch.float32)
        ctr = torch.tensor(centre, dtype=torch.float32)
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        # the activation attributes the reference model carries (scene/gaussian_model.py:34-42)
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self._xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * ext + ctr
        self._scaling = log_scale_mean + 0.4 * torch.randn(P, 3, generator=g)
        self._rotation = torch.randn(P, 4, generator=g)
        self._opacity = 1.5 * torch.randn(P, 1, generator=g)
        self._features_dc = torch.randn(P, 1, 3, generator=g)
        self._features_rest = 0.1 * torch.randn(P, M - 1, 3, generator=g)
        self.xyz_gradient_accum = torch.zeros(P, 1)
        self.denom = torch.zeros(P, 1)
        self.max_radii2D = torch.zeros(P)
        self.to(device)
        if requires_grad:
            for t in self.parameters():
                t.requires_grad_(True)

    _PARAMS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")

    def parameters(self):
        return [getattr(self, k) for k in self._PARAMS]

    def to(self, device):
        for k in self._PARAMS + ("xyz_gradient_accum", "denom", "max_radii2D"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier: float = 1.0):
        """``scene/gaussian_model.py:28-32,191-192`` on whatever device the parameters live on."""
        s = scaling_modifier * self.get_scaling
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        L = R * s[:, None, :]
        cov = L @ L.transpose(1, 2)
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


class PipelineParams:
    """``arguments/__init__.py:74-80`` defaults."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fuse_activations = True     # this build's extension: feed raw parameters to the operator when possible
    fuse_densify_stats = False  # this build's extension: the backward also takes the densification statistics


def make_heavy_tail_model(P: int, sh_degree: int, seed: int = 0, z_near: float = 0.3, z_far: float = 60.0,
                          log_footprint_mean: float = math.log(0.0013), log_footprint_sigma: float = 0.95,
                          opacity_mean: float = -0.5) -> "SyntheticGaussianModel":
    """A cloud shaped like a TRAINED scene rather than the uniform box of SURVEY §8(d): depths log-uniform over more
    than seven binades (so a frame needs all 32 bits of the depth key), a third of the Gaussians in dense blobs
    (per-tile lists thousands long), world-space scales proportional to depth times a log-normal factor with a heavy tail
    (most footprints a few pixels, a few covering hundreds of tiles: instances per Gaussian ~6-10 at 1080p, rects above
    the packed-payload and tile-mask limits, Gaussians with more than 64 gradient rows), anisotropic axes, and opacities
    from transparent to opaque.  Camera: the identity view of ``orbit_camera(0, ...)`` looking down +z."""
    g = torch.Generator().manual_seed(seed)
    m = SyntheticGaussianModel(P, sh_degree, seed=seed)
    z = torch.exp(torch.rand(P, generator=g) * (math.log(z_far) - math.log(z_near)) + math.log(z_near))
    # positions on the screen: uniform over a frustum a little wider than the image (tan half-fov 0.8 x 0.45 at 1080p)
    xy = (torch.rand(P, 2, generator=g) * 2 - 1) * torch.tensor([0.95, 0.55])
    n_blob = P // 3
    centres = torch.tensor([[0.25, -0.1, 4.0], [-0.4, 0.15, 9.0], [0.05, 0.2, 2.0]])
    which = torch.randint(0, 3, (n_blob,), generator=g)
    blob = centres[which] + torch.randn(n_blob, 3, generator=g) * torch.tensor([0.06, 0.05, 0.08])
    xy[:n_blob] = blob[:, :2]
    z[:n_blob] = blob[:, 2].clamp(min=z_near)
    m._xyz = torch.stack([xy[:, 0] * z, xy[:, 1] * z, z], dim=1).contiguous()
    foot = torch.exp(log_footprint_mean + log_footprint_sigma * torch.randn(P, 1, generator=g))      # tan-space size
    aniso = torch.exp(0.5 * torch.randn(P, 3, generator=g))
    m._scaling = torch.log(z[:, None] * foot * aniso).contiguous()
    m._opacity = (2.0 * torch.randn(P, 1, generator=g) + opacity_mean).contiguous()
    return m


def make_scene(cfg: SceneConfig, seed: int = 0, device="cpu", P: Optional[int] = None, view: int = 0,
               n_views: int = 8):
    """(model, camera, bg, target image) for a config; ``P`` overrides the Gaussian count."""
    model = SyntheticGaussianModel(P or cfg.P, cfg.sh_degree, seed=seed, log_scale_mean=cfg.log_scale_mean,
                                   device=device)
    cam = orbit_camera(view, n_views, cfg.width, cfg.height, cfg.fx, cfg.fy, device=device)
    bg = torch.zeros(3, dtype=torch.float32, device=device)
    g = torch.Generator().manual_seed(1 + view)
    target = torch.rand(3, cfg.height, cfg.width, generator=g).to(device)
    return model, cam, bg, target
