"""Generates tests/golden/model_densify.npz and tests/golden/render_contract.npz from the reference's own Python
(run ONLY in the build container, where /root/reference exists; the fixtures -- plain arrays -- are committed):

    python tests/golden/make_golden_model.py

1. scene/gaussian_model.py (the GaussianModel class) -- caller-side rows of SURVEY §8 that so far rested on this repo's
   restatement (oracle/densify_ref.py) although the reference holds the code in Python:
     * add_densification_stats              :775-777  (+ the max_radii2D update of train.py:130, restated below)
     * densify_and_prune                    :750-772  (plain branch) with densify_and_clone :580-610,
       densify_and_split :506-578 (N = 2), densification_postfix :466-504, prune_points / _prune_optimizer :401-449,
       cat_tensors_to_optimizer :451-472 -- on a model with a real torch.optim.Adam state (training_setup :240-266)
     * the getters                           :151-183
   The file is loaded by path (importlib) so that scene/__init__.py -- dataset readers, PIL, COLMAP loaders -- does not
   run.  It cannot be imported as it stands for three ordinary reasons, each handled without touching its text:
     - `from plyfile import ...` / `from simple_knn._C import distCUDA2` (absent packages, unused by the functions
       above): empty stub modules in sys.modules;
     - hard-coded device="cuda" in torch.zeros / torch.tensor: called under a patch that drops the `device` keyword
       (the same trick as make_golden_preprocess.py);
     - torch.normal(mean=means, std=stds) draws its own samples (:537-539): patched to `means + stds * z` with z from
       a seeded generator, and z is stored in the fixture so that another implementation can consume the same draws.
2. gaussian_renderer/__init__.py render() :19-90,256-313 with a RECORDING stub for the absent operator module
   `diff_gaussian_rasterization`: the 12-field settings tuple and the keyword arguments the reference hands its operator,
   for the four input modes (default, compute_cov3D_python, convert_SHs_python, override_color).
"""
import importlib.util
import os
import sys
import types
from typing import NamedTuple
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}


def _drop_device(fn):
    def wrapped(*a, **kw):
        kw.pop("device", None)
        return fn(*a, **kw)
    return wrapped


def _cpu_patches():
    """torch factory functions without their hard-coded device="cuda"."""
    return [mock.patch.object(torch, name, _drop_device(getattr(torch, name)))
            for name in ("zeros", "ones", "tensor", "zeros_like", "empty", "rand", "randn")]


class _Patched:
    def __init__(self, extra=()):
        self.ps = _cpu_patches() + list(extra)

    def __enter__(self):
        for p in self.ps:
            p.start()

    def __exit__(self, *exc):
        for p in reversed(self.ps):
            p.stop()


def load_reference_model_module():
    sys.path.insert(0, REF)
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = type("Absent", (), {})
    knn_pkg, knn_c = types.ModuleType("simple_knn"), types.ModuleType("simple_knn._C")
    knn_c.distCUDA2 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("simple_knn is absent"))
    knn_pkg._C = knn_c
    sys.modules.update({"plyfile": ply, "simple_knn": knn_pkg, "simple_knn._C": knn_c})
    spec = importlib.util.spec_from_file_location("ref_gaussian_model", os.path.join(REF, "scene", "gaussian_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_model(mod, P, sh_degree, seed):
    g = torch.Generator().manual_seed(seed)
    cg = types.SimpleNamespace(learn_split_distance=False, learn_split_scale=False, symmetric_split=False,
                               split_notreinit=False)
    m = mod.GaussianModel(sh_degree, modelcg=cg)
    M = (sh_degree + 1) ** 2
    import math
    raw = {
        "xyz": torch.randn(P, 3, generator=g) * 2.0,
        "f_dc": torch.randn(P, 1, 3, generator=g),
        "f_rest": 0.1 * torch.randn(P, M - 1, 3, generator=g),
        "opacity": 2.5 * torch.randn(P, 1, generator=g) - 1.0,              # some below inverse_sigmoid(0.005)
        "scaling": math.log(0.05) + 1.2 * torch.randn(P, 3, generator=g),    # straddles 0.01 * extent and 0.1 * extent
        "rotation": torch.randn(P, 4, generator=g),
    }
    for k, a in ATTR.items():
        setattr(m, a, torch.nn.Parameter(raw[k].clone().requires_grad_(True)))
    m.active_sh_degree = sh_degree
    m.spatial_lr_scale = 1.0
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                 opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
    with _Patched():
        m.training_setup(args)                                     # :240-266 (Adam with the six named groups)
    for k, a in ATTR.items():                                      # one step creates exp_avg / exp_avg_sq
        p = getattr(m, a)
        p.grad = torch.randn(p.shape, generator=g)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    denom = torch.randint(0, 4, (P, 1), generator=g).float()       # zeros -> 0/0 = NaN -> 0 (:751-752)
    m.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 0.0006 * denom
    m.denom = denom
    m.max_radii2D = torch.floor(torch.rand(P, generator=g) * 40)
    return m, g


def snapshot(m, prefix, out):
    for k, a in ATTR.items():
        p = getattr(m, a)
        out[f"{prefix}/param/{k}"] = p.detach().numpy().copy()
        st = m.optimizer.state[p]
        out[f"{prefix}/exp_avg/{k}"] = st["exp_avg"].numpy().copy()
        out[f"{prefix}/exp_avg_sq/{k}"] = st["exp_avg_sq"].numpy().copy()
    out[f"{prefix}/xyz_gradient_accum"] = m.xyz_gradient_accum.numpy().copy()
    out[f"{prefix}/denom"] = m.denom.numpy().copy()
    out[f"{prefix}/max_radii2D"] = m.max_radii2D.numpy().copy()


def make_model_fixture(mod):
    out = {}
    # ---- getters :151-183 -------------------------------------------------------------------------------------------
    m, g = build_model(mod, 400, 2, seed=11)
    out["getters/in/scaling"] = m._scaling.detach().numpy().copy()
    out["getters/in/rotation"] = m._rotation.detach().numpy().copy()
    out["getters/in/opacity"] = m._opacity.detach().numpy().copy()
    out["getters/in/f_dc"] = m._features_dc.detach().numpy().copy()
    out["getters/in/f_rest"] = m._features_rest.detach().numpy().copy()
    out["getters/out/get_scaling"] = m.get_scaling.detach().numpy().copy()
    out["getters/out/get_rotation"] = m.get_rotation.detach().numpy().copy()
    out["getters/out/get_opacity"] = m.get_opacity.detach().numpy().copy()
    out["getters/out/get_features"] = m.get_features.detach().numpy().copy()

    # ---- add_densification_stats :775-777 (+ train.py:130) ---------------------------------------------------------
    P = 1500
    m, g = build_model(mod, P, 1, seed=21)
    vsp = torch.zeros(P, 3, requires_grad=True)
    vsp.grad = torch.randn(P, 3, generator=g) * torch.tensor([1e-3, 5e-4, 7.0])      # z large: it must not enter the norm
    radii = torch.randint(-1, 40, (P,), generator=g, dtype=torch.int32)
    radii[torch.rand(P, generator=g) < 0.4] = 0
    visibility_filter = radii > 0
    out["stats/in/grad"] = vsp.grad.numpy().copy()
    out["stats/in/radii"] = radii.numpy().copy()
    for frame in (1, 2):
        if frame == 1:
            snapshot(m, "stats/before", out)
        # train.py:130-131
        m.max_radii2D[visibility_filter] = torch.max(m.max_radii2D[visibility_filter], radii[visibility_filter])
        m.add_densification_stats(vsp, visibility_filter)
        for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
            out[f"stats/after{frame}/{k}"] = getattr(m, k).numpy().copy()

    # ---- densify_and_prune :750-772, max_screen_size None and set ---------------------------------------------------
    for tag, max_screen_size, seed in (("densify_vs20", 20, 31), ("densify_none", None, 32)):
        m, g = build_model(mod, 2000, 1, seed=seed)
        snapshot(m, f"{tag}/in", out)
        draws = []

        def recording_normal(mean=None, std=None, **kw):
            z = torch.randn(std.shape, generator=g)
            draws.append(z.clone())
            return mean + std * z

        max_grad, min_opacity, extent = 0.0002, 0.005, 5.0
        with _Patched([mock.patch.object(torch, "normal", recording_normal)]):
            m.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)
        assert len(draws) == 1
        out[f"{tag}/noise"] = draws[0].numpy()
        out[f"{tag}/args"] = np.array([max_grad, min_opacity, extent, -1.0 if max_screen_size is None else max_screen_size,
                                       m.percent_dense], dtype=np.float64)
        snapshot(m, f"{tag}/out", out)
        n_in, n_out = out[f"{tag}/in/param/xyz"].shape[0], out[f"{tag}/out/param/xyz"].shape[0]
        assert draws[0].shape[0] > 40 and n_out != n_in, (draws[0].shape, n_in, n_out)
        for p in (grp["params"][0] for grp in m.optimizer.param_groups):          # the optimizer survives the surgery
            p.grad = torch.ones_like(p)
        m.optimizer.step()
    np.savez_compressed(os.path.join(OUT, "model_densify.npz"), **out)
    print("wrote model_densify.npz:", len(out), "arrays,",
          {k: out[k].shape for k in ("densify_vs20/in/param/xyz", "densify_vs20/out/param/xyz", "densify_vs20/noise",
                                     "densify_none/out/param/xyz")})
    return mod


# ---- render() call contract ------------------------------------------------------------------------------------------
class RecordedSettings(NamedTuple):     # the stub's own 12-field tuple: render() fills it by keyword (:42-55)
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


CALLS = []


class RecordingRasterizer:
    def __init__(self, raster_settings):
        self.raster_settings = raster_settings

    def __call__(self, **kw):
        CALLS.append((self.raster_settings, dict(kw)))
        P = kw["means3D"].shape[0]
        H, W = self.raster_settings.image_height, self.raster_settings.image_width
        radii = (torch.arange(P) % 3).to(torch.int32)                # 0, 1, 2, 0, ...: visibility_filter = radii > 0
        return torch.full((3, H, W), 0.25), radii


def make_render_fixture(model_mod):
    stub = types.ModuleType("diff_gaussian_rasterization")
    stub.GaussianRasterizationSettings = RecordedSettings
    stub.GaussianRasterizer = RecordingRasterizer
    scene_pkg = types.ModuleType("scene")
    scene_pkg.__path__ = []
    sys.modules.update({"diff_gaussian_rasterization": stub, "scene": scene_pkg, "scene.gaussian_model": model_mod})
    spec = importlib.util.spec_from_file_location("ref_gaussian_renderer", os.path.join(REF, "gaussian_renderer", "__init__.py"))
    rmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rmod)

    m, g = build_model(model_mod, 300, 3, seed=41)
    m.active_sh_degree = 2                                           # fewer active than stored degrees
    import math
    cam = types.SimpleNamespace(image_height=72, image_width=104, FoVx=1.1, FoVy=0.8,
                                world_view_transform=torch.randn(4, 4, generator=g),
                                full_proj_transform=torch.randn(4, 4, generator=g),
                                camera_center=torch.tensor([0.3, -0.2, -0.5]))
    bg = torch.tensor([0.1, 0.2, 0.3])
    override = torch.rand(300, 3, generator=g)
    out = {"cam/world_view_transform": cam.world_view_transform.numpy(), "cam/full_proj_transform": cam.full_proj_transform.numpy(),
           "cam/camera_center": cam.camera_center.numpy(), "cam/fov_hw": np.array([cam.FoVx, cam.FoVy, 72, 104], dtype=np.float64),
           "bg": bg.numpy(), "override_color": override.numpy(), "active_sh_degree": np.array([2, 3])}
    for k, a in ATTR.items():
        out[f"model/{k}"] = getattr(m, a).detach().numpy().copy()
    modes = {
        "default": dict(pipe=types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False), kw={}),
        "cov3d_python": dict(pipe=types.SimpleNamespace(compute_cov3D_python=True, convert_SHs_python=False, debug=True),
                             kw=dict(scaling_modifier=1.7)),
        "shs_python": dict(pipe=types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=True, debug=False), kw={}),
        "override_color": dict(pipe=types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False),
                               kw=dict(override_color=override, scaling_modifier=0.6)),
    }
    for name, md in modes.items():
        CALLS.clear()
        with _Patched():
            res = rmod.render(cam, m, md["pipe"], bg, **md["kw"])
        assert len(CALLS) == 1
        st, kw = CALLS[0]
        for f in RecordedSettings._fields:
            v = getattr(st, f)
            out[f"{name}/settings/{f}"] = v.detach().numpy().copy() if torch.is_tensor(v) else np.array(v)
        none_keys = []
        for k, v in kw.items():
            if v is None:
                none_keys.append(k)
            else:
                out[f"{name}/kwargs/{k}"] = v.detach().numpy().copy()
        out[f"{name}/kwargs_none"] = np.array(sorted(none_keys))
        out[f"{name}/kwargs_names"] = np.array(sorted(kw.keys()))
        out[f"{name}/result_keys"] = np.array(sorted(res.keys()))
        out[f"{name}/result/visibility_filter"] = res["visibility_filter"].numpy().copy()
        out[f"{name}/result/viewspace_points_shape"] = np.array(res["viewspace_points"].shape)
        out[f"{name}/result/viewspace_points_abs_max"] = np.array(float(res["viewspace_points"].abs().max()))
        out[f"{name}/result/selected_pts_mask_is_none"] = np.array(res["selected_pts_mask"] is None)
        assert res["viewspace_points"].requires_grad and kw["means2D"] is res["viewspace_points"]
    np.savez_compressed(os.path.join(OUT, "render_contract.npz"), **out)
    print("wrote render_contract.npz:", len(out), "arrays; kwargs of the default mode:", list(out["default/kwargs_names"]),
          "None:", list(out["default/kwargs_none"]))


if __name__ == "__main__":
    model_mod = load_reference_model_module()
    make_model_fixture(model_mod)
    make_render_fixture(model_mod)
