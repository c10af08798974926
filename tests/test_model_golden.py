"""Caller-side rows of SURVEY §8 against the reference's OWN Python (fixtures of tests/golden/make_golden_model.py, which
imports scene/gaussian_model.py and gaussian_renderer/__init__.py in the build container):

* a1-a3  ``render()`` hands its operator the settings tuple and keyword arguments the reference's ``render()`` does
         (gaussian_renderer/__init__.py:19-90,256-313), in all four input modes -- CPU, with a recording operator;
* a14    the getters (scene/gaussian_model.py:151-183) of the synthetic model;
* a13    ``add_densification_stats`` + the max_radii2D update (:775-777, train.py:130): CPU check of the formula the GPU
         tests use, and the HIP kernel (stand-alone and fused into the backward) under -m gpu;
* f3     ``densify_and_prune`` (:750-772) on a model with Adam state: the CPU restatement oracle/densify_ref.py (which the
         larger GPU tests compare the HIP path with) and the HIP path itself, against the reference's outputs, with the
         reference's own normal draws.
"""
import math
import os
import types

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}


@pytest.fixture(scope="module")
def model_gold():
    return np.load(os.path.join(GOLD, "model_densify.npz"))


@pytest.fixture(scope="module")
def render_gold():
    return np.load(os.path.join(GOLD, "render_contract.npz"))


# ---- a1-a3: the render() call contract ---------------------------------------------------------------------------------
class _Recorder:
    calls = []

    def __init__(self, raster_settings):
        self.raster_settings = raster_settings

    def __call__(self, **kw):
        _Recorder.calls.append((self.raster_settings, dict(kw)))
        P = kw["means3D"].shape[0]
        st = self.raster_settings
        return torch.full((3, st.image_height, st.image_width), 0.25), (torch.arange(P) % 3).to(torch.int32)


def _duck_model(g):
    from mvs_gaussian_splatting_amd.synthetic import SyntheticGaussianModel
    m = SyntheticGaussianModel(300, 3)
    for k, a in ATTR.items():
        setattr(m, a, torch.tensor(g[f"model/{k}"]))
    m.active_sh_degree, m.max_sh_degree = int(g["active_sh_degree"][0]), int(g["active_sh_degree"][1])
    return m


@pytest.mark.parametrize("mode", ["default", "cov3d_python", "shs_python", "override_color"])
def test_render_hands_the_operator_what_the_reference_does(render_gold, mode, monkeypatch):
    from mvs_gaussian_splatting_amd import renderer
    from mvs_gaussian_splatting_amd.rasterizer import GaussianRasterizationSettings
    g = render_gold
    monkeypatch.setattr(renderer, "GaussianRasterizer", _Recorder)
    _Recorder.calls.clear()
    m = _duck_model(g)
    fovx, fovy, H, W = (float(v) for v in g["cam/fov_hw"])
    cam = types.SimpleNamespace(image_height=int(H), image_width=int(W), FoVx=fovx, FoVy=fovy,
                                world_view_transform=torch.tensor(g["cam/world_view_transform"]),
                                full_proj_transform=torch.tensor(g["cam/full_proj_transform"]),
                                camera_center=torch.tensor(g["cam/camera_center"]))
    pipe = types.SimpleNamespace(compute_cov3D_python=mode == "cov3d_python", convert_SHs_python=mode == "shs_python",
                                 debug=mode == "cov3d_python", fuse_activations=False)     # the reference has no fused path
    kw = {"cov3d_python": dict(scaling_modifier=1.7),
          "override_color": dict(override_color=torch.tensor(g["override_color"]), scaling_modifier=0.6)}.get(mode, {})
    res = renderer.render(cam, m, pipe, torch.tensor(g["bg"]), **kw)
    assert len(_Recorder.calls) == 1
    st, got = _Recorder.calls[0]
    # the settings tuple: our NamedTuple has the reference's 12 fields in the reference's order, same values and types
    assert isinstance(st, GaussianRasterizationSettings)
    assert list(st._fields) == ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                                "projmatrix", "sh_degree", "campos", "prefiltered", "debug"]
    for f in st._fields:
        want = g[f"{mode}/settings/{f}"]
        v = getattr(st, f)
        if torch.is_tensor(v):
            assert np.array_equal(v.numpy(), want), f
        else:
            assert type(v) in (int, float, bool) and v == want.item(), (f, v, want)
    # the keyword arguments: same names, the same ones None, the same values bit for bit
    assert sorted(got.keys()) == list(g[f"{mode}/kwargs_names"])
    assert sorted(k for k, v in got.items() if v is None) == list(g[f"{mode}/kwargs_none"])
    for k, v in got.items():
        if v is not None:
            want = g[f"{mode}/kwargs/{k}"]
            assert tuple(v.shape) == want.shape, k
            if (k == "colors_precomp" and mode == "shs_python") or k == "cov3D_precomp":
                # computed values (eval_sh here; get_covariance of the duck-typed model, which builds R and L with other
                # tensor operations than utils/general_utils.py): float32 operation order may differ by an ulp
                assert np.allclose(v.detach().numpy(), want, rtol=2e-6, atol=1e-7), k
            else:
                assert np.array_equal(v.detach().numpy(), want), k
    # the result dict
    assert sorted(res.keys()) == list(g[f"{mode}/result_keys"])
    assert np.array_equal(res["visibility_filter"].numpy(), g[f"{mode}/result/visibility_filter"])
    assert list(res["viewspace_points"].shape) == list(g[f"{mode}/result/viewspace_points_shape"])
    assert float(res["viewspace_points"].detach().abs().max()) == float(g[f"{mode}/result/viewspace_points_abs_max"]) == 0.0
    assert res["viewspace_points"].requires_grad and got["means2D"] is res["viewspace_points"]
    assert res["selected_pts_mask"] is None and bool(g[f"{mode}/result/selected_pts_mask_is_none"])


# ---- a14: the getters -----------------------------------------------------------------------------------------------
def test_synthetic_model_getters_match_the_reference_class(model_gold):
    from mvs_gaussian_splatting_amd.synthetic import SyntheticGaussianModel
    g = model_gold
    m = SyntheticGaussianModel(400, 2)
    m._scaling, m._rotation, m._opacity = (torch.tensor(g[f"getters/in/{k}"]) for k in ("scaling", "rotation", "opacity"))
    m._features_dc, m._features_rest = torch.tensor(g["getters/in/f_dc"]), torch.tensor(g["getters/in/f_rest"])
    for name in ("get_scaling", "get_rotation", "get_opacity", "get_features"):
        assert np.array_equal(getattr(m, name).numpy(), g[f"getters/out/{name}"]), name


# ---- a13: densification statistics ------------------------------------------------------------------------------------
def _stats_inputs(g, dev="cpu"):
    grad = torch.tensor(g["stats/in/grad"]).to(dev)
    radii = torch.tensor(g["stats/in/radii"]).to(dev)
    model = types.SimpleNamespace(xyz_gradient_accum=torch.tensor(g["stats/before/xyz_gradient_accum"]).to(dev),
                                  denom=torch.tensor(g["stats/before/denom"]).to(dev),
                                  max_radii2D=torch.tensor(g["stats/before/max_radii2D"]).to(dev))
    return model, grad, radii


def test_reference_stats_lines_as_the_gpu_tests_restate_them(model_gold):
    """tests/test_gpu_densify_stats.py compares the HIP kernel with three torch lines retyped from the reference: those
    lines reproduce the reference class's own outputs."""
    from test_gpu_densify_stats import _reference_update
    g = model_gold
    model, grad, radii = _stats_inputs(g)
    state = (model.xyz_gradient_accum, model.denom, model.max_radii2D)
    for frame in (1, 2):
        state = _reference_update(*state, grad, radii)
        for got, k in zip(state, ("xyz_gradient_accum", "denom", "max_radii2D")):
            assert np.array_equal(got.numpy(), g[f"stats/after{frame}/{k}"]), (frame, k)


@pytest.mark.gpu
def test_hip_densify_stats_match_the_reference_class(model_gold, gpu_device):
    from mvs_gaussian_splatting_amd import add_densification_stats
    g = model_gold
    model, grad, radii = _stats_inputs(g, gpu_device)
    vsp = torch.zeros_like(grad, requires_grad=True)
    vsp.grad = grad
    for frame in (1, 2):
        add_densification_stats(model, vsp, radii)
        assert np.array_equal(model.denom.cpu().numpy(), g[f"stats/after{frame}/denom"])
        assert np.array_equal(model.max_radii2D.cpu().numpy(), g[f"stats/after{frame}/max_radii2D"])
        want = g[f"stats/after{frame}/xyz_gradient_accum"]
        err = np.abs(model.xyz_gradient_accum.cpu().numpy() - want) / np.maximum(np.abs(want), 1e-12)
        assert float(err.max()) <= 1e-6          # norm(): sqrt of a two-term sum, one rounding apart at most
        untouched = (g["stats/in/radii"] <= 0)
        assert np.array_equal(model.xyz_gradient_accum.cpu().numpy()[untouched], g["stats/before/xyz_gradient_accum"][untouched])


# ---- f3: densify_and_prune ------------------------------------------------------------------------------------------
def _densify_case(g, tag):
    params = {k: torch.tensor(g[f"{tag}/in/param/{k}"]) for k in GROUPS}
    moments = {k: (torch.tensor(g[f"{tag}/in/exp_avg/{k}"]), torch.tensor(g[f"{tag}/in/exp_avg_sq/{k}"])) for k in GROUPS}
    max_grad, min_opacity, extent, mss, percent_dense = (float(v) for v in g[f"{tag}/args"])
    return dict(params=params, moments=moments, accum=torch.tensor(g[f"{tag}/in/xyz_gradient_accum"]),
                denom=torch.tensor(g[f"{tag}/in/denom"]), radii=torch.tensor(g[f"{tag}/in/max_radii2D"]),
                noise=torch.tensor(g[f"{tag}/noise"]), max_grad=max_grad, min_opacity=min_opacity, extent=extent,
                max_screen_size=None if mss < 0 else mss, percent_dense=percent_dense)


def _check_against_reference(g, tag, params, moments, accum, denom, radii):
    n_out = g[f"{tag}/out/param/xyz"].shape[0]
    n_children = g[f"{tag}/noise"].shape[0]               # rows the split computed (upper bound of those that survive)
    for k in GROUPS:
        want = g[f"{tag}/out/param/{k}"]
        got = params[k].detach().cpu().numpy()
        assert got.shape == want.shape, (k, got.shape, want.shape)
        if k in ("xyz", "scaling"):                       # children: exp / log / 3x3 product in float32
            same = np.all(got == want, axis=tuple(range(1, got.ndim)))
            assert int((~same).sum()) <= n_children
            err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
            assert float(err.max()) <= 1e-6, k
            first_computed = n_out - min(n_children, n_out)
            assert np.array_equal(got[:first_computed], want[:first_computed]), k      # copies: bit-identical
        else:
            assert np.array_equal(got, want), k
        for which, idx in (("exp_avg", 0), ("exp_avg_sq", 1)):
            assert np.array_equal(moments[k][idx].cpu().numpy(), g[f"{tag}/out/{which}/{k}"]), (k, which)
    for got, k in ((accum, "xyz_gradient_accum"), (denom, "denom"), (radii, "max_radii2D")):
        assert np.array_equal(got.cpu().numpy(), g[f"{tag}/out/{k}"]), k


@pytest.mark.parametrize("tag", ["densify_vs20", "densify_none"])
def test_cpu_restatement_of_densify_and_prune_matches_the_reference_class(model_gold, tag):
    from oracle.densify_ref import densify_and_prune_ref, count_split_selected_ref
    g = model_gold
    c = _densify_case(g, tag)
    assert 2 * count_split_selected_ref(c["params"], c["accum"].clone(), c["denom"], c["percent_dense"], c["max_grad"],
                                        c["extent"]) == c["noise"].shape[0]
    p, m, accum, denom, radii, info = densify_and_prune_ref(c["params"], c["moments"], c["accum"], c["denom"], c["radii"],
                                                            c["percent_dense"], c["max_grad"], c["min_opacity"], c["extent"],
                                                            c["max_screen_size"], c["noise"])
    assert info["cloned"] > 0 and info["split"] > 0 and info["pruned"] > 0
    _check_against_reference(g, tag, p, m, accum, denom, radii)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["densify_vs20", "densify_none"])
def test_hip_densify_and_prune_matches_the_reference_class(model_gold, gpu_device, tag):
    from mvs_gaussian_splatting_amd.densify import densify_and_prune
    g = model_gold
    c = _densify_case(g, tag)
    dev = gpu_device
    model = types.SimpleNamespace(percent_dense=c["percent_dense"], xyz_gradient_accum=c["accum"].to(dev),
                                  denom=c["denom"].to(dev), max_radii2D=c["radii"].to(dev))
    for k, a in ATTR.items():
        setattr(model, a, torch.nn.Parameter(c["params"][k].to(dev).requires_grad_(True)))
    groups = [{"params": [getattr(model, ATTR[k])], "lr": 1e-3, "name": k} for k in GROUPS]
    model.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for k in GROUPS:              # install the reference's Adam moments
        p = getattr(model, ATTR[k])
        model.optimizer.state[p] = {"step": torch.tensor(1.0), "exp_avg": c["moments"][k][0].to(dev),
                                    "exp_avg_sq": c["moments"][k][1].to(dev)}
    info = densify_and_prune(model, c["max_grad"], c["min_opacity"], c["extent"], c["max_screen_size"],
                             noise=c["noise"].to(dev))
    assert info["points"] == g[f"{tag}/out/param/xyz"].shape[0] and 2 * info["split_selected"] == c["noise"].shape[0]
    params = {k: getattr(model, ATTR[k]) for k in GROUPS}
    moments = {k: (model.optimizer.state[params[k]]["exp_avg"], model.optimizer.state[params[k]]["exp_avg_sq"]) for k in GROUPS}
    _check_against_reference(g, tag, params, moments, model.xyz_gradient_accum, model.denom, model.max_radii2D)
    for p in params.values():     # the optimizer steps on the regrown model
        p.grad = torch.ones_like(p)
    model.optimizer.step()
