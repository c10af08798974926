"""Offline rendering in the shape of the reference's ``render.py:37-49``: load a model PLY (``save_ply`` layout), render
a set of orbit views through the drop-in ``render`` and write them as binary PPM images.

    python examples/render_ply.py point_cloud.ply out_dir [n_views] [width] [height]

By default the model is rendered in the order the PLY stores it: the images are then those of the reference's stable
index-order sort.  ``render_set(..., spatial_order=True)`` (opt-in) puts the loaded model along a Morton curve first
(``mvs_gaussian_splatting_amd/layout.py``: frames 5-7 % faster at 6 M Gaussians, but equal-depth ties inside a tile may blend
in the other order -- 279 of 2 M pixels differed by up to 1.2e-3 at 6 M Gaussians).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mvs_gaussian_splatting_amd.graphed import GraphedRenderer  # noqa: E402
from mvs_gaussian_splatting_amd.layout import reorder_gaussians_  # noqa: E402
from mvs_gaussian_splatting_amd.ply_io import load_ply  # noqa: E402
from mvs_gaussian_splatting_amd.synthetic import PipelineParams, orbit_camera  # noqa: E402


class PlyModel:
    """The attributes ``render`` reads from ``GaussianModel`` (``scene/gaussian_model.py:151-194``)."""

    def __init__(self, path, device, max_sh_degree=3):
        for k, v in load_ply(path, max_sh_degree, device).items():
            setattr(self, k, v)
        self.max_sh_degree = max_sh_degree
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))


def write_ppm(path, image):
    """image: [3, H, W] in [0, 1] -> binary PPM (P6)."""
    img = (image.clamp(0.0, 1.0) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
    with open(path, "wb") as f:
        f.write(f"P6\n{img.shape[1]} {img.shape[0]}\n255\n".encode())
        f.write(img.tobytes())


def render_set(ply_path, out_dir, n_views=8, width=256, height=160, focal=220.0, centre=(0.0, 0.0, 4.0), device="cuda:0",
               spatial_order=False):
    dev = torch.device(device)
    model = PlyModel(ply_path, dev)
    if spatial_order:
        reorder_gaussians_(model)
    bg = torch.zeros(3, device=dev)
    os.makedirs(out_dir, exist_ok=True)
    images = []
    # a fixed model seen from many cameras: the frame is captured once as a HIP graph and replayed per camera;
    # verify=True checks every frame's instance count against the workspace before the image is trusted (and re-renders
    # it if the count grew beyond it), which is what a renderer that writes every frame to disk wants
    renderer = GraphedRenderer(model, PipelineParams(), bg)
    with torch.no_grad():                                       # render.py:38
        for v in range(n_views):
            cam = orbit_camera(v, n_views, width, height, focal, focal, centre=centre, device=dev)
            images.append(renderer.render(cam, verify=True)["render"].clone())
            write_ppm(os.path.join(out_dir, f"{v:05d}.ppm"), images[-1])
    return model, images


if __name__ == "__main__":
    a = sys.argv[1:]
    render_set(a[0], a[1], *(int(x) for x in a[2:5]))
    print("wrote", a[1])
