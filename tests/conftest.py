import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_settings(cam, bg, sh_degree, cls=None, scale_modifier=1.0, debug=False):
    """Settings tuple for either the oracle (default) or the product operator."""
    if cls is None:
        from oracle import RasterSettings as cls
    return cls(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg,
               scale_modifier, cam.world_view_transform, cam.full_proj_transform, sh_degree, cam.camera_center,
               False, debug)


def small_scene(P=3000, sh_degree=3, width=320, height=176, focal=200.0, scale=0.05, seed=0, view=0):
    """A few-thousand-Gaussian scene the oracle renders in seconds (odd size: the last tile row/col are partial)."""
    from mvs_gaussian_splatting_amd.synthetic import SceneConfig, make_scene
    cfg = SceneConfig("small", P, sh_degree, width, height, focal, focal, math.log(scale))
    return make_scene(cfg, seed=seed, view=view)


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
