"""Where does the HIP image of tests/test_gpu_guard.py's 60:1 needle scene leave the float64 oracle?"""
import sys
import torch
sys.path.insert(0, "tests")
from conftest import make_settings
from test_gpu_guard import _needles
from gpu_util import forward_with_state, product_settings
from oracle import rasterize_ref

dev = torch.device("cuda:0")
model, cam, bg = _needles(1500, 240, 144, 4.0, 60.0, 3.0, 60.0, seed=21)
kw64 = dict(shs=model.get_features.double(), scales=model.get_scaling.double(), rotations=model.get_rotation.double())
col64, radii, aux = rasterize_ref(model.get_xyz.double(), None, model.get_opacity.double(), make_settings(cam, bg, 0),
                                  want_aux=True, want_margin=True, **kw64)
col32, _ = rasterize_ref(model.get_xyz, None, model.get_opacity, make_settings(cam, bg, 0), shs=model.get_features,
                         scales=model.get_scaling, rotations=model.get_rotation)
col64 = col64.float()
robust = aux["margin"] > 1e-4
print("float32 oracle vs float64 on robust pixels:", float(((col32 - col64).abs().max(dim=0).values)[robust].max()))
for mode in (0, 1, 2):
    out = forward_with_state(dev, product_settings(cam, bg, 0, dev), model.get_xyz, model.get_opacity, shs=model.get_features,
                             scales=model.get_scaling, rotations=model.get_rotation, binning_mode=mode)
    err = (out["color"] - col64).abs().max(dim=0).values
    e = err.clone(); e[~robust] = 0
    iy, ix = divmod(int(e.argmax()), e.shape[1])
    print(f"mode {mode}: max err on robust {float(e.max()):.3e} at pixel ({ix},{iy}), margin there {float(aux['margin'][iy, ix]):.3e}, "
          f"n_contrib hip {int(out['n_contrib'][iy, ix])} oracle {int(aux['n_contrib'][iy, ix])}, radii equal {bool(torch.equal(out['radii'], radii))}, "
          f"pixels above 1e-5: {int((e > 1e-5).sum())}")
    if mode == 0:
        import numpy as np
        print("  lists equal:", np.array_equal(out["point_list"], aux["point_list"]))
        # which Gaussians cover that pixel
        t = (iy // 16) * ((240 + 15) // 16) + ix // 16
        s, e_ = aux["ranges"][t]
        ids = aux["point_list"][s:e_]
        pre = aux["pre"]
        print("  tile", t, "list", len(ids))
