"""Forward of a small scene against the float32 oracle: where (which mini-blocks / pixels) do the images differ."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests")
from conftest import make_settings, small_scene
from gpu_util import forward_with_state, product_settings
from oracle import rasterize_ref

dev = torch.device("cuda:0")
P, deg, w, h = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (3000, 3, 320, 176)
model, cam, bg, _ = small_scene(P=P, sh_degree=deg, width=w, height=h)
bg = torch.tensor([0.1, 0.2, 0.3])
col, radii, aux = rasterize_ref(model.get_xyz, None, model.get_opacity, make_settings(cam, bg, deg), shs=model.get_features,
                                scales=model.get_scaling, rotations=model.get_rotation, want_aux=True, want_margin=True)
out = forward_with_state(dev, product_settings(cam, bg, deg, dev), model.get_xyz, model.get_opacity, shs=model.get_features,
                         scales=model.get_scaling, rotations=model.get_rotation)
err = (out["color"] - col).abs().max(dim=0).values
robust = aux["margin"] > 1e-4
bad = (err > 1e-5) & robust
print("bad robust pixels", int(bad.sum()), "of", int(robust.sum()), "max err", float(err[robust].max()))
print("n_contrib mismatches", int((out["n_contrib"][robust] != aux["n_contrib"][robust]).sum()))
ys, xs = np.nonzero(bad.numpy())
if len(ys):
    mb = {}
    for y, x in zip(ys, xs):
        k = ((y % 16) // 4, (x % 16) // 4)
        mb[k] = mb.get(k, 0) + 1
    print("bad pixels by mini-block (row, col):", sorted(mb.items()))
    for y, x in list(zip(ys, xs))[:10]:
        print(f"  px ({x},{y}) tile ({x // 16},{y // 16}) got {out['color'][:, y, x].tolist()} want {col[:, y, x].tolist()} "
              f"n_contrib {int(out['n_contrib'][y, x])} vs {int(aux['n_contrib'][y, x])}")
