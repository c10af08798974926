"""Generates tests/golden/*.npz from the reference's own importable Python (run ONLY in the build
container, where /root/reference exists; the fixtures -- plain input/output arrays -- are committed):

    python tests/golden/make_golden.py

Pinned sub-steps: utils/sh_utils.py (eval_sh, RGB2SH, SH2RGB), utils/graphics_utils.py
(getWorld2View2, getProjectionMatrix, focal2fov, geom_transform_points), utils/loss_utils.py (l1_loss,
l2_loss, ssim), utils/image_utils.py (psnr).  The CUDA rasterizer itself is absent from the reference,
so no fixture can pin it ("parity unpinned" at that boundary).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh, RGB2SH, SH2RGB
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix, focal2fov, fov2focal, geom_transform_points
    from utils.loss_utils import l1_loss, l2_loss, ssim
    from utils.image_utils import psnr

    g = torch.Generator().manual_seed(1234)
    # ---- SH: [P,3,16] channel-major as eval_sh expects, unit directions -------------------
    P = 257
    sh = torch.randn(P, 3, 16, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1)
    out = {"sh": sh.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        out[f"eval_deg{deg}"] = eval_sh(deg, sh, dirs).numpy()
    rgb = torch.rand(64, 3, generator=g)
    out["rgb"] = rgb.numpy()
    out["rgb2sh"] = RGB2SH(rgb).numpy()
    out["sh2rgb"] = SH2RGB(RGB2SH(rgb)).numpy()
    np.savez_compressed(os.path.join(OUT, "sh.npz"), **out)

    # ---- camera matrices ---------------------------------------------------------------
    cams = {}
    rng = np.random.default_rng(7)
    for i in range(3):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        t = rng.normal(size=3)
        fx, fy, W, H = 900.0 + 100 * i, 1000.0 - 50 * i, 1920, 1080
        fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
        w2v = getWorld2View2(Q, t)
        w2v_ts = getWorld2View2(Q, t, np.array([0.1, -0.2, 0.3]), 1.5)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy)
        wvt = torch.tensor(w2v).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.transpose(0, 1).unsqueeze(0))).squeeze(0)
        pts = torch.randn(32, 3, generator=g) + torch.tensor([0.0, 0.0, 5.0])
        cams.update({f"R{i}": Q, f"t{i}": t, f"fx{i}": fx, f"fy{i}": fy, f"fovx{i}": fovx, f"fovy{i}": fovy,
                     f"fx_back{i}": fov2focal(fovx, W), f"w2v{i}": w2v, f"w2v_ts{i}": w2v_ts, f"proj{i}": proj.numpy(),
                     f"full{i}": full.numpy(), f"center{i}": wvt.inverse()[3, :3].numpy(), f"pts{i}": pts.numpy(),
                     f"pts_proj{i}": geom_transform_points(pts, full).numpy()})
    np.savez_compressed(os.path.join(OUT, "camera.npz"), **cams)

    # ---- losses ------------------------------------------------------------------------
    a = torch.rand(3, 48, 64, generator=g)
    b = torch.rand(3, 48, 64, generator=g)
    a.requires_grad_(True)
    l1 = l1_loss(a, b)
    (g1,) = torch.autograd.grad(l1, a)
    s = ssim(a, b)
    (gs,) = torch.autograd.grad(s, a)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), a=a.detach().numpy(), b=b.numpy(), l1=l1.item(),
                        l1_grad=g1.numpy(), l2=l2_loss(a, b).item(), ssim=s.item(), ssim_grad=gs.numpy(),
                        psnr=psnr(a.detach()[None], b[None]).numpy())
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
