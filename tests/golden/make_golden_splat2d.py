"""Golden vectors for BASELINE config 1 from the reference's own function bodies (build container only).

The reference script downloads two files at import (2d_gaussian_splatting.py:208-218), so the module is
never imported or run: only its ``FunctionDef`` nodes are parsed out with ``ast`` and executed with
torch/numpy in scope.  Output: tests/golden/splat2d.npz (inputs + image, loss and gradients).

    python tests/golden/make_golden_splat2d.py
"""
import ast
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SRC = "/root/reference/2D-Gaussian-Splatting-main/2d_gaussian_splatting.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "splat2d.npz")
WANT = {"generate_2D_gaussian_splatting", "create_window", "ssim", "d_ssim_loss", "combined_loss"}


def main():
    tree = ast.parse(open(SRC).read())
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    ns = {"torch": torch, "np": np, "nn": nn, "F": F}
    exec(compile(ast.Module(body=defs, type_ignores=[]), SRC, "exec"), ns)

    g = torch.Generator().manual_seed(11)
    N, K, size = 64, 33, (128, 128, 3)
    sx = (torch.rand(N, generator=g) * 0.8 + 0.2).requires_grad_(True)
    sy = (torch.rand(N, generator=g) * 0.8 + 0.2).requires_grad_(True)
    rho = (torch.rand(N, generator=g) * 1.6 - 0.8).requires_grad_(True)
    coords = (torch.rand(N, 2, generator=g) * 1.6 - 0.8).requires_grad_(True)
    colours = torch.rand(N, 3, generator=g).requires_grad_(True)
    target = torch.rand(*size, generator=g)
    img = ns["generate_2D_gaussian_splatting"](K, sx, sy, rho, coords, colours, size, "cpu")
    loss = ns["combined_loss"](img, target, lambda_param=0.2)
    grads = torch.autograd.grad(loss, [sx, sy, rho, coords, colours])
    np.savez_compressed(OUT, K=K, size=np.array(size), sx=sx.detach().numpy(), sy=sy.detach().numpy(),
                        rho=rho.detach().numpy(), coords=coords.detach().numpy(), colours=colours.detach().numpy(),
                        target=target.numpy(), image=img.detach().numpy(), loss=loss.item(),
                        g_sx=grads[0].numpy(), g_sy=grads[1].numpy(), g_rho=grads[2].numpy(),
                        g_coords=grads[3].numpy(), g_colours=grads[4].numpy())
    print("wrote", OUT, "loss", loss.item())

    # BASELINE config 1 at its full size: N = 1000 Gaussians, K = 101 (config.yml:1), 128 x 128, the script's own
    # parameterisation (sigmoid / tanh of W's columns, :321-330) and lambda = 0.2 (:334)
    g = torch.Generator().manual_seed(12)
    N, K, size = 1000, 101, (128, 128, 3)
    Wp = torch.cat([torch.rand(N, 2, generator=g), 2 * torch.rand(N, 1, generator=g) - 1,
                    torch.logit(0.01 + 0.09 * torch.rand(N, 1, generator=g)),   # alpha: mostly unsaturated sum
                    torch.logit(torch.rand(N, 3, generator=g).clamp(0.02, 0.98)),
                    torch.atanh((torch.rand(N, 2, generator=g) * 2 - 1) * 0.98)], dim=1)
    sx = torch.sigmoid(Wp[:, 0]).requires_grad_(True)
    sy = torch.sigmoid(Wp[:, 1]).requires_grad_(True)
    rho = torch.tanh(Wp[:, 2]).requires_grad_(True)
    colours = (torch.sigmoid(Wp[:, 4:7]) * torch.sigmoid(Wp[:, 3]).view(N, 1)).requires_grad_(True)
    coords = torch.tanh(Wp[:, 7:9]).requires_grad_(True)
    target = torch.rand(*size, generator=g).half().float()      # exactly representable in the float16 fixture
    img = ns["generate_2D_gaussian_splatting"](K, sx, sy, rho, coords, colours, size, "cpu")
    loss = ns["combined_loss"](img, target, lambda_param=0.2)
    grads = torch.autograd.grad(loss, [sx, sy, rho, coords, colours])
    out = OUT.replace("splat2d.npz", "splat2d_c1.npz")
    np.savez_compressed(out, K=K, size=np.array(size), sx=sx.detach().numpy(), sy=sy.detach().numpy(),
                        rho=rho.detach().numpy(), coords=coords.detach().numpy(), colours=colours.detach().numpy(),
                        target=target.numpy().astype(np.float16), image=img.detach().numpy(), loss=loss.item(),
                        g_sx=grads[0].numpy(), g_sy=grads[1].numpy(), g_rho=grads[2].numpy(),
                        g_coords=grads[3].numpy(), g_colours=grads[4].numpy())
    print("wrote", out, "loss", loss.item(), "clamped px", int((img >= 1).sum()))


if __name__ == "__main__":
    main()
